P=$PWD/tools/experiments/lib_fft_nonop.so
for i in 1 2; do
  echo "--- without the wait states ($i)"; FAVAE_HIP_LIB=$P FFL_INPROC=wgrad timeout 300 python tools/experiments/ffl_race2.py W 80000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
  echo "--- with them ($i)"; FFL_INPROC=wgrad timeout 300 python tools/experiments/ffl_race2.py W 80000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done
