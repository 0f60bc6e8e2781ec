P=$PWD/tools/experiments/lib_fft_noslp.so
for i in 1 2; do
  echo "--- FFT compiled without SLP vectorisation (fewer packed fp32 ops) ($i)"; FAVAE_HIP_LIB=$P FFL_INPROC=wgrad timeout 300 python tools/experiments/ffl_race2.py W 60000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
  echo "--- product ($i)"; FFL_INPROC=wgrad timeout 300 python tools/experiments/ffl_race2.py W 60000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done
