O=gpurun_out/r04_race; mkdir -p $O
echo "--- FFL backward with fp16 GEMMs on a second stream of the SAME process"
FFL_INPROC=1 timeout 400 python tools/experiments/ffl_race2.py A 300000 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -5
