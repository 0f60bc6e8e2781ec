O=gpurun_out/r04_t4; mkdir -p $O
python -m pytest tests/test_gpu_model.py -q -s -x -k "gan_iteration or b1_index or flat_buffer" > $O/tests.log 2>&1; tail -12 $O/tests.log
for rep in 1 2; do
echo "--- base"; python tools/conv_bench.py 32 2>&1 | grep -v amdgpu.ids | head -3
echo "--- setprio"; FAVAE_HIP_LIB=$PWD/tools/experiments/lib_setprio.so python tools/conv_bench.py 32 2>&1 | grep -v amdgpu.ids | head -3
done
