O=gpurun_out/r04_t3; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -x -k "wino or split_precision" > $O/tests.log 2>&1; tail -8 $O/tests.log
echo "--- new kernel"; python tools/conv_bench.py 32 2>&1 | grep -v amdgpu.ids | head -6 | tee $O/bench_new.txt
echo "--- old kernel"; FAVAE_WINO_R=0 python tools/conv_bench.py 32 2>&1 | grep -v amdgpu.ids | head -6 | tee $O/bench_old.txt
