O=gpurun_out/r04_t6; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -x -k "wino or split_precision or conv" > $O/tests.log 2>&1; tail -3 $O/tests.log
python tools/conv_bench.py 32 2>&1 | grep -v amdgpu.ids | head -6 | tee $O/bench_new.txt
