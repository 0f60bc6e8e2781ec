O=gpurun_out/r04_t9; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -x -k "thin or conv or block" > $O/tests.log 2>&1; tail -3 $O/tests.log
python tools/thin_bench.py 32 2>&1 | grep -v amdgpu.ids | tee $O/thin_new.txt
