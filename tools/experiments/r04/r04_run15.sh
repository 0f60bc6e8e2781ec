O=gpurun_out/r04_t15; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -15 $O/tests.log
