O=gpurun_out/r04_t19; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -6 $O/tests.log
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-330
