O=gpurun_out/r04_last; mkdir -p $O
python -m pytest tests/test_gpu_dist.py -q -x > $O/dist.log 2>&1; tail -3 $O/dist.log
FAVAE_BENCH_DETAIL=$O/bench_detail.json python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-2300 $O/bench.json
FAVAE_BENCH_DETAIL=$O/bench_default_detail.json python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
