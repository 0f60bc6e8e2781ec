# final evidence of round 4 (GPU box): full GPU suite, smoke, default bench line, kernel trace + timeline, PMC traffic, other configurations
bash tools/r04_run.sh r04_final tests_all prof traffic bench_default configs
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-extras --precision bf16 > gpurun_out/r04_final/bench_bf16.json 2>/dev/null; cut -c1-160 gpurun_out/r04_final/bench_bf16.json
FAVAE_BENCH_DETAIL=gpurun_out/r04_final/bench_detail.json python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_final/bench20.json 2>/dev/null; cut -c1-700 gpurun_out/r04_final/bench20.json
