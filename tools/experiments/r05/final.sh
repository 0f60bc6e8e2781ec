# evidence of round 5, one call (GPU box): full GPU suite, kernel trace + timeline, PMC traffic (copied into profiles/ on the box so the
# bench lines that follow carry it), the 20/5 and default bench lines, other configurations, smoke
R=$GRAFT_REPO_ROOT; TAG=${1:-r05_final}; O=$R/gpurun_out/$TAG; mkdir -p $O
bash tools/r05_run.sh $TAG tests_all prof
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/traffic -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/f.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/traffic -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/w.log 2>&1
cd $R
F=$(find $O/traffic -name "f_counter_collection.csv" | head -1); W=$(find $O/traffic -name "w_counter_collection.csv" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then
  python tools/hbm_traffic.py $F $W 2 $O/hbm_traffic.json $O/traffic_by_grid.txt && cp $O/hbm_traffic.json profiles/r05_hbm_traffic.json
  cat $O/traffic_by_grid.txt
fi
rm -rf $O/traffic
FAVAE_BENCH_DETAIL=$O/bench_detail.json timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-2600 $O/bench.json
FAVAE_BENCH_DETAIL=$O/bench_default_detail.json timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
bash tools/r05_run.sh $TAG configs
timeout 600 python bench.py --no-cpu-baseline --no-extras --precision bf16 > $O/bench_bf16.json 2>/dev/null; cut -c1-160 $O/bench_bf16.json
FAVAE_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 4 > $O/bench_dist_world1.json 2>/dev/null; python -c "import json;print(json.dumps(json.loads(open('$O/bench_dist_world1.json').read().splitlines()[0])['comm']))"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
