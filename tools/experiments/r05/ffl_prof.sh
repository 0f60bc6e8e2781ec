# per-kernel times of the focal-frequency loss micro-benchmark (tools/ffl_bench.py) under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05_fflprof}; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ffl -- python3 $GRAFT_REPO_ROOT/tools/ffl_bench.py 32 > $O/bench.txt 2>&1
cat $O/bench.txt | grep FFL
F=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s n=%4s avg=%9.1f us total=%8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
