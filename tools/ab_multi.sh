# usage: bash tools/ab_multi.sh <tag> "<ENV..>" "<ENV..>" ...   -- same-box comparison of bench.py under several environments
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
i=0
for E in "$@"; do
  i=$((i+1))
  env $E FAVAE_BENCH_DETAIL=$O/$i.detail.json python bench.py --steps ${AB_STEPS:-6} --warmup 2 --no-cpu-baseline ${AB_ARGS:-} > $O/$i.json 2> $O/$i.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$i.json"))
except Exception as e:
    print("$i [$E]: FAILED", e); raise SystemExit
print("$i [$E]: %.2f ms/step  %.1f img/s  (single-stream %s)" % (d["ms_per_step"], d["value"], d.get("ms_per_step_single_stream")))
try:
    kt=json.load(open("$O/$i.detail.json")).get("kernel_table")
except Exception:
    kt=None
if kt:
    for k in kt["kernels"][:${AB_TOP:-8}]:
        print("    %-52s n=%4d avg=%8.1f ss=%8.1f" % (k["kernel"][:52],k["launches"]//2,k["avg_launch_us"],k.get("avg_launch_us_single_stream",0)))
PY
done
