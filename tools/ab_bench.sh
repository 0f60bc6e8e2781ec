# usage: bash tools/ab_bench.sh <tag> "<ENV=.. for A>" "<ENV=.. for B>" [bench args]   -- same-box A/B of bench.py (quick form)
TAG=$1; A="$2"; B="$3"; shift 3
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
for arm in A B; do
  if [ $arm = A ]; then E="$A"; else E="$B"; fi
  env $E FAVAE_BENCH_DETAIL=$O/$arm.detail.json python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $O/$arm.json 2> $O/$arm.err
  python - <<PY
import json
d=json.load(open("$O/$arm.json"))
print("$arm [$E]: %.2f ms/step  %.1f img/s  (single-stream %s)" % (d["ms_per_step"], d["value"], d.get("ms_per_step_single_stream")))
try:
    kt=json.load(open("$O/$arm.detail.json")).get("kernel_table")
except Exception:
    kt=None
if kt:
    for k in kt["kernels"][:10]:
        print("    %-52s n=%4d avg=%8.1f ss=%8.1f" % (k["kernel"][:52],k["launches"]//2,k["avg_launch_us"],k.get("avg_launch_us_single_stream",0)))
PY
done
