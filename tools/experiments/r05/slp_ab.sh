# library-wide -fno-slp-vectorize A/B on one box (A = product build: SLP everywhere but ffl.hip; B = no unit with SLP).
# Step time, kernel table, loss_g_last of both arms (ABAB), then the co-run victim test at 1000 repetitions on arm B.
O=gpurun_out/r05_slp; mkdir -p $O
L=tools/experiments/lib_noslp.so
for rep in 1 2; do
  for arm in A B; do
    if [ $arm = A ]; then E="FAVAE_X=0"; else E="FAVAE_HIP_LIB=$PWD/$L"; fi
    env $E FAVAE_BENCH_DETAIL=$O/$arm$rep.detail.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/$arm$rep.json 2> $O/$arm$rep.err
    python - <<PY
import json
d=json.load(open("$O/$arm$rep.json"))
print("$arm$rep: %.2f ms/step  %.1f img/s  loss_g_last %r" % (d["ms_per_step"], d["value"], d["config"].get("loss_g_last")))
kt=json.load(open("$O/$arm$rep.detail.json")).get("kernel_table")
if kt:
    for k in kt["kernels"][:40]:
        print("    %-64s n=%4d avg=%8.1f ss=%8.1f" % (k["kernel"][:64],k["launches"]//2,k["avg_launch_us"],k.get("avg_launch_us_single_stream",0)))
PY
  done
done > $O/summary.txt 2>&1
FAVAE_HIP_LIB=$PWD/$L FAVAE_CORUN_REPS=1000 timeout 1500 python -m pytest tests/test_gpu_corun.py -q -x > $O/corun_noslp.log 2>&1; tail -3 $O/corun_noslp.log >> $O/summary.txt
cat $O/summary.txt | grep -v "^    "
