O=gpurun_out/r04_t8; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "lpips or conv or wino or gan_lpips" > $O/tests.log 2>&1; tail -4 $O/tests.log
python bench.py --no-cpu-baseline --no-extras --lpips > $O/bench_lpips.json 2>/dev/null
python bench.py --no-cpu-baseline --no-extras --config ffhq_f16 --gan --lpips > $O/bench_cfg5_fp32.json 2>/dev/null
for f in lpips cfg5_fp32; do python - <<PY
import json
d=json.load(open("$O/bench_$f.json")); print("$f: %.1f img/s %.2f ms" % (d["value"], d["ms_per_step"]))
PY
done
