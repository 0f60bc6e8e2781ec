O=gpurun_out/r04_t5; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -6 $O/tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.load(open("$O/bench.json")); print("bench: %.2f ms/step %.1f img/s" % (d["ms_per_step"], d["value"]))
PY
