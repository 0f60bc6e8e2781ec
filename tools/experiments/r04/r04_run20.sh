O=gpurun_out/r04_t20; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
python tools/experiments/apply_race.py 200 wgrad 2>&1 | grep -v amdgpu.ids | tail -2
AB_STEPS=8 AB_TOP=0 timeout 300 bash tools/ab_multi.sh r04_nop "FAVAE_X=1" 2>&1 | grep ms/step
python - <<PY
import json; d=json.load(open("gpurun_out/r04_nop/1.json")); print(d["config"]["loss_g_last"])
PY
