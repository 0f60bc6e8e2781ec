O=gpurun_out/r04_t21; mkdir -p $O
echo "--- two training processes on one GPU (race_probe, 150 repetitions each)"
RACE_FRESH=0 RACE_NODEHOOKS=0 RACE_GRADHOOKS=0 timeout 600 python tools/race_probe.py a 150 > $O/a.log 2>&1 &
PA=$!
RACE_FRESH=0 RACE_NODEHOOKS=0 RACE_GRADHOOKS=0 timeout 600 python tools/race_probe.py b 150 > $O/b.log 2>&1 &
PB=$!
wait $PA; wait $PB; grep -h "done\|differs" $O/a.log $O/b.log | cut -c1-200 | head -8
echo "--- step time"; AB_STEPS=8 AB_TOP=0 timeout 300 bash tools/ab_multi.sh r04_wide "FAVAE_X=1" "FAVAE_X=2" 2>&1 | grep ms/step
python - <<PY
import json; print([json.load(open("gpurun_out/r04_wide/%d.json"%i))["config"]["loss_g_last"] for i in (1,2)])
for k in json.load(open("gpurun_out/r04_wide/1.detail.json"))["kernel_table"]["kernels"]:
    if k["kernel"].startswith("fft_"): print(k["kernel"], round(k["avg_launch_us"],1), round(k.get("frac",0),3))
PY
