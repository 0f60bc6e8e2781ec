O=gpurun_out/r04_t22; mkdir -p $O
echo "--- FFL backward next to the weight gradient, same process"; FFL_INPROC=wgrad timeout 300 python tools/experiments/ffl_race2.py W 60000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
echo "--- two training processes on one GPU (race_probe, 200 repetitions each)"
RACE_FRESH=0 RACE_NODEHOOKS=0 RACE_GRADHOOKS=0 timeout 600 python tools/race_probe.py a 200 > $O/a.log 2>&1 &
PA=$!
RACE_FRESH=0 RACE_NODEHOOKS=0 RACE_GRADHOOKS=0 timeout 600 python tools/race_probe.py b 200 > $O/b.log 2>&1 &
PB=$!
wait $PA; wait $PB; grep -h "done\|differs" $O/a.log $O/b.log | cut -c1-260 | head -8
echo "--- step"; AB_STEPS=8 AB_TOP=0 timeout 300 bash tools/ab_multi.sh r04_noslp "FAVAE_X=1" 2>&1 | grep ms/step
python - <<PY
import json; print(json.load(open("gpurun_out/r04_noslp/1.json"))["config"]["loss_g_last"])
for k in json.load(open("gpurun_out/r04_noslp/1.detail.json"))["kernel_table"]["kernels"]:
    if k["kernel"].startswith("fft_"): print(k["kernel"], round(k["avg_launch_us"],1), round(k.get("frac",0),3))
PY
