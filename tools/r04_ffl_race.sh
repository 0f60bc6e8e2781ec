O=gpurun_out/r04_race; mkdir -p $O
FFL_WAIT=${FFL_WAIT:-22} timeout 600 python tools/experiments/ffl_race2.py A ${FFL_REPS:-30000} > $O/ffl_a.log 2>&1 &
PA=$!
sleep 8
RACE_FRESH=0 RACE_NODEHOOKS=0 RACE_GRADHOOKS=0 timeout 600 python tools/race_probe.py bg 300 > $O/ffl_bg.log 2>&1 &
PB=$!
wait $PA; wait $PB
grep -v amdgpu.ids $O/ffl_a.log | cut -c1-1100 | head -30; grep "done" $O/ffl_bg.log
