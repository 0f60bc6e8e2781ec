O=gpurun_out/r04_t24; mkdir -p $O
echo "--- two training processes on one GPU, 400 repetitions each"
RACE_FRESH=0 RACE_NODEHOOKS=0 RACE_GRADHOOKS=0 timeout 900 python tools/race_probe.py a 400 > $O/a.log 2>&1 &
PA=$!
RACE_FRESH=0 RACE_NODEHOOKS=0 RACE_GRADHOOKS=0 timeout 900 python tools/race_probe.py b 400 > $O/b.log 2>&1 &
PB=$!
wait $PA; wait $PB; grep -h "done\|differs" $O/a.log $O/b.log | cut -c1-200 | head -6
echo "--- the world-2 gloo probe on one GPU, 16 x 2 variants"
FLAKE_N=16 bash tools/r04_flake.sh 2>&1 | tail -4
