O=gpurun_out/r04_race; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w tools/experiments/mfma_power.hip -o /tmp/mfma_power || exit 1
echo "--- FFL backward next to an MFMA power loop of ANOTHER program"
FFL_WAIT=10 timeout 300 python tools/experiments/ffl_race2.py A 400000 > $O/v2_a.log 2>&1 &
PA=$!
sleep 13
( for i in $(seq 1 400); do /tmp/mfma_power 40000 > /dev/null 2>&1; done ) &
PB=$!
wait $PA; kill $PB 2>/dev/null; wait $PB 2>/dev/null
date +%s.%N; grep -v amdgpu.ids $O/v2_a.log | cut -c1-300 | tail -4
echo "--- FFL backward next to a second copy of ITSELF"
FFL_WAIT=14 timeout 300 python tools/experiments/ffl_race2.py A 400000 > $O/v2_b.log 2>&1 &
PA=$!
sleep 12
FFL_WAIT=0 timeout 300 python tools/experiments/ffl_race2.py B 700000 > $O/v2_c.log 2>&1 &
PB=$!
wait $PA; wait $PB
grep -v amdgpu.ids $O/v2_b.log | cut -c1-300 | tail -3; grep -v amdgpu.ids $O/v2_c.log | cut -c1-300 | tail -2
