O=gpurun_out/r04_race; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o /tmp/lds_victim tools/experiments/lds_victim.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o /tmp/memset_order tools/experiments/memset_order.hip || exit 1
echo "--- minimal victim alone"; timeout 120 /tmp/lds_victim 30000 0
echo "--- minimal victim (LDS) and (no LDS) next to a training process"
RACE_FRESH=0 RACE_NODEHOOKS=0 RACE_GRADHOOKS=0 timeout 600 python tools/race_probe.py bg 400 > $O/v_bg.log 2>&1 &
PB=$!
sleep 14
timeout 120 /tmp/lds_victim 150000 0
timeout 120 /tmp/lds_victim 150000 1
wait $PB; grep done $O/v_bg.log
echo "--- FFL backward (product kernels) next to a process that only runs simple kernels (no LDS)"
timeout 300 /tmp/memset_order 600000 1 1 > $O/v_ms.log 2>&1 &
PB=$!
sleep 2
timeout 300 python tools/experiments/ffl_race2.py A 60000 2>&1 | grep -v amdgpu.ids | tail -2
wait $PB
