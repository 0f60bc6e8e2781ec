O=gpurun_out/r04_race; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o /tmp/memset_order tools/experiments/memset_order.hip || exit 1
echo "--- alone"; timeout 120 /tmp/memset_order 20000 0 0; timeout 120 /tmp/memset_order 20000 0 1; timeout 120 /tmp/memset_order 20000 1 1
echo "--- two processes"
timeout 200 /tmp/memset_order 40000 0 1 > $O/ms_a.log 2>&1 &
PA=$!
timeout 200 /tmp/memset_order 40000 0 1 > $O/ms_b.log 2>&1 &
PB=$!
wait $PA; wait $PB; cat $O/ms_a.log $O/ms_b.log
timeout 200 /tmp/memset_order 40000 1 1 > $O/ms_c.log 2>&1 &
PA=$!
timeout 200 /tmp/memset_order 40000 1 1 > $O/ms_d.log 2>&1 &
PB=$!
wait $PA; wait $PB; cat $O/ms_c.log $O/ms_d.log
