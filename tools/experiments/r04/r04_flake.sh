# reproduce the intermittent failure of test_product_trainstep_two_ranks_on_one_gpu: loop the world-2 gloo probe, keep full logs of failures
O=gpurun_out/r04_flake; mkdir -p $O
fail=0
for i in $(seq 1 ${FLAKE_N:-24}); do
  for v in "gauss_resblock 1" "gauss_resblock 0"; do
    set -- $v
    port=$((29600 + (i % 50) * 3 + $2))
    HSA_ENABLE_IPC_MODE_LEGACY=0 FAVAE_PROBE_VARIANT=$1 FAVAE_OVERLAP_COMM=$2 FAVAE_PROBE_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port tests/dist_probe.py > $O/run.out 2> $O/run.err
    rc=$?
    if [ $rc -ne 0 ]; then fail=$((fail+1)); cp $O/run.err $O/fail_${i}_$1_$2.err; cp $O/run.out $O/fail_${i}_$1_$2.out; echo "iteration $i $v: rc=$rc"; grep -n "DIAG\|AssertionError" $O/run.err | head -40; fi
  done
done
echo "failures: $fail"
