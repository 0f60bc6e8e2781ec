# VQ near-tie re-score restricted to the qualifying 128-code tiles: tests, then rocprof kernel trace of the first 3 steps
O=gpurun_out/r04_t14; mkdir -p $O; rm -rf $O/prof
timeout 600 python -m pytest tests -m gpu -q -x -k "vq or quant or indices or golden or train_step or smoke" > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o vq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 120 python tools/rocpd_stats.py $O/prof/vq_results.db 2>&1 | grep -E "vq_|l2norm" | cut -c1-150
AB_STEPS=20 AB_TOP=0 timeout 300 bash tools/ab_multi.sh r04_vq3 "FAVAE_X=1" 2>&1 | grep ms/step
