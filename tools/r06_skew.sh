# usage (GPU box): bash tools/r06_skew.sh <tag> -- Winograd kernel with out-of-phase wave groups (FAVAE_WINO_SKEW) against the lock-step loop
TAG=${1:-sk0}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "winograd" > $O/wino_tests.log 2>&1; tail -3 $O/wino_tests.log
for sk in 0 1 0 1; do
  echo "== FAVAE_WINO_SKEW=$sk"
  FAVAE_WINO_SKEW=$sk timeout 300 python tools/wino_wide_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_skew$sk.txt
done
