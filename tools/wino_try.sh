# usage (GPU box): bash tools/wino_try.sh <tag> [prec|dbg] -- Winograd conv kernel: layer micro-benchmark (precision table with "prec", ablations with "dbg")
TAG=${1:-w0}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
if [ "$2" = "prec" ]; then
FAVAE_WINO=1 timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "precision_is_fp32_grade" -s > $O/prec_on.log 2>&1; grep "h3\|passed\|failed\|Error" $O/prec_on.log | cut -c1-200
fi
FAVAE_WINO=1 timeout 300 python tools/conv_bench.py > $O/cb_on.log 2>&1; grep "k3" $O/cb_on.log | cut -c1-200
if [ "$2" = "dbg" ]; then
for d in ${WDBG:-1 2 3 4}; do echo "DBG $d"; FAVAE_WINO_DBG=$d timeout 300 python tools/conv_bench.py 2>&1 | grep "k3" | cut -c1-60,120-200; done
fi
