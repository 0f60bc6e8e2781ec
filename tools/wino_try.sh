# usage (GPU box): bash tools/wino_try.sh <tag> [prec] -- conv layer micro-benchmark with the Winograd kernel on and off (precision table with "prec")
TAG=${1:-w0}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
if [ "$2" = "prec" ]; then
FAVAE_WINO=1 timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "precision_is_fp32_grade or winograd" -s > $O/prec_on.log 2>&1; grep "h3\|passed\|failed\|Error" $O/prec_on.log | cut -c1-200
fi
FAVAE_WINO=1 timeout 300 python tools/conv_bench.py > $O/cb_on.log 2>&1; grep "k3\|k1" $O/cb_on.log | cut -c1-200
FAVAE_WINO=0 timeout 300 python tools/conv_bench.py > $O/cb_off.log 2>&1; grep "k3" $O/cb_off.log | cut -c1-200
