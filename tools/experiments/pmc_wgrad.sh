cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcw
for v in 0 1; do
export FAVAE_WGRAD_XCD=$v
timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmcw/x$v -- python3 $R/tools/conv_one.py 128 128 256 3 32 > $R/gpurun_out/pmcw/x$v.log 2>&1
done
