# run on the GPU box from the repo root: two PMC passes (separately) + the default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/traffic
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/traffic -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traffic/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/traffic -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traffic/w.log 2>&1
ls $R/gpurun_out/traffic
