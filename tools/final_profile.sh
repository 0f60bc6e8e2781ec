cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/traffic -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/traffic -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/w.log 2>&1
cd $R && python bench.py --no-cpu-baseline --precision fp16 > $O/bench_fp16.json 2>/dev/null
python bench.py --no-cpu-baseline --config ffhq_f16 --gan --lpips > $O/bench_cfg5_fp32.json 2>/dev/null
python bench.py --no-cpu-baseline --config ffhq_f16 --gan --lpips --precision fp16 > $O/bench_cfg5_fp16.json 2>/dev/null
python bench.py --no-cpu-baseline --config imagenet_f4 > $O/bench_f4.json 2>/dev/null
ls -la $O $O/kt $O/traffic | head -40
cat $O/bench.json | cut -c1-400
