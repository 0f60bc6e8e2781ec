# usage (on the GPU box, through gpurun): bash tools/r06_run.sh <tag> [tests|bench|prof|traffic|configs|bf16detail ...] -> gpurun_out/<tag>/
TAG=${1:-run}; shift
WHAT=${@:-tests bench}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for w in $WHAT; do
case $w in
tests)
  rm -f $R/gpurun_out/parity_margins.txt
  FAVAE_PARITY_MARGINS=$O/parity_margins.txt timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log ;;
tests_all)
  FAVAE_PARITY_MARGINS=$O/parity_margins.txt timeout 1800 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -60 $O/pytest.log ;;
bench)
  FAVAE_BENCH_DETAIL=$O/bench_detail.json timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; wc -c $O/bench.json; cat $O/bench.json
  python tools/kernel_table.py $O/bench_detail.json | head -40 ;;
bench_default)
  FAVAE_BENCH_DETAIL=$O/bench_default_detail.json timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err; wc -c $O/bench_default.json; cat $O/bench_default.json ;;
bench_quick)
  FAVAE_BENCH_DETAIL=$O/bench_quick_detail.json timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc=$?"; tail -3 $O/bench_quick.err; cat $O/bench_quick.json
  python tools/kernel_table.py $O/bench_quick_detail.json | head -30 ;;
bf16detail)
  FAVAE_BENCH_DETAIL=$O/bench_bf16_detail.json timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench rc=$?"; cut -c1-300 $O/bench_bf16.json
  python tools/kernel_table.py $O/bench_bf16_detail.json | head -45
  FAVAE_BENCH_DETAIL=$O/bench_cfg5_bf16_detail.json timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --config ffhq_f16 --gan --lpips --precision bf16 > $O/bench_cfg5_bf16.json 2> $O/bench_cfg5_bf16.err; echo "bench rc=$?"; cut -c1-300 $O/bench_cfg5_bf16.json
  python tools/kernel_table.py $O/bench_cfg5_bf16_detail.json | head -45 ;;
prof)
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
  cd $R
  DB=$(ls $O/kt/*/*.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(ls $O/kt/*.db | head -1)
  python tools/rocpd_stats.py $DB $O/kernel_stats.csv; head -25 $O/kernel_stats.csv
  python tools/timeline.py $DB > $O/timeline.txt 2>&1; tail -30 $O/timeline.txt
  rm -rf $O/kt ;;
traffic)
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/traffic -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/f.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/traffic -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/w.log 2>&1
  cd $R; ls $O/traffic | head
  F=$(find $O/traffic -name "f_counter_collection.csv" | head -1); W=$(find $O/traffic -name "w_counter_collection.csv" | head -1)
  python tools/hbm_traffic.py $F $W 2 $O/hbm_traffic.json ;;
configs)
  python bench.py --no-cpu-baseline --no-extras --precision fp16 > $O/bench_fp16.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-extras --precision bf16 > $O/bench_bf16.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-extras --config ffhq_f16 --gan --lpips > $O/bench_cfg5_fp32.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-extras --config ffhq_f16 --gan --lpips --precision fp16 > $O/bench_cfg5_fp16.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-extras --config ffhq_f16 --gan --lpips --precision bf16 > $O/bench_cfg5_bf16.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-extras --config imagenet_f4 > $O/bench_f4.json 2>/dev/null
  for f in fp16 bf16 cfg5_fp32 cfg5_fp16 cfg5_bf16 f4; do cut -c1-240 $O/bench_$f.json; echo; done ;;
*) echo "unknown step $w" ;;
esac
done
