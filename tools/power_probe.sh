# samples rocm-smi power / clocks while the conv micro-benchmark loops (random vs all-zero operands): evidence for the power wall
cd $GRAFT_REPO_ROOT
for mode in 0 1; do
  (for i in 1 2 3; do CONV_BENCH_ZEROS=$mode python tools/conv_bench.py > /dev/null 2>&1; done) &
  PID=$!
  sleep 12
  echo "== CONV_BENCH_ZEROS=$mode"
  for k in 1 2 3 4; do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "sclk|Power|Max Graphics" | tr '\n' ' '; echo; sleep 1.5; done
  wait $PID
done
