O=gpurun_out/r04_race; mkdir -p $O
run2() {   # two processes at once: $1 tag, rest = env
  tag=$1; shift
  env "$@" timeout 900 python tools/race_probe.py ${tag}a ${RACE_REPS:-120} > $O/${tag}a.log 2>&1 &
  PA=$!
  env "$@" timeout 900 python tools/race_probe.py ${tag}b ${RACE_REPS:-120} > $O/${tag}b.log 2>&1 &
  PB=$!
  wait $PA; wait $PB
  echo "=== $tag [$@]"; grep -h "differs\|done\|Error\|FFL site" $O/${tag}a.log $O/${tag}b.log | cut -c1-900 | head -${RACE_SHOW:-24}
}
run2 base RACE_FRESH=0
