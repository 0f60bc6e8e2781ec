O=gpurun_out/r04_race; mkdir -p $O
run() {  # $1 tag, $2 lib ("" = product)
  if [ -n "$2" ]; then export FAVAE_HIP_LIB=$2; else unset FAVAE_HIP_LIB; fi
  FFL_WAIT=${FFL_WAIT:-20} timeout 600 python tools/experiments/ffl_race2.py $1 ${FFL_REPS:-200000} > $O/ffl_$1.log 2>&1 &
  PA=$!
  sleep 8
  ( unset FAVAE_HIP_LIB; RACE_FRESH=0 RACE_NODEHOOKS=0 RACE_GRADHOOKS=0 timeout 600 python tools/race_probe.py bg 260 > $O/ffl_bg_$1.log 2>&1 ) &
  PB=$!
  wait $PA; wait $PB
  grep -v amdgpu.ids $O/ffl_$1.log | tail -1 | cut -c1-300; grep "done" $O/ffl_bg_$1.log
}
echo "--- product FFT next to a training process"; run prod ""
echo "--- FFT with a wait state behind its 64-bit stores"; run nop $PWD/tools/experiments/lib_fftnop.so
