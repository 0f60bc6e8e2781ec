O=gpurun_out/r04_t2; mkdir -p $O
python -m pytest tests/test_gpu_model.py -q -s -x -k "gan_iteration or b1_index or flat_buffer" > $O/tests.log 2>&1; tail -15 $O/tests.log
python -m pytest tests/test_gpu_dist.py -q -x -k "bench_line_schema" > $O/tests2.log 2>&1; tail -5 $O/tests2.log
bash tools/ab_multi.sh r04_sched "FAVAE_NOP=1" "FAVAE_SERIALIZE_MFMA=1" 2>&1 | tee $O/ab.txt
