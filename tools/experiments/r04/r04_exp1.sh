# round-4 first probe: do specialised MFMA / vector-ALU waves overlap on a SIMD?  + per-shape conv baseline
O=gpurun_out/r04_e1; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/roles tools/experiments/mfma_valu_roles.hip && /tmp/roles > $O/roles.txt 2>&1
/tmp/roles >> $O/roles.txt 2>&1
cat $O/roles.txt
python tools/conv_bench.py 32 > $O/conv_bench.txt 2>&1
cat $O/conv_bench.txt
