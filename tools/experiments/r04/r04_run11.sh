O=gpurun_out/r04_t11; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "disc or gan or lpips or thin or hinge" > $O/tests.log 2>&1; tail -3 $O/tests.log
AB_STEPS=8 AB_TOP=2 bash tools/ab_multi.sh r04_c1 "FAVAE_NOP=1" 2>&1 | head -3
python - <<PY
import json
d=json.load(open("gpurun_out/r04_c1/1.detail.json"))
print(d["line"]["config"]["loss_g_last"])
for k in d['kernel_table']['kernels']:
    if k['kernel'].startswith(('conv_cout1','conv_fwd_buf','conv_fwd_kernel','thin_out','lpips')): print(k['kernel'][:50], k['launches']//2, round(k['avg_launch_us'],1))
PY
python bench.py --no-cpu-baseline --no-extras --lpips 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lpips', round(d['value'],1), round(d['ms_per_step'],2))"
