O=gpurun_out/r04_t10; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "vq or quant or indices or golden" > $O/tests.log 2>&1; tail -3 $O/tests.log
AB_STEPS=8 AB_TOP=3 bash tools/ab_multi.sh r04_vq "FAVAE_NOP=1" 2>&1 | head -3
python - <<PY
import json
d=json.load(open("gpurun_out/r04_vq/1.detail.json"))
for k in d['kernel_table']['kernels']:
    if k['kernel'].startswith(('vq_','thin_out')): print(k['kernel'][:50], k['launches']//2, round(k['avg_launch_us'],1))
PY
