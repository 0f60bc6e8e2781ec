O=gpurun_out/r04_t12; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "vq or quant or indices or golden or train_step or smoke" > $O/tests.log 2>&1; tail -3 $O/tests.log
AB_STEPS=8 AB_TOP=1 bash tools/ab_multi.sh r04_vq2 "FAVAE_VQ_H3=1" "FAVAE_VQ_H3=0" "FAVAE_VQ_H3=1" "FAVAE_VQ_H3=0" 2>&1 | grep "ms/step"
python - <<PY
import json
for i in (1,2):
    d=json.load(open("gpurun_out/r04_vq2/%d.detail.json"%i))
    print(d["line"]["config"]["loss_g_last"], [(k['kernel'][:28], round(k['avg_launch_us'],1)) for k in d['kernel_table']['kernels'] if k['kernel'].startswith('vq_dist')])
PY
