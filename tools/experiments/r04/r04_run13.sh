# mixlo/mixhi operand split (6 instead of 8 vector instructions per four values): bit identity and same-box A/B against the previous build
O=gpurun_out/r04_t13; mkdir -p $O
P=$PWD/tools/experiments/lib_prev.so
AB_STEPS=8 AB_TOP=6 bash tools/ab_multi.sh r04_mix "FAVAE_X=1" "FAVAE_HIP_LIB=$P" "FAVAE_X=1" "FAVAE_HIP_LIB=$P" 2>&1 | tee $O/ab.txt
python - <<PY
import json
for i in (1,2,3,4):
    d=json.load(open("gpurun_out/r04_mix/%d.json"%i)); print(i, d["config"]["loss_g_last"], d["ms_per_step"])
PY
echo "--- conv_bench new"; python tools/conv_bench.py 32 2>&1 | grep -v amdgpu.ids > $O/conv_new.txt; head -40 $O/conv_new.txt
echo "--- conv_bench prev"; FAVAE_HIP_LIB=$P python tools/conv_bench.py 32 2>&1 | grep -v amdgpu.ids > $O/conv_prev.txt; head -40 $O/conv_prev.txt
python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
