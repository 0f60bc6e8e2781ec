timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -x -k "ab_switches or gn_epilogue or dy_byproducts or f44 or golden" 2>&1 | tail -4
AB_STEPS=8 AB_TOP=6 bash tools/ab_multi.sh r05_premul "FAVAE_GB_PREMUL=0" "FAVAE_GB_PREMUL=1" "FAVAE_GB_PREMUL=0" "FAVAE_GB_PREMUL=1" 2>&1 | grep -v "^    conv\|wgrad"
