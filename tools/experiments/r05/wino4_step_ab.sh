# step-level A/B of the F(4x4,3x3) kernel: FAVAE_WINO4=0 (off) / 1 (data gradients) / 2 (+ decoder forward at 256^2)
AB_STEPS=8 AB_TOP=6 bash tools/ab_multi.sh r05_w4ab "FAVAE_WINO4=0" "FAVAE_WINO4=1" "FAVAE_WINO4=2" "FAVAE_WINO4=0" "FAVAE_WINO4=1" "FAVAE_WINO4=2"
