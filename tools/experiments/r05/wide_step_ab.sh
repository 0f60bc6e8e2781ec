# step-level A/B of the 16 x 8 x 128 tiling of the F(2x2) kernel, and of F(4x4) data gradients next to it
AB_STEPS=8 AB_TOP=6 bash tools/ab_multi.sh r05_wideab "FAVAE_WINO_WIDE=0 FAVAE_WINO4=1" "FAVAE_WINO_WIDE=1 FAVAE_WINO4=1" "FAVAE_WINO_WIDE=1 FAVAE_WINO4=0" "FAVAE_WINO_WIDE=0 FAVAE_WINO4=1" "FAVAE_WINO_WIDE=1 FAVAE_WINO4=1" "FAVAE_WINO_WIDE=1 FAVAE_WINO4=0"
