# step-level A/B of the fused operand split (v_fma_mixlo/mixhi_f16: 8 instead of 12 vector instructions per four values in the direct kernels,
# 6 instead of 8 in the Winograd kernels): the previous build (tools/experiments/libfavae_prev.so, a copy of commit f046d98's library) against this one
P=$GRAFT_REPO_ROOT/tools/experiments/libfavae_prev.so
AB_STEPS=8 AB_TOP=8 bash tools/ab_multi.sh r05_splitab "FAVAE_HIP_LIB=$P" "FAVAE_AB=new" "FAVAE_HIP_LIB=$P" "FAVAE_AB=new" "FAVAE_HIP_LIB=$P" "FAVAE_AB=new"
