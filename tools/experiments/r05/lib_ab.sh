# step-level A/B of two builds: tools/experiments/libfavae_prev.so (a copy of an earlier commit's library) against the in-tree one
P=$GRAFT_REPO_ROOT/tools/experiments/libfavae_prev.so
AB_STEPS=8 AB_TOP=0 bash tools/ab_multi.sh ${1:-r05_libab} "FAVAE_HIP_LIB=$P" "FAVAE_AB=new" "FAVAE_HIP_LIB=$P" "FAVAE_AB=new" "FAVAE_HIP_LIB=$P" "FAVAE_AB=new"
