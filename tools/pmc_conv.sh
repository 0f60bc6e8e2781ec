# PMC passes over the three dominant conv kernels on one layer shape (tools/conv_one.py); separate rocprofv3 --pmc runs, kernel trace only.
# usage (GPU box): bash tools/pmc_conv.sh [out dir under gpurun_out] [cin cout hw k batch]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_r03}; shift
ARGS=${@:-128 128 256 3 32}
mkdir -p $O
for i in 1 2 3 4; do
case $i in
1) C="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES";;
2) C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU";;
3) C="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS";;
4) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC";;
esac
timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -- python3 $R/tools/conv_one.py $ARGS > $O/log$i.txt 2>&1
done
cd $R
python tools/pmc_summary.py $O > $O/summary.md; cat $O/summary.md
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
