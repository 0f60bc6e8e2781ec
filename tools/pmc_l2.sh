# L2 / L1 counters of the conv kernels on one layer shape. usage (GPU box): bash tools/pmc_l2.sh [out dir under gpurun_out]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_l2}; mkdir -p $O
timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/p1 -- python3 $R/tools/conv_one.py 128 128 256 3 32 > $O/log1.txt 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr --output-format csv -d $O/p2 -- python3 $R/tools/conv_one.py 128 128 256 3 32 > $O/log2.txt 2>&1
cd $R
python - <<PY
import csv,glob,collections,re
csv.field_size_limit(1<<30)
v=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"^void ","",k).split("(")[0][:60]
        if "wino" in k or "nine" in k or "halo" in k:
            v[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in v:
    print(k)
    for c,x in sorted(v[k].items()):
        print("   %-34s %.4g"%(c,sum(x)/len(x)))
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
