# K-loop start rotated per channel tile (FAVAE_WINO_ROT): tests, single-layer times, FETCH_SIZE per launch, step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_t18; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "wino or conv or golden or blocks" > $O/tests.log 2>&1; tail -3 $O/tests.log
for r in 1 0; do echo "--- conv_bench FAVAE_WINO_ROT=$r"; FAVAE_WINO_ROT=$r timeout 300 python tools/conv_bench.py 32 2>&1 | grep -v amdgpu.ids | head -6; done
cd /tmp && export TMPDIR=/tmp
for r in 1 0; do
  for shp in "128 128 256 3 32" "256 256 64 3 32" "512 512 16 3 32"; do
    tag=$(echo $shp | tr ' ' '_')
    FAVAE_WINO_ROT=$r timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f${r}_$tag -o f -- python3 $R/tools/conv_one.py $shp > $O/f${r}_$tag.log 2>&1
  done
done
cd $R
python - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob("$O/f?_*/")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "wino_sp_kernel" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[1:] or v
        print(os.path.basename(d.rstrip("/")), k, "launches", len(v), "fetch MB/launch (x2 corrected) %.1f" % (2 * 1024 * sum(v) / len(v) / 1e6))
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
AB_STEPS=8 AB_TOP=3 timeout 600 bash tools/ab_multi.sh r04_rot "FAVAE_WINO_ROT=1" "FAVAE_WINO_ROT=0" "FAVAE_WINO_ROT=1" "FAVAE_WINO_ROT=0" 2>&1 | tee $O/ab.txt
