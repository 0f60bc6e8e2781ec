O=gpurun_out/r05_dist; mkdir -p $O
FAVAE_BENCH_DETAIL=$O/plain.json python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/plain.line 2>/dev/null
FAVAE_FORCE_DIST=1 FAVAE_DIST_DEBUG=pg_only FAVAE_BENCH_DETAIL=$O/pg.json python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-comm-diag > $O/pg.line 2>/dev/null
python - <<PY
import json
def tab(f):
    d=json.load(open(f)); kt=d["kernel_table"]
    return kt, {k["kernel"]:(k["launches"]//2,k["avg_launch_us"],k.get("avg_launch_us_single_stream",0)) for k in kt["kernels"]}
ka,A=tab("$O/plain.json"); kb,B=tab("$O/pg.json")
print("ms/step two-stream plain %.2f pg %.2f | single-stream plain %.2f pg %.2f" % (ka["ms_per_step"],kb["ms_per_step"],ka["ms_per_step_single_stream"],kb["ms_per_step_single_stream"]))
sa=sum(n*a for n,a,s in A.values())/1e3; sb=sum(n*a for n,a,s in B.values())/1e3
ssa=sum(n*s for n,a,s in A.values())/1e3; ssb=sum(n*s for n,a,s in B.values())/1e3
print("sum of kernel times in step: plain %.1f pg %.1f ms | single-stream sums plain %.1f pg %.1f" % (sa,sb,ssa,ssb))
rows=sorted(((n*(B[k][1]-a)/1e3,k,n,a,B[k][1],s,B[k][2]) for k,(n,a,s) in A.items() if k in B),reverse=True)
for r in rows[:12]: print("%+6.2f ms  %-55s n=%3d step %7.1f -> %7.1f  excl %7.1f -> %7.1f" % (r[0],r[1][:55],r[2],r[3],r[4],r[5],r[6]))
PY
