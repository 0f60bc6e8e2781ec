# does the distributed code path cost time at world 1?  (the N >= 2 scaling runs use it, the N = 1 run does not)
for i in 1 2; do
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().splitlines()[-1]); print('plain      ', r['ms_per_step'])"
FAVAE_FORCE_DIST=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-comm-diag 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('dist world 1', r['ms_per_step'])"
FAVAE_FORCE_DIST=1 FAVAE_OVERLAP_COMM=0 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-comm-diag 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('dist world 1, one all-reduce (no marks)', r['ms_per_step'])"
done
