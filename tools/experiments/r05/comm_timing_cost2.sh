run() { env "$@" python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-comm-diag 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$*', round(r['ms_per_step'],2))"; }
run FAVAE_FORCE_DIST=1 FAVAE_COMM_TIMING=1




