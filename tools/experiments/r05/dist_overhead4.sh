run() { env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-comm-diag 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$*', r['ms_per_step'])"; }
run FAVAE_X=1
run FAVAE_FORCE_DIST=1 FAVAE_DIST_DEBUG=pg_only
run FAVAE_FORCE_DIST=1 FAVAE_DIST_DEBUG=pg_only GPU_MAX_HW_QUEUES=8
run FAVAE_FORCE_DIST=1 FAVAE_DIST_DEBUG=pg_only GPU_MAX_HW_QUEUES=16
run FAVAE_X=1 GPU_MAX_HW_QUEUES=8
run FAVAE_FORCE_DIST=1 GPU_MAX_HW_QUEUES=8
