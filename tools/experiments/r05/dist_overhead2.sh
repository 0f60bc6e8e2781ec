run() { env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-comm-diag 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$*', r['ms_per_step'])"; }
run FAVAE_X=1
run FAVAE_FORCE_DIST=1 FAVAE_DIST_DEBUG=pg_only
run FAVAE_FORCE_DIST=1 FAVAE_DIST_DEBUG=no_codebook FAVAE_OVERLAP_COMM=0
run FAVAE_FORCE_DIST=1 FAVAE_OVERLAP_COMM=0
run FAVAE_FORCE_DIST=1 FAVAE_DIST_DEBUG=pg_only NCCL_MAX_NCHANNELS=1
