# the comm diagnostics add a communication stream + 4 watch streams: do 8 hardware queues still hold the two compute streams apart?
for q in 8 16 32; do
GPU_MAX_HW_QUEUES=$q FAVAE_FORCE_DIST=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=r['comm']['arms']
print('queues $q: timed %.2f ms | comm arms: defer %.2f eager %.2f' % (r['ms_per_step'], c['defer']['ms_per_step'], c['eager']['ms_per_step']))"
done
