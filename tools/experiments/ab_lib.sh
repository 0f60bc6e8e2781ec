# same-box A/B of two builds of libfavae_hip.so: tools/experiments/old.so vs the in-tree build
L=fa-vae_amd/favae_hip/libfavae_hip.so
cp $L /tmp/new.so
for rep in 1 2; do
for which in old new; do
  if [ $which = old ]; then cp tools/experiments/old.so $L; else cp /tmp/new.so $L; fi
  python bench.py --no-cpu-baseline --steps 6 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$which', round(r['value'],2), round(r['ms_per_step'],2))
"
done
done
cp /tmp/new.so $L
