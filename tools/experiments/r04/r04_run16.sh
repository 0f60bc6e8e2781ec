O=gpurun_out/r04_t16; mkdir -p $O
python - <<PY
import torch
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,"priority_range") else None)
for p in (-2,-1,0,1,2):
    s=torch.cuda.Stream(priority=p); print(p, s.priority)
PY
AB_STEPS=8 AB_TOP=5 timeout 900 bash tools/ab_multi.sh r04_prio "FAVAE_SIDE_PRIORITY=0" "FAVAE_SIDE_PRIORITY=1" "FAVAE_SIDE_PRIORITY=-1" "FAVAE_SIDE_PRIORITY=0" "FAVAE_SIDE_PRIORITY=1" 2>&1 | tee $O/ab.txt
