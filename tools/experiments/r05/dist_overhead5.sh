run() { env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-comm-diag 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$*', r['ms_per_step'])"; }
run FAVAE_X=1
run FAVAE_FORCE_DIST=1
run FAVAE_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4
run FAVAE_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4 FAVAE_SIDE_PROBE=0
run FAVAE_X=1 GPU_MAX_HW_QUEUES=4
python - <<PY
import os, sys
sys.path.insert(0, "fa-vae_amd")
os.environ["GPU_MAX_HW_QUEUES"] = "4"
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from favae_hip import ops as K
st = [torch.cuda.Stream() for _ in range(8)]
print("overlap of 8 fresh streams with the current stream (4 hardware queues, process group initialised):", [K.streams_overlap(torch.cuda.current_stream(), s) for s in st])
print("picked:", K._side_stream() is not None, K._SIDE["probe"])
dist.destroy_process_group()
PY
