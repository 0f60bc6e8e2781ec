"""times favae_gn_act_bwd on a few (N, C, H, W, G) shapes and prints the launch profiler's per-kernel breakdown"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "fa-vae_amd"))
import torch, favae_hip as H
from favae_hip import ops as K
lib = H.load()
dev = torch.device("cuda:0")
for (N, C, Hh, W, G) in [(16, 128, 256, 256, 32), (16, 256, 128, 128, 32), (16, 512, 64, 64, 32), (32, 128, 256, 256, 32)]:
    x = torch.randn(N, C, Hh, W, device=dev).contiguous(memory_format=torch.channels_last)
    da = torch.randn_like(x)
    gw, gb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mean, rstd, scale, shift = K.gn_stats(x, gw, gb, G)
    dx = torch.empty_like(x)
    dgw, dgb = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws = H.workspace(H.query("favae_gn_workspace", N, Hh * W, C), dev)
    def run():
        H.call("favae_gn_act_bwd", H.ptr(da), H.ptr(x), H.ptr(gw), H.ptr(gb), H.ptr(mean), H.ptr(rstd), N, Hh * W, C, G, 1, None, H.ptr(dx),
               H.ptr(dgw), H.ptr(dgb), 0, H.ptr(ws), ws.numel())
    run(); torch.cuda.synchronize()
    lib.favae_prof_reset(); lib.favae_prof_enable(2)
    for _ in range(5): run()
    torch.cuda.synchronize()
    lib.favae_prof_enable(0)
    n = lib.favae_prof_report(None, 0); buf = ctypes.create_string_buffer(int(n) + 8); lib.favae_prof_report(buf, len(buf)); lib.favae_prof_reset()
    print((N, C, Hh, W, G), " | ".join("%s %.1f us" % (l.split("\t")[0][:28], float(l.split("\t")[2]) / int(l.split("\t")[1])) for l in buf.value.decode().splitlines()))
