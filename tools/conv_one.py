#!/usr/bin/env python3
"""Launch the three conv kernels a few times on one layer shape (for PMC collection). usage: conv_one.py cin cout hw k batch"""
import os, sys
from ctypes import byref
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import torch
import favae_hip as H
from favae_hip import ops as K
cin, cout, hw, k, B = [int(v) for v in sys.argv[1:6]]
dev = torch.device("cuda:0")
x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.zeros(cout, device=dev)
gw, gb = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
mean, rstd, scale, shift, xb = K.gn_stats(x, gw, gb, 32, with_bound=True)
y = K.new_cl(B, cout, hw, hw, dev)
d = H.make_conv_desc(B, hw, hw, cin, hw, hw, cout, k, k, 1, k // 2, 0, H.ACT_SILU, 1)
wt = torch.empty(cin, k, k, cout, device=dev)
H.call("favae_weight_flip", H.ptr(w), H.ptr(wt), cout, k, k, cin)
d2 = H.make_conv_desc(B, hw, hw, cout, hw, hw, cin, k, k, 1, k // 2, 0, 0, 1)
dx = K.new_cl(B, cin, hw, hw, dev)
dw = torch.empty(cout, k, k, cin, device=dev)
ws = H.workspace(H.query("favae_conv_wgrad_workspace", byref(d)), dev)
# CONV_ONE_F44=1: the forward and the data gradient (launched as a forward conv on the flipped weights) on the F(4x4, 3x3) kernel too
f44 = os.environ.get("CONV_ONE_F44") == "1"
K._conv_launch(d, x, w, b, None, scale, shift, y, xb)
yb = K.absmax(y)
for _ in range(3):
    K._conv_launch(d, x, w, b, None, scale, shift, y, xb)
    K._conv_launch(d2, y, wt, None, None, None, None, dx, yb)
    if f44:
        prev = K.set_wino4("2")
        with K.wino4_forward(True):
            K._conv_launch(d, x, w, b, None, scale, shift, y, xb)
            K._conv_launch(d2, y, wt, None, None, None, None, dx, yb)
        K.set_wino4(prev)
    H.call("favae_conv_wgrad", byref(d), H.ptr(x), H.ptr(y), H.ptr(scale), H.ptr(shift), H.ptr(xb), H.ptr(yb), H.ptr(dw), 0, H.ptr(ws), ws.numel())
torch.cuda.synchronize()
