"""decode the F(4x4,3x3) weight records (csrc/conv_wino4.h wino4_weights_body) and compare hi + lo with G g G^T in fp64"""
import os, sys, math, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "fa-vae_amd"))
import favae_hip as H
from favae_hip import ops as K
d = torch.device("cuda:0")
torch.manual_seed(0)
co, ci = 128, 64
w = (torch.randn(co, ci, 3, 3) * 0.03).to(d).contiguous(memory_format=torch.channels_last)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
for flip in (2, 3):
    amax = K.absmax(w)
    rec = K._wino_records(w, co, ci, flip, amax)
    torch.cuda.synchronize()
    raw = rec.cpu().numpy()
    S = 2.0 ** (14 - math.floor(math.log2(float(amax))))
    body = raw[256:].view(np.float16).astype(np.float64)
    wd = w.cpu().double()
    if flip & 1:
        g = wd.flip(2, 3).permute(1, 0, 2, 3)      # [o = ci][i = co][kh][kw], taps flipped
    else:
        g = wd
    O, I = g.shape[0], g.shape[1]
    U = torch.einsum("ak,oikl,bl->oiab", G, g, G) * S        # [o][i][a][b]
    KC = I // 16
    # layout: [o/64][i/16][pos][cb][plane][lane][8]
    arr = body.reshape(O // 64, KC, 36, 2, 2, 64, 8)
    hi, lo = arr[:, :, :, :, 0], arr[:, :, :, :, 1]          # [ct][kc][pos][cb][lane][8]
    got = np.zeros((O, I, 36))
    gothi = np.zeros((O, I, 36))
    for ct in range(O // 64):
        for kc in range(KC):
            for cb in range(2):
                for ln in range(64):
                    o = ct * 64 + cb * 32 + (ln & 31)
                    i0 = kc * 16 + 8 * (ln >> 5)
                    got[o, i0:i0 + 8, :] = (hi[ct, kc, :, cb, ln, :] + lo[ct, kc, :, cb, ln, :]).T
                    gothi[o, i0:i0 + 8, :] = hi[ct, kc, :, cb, ln, :].T
    ref = U.reshape(O, I, 36).numpy()
    e = np.abs(got - ref)
    eh = np.abs(gothi - ref)
    print("flip", flip, "max|U| %.1f  err(hi+lo) max %.3e rms %.3e  | err(hi only) max %.3e rms %.3e | rel rms (hi+lo) %.2e" % (
        np.abs(ref).max(), e.max(), np.sqrt((e ** 2).mean()), eh.max(), np.sqrt((eh ** 2).mean()), np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean())))
    pos_err = e.reshape(-1, 36).max(0).reshape(6, 6)
    print(" max err by position (a rows, b cols):\n", np.array2string(pos_err, precision=3))
    bad = np.argwhere(e > 0.01)
    print(" outliers:", len(bad), "of", e.size)
    for (o, i, p) in bad[:12]:
        ct, cb, kc = o // 64, (o >> 5) & 1, i // 16
        ln = (o & 31) + 32 * ((i // 8) & 1)
        print("   o %3d i %3d pos (%d,%d): ref %.6f  hi %.6f lo %.6f  hi+lo-ref %.6f" % (o, i, p // 6, p % 6, ref[o, i, p], hi[ct, kc, p, cb, ln, i % 8], lo[ct, kc, p, cb, ln, i % 8], got[o, i, p] - ref[o, i, p]))
