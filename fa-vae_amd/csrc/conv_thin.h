// 3x3 stride-1 convolutions with a THIN channel side (<= 4 channels: the RGB ends of the codec -- conv_in, conv_out and
// their gradients; models/codec.py:140,175).  These layers are HBM-bound (one side is a 128-channel 256x256 tensor, the
// arithmetic is 27 multiply-adds per wide element), so they do not go to the matrix pipe: the implicit-GEMM kernels pad the
// thin side to a 32-wide MFMA tile and re-read the wide tensor once per filter tap (9.3 GB of fabric traffic for a 1.07 GB
// tensor, profiles/r01_hbm_traffic.json).  Here every wide element is read once (~3x through L2 for the row halo), lanes run
// along the wide channel dimension (4 channels = one float4 per lane, 128-byte..512-byte coalesced segments), the thin side
// lives in registers, and all arithmetic is fp32 FMA on the vector ALUs.
//
//   thin_in_fwd    y[p][co4]   = bias + sum_{tap,ci} x[p+tap][ci] w[co][tap][ci]                  Cin  <= 4 (also: dgrad of a thin-out conv)
//   thin_in_wgrad  dw[co4][tap][ci] = sum_p dy[p][co4] x[p+tap][ci]
//   thin_out_fwd   y[p][co]    = bias + sum_lanes sum_tap <T(x)[p+tap][c4], w[co][tap][c4]>     Cout <= 4, T = fused GroupNorm/SiLU
//   thin_out_wgrad dw[co][tap][c4] = sum_p dy[p][co] T(x)[p+tap][c4]
// The thin-out kernels march along an image row keeping the 3x3 window of transformed float4s in registers (3 new loads per
// pixel).  Weight gradients: grid-stride workgroups, one partial slab per workgroup (waves folded through LDS in a fixed
// order), slabs summed in a fixed order by reduce_slabs_kernel -> deterministic.
// Preconditions (dispatcher): KH = KW = 3, stride 1, pad 1, plain gather; thin side = 3 channels, wide side 64 or 128
// channels, 16-byte aligned activations (parameters are read with scalar loads: flat-buffer views may be 4-byte aligned).
#pragma once

struct ThinArgs {
    const float* x;       // conv input (thin_in: N,H,W,CT ; thin_out: N,H,W,Cw)
    const float* w;       // OHWI weights
    const float* bias;
    const float* resid;
    const float* scale;   // thin_out: fused input transform (per image, per wide channel)
    const float* shift;
    const float* dy;      // wgrad: output gradient
    float* y;             // forward output
    float* part;          // wgrad: partial slabs [slab][Cout][9][Cin]
    int N, H, W, Cw;      // Cw = wide channel count
    int aff_stride;       // Cw if the affine is per image, 0 otherwise
    int xb;               // pixels of one row per workgroup
    int act;              // FAVAE_ACT_* of the fused input transform (thin_out)
};


// pixel-group sum of thin_out_kernel: common.h group_sum_xor (DPP butterfly, bit-identical to the __shfl_xor loop of rounds 1-3)
__device__ __forceinline__ float thin_group_sum(float acc, int QW) { return QW == 32 ? group_sum_xor<32>(acc) : group_sum_xor<16>(acc); }

template <int XFORM>
__device__ __forceinline__ float4 thin_xform(float4 v, float4 sc, float4 sh, bool ok, int act) {
    if (XFORM == 0) return v;                                   // out-of-image loads already returned zeros
    v = xform4_t<XFORM>(v, sc, sh, act);
    return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- thin input --------------------------------------------------------------------------------------------------------
// thread = (pixel lane, output-channel quad); the 9*CT inputs of a pixel are the same for all lanes of the quad group
// (broadcast loads), the 9*CT*4 weights of the thread stay in registers.
template <int CT, bool WGRAD>
__global__ __launch_bounds__(256) void thin_in_kernel(ThinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float thin_red[];
    const int QW = a.Cw >> 2, PL = 256 / QW;
    const int q = threadIdx.x % QW, pl = threadIdx.x / QW;
    const int xblocks = (a.W + a.xb - 1) / a.xb;
    const int items = a.N * a.H * xblocks;                      // work item = (image, row, block of xb pixels)
    float4 wr[9 * CT];                                          // forward: weights ; wgrad: accumulators  [tap][ci] x 4 co
#pragma unroll
    for (int i = 0; i < 9 * CT; ++i) {
        if (WGRAD) wr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        else {
            const float* wp = a.w + (size_t)(4 * q) * 9 * CT + i;
            wr[i] = make_float4(wp[0], wp[9 * CT], wp[18 * CT], wp[27 * CT]);
        }
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!WGRAD && a.bias) bv = make_float4(a.bias[4 * q], a.bias[4 * q + 1], a.bias[4 * q + 2], a.bias[4 * q + 3]);
    // The 3 x (xb + 2) x CT input window of a work item is staged through LDS once (coalesced, zero padded) and every thread
    // then gathers its 9*CT inputs with LDS broadcasts: 27 global load instructions per pixel and quad group (same address in
    // all lanes of the group, ~16 address-pipe cycles each) made this HBM-write-bound kernel load-issue-bound (0.82 ms for the
    // 1.07 GB conv_in output of the f=16 model at batch 32).  The staging area sits in front of the reduction scratch.
    const int RW = (a.xb + 2) * CT;                             // floats per staged row
    float* stage = thin_red;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
    int b = item;
    const int xb0 = (b % xblocks) * a.xb; b /= xblocks;
    const int y = b % a.H;
    const int n = b / a.H;
    const int x_end = min(a.W, xb0 + a.xb);
    __syncthreads();                                            // readers of the previous item are done
    for (int i = threadIdx.x; i < 3 * RW; i += 256) {
        const int r = i / RW, j = i - r * RW;
        const int yy = y + r - 1;
        const long xf = (long)(xb0 - 1) * CT + j;               // float index inside the image row
        const bool ok = (unsigned)yy < (unsigned)a.H && xf >= 0 && xf < (long)a.W * CT;
        stage[i] = ok ? a.x[((size_t)(n * a.H + yy) * a.W) * CT + xf] : 0.f;
    }
    __syncthreads();
    for (int xx = xb0 + pl; xx < x_end; xx += PL) {
        float in[9 * CT];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const float* sp = stage + kh * RW + (xx - xb0) * CT;
#pragma unroll
            for (int j = 0; j < 3 * CT; ++j) in[kh * 3 * CT + j] = sp[j];
        }
        const size_t o = ((size_t)(n * a.H + y) * a.W + xx) * a.Cw + 4 * q;
        if (WGRAD) {
            const float4 d = *reinterpret_cast<const float4*>(a.dy + o);
#pragma unroll
            for (int i = 0; i < 9 * CT; ++i) {
                wr[i].x = fmaf(d.x, in[i], wr[i].x); wr[i].y = fmaf(d.y, in[i], wr[i].y);
                wr[i].z = fmaf(d.z, in[i], wr[i].z); wr[i].w = fmaf(d.w, in[i], wr[i].w);
            }
        } else {
            float4 acc = bv;
#pragma unroll
            for (int i = 0; i < 9 * CT; ++i) {
                acc.x = fmaf(in[i], wr[i].x, acc.x); acc.y = fmaf(in[i], wr[i].y, acc.y);
                acc.z = fmaf(in[i], wr[i].z, acc.z); acc.w = fmaf(in[i], wr[i].w, acc.w);
            }
            if (a.resid) {
                const float4 r = *reinterpret_cast<const float4*>(a.resid + o);
                acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
            }
            *reinterpret_cast<float4*>(a.y + o) = acc;
        }
    }
    }
    if (WGRAD) {
        // pixel lanes of one wave hold partial sums of the same channels: fold them with shuffles, fold the four waves through
        // LDS in a fixed order, then one slab per workgroup
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        for (int off = QW; off < 64; off <<= 1) {
#pragma unroll
            for (int i = 0; i < 9 * CT; ++i) {
                wr[i].x += __shfl_xor(wr[i].x, off); wr[i].y += __shfl_xor(wr[i].y, off);
                wr[i].z += __shfl_xor(wr[i].z, off); wr[i].w += __shfl_xor(wr[i].w, off);
            }
        }
        const bool owner = lane < QW;                           // QW = 64: every lane; the waves then hold different pixel lanes
        __syncthreads();                                        // the staging area (aliased by `red`) is no longer read
        float4* red = reinterpret_cast<float4*>(thin_red);      // [3 waves][9*CT][QW]
        if (owner && wid > 0) {
#pragma unroll
            for (int i = 0; i < 9 * CT; ++i) red[((wid - 1) * 9 * CT + i) * QW + q] = wr[i];
        }
        __syncthreads();
        if (owner && wid == 0) {
            float* out = a.part + (size_t)blockIdx.x * (size_t)a.Cw * 9 * CT;
#pragma unroll
            for (int i = 0; i < 9 * CT; ++i) {
                float4 v = wr[i];
                for (int w2 = 0; w2 < 3; ++w2) {
                    const float4 t = red[(w2 * 9 * CT + i) * QW + q];
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                float* o4 = out + (size_t)(4 * q) * 9 * CT + i;
                o4[0] = v.x; o4[9 * CT] = v.y; o4[18 * CT] = v.z; o4[27 * CT] = v.w;
            }
        }
    }
}

// ---- thin output -------------------------------------------------------------------------------------------------------
// thread = (pixel lane, input-channel quad); a pixel lane marches along a row segment of `seg` pixels.
template <int CT, int XFORM, bool WGRAD>
__global__ __launch_bounds__(256) void thin_out_kernel(ThinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float thin_red[];
    const int QW = a.Cw >> 2, PL = 256 / QW;
    const int q = threadIdx.x % QW, pl = threadIdx.x / QW;
    const int seg = a.xb / PL;                                  // pixels per pixel lane
    const int xblocks = (a.W + a.xb - 1) / a.xb;
    const int items = a.N * a.H * xblocks;

    float4 wr[CT][9];                                           // forward: weights ; wgrad: accumulators
#pragma unroll
    for (int co = 0; co < CT; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (WGRAD) wr[co][t] = make_float4(0.f, 0.f, 0.f, 0.f);
            else {
                const float* wp = a.w + ((size_t)(co * 9 + t)) * a.Cw + 4 * q;
                wr[co][t] = make_float4(wp[0], wp[1], wp[2], wp[3]);
            }
    float bias[CT];
#pragma unroll
    for (int co = 0; co < CT; ++co) bias[co] = (!WGRAD && a.bias) ? a.bias[co] : 0.f;

    // Items = image rows (x blocks of a row): a row's neighbours y - 1, y + 1 are read again by the items above and below.  Consecutive
    // block ids go to different XCDs (private L2s), so with item = blockIdx the 128-channel tensor came from HBM three times (traffic
    // 3.1 x); with the XCD-contiguous remap the three readers of a row share one L2 at nearly the same time.
    for (int item = xcd_remap(blockIdx.x, gridDim.x); item < items; item += gridDim.x) {
    int b = item;
    const int xs = (b % xblocks) * a.xb + pl * seg; b /= xblocks;
    const int y = b % a.H;
    const int n = b / a.H;
    const int x_end = min(a.W, xs + seg);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
    if (XFORM) {
        sc = *reinterpret_cast<const float4*>(a.scale + (size_t)n * a.aff_stride + 4 * q);
        sh = *reinterpret_cast<const float4*>(a.shift + (size_t)n * a.aff_stride + 4 * q);
    }
    // transformed input column xc (rows y-1, y, y+1)
    auto load_col = [&](int xc, float4 (&c)[3]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xc < (unsigned)a.W;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + yy) * a.W + xc) * a.Cw + 4 * q);
            c[r] = thin_xform<XFORM>(v, sc, sh, ok, a.act);
        }
    };
    auto pixel = [&](int xx, const float4 (&L)[3], const float4 (&M)[3], const float4 (&R)[3]) {
        if (WGRAD) {
            float d[CT];
            const float* dp = a.dy + ((size_t)(n * a.H + y) * a.W + xx) * CT;
#pragma unroll
            for (int co = 0; co < CT; ++co) d[co] = dp[co];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int co = 0; co < CT; ++co) {
                    float4& a0 = wr[co][r * 3 + 0];
                    float4& a1 = wr[co][r * 3 + 1];
                    float4& a2 = wr[co][r * 3 + 2];
                    a0.x = fmaf(d[co], L[r].x, a0.x); a0.y = fmaf(d[co], L[r].y, a0.y); a0.z = fmaf(d[co], L[r].z, a0.z); a0.w = fmaf(d[co], L[r].w, a0.w);
                    a1.x = fmaf(d[co], M[r].x, a1.x); a1.y = fmaf(d[co], M[r].y, a1.y); a1.z = fmaf(d[co], M[r].z, a1.z); a1.w = fmaf(d[co], M[r].w, a1.w);
                    a2.x = fmaf(d[co], R[r].x, a2.x); a2.y = fmaf(d[co], R[r].y, a2.y); a2.z = fmaf(d[co], R[r].z, a2.z); a2.w = fmaf(d[co], R[r].w, a2.w);
                }
        } else {
            float s[CT];
#pragma unroll
            for (int co = 0; co < CT; ++co) {
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float4 w0 = wr[co][r * 3 + 0], w1 = wr[co][r * 3 + 1], w2 = wr[co][r * 3 + 2];
                    acc = fmaf(L[r].x, w0.x, fmaf(L[r].y, w0.y, fmaf(L[r].z, w0.z, fmaf(L[r].w, w0.w, acc))));
                    acc = fmaf(M[r].x, w1.x, fmaf(M[r].y, w1.y, fmaf(M[r].z, w1.z, fmaf(M[r].w, w1.w, acc))));
                    acc = fmaf(R[r].x, w2.x, fmaf(R[r].y, w2.y, fmaf(R[r].z, w2.z, fmaf(R[r].w, w2.w, acc))));
                }
                acc = thin_group_sum(acc, QW);
                s[co] = acc;
            }
            if (q == 0) {
                float* op = a.y + ((size_t)(n * a.H + y) * a.W + xx) * CT;
#pragma unroll
                for (int co = 0; co < CT; ++co) {
                    float v = s[co] + bias[co];
                    if (a.resid) v += a.resid[((size_t)(n * a.H + y) * a.W + xx) * CT + co];
                    op[co] = v;
                }
            }
        }
    };

    if (xs < x_end) {                                           // (uniform per pixel lane; shuffles stay inside the lane group)
        float4 c0[3], c1[3], c2[3];
        load_col(xs - 1, c0);
        load_col(xs, c1);
        int xx = xs;
        // rotate the roles of the three column registers instead of moving them
        while (true) {
            load_col(xx + 1, c2); pixel(xx, c0, c1, c2); if (++xx >= x_end) break;
            load_col(xx + 1, c0); pixel(xx, c1, c2, c0); if (++xx >= x_end) break;
            load_col(xx + 1, c1); pixel(xx, c2, c0, c1); if (++xx >= x_end) break;
        }
    }
    }
    if (WGRAD) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        for (int off = QW; off < 64; off <<= 1) {
#pragma unroll
            for (int co = 0; co < CT; ++co)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    wr[co][t].x += __shfl_xor(wr[co][t].x, off); wr[co][t].y += __shfl_xor(wr[co][t].y, off);
                    wr[co][t].z += __shfl_xor(wr[co][t].z, off); wr[co][t].w += __shfl_xor(wr[co][t].w, off);
                }
        }
        const bool owner = lane < QW;
        float4* red = reinterpret_cast<float4*>(thin_red);      // [3 waves][CT*9][QW]
        if (owner && wid > 0) {
#pragma unroll
            for (int co = 0; co < CT; ++co)
#pragma unroll
                for (int t = 0; t < 9; ++t) red[((wid - 1) * CT * 9 + co * 9 + t) * QW + q] = wr[co][t];
        }
        __syncthreads();
        if (owner && wid == 0) {
            float* out = a.part + (size_t)blockIdx.x * (size_t)CT * 9 * a.Cw;
#pragma unroll
            for (int co = 0; co < CT; ++co)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    float4 v = wr[co][t];
                    for (int w2 = 0; w2 < 3; ++w2) {
                        const float4 u = red[(w2 * CT * 9 + co * 9 + t) * QW + q];
                        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    }
                    *reinterpret_cast<float4*>(out + ((size_t)(co * 9 + t)) * a.Cw + 4 * q) = v;
                }
        }
    }
}

// ---- one output channel (the PatchGAN head, models/discriminator.py:213: Conv2d(512, 1, 4, 1, 1)) ----------------------------------------
// y[p] = bias + sum_{tap, ci} T(x)[p + tap][ci] w[tap][ci]: a dot product of K = KH KW Cin terms per output pixel.  The implicit-GEMM
// kernels pad the single output channel to a 32-wide MFMA tile and walk K in 512 barrier-separated steps with 225 workgroups on 256 CUs
// (285 us for 0.5 GFLOP at batch 32).  Here ONE WAVE owns an output pixel: its lanes stride over the (tap, channel quad) pairs -- weights
// are one contiguous float4 stream (OHWI with O = 1), the input rows of a tap are contiguous channel runs -- and fold with the fixed-order
// wave sum.  Any kernel size / stride / pad, plain gather, fused input transform (BatchNorm / activation on load).
template <int XFORM>
__global__ __launch_bounds__(256) void conv_cout1_kernel(ThinArgs a, int Hout, int Wout, int KH, int KW, int stride, int pad) {
    const int lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long M = (long)a.N * Hout * Wout;
    if (p >= M) return;
    const int ow = (int)(p % Wout), oh = (int)((p / Wout) % Hout), n = (int)(p / ((long)Wout * Hout));
    const int C4 = a.Cw >> 2, K4 = KH * KW * C4;
    const float4* w4 = reinterpret_cast<const float4*>(a.w);
    // four independent partial sums (e, e + 64, e + 128, e + 192: four loads in flight per lane), folded in a fixed order
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    for (int e0 = lane; e0 < K4; e0 += 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 64 * u;
            if (e >= K4) continue;
            const int tap = e / C4, cq = e - tap * C4;
            const int kh = tap / KW, kw = tap - kh * KW;
            const int ih = oh * stride + kh - pad, iw = ow * stride + kw - pad;
            if ((unsigned)ih >= (unsigned)a.H || (unsigned)iw >= (unsigned)a.W) continue;   // zero padding of the TRANSFORMED tensor
            float4 v = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + ih) * a.W + iw) * a.Cw + 4 * cq);
            if (XFORM) {
                const float4 sc = *reinterpret_cast<const float4*>(a.scale + (size_t)n * a.aff_stride + 4 * cq);
                const float4 sh = *reinterpret_cast<const float4*>(a.shift + (size_t)n * a.aff_stride + 4 * cq);
                v = xform4_t<XFORM>(v, sc, sh, a.act);
            }
            const float4 w = w4[e];
            ps[u] = fmaf(v.x, w.x, fmaf(v.y, w.y, fmaf(v.z, w.z, fmaf(v.w, w.w, ps[u]))));
        }
    }
    float acc = (ps[0] + ps[1]) + (ps[2] + ps[3]);
    acc = wave_sum(acc);
    if (lane == 0) a.y[p] = acc + (a.bias ? a.bias[0] : 0.f);
}
