// GroupNorm statistics / fused GroupNorm(+SiLU) backward for NHWC tensors (HBM-bound streaming kernels).
//
//   stats : per (image n, group g) mean and 1/sqrt(var+eps) over cpg = C/G channels x HW pixels, emitted both as
//           (mean, rstd) and as the per-(n,c) affine  scale = rstd*gamma, shift = beta - mean*rstd*gamma  that the conv
//           kernels apply while they load their input tile (the normalised tensor is never written to HBM).
//   bwd   : da = dL/d act(GN(x))  ->  dx, dgamma, dbeta  (SURVEY Appendix C):
//           dy = da * act'(y);  S1[n,c] = sum_hw dy, S2[n,c] = sum_hw dy*xhat  (pass 1, streaming)
//           k1[n,g] = mean_g(gamma*S1), k2[n,g] = mean_g(gamma*S2), dgamma = sum_n S2, dbeta = sum_n S1   (tiny)
//           dx = rstd * (dy*gamma - k1 - xhat*k2) (+ dx_add)                                                (pass 2)
// Accumulation is in fp64 (cheap next to the HBM traffic) so E[x^2]-mean^2 does not cancel.
// Layout of the work: a block owns a contiguous range of pixels of ONE image and all C channels; lanes run along the
// channel dimension with 16-byte loads (C % 4 == 0), so every global access is a full contiguous row segment.
#include "common.h"

namespace {

__device__ __forceinline__ float act_grad(float y, int act) { return favae_act_grad(y, act); }

// part[n][split][c][2] (double): MODE 0: sum x, sum x^2 ; MODE 1: S1, S2 of the backward.
// VEC: thread = (channel quad, row lane); otherwise thread = (channel, row lane); C > 256*V loops over channel blocks.
// AT (round 6): storage type of x / da -- float, or bf16_t (V = 4 only): common.h
template <int MODE, int V, typename AT = float>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, const float* __restrict__ da,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         double* __restrict__ part, long HW, int C, int G, int act,
                                                         long rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) double sm[];     // [RL][Q][2*V]
    const int n = blockIdx.y, split = blockIdx.x, S = gridDim.x;
    const int QT = C / V;                                           // column items in total
    const int Q = QT < 256 ? QT : 256;                              // column items per pass
    const int RL = 256 / Q;
    const int qi = threadIdx.x % Q, li = threadIdx.x / Q;
    const long r0 = (long)split * rows_per_block;
    const long r1 = min(HW, r0 + rows_per_block);
    const int cpg = C / G;
    for (int qb = 0; qb < QT; qb += Q) {
        const int c = (qb + qi) * V;
        const bool on = (qb + qi) < QT && li < RL;
        double s1[V], s2[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
        if (on) {
            float mu[V], rs[V], ga[V], be[V];
            if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    mu[e] = mean[n * G + (c + e) / cpg];
                    rs[e] = rstd[n * G + (c + e) / cpg];
                    ga[e] = gamma[c + e];
                    be[e] = beta[c + e];
                }
            }
            static_assert(sizeof(AT) == 4 || V == 4, "bf16 storage: the vector path only");
            const size_t e0 = ((size_t)n * HW) * C + c;                 // element index of (image n, row 0, channel c)
            const float* xp = x + e0;                                   // (fp32 storage; the bf16 path indexes from x / da by element)
            const float* dp = MODE == 1 ? da + e0 : nullptr;
            auto accum = [&](const float (&xv)[V], const float (&dv)[V]) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    if (MODE == 0) {
                        s1[e] += (double)xv[e];
                        s2[e] += (double)xv[e] * (double)xv[e];
                    } else {
                        const float xh = (xv[e] - mu[e]) * rs[e];
                        const float y = fmaf(xh, ga[e], be[e]);
                        const float dy = dv[e] * act_grad(y, act);
                        s1[e] += (double)dy;
                        s2[e] += (double)dy * (double)xh;
                    }
                }
            };
            long r = r0 + li;
            if constexpr (V == 4) {
                constexpr int U = 4;                                  // rows in flight per thread (HBM latency hiding)
                for (; r + (U - 1) * RL < r1; r += U * RL) {
                    float4 t[U], d[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        t[u] = act_get4<AT>(x, e0 + (size_t)(r + u * RL) * C);
                        if (MODE == 1) d[u] = act_get4<AT>(da, e0 + (size_t)(r + u * RL) * C);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float xv[V] = {t[u].x, t[u].y, t[u].z, t[u].w};
                        float dv[V] = {0.f, 0.f, 0.f, 0.f};
                        if (MODE == 1) { dv[0] = d[u].x; dv[1] = d[u].y; dv[2] = d[u].z; dv[3] = d[u].w; }
                        accum(xv, dv);
                    }
                }
            }
            for (; r < r1; r += RL) {
                float xv[V], dv[V];
                if (V == 4) {
                    const float4 t = act_get4<AT>(x, e0 + (size_t)r * C);
                    xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                    if (MODE == 1) {
                        const float4 d = act_get4<AT>(da, e0 + (size_t)r * C);
                        dv[0] = d.x; dv[1] = d.y; dv[2] = d.z; dv[3] = d.w;
                    }
                } else {
                    xv[0] = xp[r * C];
                    if (MODE == 1) dv[0] = dp[r * C];
                }
                accum(xv, dv);
            }
            double* o = sm + ((size_t)li * Q + qi) * 2 * V;
#pragma unroll
            for (int e = 0; e < V; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
        }
        __syncthreads();
        if (on && li == 0) {
            for (int l = 1; l < RL; ++l) {
                const double* o = sm + ((size_t)l * Q + qi) * 2 * V;
#pragma unroll
                for (int e = 0; e < V; ++e) { s1[e] += o[2 * e]; s2[e] += o[2 * e + 1]; }
            }
            double* out = part + (((size_t)n * S + split) * C + c) * 2;
#pragma unroll
            for (int e = 0; e < V; ++e) { out[2 * e] = s1[e]; out[2 * e + 1] = s2[e]; }
        }
        __syncthreads();
    }
}

// one block (256 threads: it must fit on a CU next to a resident 512-thread weight-gradient workgroup -- 1024-thread blocks waited
// up to 0.2 ms for a free CU in the f=4 step) per (image, group slab): sum the splits per channel -> acc[n][c][2] (double), then group statistics + per-channel
// affine.  Split sums as in gn_bwd_finalize_kernel: 256 / min(C, 256) interleaved slices per channel, slices added in ascending
// order (deterministic); S is ~32 for the streaming pass and (H/8)(W/16) for tile partials from a conv epilogue.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ mean,
                                                           float* __restrict__ rstd, float* __restrict__ scale,
                                                           float* __restrict__ shift, double* __restrict__ acc, long HW, int C,
                                                           int G, int S, float eps, unsigned* __restrict__ absmax) {
    __shared__ double sl[2 * 256];
    const int n = blockIdx.x;
    const int cpg = C / G;
    // blockIdx.y = a slab of whole groups (gridDim.y divides G): more blocks in flight for this latency-bound kernel
    const int GS = G / gridDim.y, gb = blockIdx.y * GS, ge = gb + GS, cb = gb * cpg, ce = ge * cpg;
    double* a = acc + (size_t)n * C * 2;
    {
        const int CS = ce - cb;
        const int C2 = CS < 256 ? CS : 256, NS = 256 / C2;
        const int ci = threadIdx.x % C2, si = threadIdx.x / C2;
        for (int c0 = cb; c0 < ce; c0 += C2) {
            const int c = c0 + ci;
            double s1 = 0.0, s2 = 0.0;
            if (si < NS && c < ce) {
                for (int s = si; s < S; s += NS) {
                    const double* p = part + (((size_t)n * S + s) * C + c) * 2;
                    s1 += p[0];
                    s2 += p[1];
                }
            }
            sl[2 * threadIdx.x] = s1;
            sl[2 * threadIdx.x + 1] = s2;
            __syncthreads();
            if (si == 0 && c < ce) {
                for (int q = 1; q < NS; ++q) { s1 += sl[2 * (q * C2 + ci)]; s2 += sl[2 * (q * C2 + ci) + 1]; }
                a[2 * c] = s1;
                a[2 * c + 1] = s2;
            }
            __syncthreads();
        }
    }
    for (int g = gb + threadIdx.x; g < ge; g += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s1 += a[2 * c]; s2 += a[2 * c + 1]; }
        const double cnt = (double)cpg * (double)HW;
        const double mu = s1 / cnt;
        double var = s2 / cnt - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[n * G + g] = (float)mu;
        rstd[n * G + g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    if (scale) {
        for (int c = cb + threadIdx.x; c < ce; c += 256) {
            const int g = c / cpg;
            const float mu = mean[n * G + g], rs = rstd[n * G + g];
            const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
            const float sc = rs * ga;
            scale[(size_t)n * C + c] = sc;
            shift[(size_t)n * C + c] = be - mu * sc;
        }
    }
    if (absmax && blockIdx.x == 0 && blockIdx.y == 0) {
        // |gamma (x - mu) rstd + beta| <= |gamma| sqrt(count) + |beta|  (sum of squares of the normalised group = count), and
        // |SiLU(t)|, |LeakyReLU(t)| <= |t|: an upper bound of the transformed activations for the fp16 split scale (conv_split.h).
        // It depends on gamma, beta and the group size only -- not on the data, not on the image: ONE block computes it over all
        // channels and stores it (round 3: was an atomicMax from every block into a target zeroed by a memset launch per GroupNorm)
        const float root = sqrtf((float)cpg * (float)HW);
        float bound = 0.f;
        for (int c = threadIdx.x; c < C; c += 256)
            bound = fmaxf(bound, fmaf(fabsf(gamma ? gamma[c] : 1.f), root, fabsf(beta ? beta[c] : 0.f)));
        bound = wave_max(bound);
        float* wm = reinterpret_cast<float*>(sl);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = bound;
        __syncthreads();
        if (threadIdx.x == 0) *absmax = __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
    }
}

// backward, one block (256 threads) per image: acc[n][c] = sum over splits of (S1,S2); k1/k2 per group.  The S partials of a
// channel are summed by 256 / C2 threads (C2 = min(C, 256)) over interleaved slices, then the slices in ascending order --
// a fixed order for a given (S, C): deterministic.  (S is ~32 for the streaming pass 1 and (H/8)(W/16) = up to 512 for the
// per-tile partials of the data-gradient epilogue, where one thread per channel took 60 us.)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const double* __restrict__ part, const float* __restrict__ gamma,
                                                               float* __restrict__ k1, float* __restrict__ k2,
                                                               double* __restrict__ acc, long HW, int C, int G, int S,
                                                               unsigned* __restrict__ zero_out) {
    __shared__ double sl[2 * 256];
    if (zero_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *zero_out = 0u;   // max |dx| of the apply pass behind
    const int n = blockIdx.x;
    const int cpg = C / G;
    const int GS = G / gridDim.y, gb = blockIdx.y * GS, ge = gb + GS, cb = gb * cpg, ce = ge * cpg;   // slab of whole groups
    double* a = acc + (size_t)n * C * 2;
    const int CS = ce - cb;
    const int C2 = CS < 256 ? CS : 256, NS = 256 / C2;         // slices per channel
    const int ci = threadIdx.x % C2, si = threadIdx.x / C2;
    for (int c0 = cb; c0 < ce; c0 += C2) {
        const int c = c0 + ci;
        double s1 = 0.0, s2 = 0.0;
        if (si < NS && c < ce) {
            for (int s = si; s < S; s += NS) {
                const double* p = part + (((size_t)n * S + s) * C + c) * 2;
                s1 += p[0];
                s2 += p[1];
            }
        }
        sl[2 * threadIdx.x] = s1;
        sl[2 * threadIdx.x + 1] = s2;
        __syncthreads();
        if (si == 0 && c < ce) {
            for (int q = 1; q < NS; ++q) { s1 += sl[2 * (q * C2 + ci)]; s2 += sl[2 * (q * C2 + ci) + 1]; }
            a[2 * c] = s1;
            a[2 * c + 1] = s2;
        }
        __syncthreads();
    }
    for (int g = gb + threadIdx.x; g < ge; g += 256) {
        double u = 0.0, v = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            u += (double)gamma[c] * a[2 * c];
            v += (double)gamma[c] * a[2 * c + 1];
        }
        const double cnt = (double)cpg * (double)HW;
        k1[n * G + g] = (float)(u / cnt);
        k2[n * G + g] = (float)(v / cnt);
    }
}

// dgamma[c] = sum_n S2[n][c], dbeta[c] = sum_n S1[n][c]
__global__ __launch_bounds__(256) void gn_param_grad_kernel(const double* __restrict__ acc, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int N, int C, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n) {
        s1 += acc[((size_t)n * C + c) * 2];
        s2 += acc[((size_t)n * C + c) * 2 + 1];
    }
    dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
}

// dx = rstd * (dy*gamma - k1 - xhat*k2) + dx_add ; grid-stride over float4 (VEC) or scalars
template <bool VEC>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ da, const float* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ k1, const float* __restrict__ k2,
                                                           const float* __restrict__ dx_add, float* __restrict__ dx, int N,
                                                           long HW, int C, int G, int act) {
    const int cpg = C / G;
    constexpr int V = VEC ? 4 : 1;
    const size_t per_img = (size_t)HW * C / V;
    const size_t total = (size_t)N * per_img;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / per_img);
        const int c0 = (int)((i % (C / V)) * V);
        float xv[4], dv[4], av[4], ov[4];
        if (VEC) {
            const float4 t = reinterpret_cast<const float4*>(x)[i];
            const float4 d = reinterpret_cast<const float4*>(da)[i];
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            dv[0] = d.x; dv[1] = d.y; dv[2] = d.z; dv[3] = d.w;
            if (dx_add) {
                const float4 q = reinterpret_cast<const float4*>(dx_add)[i];
                av[0] = q.x; av[1] = q.y; av[2] = q.z; av[3] = q.w;
            }
        } else {
            xv[0] = x[i]; dv[0] = da[i];
            if (dx_add) av[0] = dx_add[i];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int c = c0 + e;
            const int g = c / cpg;
            const float mu = mean[n * G + g], rs = rstd[n * G + g];
            const float ga = gamma[c], be = beta[c];
            const float xh = (xv[e] - mu) * rs;
            const float y = fmaf(xh, ga, be);
            const float dy = dv[e] * act_grad(y, act);
            float o = rs * (dy * ga - k1[n * G + g] - xh * k2[n * G + g]);
            if (dx_add) o += av[e];
            ov[e] = o;
        }
        if (VEC) reinterpret_cast<float4*>(dx)[i] = make_float4(ov[0], ov[1], ov[2], ov[3]);
        else dx[i] = ov[0];
    }
}

// Same result as gn_bwd_apply_kernel<true>, organised like gn_partial_kernel: thread = (channel quad, row lane) of image
// blockIdx.y, so the per-channel constants live in registers (no per-element parameter loads, no 64-bit index divisions)
// and four rows are in flight per thread.  Requires C % 4 == 0, C / 4 <= 256 and 16-byte aligned tensors.
// Round 3: at most 96 registers (launch bound 5 waves per SIMD): the kernel has to find a wave slot NEXT to the register-heavy
// weight-gradient kernel of the second stream (2 x 208 of the 512 registers of a SIMD lane), which is the whole point of that stream.
// CS: the pass also emits, for the tensor dx it writes, the per-block column sums cs_part[block][C] (block = blockIdx.y * gridDim.x
// + blockIdx.x) and max |dx| -- dx is the `dy` of the conv in front of this GroupNorm, whose bias gradient and fp16 operand
// range are exactly these two (favae_colsum read the tensor once more for them).
// AT (round 6): storage type of da / x / dx_add / dx (float, or bf16_t: every tensor of the pass at half the bytes; same arithmetic)
template <bool SKIP, bool CS, typename AT = float>
__global__ __launch_bounds__(256, 5) void gn_bwd_apply_rows_kernel(const float* __restrict__ da, const float* __restrict__ x,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ k1, const float* __restrict__ k2,
                                                                   const float* __restrict__ dx_add, float* __restrict__ dx,
                                                                   long HW, int C, int G, int act, long rows_per_block,
                                                                   float* __restrict__ cs_part, unsigned* __restrict__ cs_amax) {
    __shared__ float4 sm[CS ? 256 : 1];
    const int n = blockIdx.y;
    const int Q = C / 4, RL = 256 / Q;
    const int qi = threadIdx.x % Q, li = threadIdx.x / Q;
    const bool on = li < RL;
    const int cpg = C / G, c = qi * 4;
    float mu[4], rs[4], ga[4], be[4], rk1[4], rk2[4];                // xh = (x - mu) rs ; dx = rs (dy ga - k1 - xh k2)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int g = (c + e) / cpg;
        mu[e] = mean[n * G + g];
        rs[e] = rstd[n * G + g];
        ga[e] = gamma[c + e];
        be[e] = beta[c + e];
        rk1[e] = k1[n * G + g];
        rk2[e] = k2[n * G + g];
    }
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    // per-image descriptors (an image is < 4 GiB); lane offset = (row lane, channel quad), the row block advances in an SGPR
    constexpr unsigned EB = ActT<AT>::B;                              // bytes per stored element
    const size_t img = ((size_t)n * HW) * C * EB;                     // byte offset of image n
    const unsigned img_bytes = (unsigned)((size_t)HW * C * EB);
    auto at = [&](const float* p) { return reinterpret_cast<const char*>(p) + img; };
    const auto rx = make_rsrc(at(x), img_bytes), rda = make_rsrc(at(da), img_bytes), rdx = make_rsrc(at(dx), img_bytes);
    const auto rsk = make_rsrc(SKIP ? at(dx_add) : (const char*)x, SKIP ? img_bytes : 0u);
    const unsigned voff = on ? (unsigned)((li * C + c) * EB) : FAVAE_OOB;
    const unsigned row_b = (unsigned)C * EB;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float mx = 0.f;
    auto one = [&](const float4 t, const float4 d, const float4 q) -> float4 {
        const float xv[4] = {t.x, t.y, t.z, t.w}, dv[4] = {d.x, d.y, d.z, d.w}, av[4] = {q.x, q.y, q.z, q.w};
        float ov[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // explicit fused multiply-adds: every instantiation rounds the same way (left to the compiler, the contraction of
            // rs (dy ga - k1 - xh k2) changed with the code around it)
            const float xh = (xv[e] - mu[e]) * rs[e];
            const float dy = dv[e] * act_grad(fmaf(xh, ga[e], be[e]), act);
            const float t = fmaf(-xh, rk2[e], fmaf(dy, ga[e], -rk1[e]));
            ov[e] = SKIP ? fmaf(rs[e], t, av[e]) : rs[e] * t;
        }
        return make_float4(ov[0], ov[1], ov[2], ov[3]);
    };
    auto tally = [&](const float4 o) {
        cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
    };
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 4;
    // rows r0 + li + k RL (k = 0, 1, ...) of this lane; rows >= r1 must not be touched: their lane offset becomes out-of-range
    for (long rb = r0; rb < r1; rb += U * RL) {
        float4 t[U], d[U], q[U];
        unsigned vo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vo[u] = (rb + u * RL + li < r1) ? voff : FAVAE_OOB;
            const unsigned so = (unsigned)(rb + u * RL) * row_b;
            t[u] = act_load4<AT>(rx, vo[u], so);
            d[u] = act_load4<AT>(rda, vo[u], so);
            q[u] = SKIP ? act_load4<AT>(rsk, vo[u], so) : z4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 o = one(t[u], d[u], q[u]);
            act_store4<AT>(rdx, vo[u], (unsigned)(rb + u * RL) * row_b, o);
            if (CS && vo[u] != FAVAE_OOB) tally(o);
        }
    }
    if constexpr (CS) {
        // column sums of this block's rows: the RL row lanes of a channel quad in ascending order (fixed order: deterministic)
        sm[threadIdx.x] = cs;
        mx = wave_max(mx);
        __syncthreads();
        if (on && li == 0) {
            for (int l = 1; l < RL; ++l) {
                const float4 v = sm[l * Q + qi];
                cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
            }
            const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            reinterpret_cast<float4*>(cs_part + blk * C)[qi] = cs;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) reinterpret_cast<float*>(sm)[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {                  // one same-address atomic per block; non-negative floats order like their bits
            const float* w = reinterpret_cast<const float*>(sm);
            atomicMax(cs_amax, __float_as_uint(fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]))));
        }
    }
}

__global__ void bn_update_running_kernel(const float* mean, const float* rstd, int C, double count, float eps, float mom,
                                         float* rm, float* rv) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double rs = rstd[c];
    double var = 1.0 / (rs * rs) - (double)eps;          // biased batch variance
    if (var < 0) var = 0;
    const double unb = count > 1 ? var * count / (count - 1.0) : var;
    rm[c] = (1.f - mom) * rm[c] + mom * mean[c];
    rv[c] = (1.f - mom) * rv[c] + mom * (float)unb;
}

// group slabs per image for the finalize kernels: up to 8 blocks per image, whole groups per block
int gn_slabs(int G) {
    int s = 8;
    while (s > 1 && G % s) s >>= 1;
    return s;
}

int gn_splits(int N, long HW) {
    long s = (HW + 1023) / 1024;          // >= 1024 pixels per block
    long cap = (1024 + N - 1) / N;        // ~1024 blocks in total
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return (int)s;
}

size_t part_bytes(int N, long HW, int C) { return (size_t)N * gn_splits(N, HW) * C * 2 * sizeof(double); }
size_t acc_bytes(int N, int C) { return (size_t)N * C * 2 * sizeof(double); }

template <int MODE>
void launch_partial(const float* x, const float* da, const float* gamma, const float* beta, const float* mean,
                    const float* rstd, double* part, int N, long HW, int C, int G, int act, hipStream_t s, bool bf16io = false) {
    const int S = gn_splits(N, HW);
    const long rpb = (HW + S - 1) / S;
    FAVAE_PROF_NOTE(0, (MODE == 0 ? 4.0 : 8.0) * (bf16io ? 0.5 : 1.0) * N * HW * C);   // one read of x (+ one of da)
    const bool vec = (C % 4 == 0) && ((((uintptr_t)x) & 15) == 0) && (MODE == 0 || (((uintptr_t)da) & 15) == 0);
    if (bf16io) {                         // bf16 storage: the caller guarantees C % 4 == 0 and 8-byte alignment
        const int QT = C / 4, Q = QT < 256 ? QT : 256, RL = 256 / Q;
        const size_t shm = (size_t)RL * Q * 8 * sizeof(double);
        FAVAE_KLAUNCH((gn_partial_kernel<MODE, 4, bf16_t>), dim3(S, N), dim3(256), shm, s, x, da, gamma, beta, mean, rstd, part, HW, C,
                           G, act, rpb);
    } else if (vec) {
        const int QT = C / 4, Q = QT < 256 ? QT : 256, RL = 256 / Q;
        const size_t shm = (size_t)RL * Q * 8 * sizeof(double);
        FAVAE_KLAUNCH((gn_partial_kernel<MODE, 4>), dim3(S, N), dim3(256), shm, s, x, da, gamma, beta, mean, rstd, part, HW, C,
                           G, act, rpb);
    } else {
        const int Q = C < 256 ? C : 256, RL = 256 / Q;
        const size_t shm = (size_t)RL * Q * 2 * sizeof(double);
        FAVAE_KLAUNCH((gn_partial_kernel<MODE, 1>), dim3(S, N), dim3(256), shm, s, x, da, gamma, beta, mean, rstd, part, HW, C,
                           G, act, rpb);
    }
}

}  // namespace

extern "C" size_t favae_gn_workspace(int N, int64_t HW, int C) {
    // split partials + per-(n,c) accumulators + k1/k2 (N*C floats upper bound each)
    return part_bytes(N, HW, C) + acc_bytes(N, C) + 2 * (size_t)N * C * sizeof(float) + 512;
}

static int gn_stats_impl(const float* x, const float* gamma, const float* beta, int N, int64_t HW, int C, int G, float eps, float* mean,
                         float* rstd, float* scale, float* shift, float* absmax_out, void* ws, size_t ws_bytes, favae_stream_t stream,
                         bool bf16io);
extern "C" int favae_gn_stats(const float* x, const float* gamma, const float* beta, int N, int64_t HW, int C, int G,
                              float eps, float* mean, float* rstd, float* scale, float* shift, float* absmax_out, void* ws,
                              size_t ws_bytes, favae_stream_t stream) {
    return gn_stats_impl(x, gamma, beta, N, HW, C, G, eps, mean, rstd, scale, shift, absmax_out, ws, ws_bytes, stream, false);
}
// the same statistics of a tensor STORED as bf16 (round 6, bf16 activation storage): x points at bf16 elements; C % 4 == 0
extern "C" int favae_gn_stats_bf16(const void* x, const float* gamma, const float* beta, int N, int64_t HW, int C, int G,
                                   float eps, float* mean, float* rstd, float* scale, float* shift, float* absmax_out, void* ws,
                                   size_t ws_bytes, favae_stream_t stream) {
    if (C % 4 != 0 || (((uintptr_t)x) & 7) != 0) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    return gn_stats_impl((const float*)x, gamma, beta, N, HW, C, G, eps, mean, rstd, scale, shift, absmax_out, ws, ws_bytes, stream, true);
}
static int gn_stats_impl(const float* x, const float* gamma, const float* beta, int N, int64_t HW, int C, int G, float eps, float* mean,
                         float* rstd, float* scale, float* shift, float* absmax_out, void* ws, size_t ws_bytes, favae_stream_t stream,
                         bool bf16io) {
    FAVAE_REQUIRE(x && mean && rstd && ws && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0);
    FAVAE_REQUIRE((scale == nullptr) == (shift == nullptr));
    if (ws_bytes < favae_gn_workspace(N, HW, C)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)ws;
    double* acc = (double*)((char*)ws + part_bytes(N, HW, C));
    FAVAE_REQUIRE(!absmax_out || scale);
    launch_partial<0>(x, nullptr, gamma, beta, nullptr, nullptr, part, N, (long)HW, C, G, 0, s, bf16io);
    FAVAE_CHECK_LAUNCH();
    FAVAE_KLAUNCH(gn_finalize_kernel, dim3(N, gn_slabs(G)), dim3(256), 0, s, (const double*)part, gamma, beta, mean, rstd, scale, shift,
                       acc, (long)HW, C, G, gn_splits(N, HW), eps, (unsigned*)absmax_out);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

// GroupNorm statistics from per-tile partial sums part[N][tiles][C][2] (double) that the conv producing x emitted from its
// epilogue (favae_conv_fwd_split_stats): the streaming pass over x is not needed.  ws: favae_gn_workspace(N, HW, C) bytes.
extern "C" int favae_gn_stats_tiles(const void* part, int tiles, const float* gamma, const float* beta, int N, int64_t HW, int C,
                                    int G, float eps, float* mean, float* rstd, float* scale, float* shift, float* absmax_out,
                                    void* ws, size_t ws_bytes, favae_stream_t stream) {
    FAVAE_REQUIRE(part && tiles > 0 && mean && rstd && ws && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0);
    FAVAE_REQUIRE((scale == nullptr) == (shift == nullptr));
    FAVAE_REQUIRE(!absmax_out || scale);
    if (ws_bytes < acc_bytes(N, C)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    hipStream_t s = (hipStream_t)stream;
    FAVAE_KLAUNCH(gn_finalize_kernel, dim3(N, gn_slabs(G)), dim3(256), 0, s, (const double*)part, gamma, beta, mean, rstd, scale, shift,
                       (double*)ws, (long)HW, C, G, tiles, eps, (unsigned*)absmax_out);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

// tile_partials > 0: pass 1 was done by the data-gradient conv's epilogue (favae_conv_dgrad_gnbwd): ws starts with
// part[N][tile_partials][C][2]
static int gn_act_bwd_impl(const float* da, const float* x, const float* gamma, const float* beta, const float* mean,
                           const float* rstd, int N, int64_t HW, int C, int G, int act, const float* dx_add, float* dx,
                           float* dgamma, float* dbeta, int accumulate, int tile_partials, void* ws, size_t ws_bytes,
                           favae_stream_t stream, float* cs_part = nullptr, float* cs_amax = nullptr);

// blocks per image of the row-organised apply pass (0: the shape does not run it)
static long apply_rows_blocks(int N, int64_t HW, int C) {
    if (C % 4 != 0 || C / 4 > 256 || 256 % (C / 4) != 0 || (size_t)HW * C * 4 >= ((size_t)1 << 31)) return 0;
    long S = (HW + 255) / 256;                           // >= 256 pixels per block, ~2048 blocks in total
    const long cap = (2048 + N - 1) / N;
    return S > cap ? cap : (S < 1 ? 1 : S);
}

extern "C" int favae_gn_bwd_colsum_blocks(int N, int64_t HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    return (int)(apply_rows_blocks(N, HW, C) * N);
}

// favae_gn_act_bwd / favae_gn_act_bwd_tiles (tiles == 0 / > 0) whose apply pass also emits, for the dx it writes, per-block column
// sums cs_part[favae_gn_bwd_colsum_blocks(N, HW, C)][C] and max |dx| (cs_absmax, one float): the bias gradient (favae_colsum_finish)
// and the fp16 operand range of the conv whose output gradient dx is -- no second pass over the tensor.
extern "C" int favae_gn_act_bwd_colsum(const float* da, const float* x, const float* gamma, const float* beta, const float* mean,
                                       const float* rstd, int N, int64_t HW, int C, int G, int act, const float* dx_add, float* dx,
                                       float* dgamma, float* dbeta, int accumulate, int tiles, void* ws, size_t ws_bytes,
                                       float* cs_part, float* cs_absmax, favae_stream_t stream) {
    FAVAE_REQUIRE(cs_part && cs_absmax && tiles >= 0);
    if (!favae_gn_bwd_colsum_blocks(N, HW, C)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    return gn_act_bwd_impl(da, x, gamma, beta, mean, rstd, N, HW, C, G, act, dx_add, dx, dgamma, dbeta, accumulate, tiles, ws, ws_bytes,
                           stream, cs_part, cs_absmax);
}

extern "C" size_t favae_gn_bwd_tiles_workspace(int N, int tiles, int C) {
    return (size_t)N * tiles * C * 2 * sizeof(double) + acc_bytes(N, C) + 2 * (size_t)N * C * sizeof(float) + 512;
}

extern "C" int favae_gn_act_bwd(const float* da, const float* x, const float* gamma, const float* beta, const float* mean,
                                const float* rstd, int N, int64_t HW, int C, int G, int act, const float* dx_add, float* dx,
                                float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes,
                                favae_stream_t stream) {
    return gn_act_bwd_impl(da, x, gamma, beta, mean, rstd, N, HW, C, G, act, dx_add, dx, dgamma, dbeta, accumulate, 0, ws, ws_bytes,
                           stream);
}

extern "C" int favae_gn_act_bwd_tiles(const float* da, const float* x, const float* gamma, const float* beta, const float* mean,
                                      const float* rstd, int N, int64_t HW, int C, int G, int act, const float* dx_add, float* dx,
                                      float* dgamma, float* dbeta, int accumulate, int tiles, void* ws, size_t ws_bytes,
                                      favae_stream_t stream) {
    FAVAE_REQUIRE(tiles > 0);
    return gn_act_bwd_impl(da, x, gamma, beta, mean, rstd, N, HW, C, G, act, dx_add, dx, dgamma, dbeta, accumulate, tiles, ws,
                           ws_bytes, stream);
}

static int gn_act_bwd_impl(const float* da, const float* x, const float* gamma, const float* beta, const float* mean,
                           const float* rstd, int N, int64_t HW, int C, int G, int act, const float* dx_add, float* dx,
                           float* dgamma, float* dbeta, int accumulate, int tile_partials, void* ws, size_t ws_bytes,
                           favae_stream_t stream, float* cs_part, float* cs_amax) {
    FAVAE_REQUIRE(da && x && gamma && beta && mean && rstd && dx && ws && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0);
    FAVAE_REQUIRE((dgamma == nullptr) == (dbeta == nullptr));
    // FAVAE_ACT_BF16IO (round 6): da, x, dx_add and dx are bf16 tensors (bf16 activation storage); the row-organised pass only
    const bool bf = (act & FAVAE_ACT_BF16IO) != 0;
    act &= ~FAVAE_ACT_BF16IO;
    if (bf && (C % 4 != 0 || ((((uintptr_t)da) | ((uintptr_t)x) | ((uintptr_t)dx) | ((uintptr_t)dx_add)) & 7) != 0 ||
               !apply_rows_blocks(N, HW, C)))
        return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (ws_bytes < (tile_partials ? favae_gn_bwd_tiles_workspace(N, tile_partials, C) : favae_gn_workspace(N, HW, C)))
        return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)ws;
    double* acc = (double*)((char*)ws + (tile_partials ? (size_t)N * tile_partials * C * 2 * sizeof(double) : part_bytes(N, HW, C)));
    float* k1 = (float*)((char*)acc + acc_bytes(N, C));
    float* k2 = k1 + (size_t)N * C;
    if (!tile_partials) {
        launch_partial<1>(x, da, gamma, beta, mean, rstd, part, N, (long)HW, C, G, act, s, bf);
        FAVAE_CHECK_LAUNCH();
    }
    FAVAE_KLAUNCH(gn_bwd_finalize_kernel, dim3(N, gn_slabs(G)), dim3(256), 0, s, (const double*)part, gamma, k1, k2, acc, (long)HW, C, G,
                       tile_partials ? tile_partials : gn_splits(N, HW), (unsigned*)cs_amax);
    FAVAE_CHECK_LAUNCH();
    if (dgamma) {
        FAVAE_KLAUNCH(gn_param_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, (const double*)acc, dgamma, dbeta, N, C, accumulate);
        FAVAE_CHECK_LAUNCH();
    }
    const size_t total = (size_t)N * HW * C;
    FAVAE_PROF_NOTE(0, (dx_add ? 16.0 : 12.0) * (bf ? 0.5 : 1.0) * total);    // reads da, x (+ dx_add), writes dx
    const bool vec = bf || ((C % 4 == 0) && (((((uintptr_t)da) | ((uintptr_t)x) | ((uintptr_t)dx) | ((uintptr_t)dx_add)) & 15) == 0));
    const long S = apply_rows_blocks(N, HW, C);
    if (cs_part && !(vec && S)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (vec && S) {
        const long rpb = (HW + S - 1) / S;
        const dim3 grid((unsigned)S, N);
#define FAVAE_LAUNCH_APPLY(SK, CS)                                                                                             \
    do {                                                                                                                       \
        if (bf) FAVAE_KLAUNCH((gn_bwd_apply_rows_kernel<SK, CS, bf16_t>), grid, dim3(256), 0, s, da, x, gamma, beta, mean, rstd, k1, \
                              k2, dx_add, dx, (long)HW, C, G, act, rpb, cs_part, (unsigned*)cs_amax);                          \
        else FAVAE_KLAUNCH((gn_bwd_apply_rows_kernel<SK, CS>), grid, dim3(256), 0, s, da, x, gamma, beta, mean, rstd, k1, k2, dx_add, \
                           dx, (long)HW, C, G, act, rpb, cs_part, (unsigned*)cs_amax);                                         \
    } while (0)
        if (dx_add && cs_part) FAVAE_LAUNCH_APPLY(true, true);
        else if (dx_add) FAVAE_LAUNCH_APPLY(true, false);
        else if (cs_part) FAVAE_LAUNCH_APPLY(false, true);
        else FAVAE_LAUNCH_APPLY(false, false);
#undef FAVAE_LAUNCH_APPLY
    } else if (vec) {
        int blocks = (int)((total / 4 + 255) / 256);
        if (blocks > 8192) blocks = 8192;
        FAVAE_KLAUNCH((gn_bwd_apply_kernel<true>), dim3(blocks), dim3(256), 0, s, da, x, gamma, beta, mean, rstd, k1, k2,
                           dx_add, dx, N, (long)HW, C, G, act);
    } else {
        int blocks = (int)((total + 255) / 256);
        if (blocks > 8192) blocks = 8192;
        FAVAE_KLAUNCH((gn_bwd_apply_kernel<false>), dim3(blocks), dim3(256), 0, s, da, x, gamma, beta, mean, rstd, k1, k2,
                           dx_add, dx, N, (long)HW, C, G, act);
    }
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_bn_update_running(const float* mean, const float* rstd, int C, int64_t count, float eps, float momentum,
                                       float* running_mean, float* running_var, favae_stream_t stream) {
    FAVAE_REQUIRE(mean && rstd && running_mean && running_var && C > 0 && count > 0);
    FAVAE_KLAUNCH(bn_update_running_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, rstd, C,
                       (double)count, eps, momentum, running_mean, running_var);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}
