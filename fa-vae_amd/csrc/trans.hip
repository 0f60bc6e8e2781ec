// Token-wise kernels of the attention FCM (TransEncoderBlock, models/codec.py:108-122 = GroupNorm + nn.TransformerEncoderLayer
// (d_model, nhead 8, FFN 2048, ReLU, dropout 0.1, post-norm); DecoderFcmAttnGauss :1011-1129 and its ResnetBlock(dropout=0.1)).
// All three are HBM-bound elementwise / row passes over NHWC rows (row = one token = C contiguous floats):
//   affine_rows   y = act(x * scale[n][c] + shift[n][c])            -- materialised GroupNorm (the block's residual is the
//                                                                      NORMALISED tensor, so it has to exist in memory)
//   layernorm     per-row mean / rstd over C in fp64, one wave per row; backward emits dx and dy * xhat (the dgamma operand;
//                                                                      dgamma / dbeta are favae_colsum passes: deterministic)
//   dropout       y = keep(seed, i) [&& gate[i] > 0] ? x * 1/(1-p) : 0 -- counter-based mask (no state, regenerated in the
//                                                                      backward from the same seed), optional fused ReLU gate
#include "common.h"

namespace {

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == FAVAE_ACT_SILU) return silu_f(v);
    if (act == FAVAE_ACT_LEAKY02) return v > 0.f ? v : 0.2f * v;
    if (act == FAVAE_ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

// grid-stride over float4s; scale/shift rows are (n, c): n = pixel / HW
__global__ __launch_bounds__(256) void affine_rows_kernel(const float4* __restrict__ x, const float4* __restrict__ scale,
                                                          const float4* __restrict__ shift, float4* __restrict__ y, long total4,
                                                          long hw_c4, int c4, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long n = i / hw_c4;
        const int q = (int)(i % c4);
        const float4 v = x[i], sc = scale[n * c4 + q], sh = shift[n * c4 + q];
        y[i] = make_float4(act_f(fmaf(v.x, sc.x, sh.x), act), act_f(fmaf(v.y, sc.y, sh.y), act), act_f(fmaf(v.z, sc.z, sh.z), act),
                           act_f(fmaf(v.w, sc.w, sh.w), act));
    }
}

constexpr int LN_MAXQ = 8;          // float4s per lane: C <= 64 * 4 * 8 = 2048

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, long rows, int C,
                                                            float eps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nq = C >> 2;
    for (long r = (long)blockIdx.x * 4 + w; r < rows; r += (long)gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + r * C);
        float4 v[LN_MAXQ];
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < LN_MAXQ; ++k) {
            const int q = lane + 64 * k;
            if (q < nq) {
                v[k] = xr[q];
                s += (double)v[k].x + (double)v[k].y + (double)v[k].z + (double)v[k].w;
            }
        }
        const double mu = wave_sum_d(s) / C;
        double ss = 0.0;
#pragma unroll
        for (int k = 0; k < LN_MAXQ; ++k) {
            const int q = lane + 64 * k;
            if (q < nq) {
                const double a = v[k].x - mu, b = v[k].y - mu, c = v[k].z - mu, d = v[k].w - mu;
                ss += a * a + b * b + c * c + d * d;
            }
        }
        const double var = wave_sum_d(ss) / C;                         // biased, as nn.LayerNorm
        const float mf = (float)mu, rs = (float)(1.0 / sqrt(var + (double)eps));
        if (lane == 0) { mean[r] = mf; rstd[r] = rs; }
        float4* yr = reinterpret_cast<float4*>(y + r * C);
#pragma unroll
        for (int k = 0; k < LN_MAXQ; ++k) {
            const int q = lane + 64 * k;
            if (q < nq) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[q], b = reinterpret_cast<const float4*>(beta)[q];
                yr[q] = make_float4(fmaf((v[k].x - mf) * rs, g.x, b.x), fmaf((v[k].y - mf) * rs, g.y, b.y),
                                    fmaf((v[k].z - mf) * rs, g.z, b.z), fmaf((v[k].w - mf) * rs, g.w, b.w));
            }
        }
    }
}

// dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)), g = dy * gamma ; t = dy * xhat (operand of the dgamma column sum)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            float* __restrict__ t, long rows, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nq = C >> 2;
    for (long r = (long)blockIdx.x * 4 + w; r < rows; r += (long)gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + r * C);
        const float4* dr = reinterpret_cast<const float4*>(dy + r * C);
        const float mf = mean[r], rs = rstd[r];
        float4 xh[LN_MAXQ], g[LN_MAXQ];
        double s1 = 0.0, s2 = 0.0;
        float4* tr = reinterpret_cast<float4*>(t + r * C);
#pragma unroll
        for (int k = 0; k < LN_MAXQ; ++k) {
            const int q = lane + 64 * k;
            if (q < nq) {
                const float4 xv = xr[q], dv = dr[q], gm = reinterpret_cast<const float4*>(gamma)[q];
                xh[k] = make_float4((xv.x - mf) * rs, (xv.y - mf) * rs, (xv.z - mf) * rs, (xv.w - mf) * rs);
                g[k] = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
                tr[q] = make_float4(dv.x * xh[k].x, dv.y * xh[k].y, dv.z * xh[k].z, dv.w * xh[k].w);
                s1 += (double)g[k].x + (double)g[k].y + (double)g[k].z + (double)g[k].w;
                s2 += (double)g[k].x * xh[k].x + (double)g[k].y * xh[k].y + (double)g[k].z * xh[k].z + (double)g[k].w * xh[k].w;
            }
        }
        const float m1 = (float)(wave_sum_d(s1) / C), m2 = (float)(wave_sum_d(s2) / C);
        float4* dxr = reinterpret_cast<float4*>(dx + r * C);
#pragma unroll
        for (int k = 0; k < LN_MAXQ; ++k) {
            const int q = lane + 64 * k;
            if (q < nq)
                dxr[q] = make_float4(rs * (g[k].x - m1 - xh[k].x * m2), rs * (g[k].y - m1 - xh[k].y * m2),
                                     rs * (g[k].z - m1 - xh[k].z * m2), rs * (g[k].w - m1 - xh[k].w * m2));
        }
    }
}

// counter-based keep mask: 32-bit mix of (element index, seed); keep <=> hash >= p * 2^32.  The oracle restates the same
// integer arithmetic (oracle/favae_oracle.py: dropout_keep), so train-mode parity is testable without torch's RNG stream.
__device__ __forceinline__ unsigned mix32(unsigned idx, unsigned seed) {
    unsigned h = idx * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x21F0AAADu;
    h ^= h >> 15; h *= 0x735A2D97u;
    h ^= h >> 15;
    return h;
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                      float* __restrict__ y, long n, unsigned thresh, float scale, unsigned seed) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        bool keep = thresh == 0u || mix32((unsigned)i, seed) >= thresh;
        if (gate) keep = keep && gate[i] > 0.f;
        y[i] = keep ? x[i] * scale : 0.f;
    }
}

int ew_grid(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int favae_affine_rows(const float* x, const float* scale, const float* shift, float* y, int N, int64_t HW, int C, int act,
                                 favae_stream_t stream) {
    FAVAE_REQUIRE(x && scale && shift && y && N > 0 && HW > 0 && C > 0);
    if (C % 4 || ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)scale) | ((uintptr_t)shift)) & 15)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    const long total4 = (long)N * HW * (C / 4);
    FAVAE_KLAUNCH(affine_rows_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (const float4*)scale, (const float4*)shift, (float4*)y, total4, (long)HW * (C / 4), C / 4, act);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                   int64_t rows, int C, float eps, favae_stream_t stream) {
    FAVAE_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0 && C > 0);
    if (C % 4 || C > 256 * LN_MAXQ || ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15))
        return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    const long blocks = (rows + 3) / 4;
    FAVAE_KLAUNCH(layernorm_fwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, x,
                       gamma, beta, y, mean, rstd, (long)rows, C, eps);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                   float* dx, float* dy_xhat, int64_t rows, int C, favae_stream_t stream) {
    FAVAE_REQUIRE(dy && x && gamma && mean && rstd && dx && dy_xhat && rows > 0 && C > 0);
    if (C % 4 || C > 256 * LN_MAXQ ||
        ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)dy_xhat) | ((uintptr_t)gamma)) & 15))
        return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    const long blocks = (rows + 3) / 4;
    FAVAE_KLAUNCH(layernorm_bwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, dy, x,
                       gamma, mean, rstd, dx, dy_xhat, (long)rows, C);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_dropout(const float* x, const float* gate, float* y, int64_t n, float p, uint32_t seed, favae_stream_t stream) {
    FAVAE_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f);
    if (n >= ((int64_t)1 << 32)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);          // the mask hashes a 32-bit element index
    const double t = (double)p * 4294967296.0;
    const unsigned thresh = p > 0.f ? (unsigned)(t > 4294967295.0 ? 4294967295.0 : t) : 0u;
    const float scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    FAVAE_KLAUNCH(dropout_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, gate, y, (long)n, thresh, scale, seed);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}
