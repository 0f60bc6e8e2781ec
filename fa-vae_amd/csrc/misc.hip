// Streaming reductions and element-wise glue of the training step (all HBM-bound, 16-byte accesses, grid-stride):
//   L1 reconstruction loss and its gradient        favae_scripts/train_favae.py:76
//   commitment (MSE) loss + straight-through bwd   models/l2_quantize.py:554,560-561
//   Adam(betas=(0.5,0.9)) over one flat buffer     favae_scripts/train_favae.py:297-305
//   NCHW <-> NHWC conversion at the module boundary
// Scalar losses are produced deterministically: per-block fp64 partials -> one finishing block.
#include "common.h"

namespace {

constexpr int RED_BLOCKS = 1024;

template <int SQ>
__global__ __launch_bounds__(256) void diff_reduce_kernel(const float* a, const float* b, long n, double* part) {
    __shared__ double red[4];
    double acc = 0.0;
    const long n4 = n / 4;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    const bool vec = ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0;
    if (vec) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            const float4 x = a4[i], y = b4[i];
            const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            if (SQ) acc += (double)(d0 * d0 + d1 * d1) + (double)(d2 * d2 + d3 * d3);
            else acc += (double)(fabsf(d0) + fabsf(d1)) + (double)(fabsf(d2) + fabsf(d3));
        }
        for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            const float d = a[i] - b[i];
            acc += SQ ? (double)(d * d) : (double)fabsf(d);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            const float d = a[i] - b[i];
            acc += SQ ? (double)(d * d) : (double)fabsf(d);
        }
    }
    const double tot = block_sum_d256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void finish_kernel(const double* part, int nparts, double scale, float* out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    const double tot = block_sum_d256(acc, red);
    if (threadIdx.x == 0) out[0] = (float)(tot * scale);
}

// MODE 0: g*scale*sign(a-b) ; MODE 1: g*scale*(a-b)
template <int MODE>
__global__ __launch_bounds__(256) void diff_bwd_kernel(const float* a, const float* b, const float* g, float scale, long n,
                                                       const float* add, float* out) {
    const float gs = g[0] * scale;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = a[i] - b[i];
        float v;
        if (MODE == 0) v = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
        else v = gs * d;
        if (add) v += add[i];
        out[i] = v;
    }
}

// straight-through estimator value exactly as the reference forms it: out = x + (q - x)   (models/l2_quantize.py:554)
__global__ __launch_bounds__(256) void ste_kernel(const float* x, const float* q, float* out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] + (q[i] - x[i]);
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* x, float alpha, float* y, float beta, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        y[i] = (beta == 0.f) ? alpha * x[i] : fmaf(alpha, x[i], beta * y[i]);
}

// tiled transpose between [C][HW] and [HW][C] per image
__global__ __launch_bounds__(256) void transpose_kernel(const float* x, float* y, int rows, int cols) {
    // x: [rows][cols] -> y: [cols][rows], batched over blockIdx.z
    __shared__ float tile[32][33];
    const size_t off = (size_t)blockIdx.z * rows * cols;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = x[off + (size_t)(r0 + r) * cols + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < cols && r0 + tx < rows) y[off + (size_t)(c0 + c) * rows + r0 + tx] = tile[tx][c];
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1,
                                                   float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

int ew_blocks(long n) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

template <int SQ>
int diff_sum(const float* a, const float* b, int64_t n, float scale, float* loss, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!(a && b && loss && ws && n > 0)) return favae_prof_fail_(FAVAE_ERR_BAD_ARG);
    if (ws_bytes < RED_BLOCKS * sizeof(double)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    long nb = (n / 4 + 255) / 256;
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    if (nb < 1) nb = 1;
    FAVAE_KLAUNCH((diff_reduce_kernel<SQ>), dim3((int)nb), dim3(256), 0, s, a, b, (long)n, (double*)ws);
    FAVAE_CHECK_LAUNCH();
    FAVAE_KLAUNCH(finish_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, (int)nb, (double)scale, loss);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

// hinge GAN terms (losses/hinge.py:5-14) as means of one logits tensor: MODE 0: -x (generator), 1: relu(1 - x) (real),
// 2: relu(1 + x) (fake).  Forward: fp64 block partials -> finish_kernel; backward: d/dx * g * scale.
__device__ __forceinline__ float hinge_val(float x, int mode) {
    return mode == 0 ? -x : (mode == 1 ? fmaxf(1.f - x, 0.f) : fmaxf(1.f + x, 0.f));
}
__global__ __launch_bounds__(256) void hinge_reduce_kernel(const float* x, long n, int mode, double* part) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += (double)hinge_val(x[i], mode);
    const double tot = block_sum_d256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void hinge_bwd_kernel(const float* x, const float* g, float scale, long n, int mode, float* dx) {
    const float gs = g[0] * scale;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        const float d = mode == 0 ? -1.f : (mode == 1 ? (1.f - v > 0.f ? -1.f : 0.f) : (1.f + v > 0.f ? 1.f : 0.f));
        dx[i] = d * gs;
    }
}

// dx = da * act'(x) for an activation applied without normalisation (first LeakyReLU of the discriminator)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* da, const float* x, int act, long n, float* dx) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        float d = 1.f;
        if (act == FAVAE_ACT_SILU) {
            const float sg = 1.f / (1.f + __expf(-v));
            d = sg * (1.f + v * (1.f - sg));
        } else if (act == FAVAE_ACT_LEAKY02) {
            d = v > 0.f ? 1.f : 0.2f;
        } else if (act == FAVAE_ACT_RELU) {
            d = v > 0.f ? 1.f : 0.f;
        }
        dx[i] = da[i] * d;
    }
}

// uint8 HWC pixels -> normalised fp32 NHWC: y = (u/255 - mean[c]) / std[c], the ToTensor + Normalize tail of the reference's
// input pipeline (datasets/general_dataloader.py:33-38) in the layout the convs consume.  Same fp32 operations in the same
// order as torchvision (true division by 255, subtract, true division) -> bit-identical values.  A thread converts 4 pixels of
// an RGB image (12 bytes in, three float4 out); other channel counts take the scalar loop.
struct U8Norm { float mean[4], std[4]; };
__global__ __launch_bounds__(256) void u8_norm_rgb_kernel(const unsigned* __restrict__ in, float4* __restrict__ out, long groups,
                                                          U8Norm p) {
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long)gridDim.x * 256) {
        const unsigned w0 = in[3 * g], w1 = in[3 * g + 1], w2 = in[3 * g + 2];
        float v[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = (float)((w0 >> (8 * k)) & 255u);
            v[4 + k] = (float)((w1 >> (8 * k)) & 255u);
            v[8 + k] = (float)((w2 >> (8 * k)) & 255u);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = __fdiv_rn(__fsub_rn(__fdiv_rn(v[k], 255.f), p.mean[k % 3]), p.std[k % 3]);
        out[3 * g] = make_float4(v[0], v[1], v[2], v[3]);
        out[3 * g + 1] = make_float4(v[4], v[5], v[6], v[7]);
        out[3 * g + 2] = make_float4(v[8], v[9], v[10], v[11]);
    }
}
__global__ __launch_bounds__(256) void u8_norm_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, long n, int C,
                                                      U8Norm p) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        out[i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)in[i], 255.f), p.mean[c]), p.std[c]);
    }
}

}  // namespace

extern "C" int favae_abi_version(void) { return 22; }

// ---- bf16 activation storage: conversion passes at the boundaries of the kernels that have no bf16 instantiation (round 6) ---------
namespace {
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float4* __restrict__ in, uint2* __restrict__ out, long n4, const float* in1,
                                                        unsigned short* out1, long tail0, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = in[i];
        out[i] = make_uint2(bf16_pack_rne(v.x, v.y), bf16_pack_rne(v.z, v.w));
    }
    if (blockIdx.x == 0)
        for (long i = tail0 + threadIdx.x; i < n; i += 256) out1[i] = bf16_rne(in1[i]);
}
__global__ __launch_bounds__(256) void cast_f32_kernel(const uint2* __restrict__ in, float4* __restrict__ out, long n4,
                                                       const unsigned short* in1, float* out1, long tail0, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const uint2 u = in[i];
        out[i] = make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                             __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
    }
    if (blockIdx.x == 0)
        for (long i = tail0 + threadIdx.x; i < n; i += 256) out1[i] = bf16_up(in1[i]);
}
}  // namespace

extern "C" int favae_cast_bf16(const float* in, void* out, int64_t n, favae_stream_t stream) {
    FAVAE_REQUIRE(in && out && n > 0 && ((((uintptr_t)in) & 15) | (((uintptr_t)out) & 7)) == 0);
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
    FAVAE_PROF_NOTE(0, 6.0 * n);
    FAVAE_KLAUNCH(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)in, (uint2*)out, n4, in,
                  (unsigned short*)out, n4 * 4, (long)n);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}
extern "C" int favae_cast_f32(const void* in, float* out, int64_t n, favae_stream_t stream) {
    FAVAE_REQUIRE(in && out && n > 0 && ((((uintptr_t)in) & 7) | (((uintptr_t)out) & 15)) == 0);
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
    FAVAE_PROF_NOTE(0, 6.0 * n);
    FAVAE_KLAUNCH(cast_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint2*)in, (float4*)out, n4,
                  (const unsigned short*)in, out, n4 * 4, (long)n);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_u8_to_float_nhwc(const unsigned char* in, float* out, int64_t pixels, int C, const float* mean, const float* std,
                                      favae_stream_t stream) {
    FAVAE_REQUIRE(in && out && pixels > 0 && C >= 1 && C <= 4 && mean && std);
    U8Norm p{};
    for (int c = 0; c < C; ++c) {
        FAVAE_REQUIRE(std[c] != 0.f);
        p.mean[c] = mean[c];
        p.std[c] = std[c];
    }
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)pixels * C;
    long done = 0;
    if (C == 3 && ((((uintptr_t)in) & 3) == 0) && ((((uintptr_t)out) & 15) == 0) && pixels >= 4) {
        const long groups = pixels / 4;
        FAVAE_KLAUNCH(u8_norm_rgb_kernel, dim3(ew_blocks(groups)), dim3(256), 0, s, (const unsigned*)in, (float4*)out, groups, p);
        FAVAE_CHECK_LAUNCH();
        done = groups * 12;
    }
    if (done < n) {
        // the remainder starts at a multiple of 12 elements, so channel phase 0 is preserved
        FAVAE_KLAUNCH(u8_norm_kernel, dim3(ew_blocks(n - done)), dim3(256), 0, s, in + done, out + done, n - done, C, p);
        FAVAE_CHECK_LAUNCH();
    }
    return FAVAE_OK;
}

extern "C" int favae_hinge_mean(const float* x, int64_t n, int mode, float* loss, void* ws, size_t ws_bytes,
                                favae_stream_t stream) {
    FAVAE_REQUIRE(x && loss && ws && n > 0 && mode >= 0 && mode <= 2);
    if (ws_bytes < RED_BLOCKS * sizeof(double)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    long nb = (n + 255) / 256;
    nb = nb > RED_BLOCKS ? RED_BLOCKS : nb;
    hipStream_t s = (hipStream_t)stream;
    FAVAE_KLAUNCH(hinge_reduce_kernel, dim3((int)nb), dim3(256), 0, s, x, (long)n, mode, (double*)ws);
    FAVAE_CHECK_LAUNCH();
    FAVAE_KLAUNCH(finish_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, (int)nb, 1.0 / (double)n, loss);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_hinge_mean_bwd(const float* x, const float* g, int64_t n, int mode, float* dx, favae_stream_t stream) {
    FAVAE_REQUIRE(x && g && dx && n > 0 && mode >= 0 && mode <= 2);
    FAVAE_KLAUNCH(hinge_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, g, (float)(1.0 / (double)n),
                       (long)n, mode, dx);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_act_bwd(const float* da, const float* x, int act, int64_t n, float* dx, favae_stream_t stream) {
    FAVAE_REQUIRE(da && x && dx && n > 0);
    FAVAE_KLAUNCH(act_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, da, x, act, (long)n, dx);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" size_t favae_reduce_workspace(int64_t n) { (void)n; return RED_BLOCKS * sizeof(double); }

extern "C" int favae_absdiff_sum(const float* a, const float* b, int64_t n, float scale, float* loss, void* ws, size_t ws_bytes,
                                 favae_stream_t stream) {
    return diff_sum<0>(a, b, n, scale, loss, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int favae_sqdiff_sum(const float* a, const float* b, int64_t n, float scale, float* loss, void* ws, size_t ws_bytes,
                                favae_stream_t stream) {
    return diff_sum<1>(a, b, n, scale, loss, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int favae_absdiff_bwd(const float* a, const float* b, const float* g, float scale, int64_t n, const float* out_add,
                                 float* out, favae_stream_t stream) {
    FAVAE_REQUIRE(a && b && g && out && n > 0);
    FAVAE_KLAUNCH((diff_bwd_kernel<0>), dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, g, scale, (long)n,
                       out_add, out);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}
extern "C" int favae_sqdiff_bwd(const float* a, const float* b, const float* g, float scale, int64_t n, const float* out_add,
                                float* out, favae_stream_t stream) {
    FAVAE_REQUIRE(a && b && g && out && n > 0);
    FAVAE_KLAUNCH((diff_bwd_kernel<1>), dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, g, scale, (long)n,
                       out_add, out);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_vq_ste(const float* x, const float* q, float* out, int64_t n, favae_stream_t stream) {
    FAVAE_REQUIRE(x && q && out && n > 0);
    FAVAE_KLAUNCH(ste_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, q, out, (long)n);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_axpby(const float* x, float alpha, float* y, float beta, int64_t n, favae_stream_t stream) {
    FAVAE_REQUIRE(x && y && n > 0);
    FAVAE_KLAUNCH(axpby_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, alpha, y, beta, (long)n);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, favae_stream_t stream) {
    FAVAE_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0);
    const int rows = C, cols = H * W;     // [C][HW] -> [HW][C]
    FAVAE_KLAUNCH(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), N), dim3(256), 0, (hipStream_t)stream, x, y, rows,
                       cols);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}
extern "C" int favae_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, favae_stream_t stream) {
    FAVAE_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0);
    const int rows = H * W, cols = C;     // [HW][C] -> [C][HW]
    FAVAE_KLAUNCH(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), N), dim3(256), 0, (hipStream_t)stream, x, y, rows,
                       cols);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, int step, float grad_scale, favae_stream_t stream) {
    FAVAE_REQUIRE(p && g && m && v && n > 0 && step >= 1);
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    FAVAE_PROF_NOTE(0, 28.0 * n);                                          // reads p, g, m, v; writes p, m, v
    FAVAE_KLAUNCH(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, beta1, beta2,
                       eps, (float)bc1, (float)sqrt(bc2), grad_scale);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}
