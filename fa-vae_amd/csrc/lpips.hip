// LPIPS perceptual distance (reference losses/lpips.py:17-110) -- the pieces that are not convolutions.
//
// The VGG16 feature stack runs on the fused conv kernels with ReLU applied on the operand load of the NEXT conv
// (FAVAE_ACT_RELU), so every feature tensor in HBM is the PRE-activation output of its conv; the kernels here apply the
// ReLU themselves where the reference reads a post-ReLU tensor:
//   * 2x2/2 max pooling (torchvision vgg16 features[4,9,16,23]): max commutes with the monotone ReLU, so pooling the
//     pre-activations and letting the next conv's load apply ReLU equals MaxPool(ReLU(.)) (forward AND gradient: if the
//     window maximum is <= 0 the ReLU derivative kills the gradient either way; otherwise the arg-max is the same element);
//   * one LPIPS level (losses/lpips.py:44-48): d = sum_c w_c (a^_c - b^_c)^2 per pixel with a^ = relu(a)/max(|relu(a)|_2, 1e-12)
//     (F.normalize over channels), spatial mean per image, summed over the 5 levels -> fused into one pass over a and b
//     (algorithmic bytes: 8*C per pixel forward, 12*C per pixel backward; HBM-bound);
//   * ScalingLayer (losses/lpips.py:55-62): (x - shift_c) / scale_c on the 3-channel image.
// Layout: NHWC fp32, C % 4 == 0 for the feature kernels.  Reductions are block-partial -> fixed-order finish: deterministic.
#include "common.h"

namespace {

constexpr int LP_MAX_CHUNKS = 64;     // workgroups per image in the level kernels

template <int C>
struct LpGeom {
    static constexpr int LP = C / 4 > 64 ? 64 : C / 4;      // lanes per pixel
    static constexpr int V = C / (4 * LP);                 // float4 per lane
    static constexpr int PPW = 64 / LP;                    // pixels per wave per iteration
};

template <int LP>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LP >= 16) return group_sum_xor<LP>(v);             // DPP butterfly, same operand order (common.h)
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float4 relu4(float4 v) {
    return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// a^ - b^ with both products rounded before the subtraction (no fma contraction): identical inputs give exactly 0,
// as F.normalize(a) - F.normalize(b) does in the reference
__device__ __forceinline__ float ndiff(float ra, float na, float rb, float nb) {
#pragma clang fp contract(off)
    const float x = ra * na;
    const float y = rb * nb;
    return x - y;
}

// MODE 0: part[n][chunk] = sum over the chunk's pixels of sum_c w_c (a^_c - b^_c)^2
// MODE 1: db = d(level value)/d(b) * g[n] / HW   (b is the PRE-activation feature of the image that carries the gradient)
template <int C, int MODE>
__global__ __launch_bounds__(256) void lpips_level_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ w, const float* __restrict__ g, int HW,
                                                          int chunks, double* __restrict__ part, float* __restrict__ db) {
    using G = LpGeom<C>;
    constexpr int LP = G::LP, V = G::V, PPW = G::PPW;
    __shared__ double red[4];
    const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int sub = lane / LP, cl = lane % LP;             // pixel slot inside the wave, channel lane
    float4 wv[V];
#pragma unroll
    for (int v = 0; v < V; ++v) wv[v] = *reinterpret_cast<const float4*>(w + 4 * (cl + LP * v));
    const float gs = MODE == 1 ? g[n] / (float)HW : 0.f;
    const size_t img = (size_t)n * HW;
    double acc = 0.0;
    for (int p = (chunk * 4 + wid) * PPW + sub; p < HW; p += chunks * 4 * PPW) {
        const size_t base = (img + p) * C;
        float4 ra[V], rb[V], pb[V];
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = 4 * (cl + LP * v);
            pb[v] = *reinterpret_cast<const float4*>(b + base + c);
            ra[v] = relu4(*reinterpret_cast<const float4*>(a + base + c));
            rb[v] = relu4(pb[v]);
            sa += dot4(ra[v], ra[v]);
            sb += dot4(rb[v], rb[v]);
        }
        sa = group_sum<LP>(sa);
        sb = group_sum<LP>(sb);
        const float na = 1.0f / fmaxf(sqrtf(sa), 1e-12f), nb = 1.0f / fmaxf(sqrtf(sb), 1e-12f);
        if (MODE == 0) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float4 d = make_float4(ndiff(ra[v].x, na, rb[v].x, nb), ndiff(ra[v].y, na, rb[v].y, nb),
                                             ndiff(ra[v].z, na, rb[v].z, nb), ndiff(ra[v].w, na, rb[v].w, nb));
                s += wv[v].x * d.x * d.x + wv[v].y * d.y * d.y + wv[v].z * d.z * d.z + wv[v].w * d.w * d.w;
            }
            acc += (double)s;
        } else {
            // u = d value / d b^ = -2 w (a^ - b^) gs ;  d r = nb (u - b^ <u, b^>)  (norm above the clamp) or u * 1e12 (below)
            float4 u[V], bh[V];
            float ub = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                bh[v] = make_float4(rb[v].x * nb, rb[v].y * nb, rb[v].z * nb, rb[v].w * nb);
                u[v] = make_float4(-2.f * gs * wv[v].x * ndiff(ra[v].x, na, rb[v].x, nb), -2.f * gs * wv[v].y * ndiff(ra[v].y, na, rb[v].y, nb),
                                   -2.f * gs * wv[v].z * ndiff(ra[v].z, na, rb[v].z, nb), -2.f * gs * wv[v].w * ndiff(ra[v].w, na, rb[v].w, nb));
                ub += dot4(u[v], bh[v]);
            }
            ub = group_sum<LP>(ub);
            if (sqrtf(sb) < 1e-12f) ub = 0.f;               // clamped norm: the denominator is a constant
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float4 o;
                o.x = pb[v].x > 0.f ? nb * (u[v].x - bh[v].x * ub) : 0.f;
                o.y = pb[v].y > 0.f ? nb * (u[v].y - bh[v].y * ub) : 0.f;
                o.z = pb[v].z > 0.f ? nb * (u[v].z - bh[v].z * ub) : 0.f;
                o.w = pb[v].w > 0.f ? nb * (u[v].w - bh[v].w * ub) : 0.f;
                *reinterpret_cast<float4*>(db + base + 4 * (cl + LP * v)) = o;
            }
        }
    }
    if (MODE == 0) {
        const double s = block_sum_d256(acc, red);
        if (threadIdx.x == 0) part[blockIdx.x] = s;
    }
}

__global__ void lpips_finish_kernel(const double* __restrict__ part, int N, int chunks, double inv_hw, int accumulate,
                                    float* __restrict__ val) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += part[(size_t)n * chunks + c];
    const float v = (float)(s * inv_hw);
    val[n] = accumulate ? val[n] + v : v;
}

// thread = one float4 of the pooled tensor
template <int BWD>
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out,
                                                       int H, int W, int C4, size_t total) {
    const int Ho = H / 2, Wo = W / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const size_t n = r / Ho;
        const float4* xp = reinterpret_cast<const float4*>(x) + ((n * H + 2 * ho) * W + 2 * wo) * C4 + c;
        const float4 v00 = xp[0], v01 = xp[C4], v10 = xp[(size_t)W * C4], v11 = xp[(size_t)W * C4 + C4];
        if (!BWD) {
            // first maximum in scan order wins, NaN propagates (aten max_pool2d: `val > max || isnan(val)`)
            auto mx = [](float a, float b2, float c2, float d) {
                float m = a;
                if (b2 > m || b2 != b2) m = b2;
                if (c2 > m || c2 != c2) m = c2;
                if (d > m || d != d) m = d;
                return m;
            };
            reinterpret_cast<float4*>(out)[i] = make_float4(mx(v00.x, v01.x, v10.x, v11.x), mx(v00.y, v01.y, v10.y, v11.y),
                                                            mx(v00.z, v01.z, v10.z, v11.z), mx(v00.w, v01.w, v10.w, v11.w));
        } else {
            const float4 g = reinterpret_cast<const float4*>(dy)[i];
            float4 o00, o01, o10, o11;
            auto route = [](float a, float b2, float c2, float d, float gg, float& oa, float& ob, float& oc, float& od) {
                int k = 0;
                float m = a;
                if (b2 > m || b2 != b2) { m = b2; k = 1; }
                if (c2 > m || c2 != c2) { m = c2; k = 2; }
                if (d > m || d != d) { m = d; k = 3; }
                oa = k == 0 ? gg : 0.f; ob = k == 1 ? gg : 0.f; oc = k == 2 ? gg : 0.f; od = k == 3 ? gg : 0.f;
            };
            route(v00.x, v01.x, v10.x, v11.x, g.x, o00.x, o01.x, o10.x, o11.x);
            route(v00.y, v01.y, v10.y, v11.y, g.y, o00.y, o01.y, o10.y, o11.y);
            route(v00.z, v01.z, v10.z, v11.z, g.z, o00.z, o01.z, o10.z, o11.z);
            route(v00.w, v01.w, v10.w, v11.w, g.w, o00.w, o01.w, o10.w, o11.w);
            float4* op = reinterpret_cast<float4*>(out) + ((n * H + 2 * ho) * W + 2 * wo) * C4 + c;
            op[0] = o00; op[C4] = o01; op[(size_t)W * C4] = o10; op[(size_t)W * C4 + C4] = o11;
        }
    }
}

// y = (x - shift_c) / scale_c   (shift == nullptr: y = x / scale_c -- the gradient of the forward)
__global__ __launch_bounds__(256) void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                                             const float* __restrict__ scale, size_t n, int C, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        y[i] = (x[i] - (shift ? shift[c] : 0.f)) / scale[c];
    }
}

int level_chunks(int HW, int C) {
    const int ppi = 4 * (64 / (C / 4 > 64 ? 64 : C / 4));   // pixels per workgroup iteration
    int ch = (HW + ppi * 4 - 1) / (ppi * 4);                // >= 4 iterations per workgroup
    return ch < 1 ? 1 : (ch > LP_MAX_CHUNKS ? LP_MAX_CHUNKS : ch);
}

bool level_c_ok(int C) { return C == 64 || C == 128 || C == 256 || C == 512; }

}  // namespace

extern "C" size_t favae_lpips_level_workspace(int N) { return (size_t)(N > 0 ? N : 0) * LP_MAX_CHUNKS * sizeof(double); }

extern "C" int favae_lpips_level(const float* a, const float* b, const float* w, int N, int HW, int C, float* val, int accumulate,
                                 void* ws, size_t ws_bytes, favae_stream_t stream) {
    FAVAE_REQUIRE(a && b && w && val && ws && N > 0 && HW > 0 && level_c_ok(C));
    if (ws_bytes < favae_lpips_level_workspace(N)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    const int chunks = level_chunks(HW, C);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(N * chunks));
#define FAVAE_LP(CC) FAVAE_KLAUNCH((lpips_level_kernel<CC, 0>), grid, dim3(256), 0, s, a, b, w, (const float*)nullptr, HW, chunks, (double*)ws, (float*)nullptr)
    if (C == 64) FAVAE_LP(64);
    else if (C == 128) FAVAE_LP(128);
    else if (C == 256) FAVAE_LP(256);
    else FAVAE_LP(512);
#undef FAVAE_LP
    FAVAE_CHECK_LAUNCH();
    FAVAE_KLAUNCH(lpips_finish_kernel, dim3(cdiv(N, 64)), dim3(64), 0, s, (const double*)ws, N, chunks, 1.0 / (double)HW,
                       accumulate, val);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_lpips_level_bwd(const float* a, const float* b, const float* w, const float* g, int N, int HW, int C, float* db,
                                     favae_stream_t stream) {
    FAVAE_REQUIRE(a && b && w && g && db && N > 0 && HW > 0 && level_c_ok(C));
    const int chunks = level_chunks(HW, C);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(N * chunks));
#define FAVAE_LP(CC) FAVAE_KLAUNCH((lpips_level_kernel<CC, 1>), grid, dim3(256), 0, s, a, b, w, g, HW, chunks, (double*)nullptr, db)
    if (C == 64) FAVAE_LP(64);
    else if (C == 128) FAVAE_LP(128);
    else if (C == 256) FAVAE_LP(256);
    else FAVAE_LP(512);
#undef FAVAE_LP
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_maxpool2(const float* x, int N, int H, int W, int C, float* y, favae_stream_t stream) {
    FAVAE_REQUIRE(x && y && N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    const size_t nb = (total + 255) / 256;
    FAVAE_KLAUNCH((maxpool2_kernel<0>), dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, (hipStream_t)stream, x,
                       (const float*)nullptr, y, H, W, C / 4, total);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_maxpool2_bwd(const float* x, const float* dy, int N, int H, int W, int C, float* dx, favae_stream_t stream) {
    FAVAE_REQUIRE(x && dy && dx && N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0);
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    const size_t nb = (total + 255) / 256;
    FAVAE_KLAUNCH((maxpool2_kernel<1>), dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, (hipStream_t)stream, x, dy,
                       dx, H, W, C / 4, total);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_channel_affine(const float* x, const float* shift, const float* scale, int64_t n, int C, float* y,
                                    favae_stream_t stream) {
    FAVAE_REQUIRE(x && scale && y && n > 0 && C > 0 && n % C == 0);
    const size_t nb = ((size_t)n + 255) / 256;
    FAVAE_KLAUNCH(channel_affine_kernel, dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, (hipStream_t)stream, x, shift,
                       scale, (size_t)n, C, y);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}
