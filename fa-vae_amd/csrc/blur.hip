// Learnable-sigma Gaussian blur of the Frequency Complement Module taps (models/codec.py:255-277, models/vqgan_fcm.py:20-41):
//   g = exp(-0.5 (t/sigma)^2) / sum,  t = linspace(-(k-1)/2, (k-1)/2, k);   K = g g^T;   y = K (*) reflectpad(x), depthwise.
// NHWC, HBM-bound.  A block owns a TH x TW pixel tile x CC channels of one image; the (TH+k-1) x (TW+k-1) x CC halo tile is
// staged once in LDS with the reflect indexing applied while loading (forward, dK) or zero-extended (dx adjoint), lanes run
// along channels (contiguous, conflict-free LDS reads), each thread produces 16 pixels of one channel.
//   dx     = adjoint(reflect-pad o correlation): every padded position j that reflects onto i contributes
//            Z[j] = sum_ab K[a][b] dy0[j+p-a]  (dy0 = dy zero-extended)  -> dx[i] = sum_{j in pre(i)} Z[j]
//   dsigma : dK[a][b] = sum dy * xpad[.+a,.+b]  (k^2 block-reduced partials -> deterministic column sum) then
//            dg_a = sum_b (dK[a][b]+dK[b][a]) g_b ; dp = (dg - <dg,g>)/sum(p) ; dsigma = sum_j dp_j p_j t_j^2 / sigma^3
#include "common.h"

extern "C" size_t favae_colsum_workspace(int64_t M, int C);
extern "C" int favae_colsum(const float* a, float* out, int64_t M, int C, void* ws, size_t ws_bytes, favae_stream_t stream);

namespace {

constexpr int MAXK = 31;

struct BlurArgs {
    const float* x;
    const float* dy;
    const float* sigma;
    float* out;       // y / dx / dK partials
    int N, H, W, C, k, TH, TW, CC, tiles_h, tiles_w, cchunks;
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ void make_kernel1d(const float* sigma, int k, float* g /*LDS, k floats*/) {
    if (threadIdx.x == 0) {
        const float s = sigma[0];
        const float half = (k - 1) * 0.5f;
        float sum = 0.f;
        for (int j = 0; j < k; ++j) {
            const float t = (float)j - half;
            const float q = t / s;
            const float p = expf(-0.5f * (q * q));
            g[j] = p;
            sum += p;
        }
        for (int j = 0; j < k; ++j) g[j] = g[j] / sum;
    }
}

// MODE 0: forward; MODE 1: dx adjoint; MODE 2: dK partials
template <int MODE>
__global__ __launch_bounds__(256) void blur_kernel(BlurArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* g = sm;                         // [32]
    float* red = sm + 32;                  // [4] cross-wave scratch (MODE 2)
    float* halo = sm + 64;                 // [(TH+k-1)][(TW+k-1)][CC]
    const int k = a.k, p = k / 2;
    const int HH = a.TH + k - 1, HW = a.TW + k - 1;
    int b = blockIdx.x;
    const int cchunk = b % a.cchunks; b /= a.cchunks;
    const int tw = b % a.tiles_w; b /= a.tiles_w;
    const int th = b % a.tiles_h;
    const int n = b / a.tiles_h;
    const int y0 = th * a.TH, x0 = tw * a.TW, c0 = cchunk * a.CC;
    const int CC = a.CC;
    const int tid = threadIdx.x;

    make_kernel1d(a.sigma, k, g);
    // ---- stage the halo tile ------------------------------------------------------------------------------------
    const float* src = (MODE == 1) ? a.dy : a.x;
    const float* img = src + (size_t)n * a.H * a.W * a.C;
    const int total = HH * HW * CC;
    for (int i = tid; i < total; i += 256) {
        const int c = i % CC;
        const int hx = (i / CC) % HW;
        const int hy = i / (CC * HW);
        int sy = y0 + hy - p, sx = x0 + hx - p;
        float v = 0.f;
        if (c0 + c < a.C) {
            if (MODE == 1) {
                if (sy >= 0 && sy < a.H && sx >= 0 && sx < a.W) v = img[((size_t)sy * a.W + sx) * a.C + c0 + c];
            } else {
                sy = reflect_idx(sy, a.H);
                sx = reflect_idx(sx, a.W);
                v = img[((size_t)sy * a.W + sx) * a.C + c0 + c];
            }
        }
        halo[i] = v;
    }
    __syncthreads();

    const int c = tid % CC, pl = tid / CC, PL = 256 / CC;
    const bool c_ok = (c0 + c < a.C) && pl < PL;
    const int npix = a.TH * a.TW;

    if (MODE == 0) {
        if (!c_ok) return;
        float* out = a.out + (size_t)n * a.H * a.W * a.C;
        for (int q = pl; q < npix; q += PL) {
            const int py = q / a.TW, px = q % a.TW;
            if (y0 + py >= a.H || x0 + px >= a.W) continue;
            float acc = 0.f;
            for (int u = 0; u < k; ++u) {
                const float gu = g[u];
                const float* row = halo + ((py + u) * HW + px) * CC + c;
                for (int v = 0; v < k; ++v) acc = fmaf(gu * g[v], row[v * CC], acc);
            }
            out[((size_t)(y0 + py) * a.W + x0 + px) * a.C + c0 + c] = acc;
        }
    } else if (MODE == 1) {
        if (!c_ok) return;
        float* out = a.out + (size_t)n * a.H * a.W * a.C;
        for (int q = pl; q < npix; q += PL) {
            const int py = q / a.TW, px = q % a.TW;
            const int y = y0 + py, x = x0 + px;
            if (y >= a.H || x >= a.W) continue;
            // padded-domain positions that reflect onto (y, x)
            int jy[3], jx[3], ny = 0, nx = 0;
            jy[ny++] = y;
            if (y >= 1 && y <= p) jy[ny++] = -y;
            if (y <= a.H - 2 && y >= a.H - 1 - p) jy[ny++] = 2 * (a.H - 1) - y;
            jx[nx++] = x;
            if (x >= 1 && x <= p) jx[nx++] = -x;
            if (x <= a.W - 2 && x >= a.W - 1 - p) jx[nx++] = 2 * (a.W - 1) - x;
            float acc = 0.f;
            for (int iy = 0; iy < ny; ++iy)
                for (int ix = 0; ix < nx; ++ix) {
                    for (int u = 0; u < k; ++u) {
                        const int ry = jy[iy] + p - u;                   // dy row (image coords)
                        if (ry < 0 || ry >= a.H) continue;
                        const int hy = ry - (y0 - p);
                        if (hy < 0 || hy >= HH) continue;
                        const float gu = g[u];
                        for (int v = 0; v < k; ++v) {
                            const int rx = jx[ix] + p - v;
                            if (rx < 0 || rx >= a.W) continue;
                            const int hx = rx - (x0 - p);
                            if (hx < 0 || hx >= HW) continue;
                            acc = fmaf(gu * g[v], halo[(hy * HW + hx) * CC + c], acc);
                        }
                    }
                }
            out[((size_t)y * a.W + x) * a.C + c0 + c] = acc;
        }
    } else {
        // dK[u][v] partial of this block: sum over its pixels/channels of dy * xpad
        constexpr int MAXPIX = 16;
        float dyv[MAXPIX];
        const float* dimg = a.dy + (size_t)n * a.H * a.W * a.C;
        int cnt = 0;
        for (int q = pl; q < npix && cnt < MAXPIX; q += PL, ++cnt) {
            const int py = q / a.TW, px = q % a.TW;
            const bool ok = c_ok && (y0 + py < a.H) && (x0 + px < a.W);
            dyv[cnt] = ok ? dimg[((size_t)(y0 + py) * a.W + x0 + px) * a.C + c0 + c] : 0.f;
        }
        float* out = a.out + (size_t)blockIdx.x * (k * k);
        const int lane = tid & 63, wid = tid >> 6;
        for (int u = 0; u < k; ++u)
            for (int v = 0; v < k; ++v) {
                float acc = 0.f;
                if (c_ok) {
                    int i = 0;
                    for (int q = pl; q < npix && i < MAXPIX; q += PL, ++i) {
                        const int py = q / a.TW, px = q % a.TW;
                        acc = fmaf(dyv[i], halo[((py + u) * HW + px + v) * CC + c], acc);
                    }
                }
                acc = wave_sum(acc);
                __syncthreads();
                if (lane == 0) red[wid] = acc;
                __syncthreads();
                if (tid == 0) out[u * k + v] = (red[0] + red[1]) + (red[2] + red[3]);
            }
    }
}

// dK (k*k floats, already summed over blocks) -> dsigma
__global__ void blur_dsigma_kernel(const float* dK, const float* sigma, int k, float* dsigma) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double s = sigma[0];
    const double half = (k - 1) * 0.5;
    double pz[MAXK], g[MAXK], dg[MAXK];
    double S = 0.0;
    for (int j = 0; j < k; ++j) {
        const double t = j - half;
        pz[j] = exp(-0.5 * (t / s) * (t / s));
        S += pz[j];
    }
    for (int j = 0; j < k; ++j) g[j] = pz[j] / S;
    double dot = 0.0;
    for (int u = 0; u < k; ++u) {
        double acc = 0.0;
        for (int v = 0; v < k; ++v) acc += ((double)dK[u * k + v] + (double)dK[v * k + u]) * g[v];
        dg[u] = acc;
        dot += acc * g[u];
    }
    double ds = 0.0;
    for (int j = 0; j < k; ++j) {
        const double t = j - half;
        const double dp = (dg[j] - dot) / S;
        ds += dp * pz[j] * t * t / (s * s * s);
    }
    dsigma[0] = (float)ds;
}

void plan(int ksize, int N, int H, int W, int C, BlurArgs& a) {
    a.N = N; a.H = H; a.W = W; a.C = C; a.k = ksize;
    a.CC = C >= 32 ? 32 : (C >= 16 ? 16 : (C >= 8 ? 8 : 4));
    const int PL = 256 / a.CC;                 // pixel lanes
    // 16 pixels per thread
    int npix = 16 * PL;                        // 128 @CC=32, 256 @16, 512 @8, 1024 @4
    a.TW = 16;
    a.TH = npix / a.TW;
    if (ksize > 9 && a.TH > 8) a.TH = 8;
    while (a.TH > 8 && a.TH / 2 >= H) a.TH /= 2;
    if (a.TH > 32) a.TH = 32;
    a.tiles_h = (H + a.TH - 1) / a.TH;
    a.tiles_w = (W + a.TW - 1) / a.TW;
    a.cchunks = (C + a.CC - 1) / a.CC;
}

size_t shm_bytes(const BlurArgs& a) {
    return (size_t)(64 + (a.TH + a.k - 1) * (a.TW + a.k - 1) * a.CC) * sizeof(float);
}

bool blur_ok(int ksize, int N, int H, int W, int C) {
    return ksize >= 1 && ksize <= MAXK && (ksize & 1) && N > 0 && C > 0 && H > ksize / 2 && W > ksize / 2;
}

}  // namespace

extern "C" int favae_blur_fwd(const float* x, const float* sigma, int ksize, int N, int H, int W, int C, float* y,
                              favae_stream_t stream) {
    FAVAE_REQUIRE(x && sigma && y && blur_ok(ksize, N, H, W, C));
    BlurArgs a;
    plan(ksize, N, H, W, C, a);
    a.x = x; a.dy = nullptr; a.sigma = sigma; a.out = y;
    const size_t shm = shm_bytes(a);
    if (shm > 160 * 1024) return FAVAE_ERR_UNSUPPORTED;
    static bool attr0 = false;
    if (!attr0) { (void)hipFuncSetAttribute((const void*)blur_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr0 = true; }
    const int grid = N * a.tiles_h * a.tiles_w * a.cchunks;
    hipLaunchKernelGGL((blur_kernel<0>), dim3(grid), dim3(256), shm, (hipStream_t)stream, a);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" size_t favae_blur_bwd_workspace(int ksize, int N, int H, int W, int C) {
    if (!blur_ok(ksize, N, H, W, C)) return 0;
    BlurArgs a;
    plan(ksize, N, H, W, C, a);
    const size_t blocks = (size_t)N * a.tiles_h * a.tiles_w * a.cchunks;
    const size_t kk = (size_t)ksize * ksize;
    return blocks * kk * sizeof(float) + favae_colsum_workspace((int64_t)blocks, (int)kk) + kk * sizeof(float) + 256;
}

extern "C" int favae_blur_bwd(const float* x, const float* dy, const float* sigma, int ksize, int N, int H, int W, int C,
                              float* dx, float* dsigma, void* ws, size_t ws_bytes, favae_stream_t stream) {
    FAVAE_REQUIRE(x && dy && sigma && ws && blur_ok(ksize, N, H, W, C));
    FAVAE_REQUIRE(dx || dsigma);
    if (ws_bytes < favae_blur_bwd_workspace(ksize, N, H, W, C)) return FAVAE_ERR_WORKSPACE;
    BlurArgs a;
    plan(ksize, N, H, W, C, a);
    a.x = x; a.dy = dy; a.sigma = sigma;
    const size_t shm = shm_bytes(a);
    if (shm > 160 * 1024) return FAVAE_ERR_UNSUPPORTED;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)blur_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)blur_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipStream_t s = (hipStream_t)stream;
    const int grid = N * a.tiles_h * a.tiles_w * a.cchunks;
    if (dx) {
        a.out = dx;
        hipLaunchKernelGGL((blur_kernel<1>), dim3(grid), dim3(256), shm, s, a);
        FAVAE_CHECK_LAUNCH();
    }
    if (dsigma) {
        const int kk = ksize * ksize;
        float* part = (float*)ws;
        char* p2 = (char*)ws + (size_t)grid * kk * sizeof(float);
        const size_t cws = favae_colsum_workspace(grid, kk);
        float* dK = (float*)(p2 + cws);
        a.out = part;
        hipLaunchKernelGGL((blur_kernel<2>), dim3(grid), dim3(256), shm, s, a);
        FAVAE_CHECK_LAUNCH();
        int rc = favae_colsum(part, dK, grid, kk, p2, cws, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(blur_dsigma_kernel, dim3(1), dim3(64), 0, s, (const float*)dK, sigma, ksize, dsigma);
        FAVAE_CHECK_LAUNCH();
    }
    return FAVAE_OK;
}
