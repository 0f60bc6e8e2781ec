// Learnable-sigma Gaussian blur of the Frequency Complement Module taps (models/codec.py:255-277, models/vqgan_fcm.py:20-41):
//   g = exp(-0.5 (t/sigma)^2) / sum,  t = linspace(-(k-1)/2, (k-1)/2, k);   K = g g^T;   y = K (*) reflectpad(x), depthwise.
// NHWC, HBM-bound.  K is an outer product and reflect padding acts per axis, so the blur is run as two 1-D passes
// (F_h F_w) inside one kernel: a block owns a TH x TW pixel tile x CC channels, stages the (TH+k-1) x (TW+k-1) halo tile
// in LDS once (reflect indexing applied while loading), blurs rows into a second LDS tile, then columns into the output --
// 2k LDS taps per output instead of k^2 and exactly one HBM read + one write per element (plus the halo overlap).
// Backward, one fused kernel (D = dy zero-extended, X = x reflect-extended, both staged in LDS):
//   V  = F_h^T D   (vertical adjoint incl. the reflect fold: every padded row j that reflects onto y contributes)
//   dx = F_w^T V
//   U  = F_w X ;  dg_a = sum D[y,x] U[y+a,x]  +  sum V[y,x] X[y,x+a]      (2k block partials, reduced deterministically)
//   dsigma = sum_j dp_j p_j t_j^2 / sigma^3,  dp = (dg - <dg,g>) / sum(p)
// Lanes run along channels: all global accesses are contiguous channel segments, all LDS accesses conflict-free.
#include "common.h"
#include <stdlib.h>

extern "C" size_t favae_colsum_workspace(int64_t M, int C);
extern "C" int favae_colsum(const float* a, float* out, int64_t M, int C, int accumulate, float* absmax_out, void* ws,
                            size_t ws_bytes, favae_stream_t stream);

namespace {

constexpr int MAXK = 31;

struct BlurArgs {
    const float* x;
    const float* dy;
    const float* sigma;
    float* y;         // forward output
    float* dx;        // backward: may be null
    float* part;      // backward: [blocks][2k] partials of dg, may be null
    const float* dx_add;  // backward: optional tensor added to dx (the other gradient of the blurred tensor's source)
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

// padded-domain positions that reflect onto image index i (1-D): i itself, -i, 2(n-1)-i
__device__ __forceinline__ int preimages(int i, int n, int p, int (&j)[3]) {
    int c = 0;
    j[c++] = i;
    if (i >= 1 && i <= p) j[c++] = -i;
    if (i <= n - 2 && i >= n - 1 - p) j[c++] = 2 * (n - 1) - i;
    return c;
}

// KS: compile-time kernel size (3, 5, 9, ...) so that the k-tap loops fully unroll and their LDS reads are issued back to
// back instead of one per loop trip (the runtime-k version was LDS-latency bound); KS = 0 keeps k a runtime value.
template <int MODE, int KS>   // MODE 0 forward, 1 backward
__global__ __launch_bounds__(256) void blur_sep_kernel(BlurArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int k = KS ? KS : a.k, p = k / 2, CC = a.CC;
    const int HH = a.TH + k - 1, HW = a.TW + k - 1;
    float* g = sm;                                   // [32]
    float* red = sm + 32;                            // [2*MAXK][4] cross-wave partials (backward)
    float* Xh = sm + 32 + 8 * MAXK;                  // [HH][HW][CC]  reflect-extended x
    float* U = Xh + HH * HW * CC;                    // [HH][TW][CC]  row-blurred x
    float* Dh = U + HH * a.TW * CC;                  // [HH][HW][CC]  zero-extended dy      (backward only)
    float* V = Dh + HH * HW * CC;                    // [TH][HW][CC]  F_h^T dy              (backward only)
    int b = blockIdx.x;
    const int cchunk = b % a.cchunks; b /= a.cchunks;
    const int tw = b % a.tiles_w; b /= a.tiles_w;
    const int th = b % a.tiles_h;
    const int n = b / a.tiles_h;
    const int y0 = th * a.TH, x0 = tw * a.TW, c0 = cchunk * CC;
    const int tid = threadIdx.x;
    const int c = tid % CC, pl = tid / CC, PL = 256 / CC;
    const bool c_ok = c0 + c < a.C;
    // interior tiles (no reflect fold, no image border inside the halo) take branch-free k-tap loops
    const bool int_y = y0 >= p + 1 && y0 + a.TH - 1 < a.H - 1 - p;
    const bool int_x = x0 >= p + 1 && x0 + a.TW - 1 < a.W - 1 - p;

    make_kernel1d(a.sigma, k, g);
    // ---- stage halo tiles ---------------------------------------------------------------------------------------
    {
        const float* ximg = a.x + (size_t)n * a.H * a.W * a.C;
        const float* dimg = MODE == 1 ? a.dy + (size_t)n * a.H * a.W * a.C : nullptr;
        for (int q = pl; q < HH * HW; q += PL) {
            const int hy = q / HW, hx = q - hy * HW;
            const int sy = y0 + hy - p, sx = x0 + hx - p;
            float xv = 0.f, dv = 0.f;
            if (c_ok) {
                xv = ximg[((size_t)reflect_idx(sy, a.H) * a.W + reflect_idx(sx, a.W)) * a.C + c0 + c];
                if (MODE == 1 && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W) dv = dimg[((size_t)sy * a.W + sx) * a.C + c0 + c];
            }
            Xh[q * CC + c] = xv;
            if (MODE == 1) Dh[q * CC + c] = dv;
        }
    }
    __syncthreads();
    // ---- pass 1: U = F_w X on every halo row; V = F_h^T D on every halo column ------------------------------------
    if (MODE == 0 || a.part) {
        for (int q = pl; q < HH * a.TW; q += PL) {
            const int hy = q / a.TW, x = q - hy * a.TW;
            const float* row = Xh + (hy * HW + x) * CC + c;
            float acc = 0.f;
#pragma unroll
            for (int v = 0; v < k; ++v) acc = fmaf(g[v], row[v * CC], acc);
            U[q * CC + c] = acc;
        }
    }
    if (MODE == 1) {
        for (int q = pl; q < a.TH * HW; q += PL) {
            const int yy = q / HW, hx = q - yy * HW;
            const int y = y0 + yy;
            float acc = 0.f;
            if (int_y) {                                  // interior tile: no fold, no bounds -> k taps straight down the column
                const float* col = Dh + ((yy + 2 * p) * HW + hx) * CC + c;
#pragma unroll
                for (int u = 0; u < k; ++u) acc = fmaf(g[u], col[-u * HW * CC], acc);
            } else if (y < a.H) {
                int jy[3];
                const int ny = preimages(y, a.H, p, jy);
                for (int i = 0; i < ny; ++i)
                    for (int u = 0; u < k; ++u) {
                        const int ry = jy[i] + p - u;                     // dy row in image coordinates
                        const int hy = ry - (y0 - p);
                        if (ry < 0 || ry >= a.H || hy < 0 || hy >= HH) continue;
                        acc = fmaf(g[u], Dh[(hy * HW + hx) * CC + c], acc);
                    }
            }
            V[q * CC + c] = acc;
        }
    }
    __syncthreads();
    // ---- pass 2 ---------------------------------------------------------------------------------------------------
    const int npix = a.TH * a.TW;
    if (MODE == 0) {
        if (!c_ok) return;
        float* out = a.y + (size_t)n * a.H * a.W * a.C;
        for (int q = pl; q < npix; q += PL) {
            const int py = q / a.TW, px = q - py * a.TW;
            if (y0 + py >= a.H || x0 + px >= a.W) continue;
            const float* col = U + (py * a.TW + px) * CC + c;
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < k; ++u) acc = fmaf(g[u], col[u * a.TW * CC], acc);
            out[((size_t)(y0 + py) * a.W + x0 + px) * a.C + c0 + c] = acc;
        }
        return;
    }
    if (a.dx && c_ok) {
        float* out = a.dx + (size_t)n * a.H * a.W * a.C;
        const float* add = a.dx_add ? a.dx_add + (size_t)n * a.H * a.W * a.C : nullptr;
        for (int q = pl; q < npix; q += PL) {
            const int py = q / a.TW, px = q - py * a.TW;
            const int y = y0 + py, x = x0 + px;
            if (y >= a.H || x >= a.W) continue;
            float acc = 0.f;
            if (int_x) {
                const float* row = V + (py * HW + px + 2 * p) * CC + c;
#pragma unroll
                for (int v = 0; v < k; ++v) acc = fmaf(g[v], row[-v * CC], acc);
                if (add) acc += add[((size_t)y * a.W + x) * a.C + c0 + c];
                out[((size_t)y * a.W + x) * a.C + c0 + c] = acc;
                continue;
            }
            int jx[3];
            const int nx = preimages(x, a.W, p, jx);
            for (int i = 0; i < nx; ++i)
                for (int v = 0; v < k; ++v) {
                    const int rx = jx[i] + p - v;
                    const int hx = rx - (x0 - p);
                    if (rx < 0 || rx >= a.W || hx < 0 || hx >= HW) continue;
                    acc = fmaf(g[v], V[(py * HW + hx) * CC + c], acc);
                }
            if (add) acc += add[((size_t)y * a.W + x) * a.C + c0 + c];
            out[((size_t)y * a.W + x) * a.C + c0 + c] = acc;
        }
    }
    if (a.part) {
        const int lane = tid & 63, wid = tid >> 6;
        constexpr int MAXP = 16;                              // pixels per thread (plan() keeps TH*TW/PL <= 16)
        float dc[MAXP], vc[MAXP];
        int offU[MAXP], offX[MAXP];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int q = pl + i * PL;
            const int py = q / a.TW, px = q - py * a.TW;
            const bool ok = c_ok && q < npix && y0 + py < a.H && x0 + px < a.W;
            dc[i] = ok ? Dh[((py + p) * HW + px + p) * CC + c] : 0.f;
            vc[i] = ok ? V[(py * HW + px + p) * CC + c] : 0.f;
            offU[i] = ok ? (py * a.TW + px) * CC + c : 0;
            offX[i] = ok ? ((py + p) * HW + px) * CC + c : 0;
        }
        for (int t = 0; t < k; ++t) {
            float ah = 0.f, aw = 0.f;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                ah = fmaf(dc[i], U[offU[i] + t * a.TW * CC], ah);
                aw = fmaf(vc[i], Xh[offX[i] + t * CC], aw);
            }
            ah = wave_sum(ah);
            aw = wave_sum(aw);
            if (lane == 0) red[t * 4 + wid] = ah + aw;
        }
        __syncthreads();
        if (tid < k) a.part[(size_t)blockIdx.x * k + tid] = (red[tid * 4] + red[tid * 4 + 1]) + (red[tid * 4 + 2] + red[tid * 4 + 3]);
    }
}

// dg (k floats, summed over blocks) -> dsigma
__global__ void blur_dsigma_kernel(const float* dgv, const float* sigma, int k, float* dsigma) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double s = sigma[0];
    const double half = (k - 1) * 0.5;
    double S = 0.0, dot = 0.0;
    for (int j = 0; j < k; ++j) {
        const double t = j - half;
        S += exp(-0.5 * (t / s) * (t / s));
    }
    for (int j = 0; j < k; ++j) {
        const double t = j - half;
        dot += (double)dgv[j] * exp(-0.5 * (t / s) * (t / s)) / S;
    }
    double ds = 0.0;
    for (int j = 0; j < k; ++j) {
        const double t = j - half;
        const double pj = exp(-0.5 * (t / s) * (t / s));
        ds += ((double)dgv[j] - dot) / S * pj * t * t / (s * s * s);
    }
    dsigma[0] = (float)ds;
}

}  // namespace
#include "blur_stream.h"
namespace {

size_t shm_floats(const BlurArgs& a, bool bwd) {
    const size_t HH = a.TH + a.k - 1, HW = a.TW + a.k - 1;
    size_t f = 32 + 8 * MAXK + HH * HW * a.CC + HH * a.TW * a.CC;
    if (bwd) f += HH * HW * a.CC + (size_t)a.TH * HW * a.CC;
    return f;
}

bool plan(int ksize, int N, int H, int W, int C, bool bwd, BlurArgs& a) {
    a.N = N; a.H = H; a.W = W; a.C = C; a.k = ksize;
    const int ccs[3] = {16, 8, 4};
    const int tiles[4][2] = {{16, 16}, {8, 16}, {8, 8}, {4, 8}};
    const size_t budget = (bwd ? 80 : 64) * 1024 / 4;          // floats: keeps >= 2 blocks per CU
    for (int relax = 0; relax < 2; ++relax)
        for (int ci = 0; ci < 3; ++ci)                          // prefer wide channel chunks (coalescing), then big tiles
            for (int ti = 0; ti < 4; ++ti) {
                a.CC = ccs[ci];
                if (a.CC > 4 && a.CC / 2 >= C) continue;          // do not waste lanes on tiny channel counts
                a.TH = tiles[ti][0]; a.TW = tiles[ti][1];
                if (shm_floats(a, bwd) <= (relax ? (size_t)38000 : budget)) {
                    a.tiles_h = (H + a.TH - 1) / a.TH;
                    a.tiles_w = (W + a.TW - 1) / a.TW;
                    a.cchunks = (C + a.CC - 1) / a.CC;
                    return true;
                }
            }
    return false;
}

bool blur_ok(int ksize, int N, int H, int W, int C) {
    return ksize >= 1 && ksize <= MAXK && (ksize & 1) && N > 0 && C > 0 && H > ksize / 2 && W > ksize / 2;
}

}  // namespace

extern "C" int favae_blur_fwd(const float* x, const float* sigma, int ksize, int N, int H, int W, int C, float* y,
                              favae_stream_t stream) {
    FAVAE_REQUIRE(x && sigma && y && blur_ok(ksize, N, H, W, C));
    FAVAE_PROF_NOTE(0, 8.0 * N * H * W * C);                               // one read + one write
    if (stream_ok(ksize, N, H, W, C) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        StreamArgs sa{};
        stream_plan(N, H, W, C, sa);
        sa.x = x; sa.sigma = sigma; sa.y = y;
        const long sgrid = (long)N * sa.segs * sa.strips * sa.cchunks;
        if (sgrid < (1L << 31)) {
            FAVAE_KLAUNCH((blur9_stream_kernel<0, STREAM_COLS>), dim3((unsigned)sgrid), dim3(STREAM_COLS * 8), 0,
                               (hipStream_t)stream, sa);
            FAVAE_CHECK_LAUNCH();
            return FAVAE_OK;
        }
    }
    BlurArgs a{};
    if (!plan(ksize, N, H, W, C, false, a)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    a.x = x; a.sigma = sigma; a.y = y;
    const size_t shm = shm_floats(a, false) * sizeof(float);
    static bool attr0 = false;
    if (!attr0) {
        (void)hipFuncSetAttribute((const void*)blur_sep_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)blur_sep_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)blur_sep_kernel<0, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)blur_sep_kernel<0, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr0 = true;
    }
    const long grid = (long)N * a.tiles_h * a.tiles_w * a.cchunks;
    if (grid >= (1L << 31)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    const dim3 g3((unsigned)grid), b3(256);
    hipStream_t s0 = (hipStream_t)stream;
    if (ksize == 9) FAVAE_KLAUNCH((blur_sep_kernel<0, 9>), g3, b3, shm, s0, a);
    else if (ksize == 5) FAVAE_KLAUNCH((blur_sep_kernel<0, 5>), g3, b3, shm, s0, a);
    else if (ksize == 3) FAVAE_KLAUNCH((blur_sep_kernel<0, 3>), g3, b3, shm, s0, a);
    else FAVAE_KLAUNCH((blur_sep_kernel<0, 0>), g3, b3, shm, s0, a);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" size_t favae_blur_bwd_workspace(int ksize, int N, int H, int W, int C) {
    if (!blur_ok(ksize, N, H, W, C)) return 0;
    BlurArgs a{};
    if (!plan(ksize, N, H, W, C, true, a)) return 0;
    size_t blocks = (size_t)N * a.tiles_h * a.tiles_w * a.cchunks;
    if (stream_ok(ksize, N, H, W, C)) {                       // either kernel may run (pointer alignment decides): size for both
        StreamArgs sa{};
        stream_plan(N, H, W, C, sa);
        const size_t sb = (size_t)N * sa.segs * sa.strips * sa.cchunks;
        if (sb > blocks) blocks = sb;
    }
    return blocks * ksize * sizeof(float) + favae_colsum_workspace((int64_t)blocks, ksize) + MAXK * sizeof(float) + 512;
}

static int blur_bwd_impl(const float* x, const float* dy, const float* sigma, int ksize, int N, int H, int W, int C, const float* dx_add,
                         float* dx, float* dsigma, void* ws, size_t ws_bytes, favae_stream_t stream);

extern "C" int favae_blur_bwd(const float* x, const float* dy, const float* sigma, int ksize, int N, int H, int W, int C,
                              float* dx, float* dsigma, void* ws, size_t ws_bytes, favae_stream_t stream) {
    return blur_bwd_impl(x, dy, sigma, ksize, N, H, W, C, nullptr, dx, dsigma, ws, ws_bytes, stream);
}

// favae_blur_bwd with dx = (adjoint blur of dy) + dx_add: the blurred tensor's source usually has a second consumer (the trunk of the
// codec, models/codec.py:209-215), whose gradient autograd would add with one more pass over both tensors
extern "C" int favae_blur_bwd_add(const float* x, const float* dy, const float* sigma, int ksize, int N, int H, int W, int C,
                                  const float* dx_add, float* dx, float* dsigma, void* ws, size_t ws_bytes, favae_stream_t stream) {
    FAVAE_REQUIRE(dx_add && dx);
    return blur_bwd_impl(x, dy, sigma, ksize, N, H, W, C, dx_add, dx, dsigma, ws, ws_bytes, stream);
}

static int blur_bwd_impl(const float* x, const float* dy, const float* sigma, int ksize, int N, int H, int W, int C, const float* dx_add,
                         float* dx, float* dsigma, void* ws, size_t ws_bytes, favae_stream_t stream) {
    FAVAE_REQUIRE(x && dy && sigma && ws && blur_ok(ksize, N, H, W, C));
    FAVAE_REQUIRE(dx || dsigma);
    BlurArgs a{};
    if (!plan(ksize, N, H, W, C, true, a)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (ws_bytes < favae_blur_bwd_workspace(ksize, N, H, W, C)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    StreamArgs sa{};
    bool stream_path = stream_ok(ksize, N, H, W, C) && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dx_add) & 15) == 0;
    long grid = (long)N * a.tiles_h * a.tiles_w * a.cchunks;
    if (stream_path) {
        stream_plan(N, H, W, C, sa);
        const long sgrid = (long)N * sa.segs * sa.strips * sa.cchunks;
        if (sgrid < (1L << 31)) grid = sgrid;
        else stream_path = false;
    }
    if (grid >= (1L << 31)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    float* part = (float*)ws;
    char* p2 = (char*)ws + (((size_t)grid * ksize * sizeof(float) + 255) / 256) * 256;
    const size_t cws = favae_colsum_workspace(grid, ksize);
    float* dgv = (float*)(p2 + ((cws + 255) / 256) * 256);
    if ((char*)(dgv + ksize) > (char*)ws + ws_bytes) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    a.x = x; a.dy = dy; a.sigma = sigma; a.dx = dx; a.part = dsigma ? part : nullptr; a.dx_add = dx_add;
    const size_t shm = shm_floats(a, true) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)blur_sep_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)blur_sep_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)blur_sep_kernel<1, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)blur_sep_kernel<1, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 g3((unsigned)grid), b3(256);
    FAVAE_PROF_NOTE(0, (dx_add ? 16.0 : 12.0) * N * H * W * C);            // reads x and dy (+ dx_add), writes dx
    static int bwd2 = -1;                     // FAVAE_BLUR_BWD2=0: the round-2 backward with nine tap-gradient accumulators (A/B switch)
    if (bwd2 < 0) { const char* e = getenv("FAVAE_BLUR_BWD2"); bwd2 = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }
    const bool direct = stream_path && bwd2;
    if (direct) {
        sa.x = x; sa.dy = dy; sa.sigma = sigma; sa.dx = dx; sa.part = a.part; sa.dx_add = dx_add;
        if (bwd2 == 2) FAVAE_KLAUNCH((blur9_stream_bwd_kernel<STREAM_COLS, 2>), g3, dim3(STREAM_COLS * 8), 0, s, sa);
        else FAVAE_KLAUNCH((blur9_stream_bwd_kernel<STREAM_COLS, 4>), g3, dim3(STREAM_COLS * 8), 0, s, sa);
    } else if (stream_path) {
        sa.x = x; sa.dy = dy; sa.sigma = sigma; sa.dx = dx; sa.part = a.part; sa.dx_add = dx_add;
        FAVAE_KLAUNCH((blur9_stream_kernel<1, STREAM_COLS>), g3, dim3(STREAM_COLS * 8), 0, s, sa);
    } else if (ksize == 9) FAVAE_KLAUNCH((blur_sep_kernel<1, 9>), g3, b3, shm, s, a);
    else if (ksize == 5) FAVAE_KLAUNCH((blur_sep_kernel<1, 5>), g3, b3, shm, s, a);
    else if (ksize == 3) FAVAE_KLAUNCH((blur_sep_kernel<1, 3>), g3, b3, shm, s, a);
    else FAVAE_KLAUNCH((blur_sep_kernel<1, 0>), g3, b3, shm, s, a);
    FAVAE_CHECK_LAUNCH();
    if (dsigma && direct) {                   // one d sigma partial per workgroup: their sum IS the gradient
        int rc = favae_colsum(part, dsigma, grid, 1, 0, nullptr, p2, cws, stream);
        if (rc) return rc;
    } else if (dsigma) {
        int rc = favae_colsum(part, dgv, grid, ksize, 0, nullptr, p2, cws, stream);
        if (rc) return rc;
        FAVAE_KLAUNCH(blur_dsigma_kernel, dim3(1), dim3(64), 0, s, (const float*)dgv, sigma, ksize, dsigma);
        FAVAE_CHECK_LAUNCH();
    }
    return FAVAE_OK;
}
