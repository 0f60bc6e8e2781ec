// Focal-frequency / dynamic-spectrum loss on NHWC tensors (replaces focal_frequency_loss.FocalFrequencyLoss 0.3.0 with
// alpha = 1, patch_factor = 1; call sites favae_scripts/train_favae.py:313,318,326, losses/vqgan_losses.py:14,25-26).
//
//   F = fft2_ortho(pred - target)              (linearity: ONE transform instead of the package's two)
//   w = clamp(|F| / max_{h,w}|F|, 0, 1), NaN -> 0, detached;   loss = lambda * mean(w |F|^2)
//   dL/dpred = (2 lambda / M) Re ifft2_ortho(w F),  dL/dtarget = -dL/dpred            
//
// The 2-D transform is two passes of one batched 1-D kernel over an array viewed as [outer][L][inner] (inner contiguous):
// W-axis: outer = N*H, L = W, inner = C;  H-axis: outer = N, L = H, inner = W*C.  A block owns one `outer` index and
// IC = 32 consecutive inner elements (256-byte contiguous complex segments -> coalesced), holds the L x IC complex tile
// in LDS, and runs an in-LDS radix-2 DIT FFT (bit-reversed load, log2 L butterfly stages, twiddles from an LDS table).
// HBM traffic per transform pass = one read + one write of the tile; the planes never leave LDS inside a pass.
//
// The input is real, so the spectrum is Hermitian, F[v][W-k] = conj F[-v][k]: only the W/2+1 bins k <= W/2 of the W-axis
// transform are stored ([N][H][W/2+1][C]); the H-axis pass, the weighting and the inverse H-axis pass run on that half, the
// loss counts the bins 0 < k < W/2 twice (|F| and the weight w are symmetric), and the inverse W-axis pass rebuilds the mirrored
// bins as conjugates while loading (after the inverse H-axis transform every row is still Hermitian in k).  Half the bytes in
// four of the five passes; the value and the gradient are those of the full-spectrum formulation.
#include "common.h"
#include <stdlib.h>

namespace {

struct FftArgs {
    const float* in0;      // complex in, or pred (real)
    const float* in1;      // target (real) for IN_DIFF
    float* out;            // complex out / gpred
    float* out2;           // gtarget (optional)
    unsigned* planemax;    // [N][C] bit patterns of max |F|^2
    const float* gscale;   // device scalar multiplied into the real output (bwd)
    long outer, inner;
    int L, logL, IC, C;
    int Lin, Lout;         // bins present in the input / written to the output along L (L, or L/2+1 for the half spectrum)
    long plane_outer_div;  // outer index -> n : n = outer / plane_outer_div
    float scale;           // 1/sqrt(L)
    int inverse;
    int generic;           // L is not a power of two: direct O(L^2) DFT per line instead of the radix-2 butterflies
};

enum { IN_COMPLEX = 0, IN_DIFF = 1, IN_HALF = 2 };   // IN_HALF: bins l > L/2 are the conjugates of bins L - l
enum { OUT_COMPLEX = 0, OUT_COMPLEX_MAX = 1, OUT_REAL = 2 };

// 256 or 512 threads per block (round 3: the 64 KB tiles of the 256-point lines allow two blocks per CU -- with 512 threads those
// are 16 waves per CU instead of 8 for the same tile: twice the loads in flight in the latency-bound load / store phases)
template <int IN, int OUT>
__global__ __launch_bounds__(512) void fft_lines_kernel(FftArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int L = a.L, IC = a.IC, tid = threadIdx.x, NT = blockDim.x;
    const int NTW = a.generic ? L : (L / 2 > 0 ? L / 2 : 1);    // twiddles: e^(-+2 pi i j / L), j < L/2 (all j for the direct DFT)
    float2* tw = reinterpret_cast<float2*>(sm);                 // [NTW]
    float2* data = reinterpret_cast<float2*>(sm) + NTW;         // [L][IC]
    const long chunks = (a.inner + IC - 1) / IC;
    const long o = blockIdx.x / chunks;
    const long i0 = (blockIdx.x % chunks) * IC;

    for (int j = tid; j < (a.generic ? L : L / 2); j += NT) {
        float s, c;
        sincospif(2.0f * (float)j / (float)L, &s, &c);
        tw[j] = make_float2(c, a.inverse ? s : -s);
    }
    // ---- load (bit-reversed along L) ------------------------------------------------------------------------
    const int ic = tid % IC, bl = tid / IC, BL = NT / IC;
    const bool ic_ok = i0 + ic < a.inner;
    const size_t base = (size_t)o * L * a.inner + i0 + ic;            // full-length lines (real input, real output)
    const size_t base_in = (size_t)o * a.Lin * a.inner + i0 + ic;
    const size_t base_out = (size_t)o * a.Lout * a.inner + i0 + ic;
    // U independent loads in flight per thread before the first LDS store (the loop bounds are run-time values, so the
    // compiler would otherwise serialise load -> store pairs: this pass is latency-bound with 8 waves per 64 KB tile)
#ifndef FAVAE_FFT_U
#define FAVAE_FFT_U 16
#endif
    constexpr int U = FAVAE_FFT_U;
    for (int l0 = bl; l0 < L; l0 += BL * U) {
        float2 v[U];
        float t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l = l0 + u * BL;
            v[u] = make_float2(0.f, 0.f);
            t[u] = 0.f;
            if (ic_ok && l < L) {
                if (IN == IN_DIFF) {
                    const size_t idx = base + (size_t)l * a.inner;
                    v[u].x = a.in0[idx];
                    t[u] = a.in1[idx];
                } else if (IN == IN_HALF && l >= a.Lin) {
                    v[u] = reinterpret_cast<const float2*>(a.in0)[base_in + (size_t)(L - l) * a.inner];
                    v[u].y = -v[u].y;
                } else {
                    v[u] = reinterpret_cast<const float2*>(a.in0)[base_in + (size_t)l * a.inner];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int l = l0 + u * BL;
            if (l < L) {
                if (IN == IN_DIFF) v[u].x -= t[u];
                const int r = a.generic ? l : (a.logL ? (int)(__brev((unsigned)l) >> (32 - a.logL)) : 0);
                data[r * IC + ic] = v[u];
            }
        }
    }
    __syncthreads();
    // ---- butterflies ------------------------------------------------------------------------------------------
    // Two radix-2 stages (s, s+1) are applied per pass on groups of four points held in registers -- the same butterflies in
    // the same order as stage-by-stage radix-2 (bit-identical results), with half the LDS traffic and half the barriers.
    // (Round 5 tried four stages per pass on sixteen points: the forward W-axis pass -13 %, the inverse one +8 %, the loss 3.42 -> 3.32 ms
    // on the 128-channel pair, and different gradient bits -- the compiler contracts the unrolled butterflies differently.  Not kept.)
    auto bfly = [](const float2 w, float2& x0, float2& x1) {
        const float2 t = make_float2(w.x * x1.x - w.y * x1.y, w.x * x1.y + w.y * x1.x);
        x1 = make_float2(x0.x - t.x, x0.y - t.y);
        x0 = make_float2(x0.x + t.x, x0.y + t.y);
    };
    int s = 1;
    if (a.generic) s = a.logL + 1;                             // no butterflies: the store phase evaluates the DFT sums
    else if (a.logL & 1) {                                     // odd number of stages: one plain radix-2 stage first
        for (int b = bl; b < L / 2; b += BL) {
            const int p0 = b << 1;
            float2 x0 = data[p0 * IC + ic], x1 = data[(p0 + 1) * IC + ic];
            bfly(tw[0], x0, x1);
            data[p0 * IC + ic] = x0;
            data[(p0 + 1) * IC + ic] = x1;
        }
        __syncthreads();
        s = 2;
    }
    for (; s + 1 <= a.logL; s += 2) {
        const int half = 1 << (s - 1);
        const int t1 = L >> s, t2 = L >> (s + 1);
        for (int b = bl; b < L / 4; b += BL) {
            const int grp = b >> (s - 1), j = b & (half - 1);
            const int p0 = (grp << (s + 1)) + j;
            float2 x0 = data[p0 * IC + ic], x1 = data[(p0 + half) * IC + ic];
            float2 x2 = data[(p0 + 2 * half) * IC + ic], x3 = data[(p0 + 3 * half) * IC + ic];
            const float2 w1 = tw[j * t1];
            bfly(w1, x0, x1);                                  // stage s
            bfly(w1, x2, x3);
            bfly(tw[j * t2], x0, x2);                          // stage s + 1
            bfly(tw[(j + half) * t2], x1, x3);
            data[p0 * IC + ic] = x0;
            data[(p0 + half) * IC + ic] = x1;
            data[(p0 + 2 * half) * IC + ic] = x2;
            data[(p0 + 3 * half) * IC + ic] = x3;
        }
        __syncthreads();
    }
    // ---- store ------------------------------------------------------------------------------------------------
    if (OUT != OUT_COMPLEX_MAX && !ic_ok) return;
    float mx = 0.f;
    const float gs = (OUT == OUT_REAL) ? a.gscale[0] * a.scale : a.scale;
    for (int l = bl; ic_ok && l < (OUT == OUT_REAL ? L : a.Lout); l += BL) {
        float2 v;
        if (a.generic) {
            // any length (FA-VAE at resolutions that are not powers of two, e.g. 192 -> 12 x 12 latents): bin l of the line is
            // sum_m data[m] w^(l m), the twiddle index (l m) mod L kept incrementally; data[m] is read by all bins of a column
            // (conflict-free across ic), the twiddle is a broadcast.  O(L^2) per line -- a fallback, not the fast path.
            float2 acc = make_float2(0.f, 0.f);
            int j = 0;
            for (int m = 0; m < L; ++m) {
                const float2 d = data[m * IC + ic], w = tw[j];
                acc.x = fmaf(d.x, w.x, fmaf(-d.y, w.y, acc.x));
                acc.y = fmaf(d.x, w.y, fmaf(d.y, w.x, acc.y));
                j += l;
                if (j >= L) j -= L;
            }
            v = acc;
        } else {
            v = data[l * IC + ic];
        }
        if (OUT == OUT_REAL) {
            const size_t idx = base + (size_t)l * a.inner;
            const float r = v.x * gs;
            a.out[idx] = r;
            if (a.out2) a.out2[idx] = -r;
        } else {
            v.x *= gs; v.y *= gs;
            reinterpret_cast<float2*>(a.out)[base_out + (size_t)l * a.inner] = v;
            if (OUT == OUT_COMPLEX_MAX) mx = fmaxf(mx, v.x * v.x + v.y * v.y);
        }
    }
    if (OUT == OUT_COMPLEX_MAX) {
        // one atomic per column of the tile, and only where it can raise the plane's maximum (round 5: one per THREAD were 8.4 M atomics
        // on 4096 addresses for the 128-channel feature pair -- 2000 per address; the maximum does not depend on the order)
        __syncthreads();                                   // every thread is here (no early return above): data[] is free
        float* red = reinterpret_cast<float*>(data);       // [BL][IC]
        red[bl * IC + ic] = mx;
        __syncthreads();
        if (bl == 0 && ic_ok) {
            for (int b = 1; b < BL; ++b) mx = fmaxf(mx, red[b * IC + ic]);
            const long n = o / a.plane_outer_div;
            const int c = (int)((i0 + ic) % a.C);
            unsigned* pm = &a.planemax[n * a.C + c];
            if (__float_as_uint(mx) > *reinterpret_cast<volatile unsigned*>(pm))       // a stale read only costs an atomic
                atomicMax(pm, __float_as_uint(mx));                                       // d >= 0: uint order == float order
        }
    }
}

// spec <- coef * w * F (in place), per-block partial of sum(w*d) in double
// half spectrum [N][H][Wh][C]: bins 0 < k < W/2 stand for two bins of the full spectrum
// grid (blocks per image, N); j = (h Wh + k) C + c inside the image.  Round 5: 32-bit indices and divisions by multiply-high with host-made
// reciprocals (rcp = floor(2^32 / d) + 1, 0 for d = 1; exact while j d < 2^32) -- the three 64-bit divisions per element of rounds 1-4
// made this streaming pass instruction-bound (0.41 of the HBM roofline on the 128-channel feature pair).
// MULHI = false: the same pass with ordinary divisions on IDX-wide indices, for the sizes the reciprocal trick is not exact
// for (per_img * max(C, Wh) >= 2^32: e.g. the 128-channel feature pair at 512 x 512) -- slower per element, never refused.  Images
// beyond the grid's y extent are taken by the same block in turn (n += gridDim.y): its partial is one sum in a fixed order.
template <bool MULHI, typename IDX>
__global__ __launch_bounds__(256) void ffl_weight_kernel(float* spec, const unsigned* planemax, double* part, IDX per_img,
                                                         int C, float coef, int Wh, int W, unsigned rcp_c, unsigned rcp_wh, int N) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        float2* s2 = reinterpret_cast<float2*>(spec) + (size_t)n * per_img;
        const unsigned* pm = planemax + (size_t)n * C;
        for (IDX j = (IDX)blockIdx.x * 256u + threadIdx.x; j < per_img; j += (IDX)gridDim.x * 256u) {
            int c, k;
            if constexpr (MULHI) {
                const unsigned q = rcp_c ? __umulhi((unsigned)j, rcp_c) : (unsigned)j;           // j / C
                c = (int)((unsigned)j - q * (unsigned)C);
                const unsigned r = rcp_wh ? __umulhi(q, rcp_wh) : q;                             // q / Wh
                k = (int)(q - r * (unsigned)Wh);
            } else {
                const IDX q = j / (IDX)C;
                c = (int)(j - q * (IDX)C);
                k = (int)(q % (IDX)Wh);
            }
            float2 f = s2[j];
            const float d = f.x * f.x + f.y * f.y;
            const float mx = sqrtf(__uint_as_float(pm[c]));
            float w = (mx > 0.f) ? sqrtf(d) / mx : 0.f;                   // 0/0 -> NaN -> 0 upstream
            w = fminf(fmaxf(w, 0.f), 1.f);
            acc += (double)(w * d) * ((Wh == W || k == 0 || 2 * k == W) ? 1.0 : 2.0);
            const float g = coef * w;
            s2[j] = make_float2(g * f.x, g * f.y);
        }
    }
    const double tot = block_sum_d256(acc, red);
    if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void ffl_finish_kernel(const double* part, int nparts, double scale, float* loss) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    const double tot = block_sum_d256(acc, red);
    if (threadIdx.x == 0) loss[0] = (float)(tot * scale);
}

// FAVAE_FFL_FULL=1 keeps all W bins (A/B switch; default: the W/2+1 bins of the Hermitian half)
int stored_bins(int W) {
    static int full = -1;
    if (full < 0) { const char* e = getenv("FAVAE_FFL_FULL"); full = (e && e[0] == '1') ? 1 : 0; }
    return full ? W : W / 2 + 1;
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
// inner elements per block: as many as the LDS tile allows (256-byte segments), but not more than the array has -- the W-axis pass of
// an RGB image has inner = 3: with 32 columns 29 of every 32 threads idled (0.75 ms per step for a 25 MB tensor)
int pick_ic(int L, long inner) {
    int ic = L <= 256 ? 32 : (L <= 512 ? 16 : 8);
    while (ic > 1 && ic / 2 >= inner) ic /= 2;
    return ic;
}
size_t fft_shm(int L, int IC, bool generic) {
    return ((size_t)(generic ? L : (L / 2 > 0 ? L / 2 : 1)) + (size_t)L * IC) * sizeof(float2);
}
constexpr int WEIGHT_BLOCKS = 2048;

template <int IN, int OUT>
int launch_fft(FftArgs& a, hipStream_t s) {
    a.IC = pick_ic(a.L, a.inner);
    a.logL = ilog2(a.L);
    if (a.Lin <= 0) a.Lin = a.L;
    if (a.Lout <= 0) a.Lout = a.L;
    a.scale = 1.0f / sqrtf((float)a.L);
    a.generic = pow2(a.L) ? 0 : 1;
    const size_t shm = fft_shm(a.L, a.IC, a.generic != 0);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)fft_lines_kernel<IN, OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const long chunks = (a.inner + a.IC - 1) / a.IC;
    const long blocks = a.outer * chunks;
    if (blocks <= 0 || blocks >= (1L << 31)) return favae_prof_fail_(FAVAE_ERR_BAD_ARG);
    {   // bytes this pass must move: its input lines (two real tensors for IN_DIFF, complex otherwise) + its output lines
        const double lines = (double)a.outer * a.inner;
        const double in_b = IN == IN_DIFF ? 8.0 * a.Lin : 8.0 * a.Lin, out_b = (OUT == OUT_REAL ? (a.out2 ? 8.0 : 4.0) : 8.0) * a.Lout;
        FAVAE_PROF_NOTE(0, lines * (in_b + out_b));
    }
    static int nt512 = -1;                    // FAVAE_FFT_512=0: 256 threads per block everywhere (A/B switch)
    if (nt512 < 0) { const char* e = getenv("FAVAE_FFT_512"); nt512 = (e && e[0] == '0') ? 0 : 1; }
    // (the H-axis pass with the plane maximum in its epilogue measured slower with 512 threads: 1.03 -> 1.54 ms per step)
    const int nt = (nt512 && a.L >= 128 && a.IC >= 32 && !a.generic && !(IN == IN_COMPLEX && OUT == OUT_COMPLEX_MAX)) ? 512 : 256;
    FAVAE_KLAUNCH((fft_lines_kernel<IN, OUT>), dim3((unsigned)blocks), dim3(nt), shm, s, a);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

}  // namespace

extern "C" size_t favae_ffl_spec_floats(int N, int H, int W, int C) {
    return (size_t)N * H * stored_bins(W) * C * 2;
}

extern "C" size_t favae_ffl_workspace(int N, int H, int W, int C) {
    // fwd: planemax (N*C u32) + WEIGHT_BLOCKS doubles ; bwd: complex scratch
    const size_t fwd = (size_t)N * C * sizeof(unsigned) + 256 + WEIGHT_BLOCKS * sizeof(double);
    const size_t bwd = (size_t)N * H * stored_bins(W) * C * 2 * sizeof(float);
    return fwd > bwd ? fwd : bwd;
}

extern "C" int favae_ffl_fwd(const float* pred, const float* target, int N, int H, int W, int C, float loss_weight, float* loss,
                             float* spec, void* ws, size_t ws_bytes, favae_stream_t stream) {
    FAVAE_REQUIRE(pred && target && loss && spec && ws && N > 0 && C > 0);
    if (H < 1 || W < 1 || H > 1024 || W > 1024) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (ws_bytes < favae_ffl_workspace(N, H, W, C)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    hipStream_t s = (hipStream_t)stream;
    unsigned* planemax = (unsigned*)ws;
    double* part = (double*)((char*)ws + (((size_t)N * C * sizeof(unsigned) + 255) / 256) * 256);
    if (hipMemsetAsync(planemax, 0, (size_t)N * C * sizeof(unsigned), s) != hipSuccess) return favae_prof_fail_(FAVAE_ERR_LAUNCH);
    FftArgs a{};
    a.C = C;
    const int Wh = stored_bins(W);
    // pass 1: along W, real difference in, bins 0..W/2 out
    a.in0 = pred; a.in1 = target; a.out = spec; a.out2 = nullptr; a.planemax = planemax; a.gscale = nullptr;
    a.outer = (long)N * H; a.inner = C; a.L = W; a.Lin = W; a.Lout = Wh; a.plane_outer_div = H; a.inverse = 0;
    int rc = launch_fft<IN_DIFF, OUT_COMPLEX>(a, s);
    if (rc) return rc;
    // pass 2: along H on the half spectrum, in place, with plane max of |F|^2 (= the maximum over the full spectrum)
    a.in0 = spec; a.in1 = nullptr; a.out = spec;
    a.outer = N; a.inner = (long)Wh * C; a.L = H; a.Lin = H; a.Lout = H; a.plane_outer_div = 1;
    rc = launch_fft<IN_COMPLEX, OUT_COMPLEX_MAX>(a, s);
    if (rc) return rc;
    const size_t per_img = (size_t)H * Wh * C;               // stored bins of one image
    const double M = (double)N * H * W * C;                  // elements of the mean (full spectrum)
    const int ny = N > WEIGHT_BLOCKS ? WEIGHT_BLOCKS : N;     // images beyond the grid: the same blocks take them in turn
    int bx = (int)((per_img + 255) / 256 < (size_t)(WEIGHT_BLOCKS / ny) ? (per_img + 255) / 256 : (size_t)(WEIGHT_BLOCKS / ny));
    if (bx < 1) bx = 1;
    auto rcp32 = [](int dv) { return dv == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dv + 1); };
    const float coef = (float)(2.0 * (double)loss_weight / M);
    // multiply-high by host-made reciprocals is exact while index * divisor < 2^32; beyond that ordinary divisions (32- or 64-bit indices)
    const bool exact = per_img * (size_t)(C > Wh ? C : Wh) < ((size_t)1 << 32);
    if (exact) {
        FAVAE_KLAUNCH((ffl_weight_kernel<true, unsigned>), dim3(bx, ny), dim3(256), 0, s, spec, (const unsigned*)planemax, part,
                      (unsigned)per_img, C, coef, Wh, W, rcp32(C), rcp32(Wh), N);
    } else if (per_img < ((size_t)1 << 32) - (size_t)WEIGHT_BLOCKS * 256) {
        FAVAE_KLAUNCH((ffl_weight_kernel<false, unsigned>), dim3(bx, ny), dim3(256), 0, s, spec, (const unsigned*)planemax, part,
                      (unsigned)per_img, C, coef, Wh, W, 0u, 0u, N);
    } else {
        FAVAE_KLAUNCH((ffl_weight_kernel<false, unsigned long long>), dim3(bx, ny), dim3(256), 0, s, spec, (const unsigned*)planemax, part,
                      (unsigned long long)per_img, C, coef, Wh, W, 0u, 0u, N);
    }
    FAVAE_CHECK_LAUNCH();
    const int blocks = bx * ny;
    FAVAE_KLAUNCH(ffl_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)part, blocks, (double)loss_weight / M, loss);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_ffl_bwd(const float* spec, const float* gloss, int N, int H, int W, int C, float* gpred, float* gtarget,
                             void* ws, size_t ws_bytes, favae_stream_t stream) {
    FAVAE_REQUIRE(spec && gloss && gpred && ws && N > 0 && C > 0);
    if (H < 1 || W < 1 || H > 1024 || W > 1024) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (ws_bytes < favae_ffl_workspace(N, H, W, C)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    hipStream_t s = (hipStream_t)stream;
    FftArgs a{};
    a.C = C;
    const int Wh = stored_bins(W);
    a.in0 = spec; a.in1 = nullptr; a.out = (float*)ws; a.out2 = nullptr; a.planemax = nullptr; a.gscale = nullptr;
    a.outer = N; a.inner = (long)Wh * C; a.L = H; a.Lin = H; a.Lout = H; a.plane_outer_div = 1; a.inverse = 1;
    int rc = launch_fft<IN_COMPLEX, OUT_COMPLEX>(a, s);
    if (rc) return rc;
    a.in0 = (const float*)ws; a.out = gpred; a.out2 = gtarget; a.gscale = gloss;
    a.outer = (long)N * H; a.inner = C; a.L = W; a.Lin = Wh; a.Lout = W; a.plane_outer_div = H;
    return launch_fft<IN_HALF, OUT_REAL>(a, s);
}
