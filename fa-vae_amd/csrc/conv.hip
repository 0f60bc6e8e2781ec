// Implicit-GEMM convolution family for gfx950 on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32.
//
//   forward / data-gradient : out[m][co] = sum_{tap,ci} T(x)[gather(m,tap)][ci] * w[co][tap][ci]   (+bias +resid)
//   weight-gradient         : dw[co][tap][ci] = sum_m dy[m][co] * T(x)[gather(m,tap)][ci]           (split-K over m)
//
// m runs over output pixels (n,oh,ow) in NHWC order, T() is the fused GroupNorm/BatchNorm affine + activation of the
// layer in front of the conv (models/codec.py:38-46), gather() folds zero padding, stride, nearest-x2 upsampling
// (codec.py:17) or x2 zero-dilation (data gradient of the stride-2 Downsample conv, codec.py:26-29).
//
// Tiling (one workgroup = 4 waves = 256 threads):
//   fwd : BM=128 output pixels x BN in {128,64,32} output channels, BK=16 input channels per (tap) step.
//         LDS tiles are [row][BK+4] (k contiguous, +4 floats of padding => conflict-free ds_read_b128): a lane reads
//         4 consecutive k of its row with ONE ds_read_b128 and feeds 4 MFMAs (lanes 0-31 carry k 0..3, lanes 32-63
//         carry k 4..7 of each 8-wide k group; A and B use the same split so the products pair up).
//   wgrad: 128(co) x 128(ci) per tap, 16 pixels per step, LDS tiles [k][row] read with conflict-free ds_read_b32.
//   Global loads are 16 B per lane along the channel dimension (NHWC => fully coalesced), register-staged and
//   double-buffered through LDS so that the loads of step i+1 are in flight during the MFMAs of step i.
//   blockIdx is remapped XCD-aware: tiles that share input rows / halos land on the same XCD's L2.
#include "common.h"
#include <stdlib.h>

namespace {

// debugging / A-B switch: FAVAE_CONV_GENERIC=1 forces the generic (runtime-predicated) kernels
bool force_generic() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_CONV_GENERIC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
// 128-wide convs run on the split-precision matrix path (conv_split.h).  FAVAE_CONV_MODE = h3 (default: two scaled fp16
// planes, 3 MFMAs, needs operand maxima) | b6 (three bf16 planes, 6 MFMAs) | fp32 (the fp32-MFMA kernels; also FAVAE_CONV_B6=0).
// h1 = ONE scaled fp16 plane, 1 MFMA: the 16-bit mixed-precision mode, not fp32-grade (BASELINE config 5); b1 = ONE bf16 plane
// (scheme id 4): the bf16 mixed-precision mode.
// Returns the scheme id: 0 fp32 | 1 h1 | 2 h3 | 3 b6 | 4 b1.  favae_set_conv_mode overrides the environment at run time.
static int g_conv_mode = -1;
int conv_mode() {
    if (g_conv_mode < 0) {
        const char* e = getenv("FAVAE_CONV_MODE");
        const char* b = getenv("FAVAE_CONV_B6");
        int v = 2;
        if (e && e[0] == 'b' && e[1] == '1') v = 4;
        else if (e && e[0] == 'b') v = 3;
        else if (e && e[0] == 'h' && e[1] == '1') v = 1;
        else if ((e && e[0] == 'f') || (b && b[0] == '0')) v = 0;
        g_conv_mode = v;
    }
    return g_conv_mode;
}
bool use_b6() { return conv_mode() != 0; }
// FAVAE_CONV_HALO2=0 sends the 2x2 phase convs back to the im2col split kernel (A/B switch)
bool use_halo2() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_CONV_HALO2"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
// FAVAE_WGRAD_NINE=0 sends the 3x3 weight gradients back to the per-tap kernel conv_wgrad_sp_kernel (A/B switch)
// 1 (default) = 128 co x 64 ci, 8 waves, prefetch distance 1 (204 registers: 96 of a SIMD lane's 512 stay free next to its two waves,
// which is what lets the <= 96-register GroupNorm-backward / bias-gradient passes of the main stream run beside it) | 2 = the same with
// prefetch distance 2 (220 registers) | 3 = 64 x 64, 4 waves, one workgroup per CU (A/B: half the rate, a single wave per SIMD)
int nine_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_WGRAD_NINE"); v = (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 1; }
    return v;
}
bool use_nine() { return nine_mode() != 0; }
// FAVAE_CONV_HALO=0 disables the LDS-halo 3x3 kernel (A/B switch)
bool use_halo() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_CONV_HALO"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
// FAVAE_B6_WAVES=4|8: waves per workgroup of the bf16x6 forward kernel (A/B switch)
int b6_waves() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_B6_WAVES"); v = (e && e[0] == '4') ? 4 : 8; }
    return v;
}
// FAVAE_WINO=0 keeps the dense 3x3 convs of the h3 scheme on the direct LDS-halo kernel instead of the Winograd F(2x2, 3x3) kernel
int g_wino = -1;
bool use_wino() {
    if (g_wino < 0) { const char* e = getenv("FAVAE_WINO"); g_wino = (e && e[0] == '0') ? 0 : 1; }
    return g_wino == 1;
}

// FAVAE_CONV_NOBUF=1 disables the buffer-addressed kernels (A/B against the flat-addressed fast kernels)
bool force_nobuf() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_CONV_NOBUF"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}


constexpr int BM = 128;
constexpr int BK = 16;
constexpr int LDK = BK + 4;   // padded row length (floats) of the [row][k] LDS tiles

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* resid;
    const float* scale;
    const float* shift;
    float* y;
    int N, Hin, Win, Cin, Hout, Wout, Cout;
    int KH, KW, stride, pad, gather, act, aff_stride;
    int M;            // N*Hout*Wout
    int tiles_m, tiles_n;
    int kchunks;      // ceil(Cin/BK)
    int vec;          // Cin % 4 == 0 (16-byte channel loads legal)
    unsigned x_bytes, w_bytes, aff_bytes;   // operand sizes for the buffer-addressed kernels
    const float* x_amax;                    // split-precision fp16 scheme: device-side bound of |transformed x| ...
    const float* w_amax;                    // ... and of |w| (header of the pre-split weight buffer)
    // conv_fwd_sp_kernel only: separate left padding and sub-grid addressing (favae_conv_desc lat_*): pixel (n, h, w) of the
    // input / output lives at ((n * img + h * step * row + w * step + off) * C) of its tensor (dense: step 1, row = W, off 0)
    int pad_w;
    int in_step, in_row, in_img, in_off;
    int out_step, out_row, out_img, out_off;
    // conv3x3_halo_sp_kernel<., 2, 3> only: when set, the kernel also stores the transformed + split operand T(x) it stages
    // (two scaled fp16 planes, one 16-byte record {hi[4], lo[4]} per 4 channels = the bytes of the fp32 tensor) for the
    // weight-gradient kernel, which then loads its operands without any transform / split arithmetic
    // conv3x3_halo_sp_kernel<0, 2, 3, true> only (data gradient of a conv whose input was GroupNorm(+act)'ed): the epilogue
    // also forms this tile's share of the two GroupNorm-backward sums S1 = sum dy, S2 = sum dy * xhat (dy = da * act'(y)) from the
    // da it has in registers and the matching tile of the conv input x -- the streaming pass-1 kernel (2 tensor reads) goes away.
    const float* gb_x;
    const float* gb_mean;
    const float* gb_rstd;
    const float* gb_gamma;
    const float* gb_beta;
    double* gb_part;          // [N][tiles per image][C][2]
    int gb_groups, gb_act;
    // conv3x3_halo_sp_kernel<., 2, 3, false, true>: per-tile sums (sum y, sum y^2) per output channel of the FINAL output
    // (bias and residual included) -- pass 1 of the GroupNorm that consumes this conv's output (gn_partial<0>: one tensor read)
    double* gs_part;          // [N][tiles per image][Cout][2]
    unsigned* gs_amax;        // optional (same variant): max |y| of the output, bit pattern, one atomicMax per workgroup (pre-zeroed)
    // conv3x3_wino_sp_kernel: floor(2^32 / d) + 1 for d = channel tiles, tile columns, tile rows (division by multiply-high)
    unsigned wino_rcp_n, wino_rcp_w, wino_rcp_h;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == FAVAE_ACT_SILU) return silu_f(v);
    if (act == FAVAE_ACT_LEAKY02) return v > 0.f ? v : 0.2f * v;
    if (act == FAVAE_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

// Resolve output pixel (oh,ow) + tap (kh,kw) to an input pixel; returns false for padding / dilation holes.
__device__ __forceinline__ bool gather_src(const ConvArgs& a, int oh, int ow, int kh, int kw, int& sh, int& sw) {
    int vh = oh * a.stride + kh - a.pad;
    int vw = ow * a.stride + kw - a.pad;
    if (a.gather == FAVAE_GATHER_PLAIN) {
        sh = vh; sw = vw;
        return (unsigned)vh < (unsigned)a.Hin && (unsigned)vw < (unsigned)a.Win;
    } else if (a.gather == FAVAE_GATHER_UPSAMPLE2) {
        sh = vh >> 1; sw = vw >> 1;
        return (unsigned)vh < (unsigned)(2 * a.Hin) && (unsigned)vw < (unsigned)(2 * a.Win);
    } else {  // zero-dilated x2: only even virtual coordinates carry data
        sh = vh >> 1; sw = vw >> 1;
        return vh >= 0 && vw >= 0 && !(vh & 1) && !(vw & 1) && sh < a.Hin && sw < a.Win;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ---------------------------------------------------------------------------------------------------------------
template <int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs a) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int A_LD = (BM * BK / 4) / 256;                         // float4 loads of A per thread (=2)
    constexpr int B_LD = (BN * BK / 4 + 255) / 256;                   // float4 loads of B per thread (2,1,1)
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDK];
    float* As = lds;                      // [2][BM][LDK]
    float* Bs = lds + 2 * BM * LDK;       // [2][BN][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;

    // ---- per-thread fixed A rows ---------------------------------------------------------------------------
    const int c4 = tid & 3;               // which float4 of the BK=16 chunk
    int a_n[A_LD], a_oh[A_LD], a_ow[A_LD];
    bool a_ok[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        int m = m0 + (tid >> 2) + 64 * j;
        a_ok[j] = m < a.M;
        int mm = a_ok[j] ? m : 0;
        int hw = a.Hout * a.Wout;
        a_n[j] = mm / hw;
        int r = mm - a_n[j] * hw;
        a_oh[j] = r / a.Wout;
        a_ow[j] = r - a_oh[j] * a.Wout;
    }
    const int b_row0 = tid >> 2;          // B rows handled: b_row0 + 64*j

    float4 ra[A_LD], rsc[A_LD], rsh[A_LD], rb[B_LD];
    bool rav[A_LD];
    const int T = a.KH * a.KW * a.kchunks;

    auto load_tiles = [&](int it) {
        const int tap = it / a.kchunks;
        const int k0 = (it - tap * a.kchunks) * BK + c4 * 4;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            int sh, sw;
            bool ok = a_ok[j] && gather_src(a, a_oh[j], a_ow[j], kh, kw, sh, sw) && k0 < a.Cin;
            rav[j] = ok;
            ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const float* p = a.x + ((size_t)(a_n[j] * a.Hin + sh) * a.Win + sw) * a.Cin + k0;
                if (a.vec) {
                    ra[j] = *reinterpret_cast<const float4*>(p);
                    if (a.scale) {
                        const size_t so = (size_t)a_n[j] * a.aff_stride + k0;
                        rsc[j] = *reinterpret_cast<const float4*>(a.scale + so);
                        rsh[j] = *reinterpret_cast<const float4*>(a.shift + so);
                    }
                } else {
                    float t[4], s1[4], s2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bool in = k0 + e < a.Cin;
                        t[e] = in ? p[e] : 0.f;
                        const size_t so = (size_t)a_n[j] * a.aff_stride + k0 + e;
                        s1[e] = (in && a.scale) ? a.scale[so] : 0.f;
                        s2[e] = (in && a.scale) ? a.shift[so] : 0.f;
                    }
                    ra[j] = make_float4(t[0], t[1], t[2], t[3]);
                    rsc[j] = make_float4(s1[0], s1[1], s1[2], s1[3]);
                    rsh[j] = make_float4(s2[0], s2[1], s2[2], s2[3]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int row = b_row0 + 64 * j;
            const int co = n0 + row;
            rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < BN && co < a.Cout && k0 < a.Cin) {
                const float* p = a.w + ((size_t)co * (a.KH * a.KW) + tap) * a.Cin + k0;
                if (a.vec) {
                    rb[j] = *reinterpret_cast<const float4*>(p);
                } else {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = (k0 + e < a.Cin) ? p[e] : 0.f;
                    rb[j] = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            float4 v = ra[j];
            if (a.scale && rav[j]) {
                v.x = apply_act(fmaf(v.x, rsc[j].x, rsh[j].x), a.act);
                v.y = apply_act(fmaf(v.y, rsc[j].y, rsh[j].y), a.act);
                v.z = apply_act(fmaf(v.z, rsc[j].z, rsh[j].z), a.act);
                v.w = apply_act(fmaf(v.w, rsc[j].w, rsh[j].w), a.act);
            }
            *reinterpret_cast<float4*>(&As[(buf * BM + (tid >> 2) + 64 * j) * LDK + c4 * 4]) = v;
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int row = b_row0 + 64 * j;
            if (row < BN) *reinterpret_cast<float4*>(&Bs[(buf * BN + row) * LDK + c4 * 4]) = rb[j];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 4;
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) load_tiles(it + 1);
        const float* Ab = As + (cur * BM + wm * WTM + frow) * LDK + fk;
        const float* Bb = Bs + (cur * BN + wn * WTN + frow) * LDK + fk;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK + kk * 8);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (it + 1 < T) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ---------------
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * WTN + j * 32 + (lane & 31);
            if (col >= a.Cout) continue;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < a.M) {
                    const size_t o = (size_t)row * a.Cout + col;
                    float v = acc[i][j][r] + bv;
                    if (a.resid) v += a.resid[o];
                    a.y[o] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// weight-gradient kernel: D[co][ci] (per tap) = sum over pixels
// ---------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;
    const float* dy;
    const float* scale;
    const float* shift;
    float* part;      // [splitk][Cout][taps][Cin]
    int N, Hin, Win, Cin, Hout, Wout, Cout;
    int KH, KW, stride, pad, gather, act, aff_stride;
    int M, tiles_co, tiles_ci, splitk, chunk;   // chunk = pixels per split (multiple of 16)
    int vec_i, vec_o;
    unsigned x_bytes, aff_bytes;
    const float* x_amax;                    // split-precision fp16 scheme: device-side bounds of |transformed x| and |dy|
    const float* dy_amax;
    int pad_w;                              // conv_wgrad_sp_kernel only: left padding, dy on a sub-grid (see ConvArgs)
    int dy_step, dy_row, dy_img, dy_off;
};

template <int BCO, int BCI, int WAVES_O, int WAVES_I>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int BKP = 16;                              // pixels per step
    constexpr int WTO = BCO / WAVES_O, WTI = BCI / WAVES_I;
    constexpr int MI = WTO / 32, NI = WTI / 32;
    constexpr int O_LD = (BKP * BCO / 4 + 255) / 256;    // float4 per thread for the dy tile
    constexpr int I_LD = (BKP * BCI / 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[2 * BKP * (BCO + BCI)];
    float* Os = lds;                       // [2][BKP][BCO]
    float* Is = lds + 2 * BKP * BCO;       // [2][BKP][BCI]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wo = wid / WAVES_I, wi = wid % WAVES_I;
    const int taps = a.KH * a.KW;
    int t = blockIdx.x;
    const int tap = t % taps; t /= taps;
    const int ci0 = (t % a.tiles_ci) * BCI;
    const int co0 = (t / a.tiles_ci) * BCO;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int z = blockIdx.y;
    const int p_begin = z * a.chunk;
    const int p_end = min(a.M, p_begin + a.chunk);
    const int T = (p_end > p_begin) ? (p_end - p_begin + BKP - 1) / BKP : 0;

    ConvArgs g;   // only the fields gather_src() reads
    g.stride = a.stride; g.pad = a.pad; g.gather = a.gather; g.Hin = a.Hin; g.Win = a.Win;

    float4 ro[O_LD], ri[I_LD], rsc[I_LD], rsh[I_LD];
    bool riv[I_LD];
    const int hw = a.Hout * a.Wout;

    auto load_tiles = [&](int it) {
        const int pb = p_begin + it * BKP;
#pragma unroll
        for (int j = 0; j < O_LD; ++j) {
            const int i = tid + 256 * j;
            const int p = i / (BCO / 4), c = (i % (BCO / 4)) * 4;
            const int m = pb + p, co = co0 + c;
            ro[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < BKP && m < p_end && co < a.Cout) {
                const float* q = a.dy + (size_t)m * a.Cout + co;
                if (a.vec_o) ro[j] = *reinterpret_cast<const float4*>(q);
                else {
                    float tt[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) tt[e] = (co + e < a.Cout) ? q[e] : 0.f;
                    ro[j] = make_float4(tt[0], tt[1], tt[2], tt[3]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < I_LD; ++j) {
            const int i = tid + 256 * j;
            const int p = i / (BCI / 4), c = (i % (BCI / 4)) * 4;
            const int m = pb + p, ci = ci0 + c;
            ri[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            riv[j] = false;
            if (p < BKP && m < p_end && ci < a.Cin) {
                const int n = m / hw;
                const int r = m - n * hw;
                const int oh = r / a.Wout, ow = r - oh * a.Wout;
                int sh, sw;
                if (gather_src(g, oh, ow, kh, kw, sh, sw)) {
                    riv[j] = true;
                    const float* q = a.x + ((size_t)(n * a.Hin + sh) * a.Win + sw) * a.Cin + ci;
                    const size_t so = (size_t)n * a.aff_stride + ci;
                    if (a.vec_i) {
                        ri[j] = *reinterpret_cast<const float4*>(q);
                        if (a.scale) {
                            rsc[j] = *reinterpret_cast<const float4*>(a.scale + so);
                            rsh[j] = *reinterpret_cast<const float4*>(a.shift + so);
                        }
                    } else {
                        float tt[4], s1[4], s2[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bool in = ci + e < a.Cin;
                            tt[e] = in ? q[e] : 0.f;
                            s1[e] = (in && a.scale) ? a.scale[so + e] : 0.f;
                            s2[e] = (in && a.scale) ? a.shift[so + e] : 0.f;
                        }
                        ri[j] = make_float4(tt[0], tt[1], tt[2], tt[3]);
                        rsc[j] = make_float4(s1[0], s1[1], s1[2], s1[3]);
                        rsh[j] = make_float4(s2[0], s2[1], s2[2], s2[3]);
                    }
                }
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < O_LD; ++j) {
            const int i = tid + 256 * j;
            const int p = i / (BCO / 4), c = (i % (BCO / 4)) * 4;
            if (p < BKP) *reinterpret_cast<float4*>(&Os[(buf * BKP + p) * BCO + c]) = ro[j];
        }
#pragma unroll
        for (int j = 0; j < I_LD; ++j) {
            const int i = tid + 256 * j;
            const int p = i / (BCI / 4), c = (i % (BCI / 4)) * 4;
            float4 v = ri[j];
            if (a.scale && riv[j]) {
                v.x = apply_act(fmaf(v.x, rsc[j].x, rsh[j].x), a.act);
                v.y = apply_act(fmaf(v.y, rsc[j].y, rsh[j].y), a.act);
                v.z = apply_act(fmaf(v.z, rsc[j].z, rsh[j].z), a.act);
                v.w = apply_act(fmaf(v.w, rsc[j].w, rsh[j].w), a.act);
                if (!a.vec_i) {   // zero the channel tail (silu(shift) of a non-existent channel must not leak)
                    const int ci = ci0 + c;
                    if (ci + 1 >= a.Cin) v.y = 0.f;
                    if (ci + 2 >= a.Cin) v.z = 0.f;
                    if (ci + 3 >= a.Cin) v.w = 0.f;
                }
            }
            if (p < BKP) *reinterpret_cast<float4*>(&Is[(buf * BKP + p) * BCI + c]) = v;
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (T > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    const int frow = lane & 31, fk = lane >> 5;
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) load_tiles(it + 1);
        const float* Ob = Os + (cur * BKP + fk) * BCO + wo * WTO + frow;
        const float* Ib = Is + (cur * BKP + fk) * BCI + wi * WTI + frow;
#pragma unroll
        for (int kk = 0; kk < BKP / 2; ++kk) {
            float af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = Ob[kk * 2 * BCO + i * 32];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = Ib[kk * 2 * BCI + j * 32];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < T) store_tiles(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int ci = ci0 + wi * WTI + j * 32 + (lane & 31);
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wo * WTO + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < a.Cout)
                    a.part[(((size_t)z * a.Cout + co) * taps + tap) * a.Cin + ci] = acc[i][j][r];
            }
        }
}

#include "conv_fast.h"
#include "conv_buf.h"
#include "conv_split.h"
#include "conv_wino.h"
#include "conv_wino4.h"
#include "conv_thin.h"

// out[i] (+)= sum_z part[z][i] in a fixed order (4 interleaved partial sums -> 4 loads in flight per thread)
__global__ void reduce_slabs_kernel(const float* part, float* out, size_t n, int slabs, int accumulate) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 3 < slabs; z += 4) {
        s0 += part[(size_t)z * n + i];
        s1 += part[(size_t)(z + 1) * n + i];
        s2 += part[(size_t)(z + 2) * n + i];
        s3 += part[(size_t)(z + 3) * n + i];
    }
    for (; z < slabs; ++z) s0 += part[(size_t)z * n + i];
    const float s = (s0 + s1) + (s2 + s3);
    out[i] = accumulate ? out[i] + s : s;
}

__global__ void weight_flip_kernel(const float* w, float* wt, int Cout, int KH, int KW, int Cin) {
    // wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci]; 32x32 LDS-tiled transpose per tap
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int kh = tap / KW, kw = tap % KW;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * KH * KW + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
    const int tapf = (KH - 1 - kh) * KW + (KW - 1 - kw);
    for (int r = ty; r < 32; r += 8) {
        int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) wt[((size_t)ci * KH * KW + tapf) * Cout + co] = tile[tx][r];
    }
}

// column sums: stage 1 = per-block partial over a row range, stage 2 = reduce_slabs_kernel
// amax (optional): max |a| of the whole matrix, one atomicMax per wave on the bit pattern (order-independent)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* a, float* part, long M, int C, long rows_per_block,
                                                             unsigned* amax) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    float mx = 0.f;
    if (c < C) {
        const long r0 = (long)blockIdx.y * rows_per_block;
        const long r1 = min(M, r0 + rows_per_block);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        long r = r0;
        for (; r + 3 < r1; r += 4) {
            const float v0 = a[r * C + c], v1 = a[(r + 1) * C + c], v2 = a[(r + 2) * C + c], v3 = a[(r + 3) * C + c];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
        }
        for (; r < r1; ++r) { const float v = a[r * C + c]; s0 += v; mx = fmaxf(mx, fabsf(v)); }
        part[(size_t)blockIdx.y * C + c] = (s0 + s1) + (s2 + s3);
    }
    if (amax) {
        mx = wave_max(mx);
        if ((threadIdx.x & 63) == 0) atomicMax(amax, __float_as_uint(mx));
    }
}

// vector version (C % 4 == 0): thread = (channel quad, row lane), 16-byte loads, LDS reduce over the row lanes
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const float* a, float* part, long M, int C, long rows_per_block,
                                                                 unsigned* amax) {
    __shared__ float4 sm[256];
    float mx = 0.f;
    const int QT = C / 4;
    const int Q = QT < 256 ? QT : 256, RL = 256 / Q;
    const int qi = threadIdx.x % Q, li = threadIdx.x / Q;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    for (int qb = 0; qb < QT; qb += Q) {
        const int q = qb + qi;
        const bool on = q < QT && li < RL;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        if (on) {
            const float4* p = reinterpret_cast<const float4*>(a) + q;
            long r = r0 + li;
            // four 16-byte loads in flight per thread (round 5: with two, 512 blocks of 256 threads kept 16 KB per CU in flight against
            // the ~60 KB an 8 TB/s / 2 us memory system needs: 0.35 of the roofline); the two accumulators keep their round-4 meaning
            // (even / odd row-lane steps), so the sums are taken in the same order as before
            for (; r + 3 * RL < r1; r += 4 * RL) {
                const float4 u = p[r * QT], v = p[(r + RL) * QT], u2 = p[(r + 2 * RL) * QT], v2 = p[(r + 3 * RL) * QT];
                s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s0.x += u2.x; s0.y += u2.y; s0.z += u2.z; s0.w += u2.w;
                s1.x += v2.x; s1.y += v2.y; s1.z += v2.z; s1.w += v2.w;
                mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))),
                                     fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))));
                mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(u2.x), fabsf(u2.y)), fmaxf(fabsf(u2.z), fabsf(u2.w))),
                                     fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w)))));
            }
            for (; r + RL < r1; r += 2 * RL) {
                const float4 u = p[r * QT], v = p[(r + RL) * QT];
                s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))),
                                     fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))));
            }
            if (r < r1) {
                const float4 u = p[r * QT];
                s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))));
            }
            s0.x += s1.x; s0.y += s1.y; s0.z += s1.z; s0.w += s1.w;
        }
        __syncthreads();
        sm[threadIdx.x] = s0;
        __syncthreads();
        if (on && li == 0) {
            for (int l = 1; l < RL; ++l) {
                const float4 v = sm[l * Q + qi];
                s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
            }
            reinterpret_cast<float4*>(part + (size_t)blockIdx.y * C)[q] = s0;
        }
    }
    if (amax) {
        mx = wave_max(mx);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) reinterpret_cast<float*>(sm)[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float* w = reinterpret_cast<const float*>(sm);
            atomicMax(amax, __float_as_uint(fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]))));
        }
    }
}

__global__ void upsample2x_bwd_kernel(const float* du, float* dx, int N, int H, int W, int C4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)N * H * W * C4;
    if (i >= total) return;
    int c = i % C4;
    size_t p = i / C4;
    int w = p % W; p /= W;
    int h = p % H;
    int n = p / H;
    const float4* s = reinterpret_cast<const float4*>(du);
    size_t W2 = 2 * (size_t)W;
    size_t b = (((size_t)n * 2 * H + 2 * h) * W2 + 2 * w) * C4 + c;
    float4 v0 = s[b], v1 = s[b + C4], v2 = s[b + W2 * C4], v3 = s[b + W2 * C4 + C4];
    float4 o;
    o.x = (v0.x + v1.x) + (v2.x + v3.x);
    o.y = (v0.y + v1.y) + (v2.y + v3.y);
    o.z = (v0.z + v1.z) + (v2.z + v3.z);
    o.w = (v0.w + v1.w) + (v2.w + v3.w);
    reinterpret_cast<float4*>(dx)[i] = o;
}

bool desc_ok(const favae_conv_desc* d) {
    return d && d->N > 0 && d->Hin > 0 && d->Win > 0 && d->Cin > 0 && d->Hout > 0 && d->Wout > 0 && d->Cout > 0 &&
           d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0 && d->gather >= 0 && d->gather <= 2 && d->act >= 0 &&
           d->act <= 3 && (long)d->N * d->Hout * d->Wout < (1L << 31) &&
           (d->lat_step == 0 || d->lat_step == 1 || (d->lat_step == 2 && (d->lat_side == 1 || d->lat_side == 2) &&
                                                      (unsigned)d->lat_oh < 2u && (unsigned)d->lat_ow < 2u)) &&
           d->pad + d->pad_dw >= 0 && d->w_rec_offset >= 0 && d->w_rec_offset % 16 == 0;
}

// descriptors only conv_fwd_sp_kernel / conv_wgrad_sp_kernel implement
bool desc_special(const favae_conv_desc* d) { return d->lat_step == 2 || d->pad_dw != 0; }

int wgrad_splitk(const favae_conv_desc* d, int tiles, int* chunk) {
    const long M = (long)d->N * d->Hout * d->Wout;
    long want = (1536 + tiles - 1) / tiles;                  // aim at ~6 workgroups per CU in total
    long maxk = (M + 16 * 16 - 1) / (16 * 16);               // at least 16 steps of 16 pixels per split
    long sk = want < 1 ? 1 : want;
    if (sk > maxk) sk = maxk;
    if (sk < 1) sk = 1;
    long ch = (M + sk - 1) / sk;
    ch = (ch + 15) / 16 * 16;
    sk = (M + ch - 1) / ch;
    *chunk = (int)ch;
    return (int)sk;
}

// Split-K of the nine-tap kernel (one 512-thread workgroup per CU): slabs are whole 16-pixel column strips of the images
// (N * Wout / 16 of them); the grid should be a whole number of 256-CU rounds, with equal strips per slab.
int nine_splitk(const favae_conv_desc* d, int tiles, int sk_max, int per_round, int* strips_per_slab) {
    const long NS = (long)d->N * (d->Wout / 16);
    long best_sk = 1, best_sps = NS;
    double best_eff = -1.0;
    for (int r = 1; r <= 3; ++r) {
        long sk = ((long)per_round * r) / tiles;
        if (sk < 1) sk = 1;
        if (sk > sk_max) sk = sk_max;
        if (sk > NS) sk = NS;
        const long sps = (NS + sk - 1) / sk;
        sk = (NS + sps - 1) / sps;
        const long total = sk * tiles;
        const double eff = ((double)total / ((double)per_round * ((total + per_round - 1) / per_round))) * ((double)NS / (double)(sk * sps));
        if (eff > best_eff + 0.01 || (r == 2 && eff >= best_eff - 0.01)) { best_eff = eff; best_sk = sk; best_sps = sps; }
    }
    *strips_per_slab = (int)best_sps;
    return (int)best_sk;
}

void wgrad_tiles(const favae_conv_desc* d, int* bco, int* bci) {
    *bco = d->Cout <= 32 ? 32 : 128;
    *bci = (d->Cin <= 32 && *bco == 128) ? 32 : 128;
}

}  // namespace

// RGB ends of the codec (conv_thin.h): 1 = thin input (Cin = 3), 2 = thin output (Cout = 3), 0 = neither.  FAVAE_CONV_THIN=0
// sends them back to the implicit-GEMM kernels (A/B switch).
constexpr int THIN_WGRAD_BLOCKS = 1024;
static bool thin_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_CONV_THIN"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
static int thin_kind(const favae_conv_desc* d, bool has_affine) {
    if (!desc_ok(d) || force_generic() || !thin_enabled() || desc_special(d)) return 0;
    if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->gather == FAVAE_GATHER_PLAIN && d->Hout == d->Hin &&
          d->Wout == d->Win))
        return 0;
    if ((size_t)d->N * d->Hin * d->Win >= ((size_t)1 << 31)) return 0;
    if (d->Cin == 3 && (d->Cout == 64 || d->Cout == 128) && !has_affine) return 1;
    if (d->Cout == 3 && (d->Cin == 64 || d->Cin == 128)) return 2;
    return 0;
}
static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// geometry of conv3x3_wino_sp_kernel (conv_wino.h): dense 3x3, stride 1, pad 1, 16 x 16-pixel tiles x 64 output channels, h3 scheme.  Its
// 64-channel tile also takes Cout == 64 (the direct split kernels need more than 64: their tiles are 128 wide) and, through XFORM = 3,
// any fused activation (LeakyReLU / ReLU on load: the VGG16 convs of LPIPS, losses/lpips.py:74-96).
// Conv modes that have a Winograd kernel for this conv: h3 always; the one-plane 16-bit modes h1 / b1 (round 5) where the wide tiling
// applies (Cout % 128 == 0) -- FAVAE_WINO1=0 keeps those modes on the direct kernels (A/B).
static bool use_wino1() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_WINO1"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}
static bool use_wino_wide();
static bool wino_mode(const favae_conv_desc* d) {
    const int m = conv_mode();
    // h1: wherever the kernel tiles (64-channel convs -- the first VGG16 layers of LPIPS -- otherwise fall to the fp32-MFMA kernel);
    // b1: only where the caller can still choose the direct bf16 kernel per call (the 128-channel tiles), see include/favae_hip.h
    return m == 2 || (m == 1 && use_wino1() && d->Cout % 64 == 0) || (m == 4 && use_wino1() && use_wino_wide() && d->Cout % 128 == 0);
}
static bool wino_geometry(const favae_conv_desc* d) {
    return use_wino() && wino_mode(d) && !desc_special(d) && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
           d->gather == FAVAE_GATHER_PLAIN && d->Hout == d->Hin && d->Wout == d->Win && d->Hin % 16 == 0 && d->Win % 16 == 0 &&
           d->Cout % 64 == 0 && d->Cin % 16 == 0;
}
// Cout == 64 has exactly ONE split-operand kernel, the Winograd one (the direct split tiles are 128 channels wide): such a conv wants split
// weights exactly when wino_ok() will hand it to that kernel -- the same conditions, spelled out here because wino_ok() itself is
// defined through this predicate (with FAVAE_CONV_HALO=0 the geometry alone said "eligible", Python built plain h3 records
// and the fp32 buffer kernel read them as weights).
static bool wino64_only_ok(const favae_conv_desc* d, bool has_affine) {
    return d->Cout == 64 && wino_geometry(d) && use_halo() && (size_t)d->Cout * d->Cin * 64 < (1u << 31) &&
           (!has_affine || d->Cin <= wino::AFF_C) && (size_t)d->N * d->Hout * d->Wout * d->Cout * 4 < ((size_t)1 << 32);
}
static bool sp_fwd_eligible(const favae_conv_desc* d, bool has_affine) {
    if (!desc_ok(d) || force_generic() || force_nobuf() || !use_b6()) return false;
    const size_t xb = (size_t)d->N * d->Hin * d->Win * d->Cin * 4, wb = (size_t)d->Cout * d->KH * d->KW * d->Cin * 6;
    return (d->Cout > 64 || wino64_only_ok(d, has_affine)) && d->Cin % 16 == 0 && xb < (1u << 31) && wb < (1u << 31) &&
           (d->gather == FAVAE_GATHER_PLAIN || !has_affine);
}

static int wrec_bytes(int planes) {
    return planes == 1 ? sp::Scheme<1>::WREC : (planes == 2 ? sp::Scheme<2>::WREC : (planes == 4 ? sp::Scheme<4>::WREC : sp::Scheme<3>::WREC));
}

extern "C" int favae_set_conv_mode(int planes) {
    FAVAE_REQUIRE(planes >= 0 && planes <= 4);
    g_conv_mode = planes;
    return FAVAE_OK;
}

extern "C" int favae_get_conv_mode(void) { return conv_mode(); }

extern "C" int favae_conv_wants_split_weights(const favae_conv_desc* d, int has_affine) {
    return sp_fwd_eligible(d, has_affine != 0) ? conv_mode() : 0;
}

extern "C" size_t favae_split_weights_bytes(int64_t n, int planes) {
    if (n <= 0 || n % 4 || planes < 1 || planes > 4) return 0;
    return (size_t)sp::WHDR + (size_t)(n / 4) * wrec_bytes(planes);
}

namespace {
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, int vec, unsigned* __restrict__ out) {
    float m = 0.f;
    const size_t n4 = vec ? n / 4 : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const size_t st = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * st < n4; i += 4 * st) {           // four loads in flight per thread
        const float4 v = x4[i], u = x4[i + st], w = x4[i + 2 * st], t = x4[i + 3 * st];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w)));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
    }
    for (; i < n4; i += st) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per block (thousands of same-address atomics serialise at ~10 ns each); non-negative floats order like
    // their bit patterns, and a maximum is order-independent -> deterministic
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

int launch_absmax(const float* x, int64_t n, float* out, hipStream_t s) {
    if (favae_zero_target(out, sizeof(float), s) != hipSuccess) return favae_prof_fail_(FAVAE_ERR_LAUNCH);
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    FAVAE_PROF_NOTE(0, 4.0 * n);
    FAVAE_KLAUNCH(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, (size_t)n, (((uintptr_t)x) & 15) == 0 ? 1 : 0,
                       (unsigned*)out);
    return FAVAE_OK;
}
}  // namespace

extern "C" int favae_absmax(const float* x, int64_t n, float* out, favae_stream_t stream) {
    FAVAE_REQUIRE(x && out && n > 0);
    const int rc = launch_absmax(x, n, out, (hipStream_t)stream);
    if (rc != FAVAE_OK) return rc;
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

namespace {
// max |x| per segment of a flat buffer: seg_off[nseg + 1] element offsets (ascending); one block per 4096-element chunk of a segment
// (chunks never straddle segments: chunk_seg / chunk_first map the block to its segment and first element), one atomicMax per block
__global__ __launch_bounds__(256) void segment_absmax_kernel(const float* __restrict__ x, const int64_t* __restrict__ seg_off,
                                                             const int* __restrict__ chunk_seg, const int64_t* __restrict__ chunk_first,
                                                             unsigned* __restrict__ out) {
    const int sg = chunk_seg[blockIdx.x];
    const int64_t b = chunk_first[blockIdx.x], e = min(seg_off[sg + 1], b + 4096);
    float m = 0.f;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out + sg, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}
}  // namespace

// out[s] = max |x[seg_off[s] .. seg_off[s+1])| for nseg segments of one flat buffer in ONE launch (the |max| of every weight tensor of
// a model whose parameters are views of a flat buffer: refreshed once per optimizer step instead of once per conv call).
// seg_off / chunk_seg / chunk_first: DEVICE arrays built by the caller (chunks of <= 4096 elements, none across a segment boundary).
extern "C" int favae_segment_absmax(const float* x, const int64_t* seg_off, int nseg, const int* chunk_seg, const int64_t* chunk_first,
                                    int nchunks, float* out, favae_stream_t stream) {
    FAVAE_REQUIRE(x && seg_off && chunk_seg && chunk_first && out && nseg > 0 && nchunks > 0);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * nseg, s) != hipSuccess) return favae_prof_fail_(FAVAE_ERR_LAUNCH);
    FAVAE_KLAUNCH(segment_absmax_kernel, dim3(nchunks), dim3(256), 0, s, x, seg_off, chunk_seg, chunk_first, (unsigned*)out);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

static int split_weights_impl(const float* in, void* out, int64_t n, int planes, const float* amax, favae_stream_t stream);

extern "C" int favae_split_weights(const float* in, void* out, int64_t n, int planes, favae_stream_t stream) {
    return split_weights_impl(in, out, n, planes, nullptr, stream);
}

// favae_split_weights with max |in| supplied by the caller (device float, e.g. from favae_segment_absmax): no reduction pass
extern "C" int favae_split_weights_amax(const float* in, void* out, int64_t n, int planes, const float* amax, favae_stream_t stream) {
    FAVAE_REQUIRE(amax);
    return split_weights_impl(in, out, n, planes, amax, stream);
}

static int split_weights_impl(const float* in, void* out, int64_t n, int planes, const float* amax, favae_stream_t stream) {
    FAVAE_REQUIRE(in && out && n > 0 && n % 4 == 0 && planes >= 1 && planes <= 4);
    FAVAE_REQUIRE((((uintptr_t)out) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    unsigned* rec = (unsigned*)((char*)out + sp::WHDR);
    if (planes == 4) {                          // one bf16 plane: no range
        FAVAE_KLAUNCH((split_w_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)in, rec, (size_t)(n / 4),
                      (const float*)nullptr, (float*)nullptr);
        FAVAE_CHECK_LAUNCH();
        return FAVAE_OK;
    }
    if (planes <= 2 && amax) {                 // the header must hold the maximum (later kernels read it there): the kernel copies it
        if (planes == 2)
            FAVAE_KLAUNCH((split_w_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)in, rec, (size_t)(n / 4), amax,
                          (float*)out);
        else
            FAVAE_KLAUNCH((split_w_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)in, rec, (size_t)(n / 4), amax,
                          (float*)out);
    } else if (planes <= 2) {
        const int rc = launch_absmax(in, n, (float*)out, s);
        if (rc != FAVAE_OK) return rc;
        if (planes == 2)
            FAVAE_KLAUNCH((split_w_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)in, rec, (size_t)(n / 4),
                               (const float*)out, (float*)nullptr);
        else
            FAVAE_KLAUNCH((split_w_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)in, rec, (size_t)(n / 4),
                               (const float*)out, (float*)nullptr);
    } else {
        FAVAE_KLAUNCH((split_w_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)in, rec, (size_t)(n / 4),
                           (const float*)nullptr, (float*)nullptr);
    }
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

struct GnBwdEpi {              // GroupNorm-backward partial sums in the data-gradient epilogue (ConvArgs::gb_*)
    const float *x, *mean, *rstd, *gamma, *beta;
    double* part;
    int groups, act;
};
static int conv_fwd_impl(const favae_conv_desc* d, const float* x, const float* w, const float* bias, const float* resid,
                         const float* scale, const float* shift, float* y, int wplanes, const float* x_amax,
                         favae_stream_t stream, const GnBwdEpi* gb = nullptr, double* stats_part = nullptr,
                         float* stats_amax = nullptr);

extern "C" int favae_conv_fwd(const favae_conv_desc* d, const float* x, const float* w, const float* bias,
                              const float* resid, const float* scale, const float* shift, float* y,
                              favae_stream_t stream) {
    return conv_fwd_impl(d, x, w, bias, resid, scale, shift, y, 0, nullptr, stream, nullptr);
}

static bool wino_ok(const favae_conv_desc* d, bool has_affine);
static bool wino4_ok(const favae_conv_desc* d, bool has_affine);
// planes word of a call that passes Winograd records: F(2x2) records go with wino_ok, F(4x4) records (FAVAE_PLANES_WINO4 on top) with wino4_ok
static bool wino_planes_ok(const favae_conv_desc* d, int planes, bool has_affine) {
    planes &= ~FAVAE_PLANES_BF16IO;                                                          // storage type of the activations: not a records property
    if (planes == (conv_mode() | FAVAE_PLANES_WINO)) return wino_ok(d, has_affine);          // 2 (h3), 1 (h1) or 4 (b1) | the flag
    if (planes == (2 | FAVAE_PLANES_WINO | FAVAE_PLANES_WINO4)) return wino4_ok(d, has_affine);
    return false;
}
extern "C" int favae_conv_fwd_split(const favae_conv_desc* d, const float* x, const void* wsplit, int planes,
                                    const float* x_absmax, const float* bias, const float* resid, const float* scale,
                                    const float* shift, float* y, favae_stream_t stream) {
    if (!sp_fwd_eligible(d, scale != nullptr)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (planes & FAVAE_PLANES_WINO) {
        if (!wino_planes_ok(d, planes, scale != nullptr)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
        FAVAE_REQUIRE(wsplit && (x_absmax || (planes & 0xff) == 4));
    } else {
        const int pl = planes & ~FAVAE_PLANES_BF16IO;
        FAVAE_REQUIRE(wsplit && (pl == 3 || pl == 4 || ((pl == 2 || pl == 1) && x_absmax)));
    }
    return conv_fwd_impl(d, x, (const float*)wsplit, bias, resid, scale, shift, y, planes, x_absmax, stream);
}

// 1 when favae_conv_fwd_split(d, ...) runs the dense 3x3 halo kernel with fp16 planes (two: h3, or one: h1) -- the kernel whose
// epilogue can emit GroupNorm sums (SE / GB variants)
static bool halo3_fp16_ok(const favae_conv_desc* d, bool has_affine);
static bool wino_ok(const favae_conv_desc* d, bool has_affine);

static bool halo3_fp16_ok(const favae_conv_desc* d, bool has_affine) {
    if (!sp_fwd_eligible(d, has_affine) || conv_mode() == 3 || conv_mode() == 0 || desc_special(d) || !use_halo()) return false;
    const size_t xb = (size_t)d->N * d->Hin * d->Win * d->Cin * 4, wb = (size_t)d->Cout * 9 * d->Cin * 4;
    // shapes only the Winograd kernel takes (64 output channels; a fused activation other than SiLU): eligible exactly when IT runs them
    const bool wg = wino_geometry(d) && (size_t)d->Cout * d->Cin * 64 < (1u << 31) && (!has_affine || d->Cin <= wino::AFF_C);
    if (has_affine && d->act != FAVAE_ACT_NONE && d->act != FAVAE_ACT_SILU && !wg) return false;
    return (d->Cout > 64 || wg) && d->Cin % 16 == 0 && d->stride == 1 && d->gather == FAVAE_GATHER_PLAIN && d->KH == 3 && d->KW == 3 &&
           d->pad == 1 && d->Hout == d->Hin && d->Wout == d->Win && d->Hin % 8 == 0 && d->Win % 16 == 0 && xb < (1u << 31) &&
           wb < (1u << 31) && (size_t)d->N * d->Hout * d->Wout * d->Cout * 4 < ((size_t)1 << 32);
}

// Dense 3x3 convs of the h3 scheme whose shape tiles into 16 x 16 pixels x 64 output channels run conv3x3_wino_sp_kernel
// (conv_wino.h): the caller then passes Winograd weight records (favae_wino_weights) and planes = 2 | FAVAE_PLANES_WINO.
static bool wino_ok(const favae_conv_desc* d, bool has_affine) {
    return use_wino() && wino_mode(d) && halo3_fp16_ok(d, has_affine) && d->Hin % 16 == 0 && d->Win % 16 == 0 && d->Cout % 64 == 0 &&
           (size_t)d->Cout * d->Cin * 64 < (1u << 31) && (!has_affine || d->Cin <= wino::AFF_C);
}

// 16 x 8-pixel x 128-channel tiling of the same kernel (conv_wino.h, WIDE): every F(2x2) launch whose output channels tile by 128.
// FAVAE_WINO_WIDE=0 / favae_set_wino_wide(0): the 16 x 16 x 64 tiling everywhere (A/B; the results are bit-identical, the per-tile
// partial sums of the statistics epilogues are on a finer grid).
int g_wino_wide = -1;
static bool use_wino_wide() {
    if (g_wino_wide < 0) { const char* e = getenv("FAVAE_WINO_WIDE"); g_wino_wide = (e && e[0] == '0') ? 0 : 1; }
    return g_wino_wide == 1;
}
static bool wino_wide_ok(const favae_conv_desc* d, bool has_affine) {
    return use_wino_wide() && wino_ok(d, has_affine) && d->Cout % 128 == 0;
}
extern "C" int favae_set_wino_wide(int on) {
    const int prev = use_wino_wide() ? 1 : 0;
    g_wino_wide = on ? 1 : 0;
    return prev;
}

// F(4x4, 3x3) variant (conv_wino4.h): 32 x 16-pixel tiles, K loop unrolled by four chunks, records of 144 bytes per (co, ci) pair.  The
// geometry check only: WHERE it is used (data gradients; decoder layers) is the caller's policy (ops.py, FAVAE_WINO4) because the
// accuracy bar depends on what consumes the result.  FAVAE_WINO4=0 in the environment: never.
int g_wino4 = -1;
static bool use_wino4() {
    if (g_wino4 < 0) { const char* e = getenv("FAVAE_WINO4"); g_wino4 = (e && e[0] == '0') ? 0 : 1; }
    return g_wino4 == 1;
}
static bool wino4_ok(const favae_conv_desc* d, bool has_affine) {
    return use_wino4() && conv_mode() == 2 && wino_ok(d, has_affine) && d->Win % 32 == 0 && d->Cin % 64 == 0 && d->Cin <= 736 &&
           (size_t)d->Cout * d->Cin * 144 < (1u << 31);
}
extern "C" int favae_conv_wino4_ok(const favae_conv_desc* d, int has_affine) { return desc_ok(d) && wino4_ok(d, has_affine != 0) ? 1 : 0; }
extern "C" int favae_set_wino4(int on) {
    const int prev = use_wino4() ? 1 : 0;
    g_wino4 = on ? 1 : 0;
    return prev;
}
extern "C" size_t favae_wino4_weights_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cout % 16 || Cin % 16) return 0;
    return (size_t)sp::WHDR + (size_t)Cout * Cin * 144;
}

// switch the Winograd path on / off at run time (overrides FAVAE_WINO); returns the previous setting
extern "C" int favae_set_wino(int on) {
    const int prev = use_wino() ? 1 : 0;
    g_wino = on ? 1 : 0;
    return prev;
}

#ifdef FAVAE_WINO_TRACE
// trace build only (tools/wino_trace.sh): copy the phase stamps of conv3x3_wino_sp_kernel to the host and clear them
extern "C" int favae_debug_wino_trace(void* host_dst, size_t bytes) {
    if (bytes > sizeof(g_wino_trace)) bytes = sizeof(g_wino_trace);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_wino_trace), bytes) != hipSuccess) return -2;
    static unsigned long long zeros[1] = {0};
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_wino_trace)) != hipSuccess) return -3;
    return hipMemset(dptr, 0, sizeof(g_wino_trace)) == hipSuccess ? 0 : -4;
}
#endif

// see common.h: pointers into [p, p + bytes) are reduction targets the caller keeps zeroed (bytes = 0 / p = NULL: no arena)
extern "C" int favae_set_zero_arena(const void* p, size_t bytes) {
    g_zero_lo = bytes ? (const char*)p : nullptr;
    g_zero_hi = bytes ? (const char*)p + bytes : nullptr;
    return FAVAE_OK;
}

// side-effect-free read of the switch
extern "C" int favae_get_wino(void) { return use_wino() ? 1 : 0; }

extern "C" int favae_conv_wino_ok(const favae_conv_desc* d, int has_affine) { return desc_ok(d) && wino_ok(d, has_affine != 0) ? 1 : 0; }

extern "C" size_t favae_wino_weights_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cout % 16 || Cin % 16) return 0;
    return (size_t)sp::WHDR + (size_t)Cout * Cin * 64;
}

// Winograd weight records of w (OHWI fp32 [Cout][3][3][Cin]) for conv3x3_wino_sp_kernel: U = G g G^T, scaled, split into two fp16
// planes, in MFMA fragment order.  flip = 0: the forward conv (Cout outputs); flip = 1: its data gradient (Cin outputs, taps
// flipped).  amax: device float max|w| (nullptr: computed here).  The header (float[0]) holds max|w|.
extern "C" int favae_wino_weights(const float* w, void* out, int Cout, int Cin, int flip, const float* amax, favae_stream_t stream) {
    FAVAE_REQUIRE(w && out && favae_wino_weights_bytes(Cout, Cin) && (((uintptr_t)out) & 15) == 0 && flip >= 0 && flip <= 5);
    const int vec = (((uintptr_t)w) & 15) == 0 ? 1 : 0;
    const bool f43 = (flip & 2) != 0;            // bit 1: F(4x4, 3x3) records (favae_wino4_weights_bytes) for conv3x3_wino4_sp_kernel
    const bool bf = (flip & 4) != 0;             // bit 2: F(2x2) records with a bf16 head plane (the one-plane bf16 mode b1)
    flip &= 1;
    FAVAE_REQUIRE(flip ? (Cin % 64 == 0 && Cout % 16 == 0) : (Cout % 64 == 0 && Cin % 16 == 0));
    hipStream_t s = (hipStream_t)stream;
    float* hdr = nullptr;
    if (!amax) {
        const int rc = launch_absmax(w, (int64_t)Cout * 9 * Cin, (float*)out, s);
        if (rc != FAVAE_OK) return rc;
        amax = (const float*)out;
    } else {
        hdr = (float*)out;
    }
    const unsigned blocks = (unsigned)(((size_t)Cout * Cin / 8 + 255) / 256);
    FAVAE_PROF_NOTE(0, 4.0 * Cout * 9 * Cin + (f43 ? 144.0 : 64.0) * Cout * Cin);
    if (f43) {
        if (flip) FAVAE_KLAUNCH((wino4_weights_kernel<true>), dim3(blocks), dim3(256), 0, s, w, (unsigned char*)out + sp::WHDR, Cout, Cin, amax, hdr, vec);
        else FAVAE_KLAUNCH((wino4_weights_kernel<false>), dim3(blocks), dim3(256), 0, s, w, (unsigned char*)out + sp::WHDR, Cout, Cin, amax, hdr, vec);
        FAVAE_CHECK_LAUNCH();
        return FAVAE_OK;
    }
    if (bf && flip) FAVAE_KLAUNCH((wino_weights_kernel<true, true>), dim3(blocks), dim3(256), 0, s, w, (unsigned char*)out + sp::WHDR, Cout, Cin, amax, hdr, vec);
    else if (bf) FAVAE_KLAUNCH((wino_weights_kernel<false, true>), dim3(blocks), dim3(256), 0, s, w, (unsigned char*)out + sp::WHDR, Cout, Cin, amax, hdr, vec);
    else if (flip) FAVAE_KLAUNCH((wino_weights_kernel<true>), dim3(blocks), dim3(256), 0, s, w, (unsigned char*)out + sp::WHDR, Cout, Cin, amax, hdr, vec);
    else FAVAE_KLAUNCH((wino_weights_kernel<false>), dim3(blocks), dim3(256), 0, s, w, (unsigned char*)out + sp::WHDR, Cout, Cin, amax, hdr, vec);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

// The records of many weight tensors in one launch (a model's flat parameter buffer after an optimizer step): jobs = DEVICE array of
// favae_wino_job {w, out (header + records, 16-byte aligned), amax (device float, required), Cout, Cin, flip, block0}, block_job = DEVICE
// int[nblocks]: job index of every block; job j owns blocks block0 .. block0 + ceil(Cout Cin / 8 / 256) - 1.
extern "C" int favae_wino_weights_grouped(const void* jobs, const int* block_job, int nblocks, favae_stream_t stream) {
    FAVAE_REQUIRE(jobs && block_job && nblocks > 0);
    static_assert(sizeof(WinoJob) == sizeof(favae_wino_job), "favae_wino_job layout");
    FAVAE_PROF_NOTE(0, 0);
    FAVAE_KLAUNCH(wino_weights_grouped_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const WinoJob*)jobs, block_job);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

// Data gradient of a conv whose INPUT was act(GroupNorm(x)): da = conv(dy, flipped w) as favae_conv_fwd_split, plus, from the
// epilogue, the per-tile partial sums of the GroupNorm backward (norm.hip: S1 = sum dy, S2 = sum dy xhat with dy = da act'(y))
// into part[N][tiles][C][2] (double; tiles = (H/8) (W/16) per image) -- favae_gn_act_bwd_tiles consumes them.
// `planes`: the planes word the conv call will be given -- only FAVAE_PLANES_WINO4 is looked at (F(4x4) records: that kernel's grid)
static int wino_part_tiles(const favae_conv_desc* d, bool has_affine, int planes) {
    if ((planes & FAVAE_PLANES_WINO4) && wino4_ok(d, has_affine)) return (d->Hout / 16) * (d->Wout / 16);
    if (wino_wide_ok(d, has_affine)) return (d->Hout / 8) * (d->Wout / 16);
    return (d->Hout / 16) * (d->Wout / 16);
}
// One-plane modes let the caller pick the DIRECT kernel for a shape the Winograd kernel also takes (planes without FAVAE_PLANES_WINO).
// The tile counts above are the Winograd kernel's; the direct kernel's grid is 16 x 8 pixels, which is the Winograd grid only in the
// wide tiling.  Where the two differ (Cout % 128 != 0, or FAVAE_WINO_WIDE=0) the direct kernel would write twice the tiles the caller
// sized `part` for: such a call is refused (tests/test_gpu_ops.py::test_one_plane_direct_stats_call_is_refused_off_the_wide_grid).
static bool direct_grid_mismatch(const favae_conv_desc* d, int planes, bool has_affine) {
    return conv_mode() != 2 && !(planes & FAVAE_PLANES_WINO) && wino_ok(d, has_affine) && !wino_wide_ok(d, has_affine);
}
static bool wgrad_nine_geom_ok(const favae_conv_desc* d);
// bf16 activation storage: does the kernel that would run `d` have the bf16 instantiation?  kind 0: forward / plain conv call
// (favae_conv_fwd_split, _stats), 1: data gradient with the GroupNorm-backward epilogue (favae_conv_dgrad_gnbwd), 2: weight gradient.
extern "C" int favae_conv_bf16io_ok(const favae_conv_desc* d, int has_affine, int kind) {
    if (!desc_ok(d) || conv_mode() != 4) return 0;
    if (kind == 2) {
        if (!wgrad_nine_geom_ok(d) || !use_nine() || nine_mode() != 1 || d->Hout != d->Hin || d->Wout != d->Win) return 0;
        return (has_affine && d->act != FAVAE_ACT_NONE && d->act != FAVAE_ACT_SILU) ? 0 : 1;
    }
    if (!halo3_fp16_ok(d, has_affine != 0) || d->Cout <= 64) return 0;
    if (kind == 1 && has_affine) return 0;
    return 1;
}
extern "C" int favae_conv_gnbwd_tiles(const favae_conv_desc* d, int planes) {
    if (!desc_ok(d) || !halo3_fp16_ok(d, false)) return 0;
    if (wino_ok(d, false)) return wino_part_tiles(d, false, planes);
    return (d->Hout / 8) * (d->Wout / 16);
}

// Forward conv that also emits pass 1 of the GroupNorm consuming its output: per-tile (sum y, sum y^2) per channel into
// part[N][tiles][Cout][2] (double); favae_gn_stats_tiles turns them into the statistics (norm.hip).
extern "C" int favae_conv_stats_tiles(const favae_conv_desc* d, int has_affine, int planes) {
    if (!desc_ok(d) || !halo3_fp16_ok(d, has_affine != 0)) return 0;
    if (has_affine && d->act != FAVAE_ACT_SILU) return 0;
    if (wino_ok(d, has_affine != 0)) return wino_part_tiles(d, has_affine != 0, planes);
    return (d->Hout / 8) * (d->Wout / 16);
}

extern "C" int favae_conv_fwd_split_stats(const favae_conv_desc* d, const float* x, const void* wsplit, int planes,
                                          const float* x_absmax, const float* bias, const float* resid, const float* scale,
                                          const float* shift, float* y, void* part, size_t part_bytes, float* y_absmax,
                                          favae_stream_t stream) {
    FAVAE_REQUIRE(desc_ok(d) && wsplit && (x_absmax || (planes & 0xff) == 4) && part);
    const int tiles = favae_conv_stats_tiles(d, scale != nullptr, planes);
    // the tile grid of the partial sums is the kernel's: Winograd records go with the Winograd kernel's tiles and nothing else.  In the
    // one-plane modes the caller chooses between the Winograd and the direct kernel per call (both have the 16 x 8 grid there)
    if (((planes & FAVAE_PLANES_WINO) != 0) != wino_ok(d, scale != nullptr) && !(conv_mode() != 2 && !(planes & FAVAE_PLANES_WINO)))
        return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (direct_grid_mismatch(d, planes, scale != nullptr)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (planes & FAVAE_PLANES_WINO) planes = wino_planes_ok(d, planes, scale != nullptr) ? planes : 0;
    if (!tiles || ((planes & 0xff) != 2 && (planes & 0xff) != 1 && (planes & 0xff) != 4)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (part_bytes < (size_t)d->N * tiles * d->Cout * 2 * sizeof(double)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    if (y_absmax && favae_zero_target(y_absmax, sizeof(float), (hipStream_t)stream) != hipSuccess) return favae_prof_fail_(FAVAE_ERR_LAUNCH);
    return conv_fwd_impl(d, x, (const float*)wsplit, bias, resid, scale, shift, y, planes, x_absmax, stream, nullptr,
                         (double*)part, y_absmax);
}

extern "C" int favae_conv_dgrad_gnbwd(const favae_conv_desc* d, const float* dy, const void* wsplit, int planes,
                                      const float* dy_absmax, float* da, const float* x, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, int groups, int act, void* part, size_t part_bytes,
                                      favae_stream_t stream) {
    FAVAE_REQUIRE(desc_ok(d) && dy && wsplit && (dy_absmax || (planes & 0xff) == 4) && da && x && mean && rstd && gamma && beta && part && groups > 0);
    const int tiles = favae_conv_gnbwd_tiles(d, planes);
    if (((planes & FAVAE_PLANES_WINO) != 0) != wino_ok(d, false) && !(conv_mode() != 2 && !(planes & FAVAE_PLANES_WINO)))
        return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (direct_grid_mismatch(d, planes, false)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (planes & FAVAE_PLANES_WINO) planes = wino_planes_ok(d, planes, false) ? planes : 0;
    if (!tiles || ((planes & 0xff) != 2 && (planes & 0xff) != 1 && (planes & 0xff) != 4) || d->Cout % groups != 0) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (part_bytes < (size_t)d->N * tiles * d->Cout * 2 * sizeof(double)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    GnBwdEpi gb{x, mean, rstd, gamma, beta, (double*)part, groups, act};
    return conv_fwd_impl(d, dy, (const float*)wsplit, nullptr, nullptr, nullptr, nullptr, da, planes, dy_absmax, stream, &gb);
}

static int conv_fwd_impl(const favae_conv_desc* d, const float* x, const float* w, const float* bias, const float* resid,
                         const float* scale, const float* shift, float* y, int wplanes, const float* x_amax,
                         favae_stream_t stream, const GnBwdEpi* gb, double* stats_part, float* stats_amax) {
    FAVAE_REQUIRE(desc_ok(d) && x && w && y);
    FAVAE_REQUIRE((scale == nullptr) == (shift == nullptr));
    // roofline numerators of this conv: 2*M*Cout*KH*KW*Cin FLOP; one read of x (+ resid), one write of y, the weights --
    // and, for a data gradient with the GroupNorm-backward epilogue, the forward activation the epilogue reads next to its output tile
    // (pass 1 of that GroupNorm's backward, which would otherwise read both tensors in a kernel of its own).  Rounds 1-4 left that
    // tensor out: the data-gradient line of the bench showed traffic / algorithmic = 1.94 where the kernel fetches 1.29 x its operands.
    FAVAE_PROF_NOTE(2.0 * d->N * d->Hout * d->Wout * d->Cout * d->KH * d->KW * d->Cin,
                    4.0 * ((double)d->N * d->Hin * d->Win * d->Cin +
                           (double)d->N * d->Hout * d->Wout * d->Cout * (1 + (resid ? 1 : 0) + (gb ? 1 : 0)) +
                           (double)d->Cout * d->KH * d->KW * d->Cin));
    const bool wino = (wplanes & FAVAE_PLANES_WINO) != 0;     // Winograd records (favae_wino_weights): conv3x3_wino_sp_kernel
    const bool wino4 = (wplanes & FAVAE_PLANES_WINO4) != 0;   // ... F(4x4, 3x3) records: conv3x3_wino4_sp_kernel
    // bf16 activation storage (round 6): x, resid, y (and the GroupNorm input of the GB epilogue) are bf16 tensors; scheme 4 only, and
    // only the kernels with that instantiation (the dense 3x3 halo kernel and the wide Winograd kernel): favae_conv_bf16io_ok
    const bool bf16io = (wplanes & FAVAE_PLANES_BF16IO) != 0;
    wplanes &= 0xff;
    if (bf16io && (wplanes != 4 || wino4 || ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)resid)) & 7) != 0))
        return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (bf16io)          // the activation bytes of the note above at two bytes per element
        FAVAE_PROF_NOTE(2.0 * d->N * d->Hout * d->Wout * d->Cout * d->KH * d->KW * d->Cin,
                        2.0 * ((double)d->N * d->Hin * d->Win * d->Cin +
                               (double)d->N * d->Hout * d->Wout * d->Cout * (1 + (resid ? 1 : 0) + (gb ? 1 : 0))) +
                            4.0 * (double)d->Cout * d->KH * d->KW * d->Cin);
    const bool w6 = wplanes != 0;                            // pre-split weights: records start behind the header
    if (!w6) {
        const int tk = thin_kind(d, scale != nullptr);
        const int xf0 = scale ? (d->act == FAVAE_ACT_SILU ? 2 : (d->act == FAVAE_ACT_NONE ? 1 : 3)) : 0;
        if (d->Cout == 1 && d->Cin % 4 == 0 && d->gather == FAVAE_GATHER_PLAIN && !desc_special(d) && !resid && !force_generic() &&
            thin_enabled() && al16(x) && al16(w) && al16(scale) && al16(shift)) {            // one output channel: a wave per output pixel
            ThinArgs t{};
            t.x = x; t.w = w; t.bias = bias; t.scale = scale; t.shift = shift; t.y = y;
            t.N = d->N; t.H = d->Hin; t.W = d->Win; t.Cw = d->Cin;
            t.aff_stride = d->affine_per_image ? d->Cin : 0;
            t.act = d->act;
            const long M1 = (long)d->N * d->Hout * d->Wout;
            const dim3 g1((unsigned)((M1 + 3) / 4));
#define FAVAE_LAUNCH_COUT1(X) FAVAE_KLAUNCH((conv_cout1_kernel<X>), g1, dim3(256), 0, (hipStream_t)stream, t, d->Hout, d->Wout, d->KH, d->KW, d->stride, d->pad)
            if (xf0 == 0) FAVAE_LAUNCH_COUT1(0);
            else if (xf0 == 1) FAVAE_LAUNCH_COUT1(1);
            else if (xf0 == 2) FAVAE_LAUNCH_COUT1(2);
            else FAVAE_LAUNCH_COUT1(3);
#undef FAVAE_LAUNCH_COUT1
            FAVAE_CHECK_LAUNCH();
            return FAVAE_OK;
        }
        if (tk && al16(x) && al16(y) && al16(resid) && al16(scale) && al16(shift)) {
            ThinArgs t{};
            t.x = x; t.w = w; t.bias = bias; t.resid = resid; t.scale = scale; t.shift = shift; t.y = y;
            t.N = d->N; t.H = d->Hin; t.W = d->Win;
            t.aff_stride = d->affine_per_image ? d->Cin : 0;
            t.act = d->act;
            hipStream_t s = (hipStream_t)stream;
            if (tk == 1) {
                t.Cw = d->Cout; t.xb = 64;
                const int items = d->N * d->Hin * cdiv(d->Win, t.xb);
                // persistent workgroups: the 27 x 4 weights of a thread are loaded once per workgroup, not once per item
                const int blocks = items < 4096 ? items : 4096;
                FAVAE_KLAUNCH((thin_in_kernel<3, false>), dim3(blocks), dim3(256), (size_t)3 * (t.xb + 2) * 3 * sizeof(float), s, t);
            } else {
                t.Cw = d->Cin;
                const int PL = 256 / (t.Cw / 4);
                int seg = cdiv(d->Win, PL);
                seg = seg < 4 ? 4 : (seg > 32 ? 32 : seg);
                t.xb = seg * PL;
                const int items = d->N * d->Hin * cdiv(d->Win, t.xb);
                if (xf0 == 0) FAVAE_KLAUNCH((thin_out_kernel<3, 0, false>), dim3(items), dim3(256), 0, s, t);
                else if (xf0 == 1) FAVAE_KLAUNCH((thin_out_kernel<3, 1, false>), dim3(items), dim3(256), 0, s, t);
                else if (xf0 == 2) FAVAE_KLAUNCH((thin_out_kernel<3, 2, false>), dim3(items), dim3(256), 0, s, t);
                else FAVAE_KLAUNCH((thin_out_kernel<3, 3, false>), dim3(items), dim3(256), 0, s, t);
            }
            FAVAE_CHECK_LAUNCH();
            return FAVAE_OK;
        }
    }
    ConvArgs a;
    a.gb_x = nullptr; a.gb_mean = a.gb_rstd = a.gb_gamma = a.gb_beta = nullptr; a.gb_part = nullptr; a.gb_groups = 1; a.gb_act = 0;
    a.gs_part = stats_part;
    a.gs_amax = (unsigned*)stats_amax;
    if (gb) {
        a.gb_x = gb->x; a.gb_mean = gb->mean; a.gb_rstd = gb->rstd; a.gb_gamma = gb->gamma; a.gb_beta = gb->beta;
        a.gb_part = gb->part; a.gb_groups = gb->groups; a.gb_act = gb->act;
    }
    a.x_amax = x_amax; a.w_amax = w;
    a.x = x; a.w = w6 ? (const float*)((const char*)w + sp::WHDR + d->w_rec_offset) : w;
    const bool special = desc_special(d);
    const int st = d->lat_step == 2 ? 2 : 1;
    a.pad_w = d->pad + d->pad_dw;
    a.in_step = 1; a.in_row = d->Win; a.in_img = d->Hin * d->Win; a.in_off = 0;
    a.out_step = 1; a.out_row = d->Wout; a.out_img = d->Hout * d->Wout; a.out_off = 0;
    if (st == 2 && d->lat_side == 2) {
        a.in_step = 2; a.in_row = 2 * d->Win; a.in_img = 4 * d->Hin * d->Win; a.in_off = d->lat_oh * a.in_row + d->lat_ow;
    } else if (st == 2) {
        a.out_step = 2; a.out_row = 2 * d->Wout; a.out_img = 4 * d->Hout * d->Wout; a.out_off = d->lat_oh * a.out_row + d->lat_ow;
    } a.bias = bias; a.resid = resid; a.scale = scale; a.shift = shift; a.y = y;
    a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.Cin = d->Cin; a.Hout = d->Hout; a.Wout = d->Wout; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.gather = d->gather; a.act = d->act;
    a.aff_stride = d->affine_per_image ? d->Cin : 0;
    a.M = d->N * d->Hout * d->Wout;
    a.kchunks = cdiv(d->Cin, BK);
    a.vec = (d->Cin % 4 == 0) ? 1 : 0;
    a.tiles_m = cdiv(a.M, BM);
    hipStream_t s = (hipStream_t)stream;
    const int bn = d->Cout > 64 ? 128 : (d->Cout > 32 ? 64 : 32);
    a.tiles_n = cdiv(d->Cout, bn);
    const dim3 grid(a.tiles_m * a.tiles_n), blk(256);
#define FAVAE_LAUNCH_FWD(G)                                                                                        \
    do {                                                                                                           \
        if (bn == 128) FAVAE_KLAUNCH((conv_fwd_fast_kernel<128, 2, 2, G>), grid, blk, 0, s, a);               \
        else if (bn == 64) FAVAE_KLAUNCH((conv_fwd_fast_kernel<64, 2, 2, G>), grid, blk, 0, s, a);            \
        else FAVAE_KLAUNCH((conv_fwd_fast_kernel<32, 4, 1, G>), grid, blk, 0, s, a);                          \
    } while (0)
    const size_t xb = (size_t)d->N * a.in_img * d->Cin * 4, wb = (size_t)d->Cout * d->KH * d->KW * d->Cin * 4;
    const size_t ab = (size_t)(d->affine_per_image ? d->N : 1) * d->Cin * 4;
    const int xf = scale ? (d->act == FAVAE_ACT_SILU ? 2 : (d->act == FAVAE_ACT_NONE ? 1 : 3)) : 0;
    const bool buf_ok = !force_generic() && !force_nobuf() && d->Cin % 16 == 0 && xb < (1u << 31) && wb < (1u << 31) &&
                        (size_t)d->N * a.out_img * d->Cout * 4 < ((size_t)1 << 32) && (d->gather == FAVAE_GATHER_PLAIN || xf == 0);
    if (special && !(buf_ok && use_b6() && bn == 128 && w6 && d->gather == FAVAE_GATHER_PLAIN)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    a.x_bytes = (unsigned)(bf16io ? xb / 2 : xb); a.aff_bytes = (unsigned)ab;
    a.w_bytes = (unsigned)(wplanes ? wb / 16 * wrec_bytes(wplanes) : wb);
#define FAVAE_LAUNCH_BUF(G, X)                                                                                     \
    do {                                                                                                           \
        if (bn == 128) FAVAE_KLAUNCH((conv_fwd_buf_kernel<128, 2, 2, G, X>), grid, blk, 0, s, a);             \
        else if (bn == 64) FAVAE_KLAUNCH((conv_fwd_buf_kernel<64, 2, 2, G, X>), grid, blk, 0, s, a);          \
        else FAVAE_KLAUNCH((conv_fwd_buf_kernel<32, 4, 1, G, X>), grid, blk, 0, s, a);                        \
    } while (0)
    const bool halo_common = buf_ok && use_b6() && w6 && use_halo() && bn == 128 && d->stride == 1 && d->gather == FAVAE_GATHER_PLAIN &&
                             d->Hout == d->Hin && d->Wout == d->Win && d->Hin % 8 == 0 && d->Win % 16 == 0;
    const bool halo_ok = !special && halo_common && d->KH == 3 && d->KW == 3 && d->pad == 1;
    // 2x2 phase convs (Upsample forward / data gradient, first phase of the Downsample data gradient): one side on a sub-grid
    const bool halo2_ok = special && halo_common && use_halo2() && d->KH == 2 && d->KW == 2 && d->lat_step == 2 && xf == 0 &&
                          (d->pad == 0 || d->pad == 1) && (a.pad_w == 0 || a.pad_w == 1);
    // pre-split records are only understood by the 128-channel split tiles and by the Winograd kernel: never let them reach a kernel
    // that would read them as fp32 weights
    if (w6 && !wino && bn != 128) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (wino) {
        if (!(!special && buf_ok && use_b6() && w6 && wino_geometry(d) && wplanes == conv_mode() && d->w_rec_offset == 0))
            return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
        if (gb && (xf != 0 || bias || resid)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
        if (stats_part && !(xf == 0 || xf == 2)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
        if (wino4) {
            if (!wino4_ok(d, scale != nullptr)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
            a.tiles_n = d->Cout / 64;
            a.w_bytes = (unsigned)((size_t)d->Cout * d->Cin * 144);
            auto rcp32 = [](int dv) { return dv == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dv + 1); };
            a.wino_rcp_n = rcp32(a.tiles_n); a.wino_rcp_w = rcp32(d->Win / 32); a.wino_rcp_h = rcp32(d->Hin / 16);
            const dim3 wgrid((unsigned)(d->N * (d->Hin / 16) * (d->Win / 32) * a.tiles_n));
#define FAVAE_LAUNCH_WINO4(X, GBV, SEV)                                                                                     \
    do {                                                                                                                    \
        static bool attr_set = false;                                                                                       \
        if (!attr_set) {                                                                                                    \
            (void)hipFuncSetAttribute((const void*)conv3x3_wino4_sp_kernel<X, GBV, SEV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      wino4::LDS_B);                                                                        \
            attr_set = true;                                                                                                \
        }                                                                                                                   \
        FAVAE_KLAUNCH((conv3x3_wino4_sp_kernel<X, GBV, SEV>), wgrid, dim3(512), wino4::LDS_B, s, a);                        \
    } while (0)
            if (gb) FAVAE_LAUNCH_WINO4(0, true, false);
            else if (stats_part && xf == 0) FAVAE_LAUNCH_WINO4(0, false, true);
            else if (stats_part) FAVAE_LAUNCH_WINO4(2, false, true);
            else if (xf == 0) FAVAE_LAUNCH_WINO4(0, false, false);
            else if (xf == 2) FAVAE_LAUNCH_WINO4(2, false, false);
            else return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
#undef FAVAE_LAUNCH_WINO4
            FAVAE_CHECK_LAUNCH();
            return FAVAE_OK;
        }
        const bool wide = wino_wide_ok(d, scale != nullptr);      // 16 x 8 pixels x 128 channels per workgroup
        if (wplanes == 4 && !wide) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
        const int wth = wide ? 8 : 16;
        a.tiles_n = d->Cout / (wide ? 128 : 64);
        a.w_bytes = (unsigned)((size_t)d->Cout * d->Cin * 64);
        auto rcp32 = [](int dv) { return dv == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dv + 1); };   // 0: divisor 1
        a.wino_rcp_n = rcp32(a.tiles_n); a.wino_rcp_w = rcp32(d->Win / 16); a.wino_rcp_h = rcp32(d->Hin / wth);
        const dim3 wgrid((unsigned)(d->N * (d->Hin / wth) * (d->Win / 16) * a.tiles_n));
#define FAVAE_LAUNCH_WINO_T(X, GBV, SEV, WD, PL)                                                                       \
    do {                                                                                                                    \
        static bool attr_set = false;                                                                                       \
        if (!attr_set) {                                                                                                    \
            (void)hipFuncSetAttribute((const void*)conv3x3_wino_sp_kernel<X, GBV, SEV, WD, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      wino::LDS_B);                                                                         \
            attr_set = true;                                                                                                \
        }                                                                                                                   \
        FAVAE_KLAUNCH((conv3x3_wino_sp_kernel<X, GBV, SEV, WD, PL>), wgrid, dim3(512), wino::LDS_B, s, a);              \
    } while (0)
#define FAVAE_LAUNCH_WINO_BF(X, GBV, SEV)                                                                                   \
    do {                                                                                                                    \
        static bool attr_set = false;                                                                                       \
        if (!attr_set) {                                                                                                    \
            (void)hipFuncSetAttribute((const void*)conv3x3_wino_sp_kernel<X, GBV, SEV, true, 4, bf16_t>,                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, wino::LDS_B);                             \
            attr_set = true;                                                                                                \
        }                                                                                                                   \
        FAVAE_KLAUNCH((conv3x3_wino_sp_kernel<X, GBV, SEV, true, 4, bf16_t>), wgrid, dim3(512), wino::LDS_B, s, a);         \
    } while (0)
        if (bf16io) {                       // bf16 activation storage: the wide tiling with the bf16 plane
            if (!(wplanes == 4 && wide)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
            if (gb) FAVAE_LAUNCH_WINO_BF(0, true, false);
            else if (stats_part && xf == 0) FAVAE_LAUNCH_WINO_BF(0, false, true);
            else if (stats_part) FAVAE_LAUNCH_WINO_BF(2, false, true);
            else if (xf == 0) FAVAE_LAUNCH_WINO_BF(0, false, false);
            else if (xf == 2) FAVAE_LAUNCH_WINO_BF(2, false, false);
            else return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
            FAVAE_CHECK_LAUNCH();
            return FAVAE_OK;
        }
#undef FAVAE_LAUNCH_WINO_BF
#define FAVAE_LAUNCH_WINO(X, GBV, SEV)                                                                                      \
    do {                                                                                                                    \
        if (wplanes == 1 && wide) FAVAE_LAUNCH_WINO_T(X, GBV, SEV, true, 1);     /* one fp16 plane (h1) */            \
        else if (wplanes == 1) FAVAE_LAUNCH_WINO_T(X, GBV, SEV, false, 1);                                           \
        else if (wplanes == 4) FAVAE_LAUNCH_WINO_T(X, GBV, SEV, true, 4);   /* one bf16 plane (b1): wide tiling only */ \
        else if (wide) FAVAE_LAUNCH_WINO_T(X, GBV, SEV, true, 2);                                                   \
        else FAVAE_LAUNCH_WINO_T(X, GBV, SEV, false, 2);                                                             \
    } while (0)
        if (gb) FAVAE_LAUNCH_WINO(0, true, false);
        else if (stats_part && xf == 0) FAVAE_LAUNCH_WINO(0, false, true);
        else if (stats_part) FAVAE_LAUNCH_WINO(2, false, true);
        else if (xf == 0) FAVAE_LAUNCH_WINO(0, false, false);
        else if (xf == 1) FAVAE_LAUNCH_WINO(1, false, false);
        else if (xf == 2) FAVAE_LAUNCH_WINO(2, false, false);
        else FAVAE_LAUNCH_WINO(3, false, false);
#undef FAVAE_LAUNCH_WINO
#undef FAVAE_LAUNCH_WINO_T
    } else if (halo_ok || halo2_ok) {
        a.tiles_n = cdiv(d->Cout, 128);
        const dim3 hgrid((unsigned)(d->N * (d->Hin / 8) * (d->Win / 16) * a.tiles_n));
#define FAVAE_LAUNCH_HALO_K(X, KS)                                                                            \
    do {                                                                                                      \
        if (wplanes == 2) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<X, 2, KS>), hgrid, dim3(512), 0, s, a);  \
        else if (wplanes == 1) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<X, 1, KS>), hgrid, dim3(512), 0, s, a); \
        else if (wplanes == 4) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<X, 4, KS>), hgrid, dim3(512), 0, s, a); \
        else FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<X, 3, KS>), hgrid, dim3(512), 0, s, a);               \
    } while (0)
#define FAVAE_LAUNCH_HALO(X) FAVAE_LAUNCH_HALO_K(X, 3)
        if (bf16io) {                       // bf16 activation storage: the dense 3x3 kernel with the bf16 plane
            if (!(halo_ok && wplanes == 4) || (gb && (xf != 0 || bias || resid)) || (stats_part && (gb || !(xf == 0 || xf == 2))))
                return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
#define FAVAE_LAUNCH_HALO_BF(X, GBV, SEV) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<X, 4, 3, GBV, SEV, bf16_t>), hgrid, dim3(512), 0, s, a)
            if (gb) FAVAE_LAUNCH_HALO_BF(0, true, false);
            else if (stats_part && xf == 0) FAVAE_LAUNCH_HALO_BF(0, false, true);
            else if (stats_part) FAVAE_LAUNCH_HALO_BF(2, false, true);
            else if (xf == 0) FAVAE_LAUNCH_HALO_BF(0, false, false);
            else if (xf == 1) FAVAE_LAUNCH_HALO_BF(1, false, false);
            else if (xf == 2) FAVAE_LAUNCH_HALO_BF(2, false, false);
            else FAVAE_LAUNCH_HALO_BF(3, false, false);
#undef FAVAE_LAUNCH_HALO_BF
            FAVAE_CHECK_LAUNCH();
            return FAVAE_OK;
        }
        const bool fp16p = wplanes == 2 || wplanes == 1 || wplanes == 4;           // schemes with the epilogue variants
        if (gb && !(halo_ok && fp16p && xf == 0 && !bias && !resid)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
        if (stats_part && !(halo_ok && fp16p && (xf == 0 || xf == 2) && !gb)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
        if (gb && wplanes == 2) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<0, 2, 3, true>), hgrid, dim3(512), 0, s, a);
        else if (gb && wplanes == 4) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<0, 4, 3, true>), hgrid, dim3(512), 0, s, a);
        else if (gb) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<0, 1, 3, true>), hgrid, dim3(512), 0, s, a);
        else if (stats_part && xf == 0 && wplanes == 2) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<0, 2, 3, false, true>), hgrid, dim3(512), 0, s, a);
        else if (stats_part && wplanes == 2) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<2, 2, 3, false, true>), hgrid, dim3(512), 0, s, a);
        else if (stats_part && xf == 0 && wplanes == 4) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<0, 4, 3, false, true>), hgrid, dim3(512), 0, s, a);
        else if (stats_part && wplanes == 4) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<2, 4, 3, false, true>), hgrid, dim3(512), 0, s, a);
        else if (stats_part && xf == 0) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<0, 1, 3, false, true>), hgrid, dim3(512), 0, s, a);
        else if (stats_part) FAVAE_KLAUNCH((conv3x3_halo_sp_kernel<2, 1, 3, false, true>), hgrid, dim3(512), 0, s, a);
        else if (halo2_ok) FAVAE_LAUNCH_HALO_K(0, 2);
        else if (xf == 0) FAVAE_LAUNCH_HALO(0);
        else if (xf == 1) FAVAE_LAUNCH_HALO(1);
        else if (xf == 2) FAVAE_LAUNCH_HALO(2);
        else FAVAE_LAUNCH_HALO(3);
#undef FAVAE_LAUNCH_HALO
#undef FAVAE_LAUNCH_HALO_K
    } else if (bf16io) {
        return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    } else if (buf_ok && use_b6() && bn == 128) {
#define FAVAE_LAUNCH_B6(G, X)                                                                     \
    do {                                                                                          \
        if (wplanes == 2) FAVAE_KLAUNCH((conv_fwd_sp_kernel<G, X, true, 8, 2>), grid, dim3(512), 0, s, a);        \
        else if (wplanes == 1) FAVAE_KLAUNCH((conv_fwd_sp_kernel<G, X, true, 8, 1>), grid, dim3(512), 0, s, a);   \
        else if (wplanes == 4) FAVAE_KLAUNCH((conv_fwd_sp_kernel<G, X, true, 8, 4>), grid, dim3(512), 0, s, a);   \
        else if (b6_waves() == 8) {                                                               \
            if (w6) FAVAE_KLAUNCH((conv_fwd_sp_kernel<G, X, true, 8, 3>), grid, dim3(512), 0, s, a);   \
            else FAVAE_KLAUNCH((conv_fwd_sp_kernel<G, X, false, 8, 3>), grid, dim3(512), 0, s, a);     \
        } else {                                                                                  \
            if (w6) FAVAE_KLAUNCH((conv_fwd_sp_kernel<G, X, true, 4, 3>), grid, blk, 0, s, a);  \
            else FAVAE_KLAUNCH((conv_fwd_sp_kernel<G, X, false, 4, 3>), grid, blk, 0, s, a);    \
        }                                                                                         \
    } while (0)
        if (d->gather == FAVAE_GATHER_UPSAMPLE2) FAVAE_LAUNCH_B6(FAVAE_GATHER_UPSAMPLE2, 0);
        else if (d->gather == FAVAE_GATHER_DILATE2) FAVAE_LAUNCH_B6(FAVAE_GATHER_DILATE2, 0);
        else if (xf == 0) FAVAE_LAUNCH_B6(FAVAE_GATHER_PLAIN, 0);
        else if (xf == 1) FAVAE_LAUNCH_B6(FAVAE_GATHER_PLAIN, 1);
        else if (xf == 2) FAVAE_LAUNCH_B6(FAVAE_GATHER_PLAIN, 2);
        else FAVAE_LAUNCH_B6(FAVAE_GATHER_PLAIN, 3);
#undef FAVAE_LAUNCH_B6
    } else if (buf_ok) {
        if (d->gather == FAVAE_GATHER_UPSAMPLE2) FAVAE_LAUNCH_BUF(FAVAE_GATHER_UPSAMPLE2, 0);
        else if (d->gather == FAVAE_GATHER_DILATE2) FAVAE_LAUNCH_BUF(FAVAE_GATHER_DILATE2, 0);
        else if (xf == 0) FAVAE_LAUNCH_BUF(FAVAE_GATHER_PLAIN, 0);
        else if (xf == 1) FAVAE_LAUNCH_BUF(FAVAE_GATHER_PLAIN, 1);
        else if (xf == 2) FAVAE_LAUNCH_BUF(FAVAE_GATHER_PLAIN, 2);
        else FAVAE_LAUNCH_BUF(FAVAE_GATHER_PLAIN, 3);
    } else if (a.vec && !force_generic()) {
        if (d->gather == FAVAE_GATHER_PLAIN) FAVAE_LAUNCH_FWD(FAVAE_GATHER_PLAIN);
        else if (d->gather == FAVAE_GATHER_UPSAMPLE2) FAVAE_LAUNCH_FWD(FAVAE_GATHER_UPSAMPLE2);
        else FAVAE_LAUNCH_FWD(FAVAE_GATHER_DILATE2);
    } else if (bn == 128) {
        FAVAE_KLAUNCH((conv_fwd_kernel<128, 2, 2>), grid, blk, 0, s, a);
    } else if (bn == 64) {
        FAVAE_KLAUNCH((conv_fwd_kernel<64, 2, 2>), grid, blk, 0, s, a);
    } else {
        FAVAE_KLAUNCH((conv_fwd_kernel<32, 4, 1>), grid, blk, 0, s, a);
    }
#undef FAVAE_LAUNCH_FWD
#undef FAVAE_LAUNCH_BUF
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" size_t favae_conv_wgrad_workspace(const favae_conv_desc* d) {
    if (!desc_ok(d)) return 0;
    int bco, bci, chunk;
    wgrad_tiles(d, &bco, &bci);
    const int tiles = cdiv(d->Cout, bco) * cdiv(d->Cin, bci) * d->KH * d->KW;
    int sk = wgrad_splitk(d, tiles, &chunk);
    if (thin_kind(d, false) && sk < THIN_WGRAD_BLOCKS) sk = THIN_WGRAD_BLOCKS;     // either kernel family may run: size for both
    return (size_t)sk * d->Cout * d->KH * d->KW * d->Cin * sizeof(float);
}

static int conv_wgrad_impl(const favae_conv_desc* d, const float* x, const float* dy, const float* scale, const float* shift,
                           const float* x_absmax, const float* dy_absmax, float* dw,
                           int accumulate, void* ws, size_t ws_bytes, favae_stream_t stream, int* slabs_out = nullptr);

// Weight gradient WITHOUT its slab reduction: the partial slabs stay in `ws` ([*slabs][Cout][KH][KW][Cin] floats, *slabs written on
// the host) for a later favae_reduce_slabs_grouped -- one launch for the slabs of many layers instead of one latency-bound launch
// per layer (214 per step).  The sums are taken in the same order either way: bit-identical gradients.
extern "C" int favae_conv_wgrad_slabs(const favae_conv_desc* d, const float* x, const float* dy, const float* scale,
                                      const float* shift, const float* x_absmax, const float* dy_absmax, void* ws, size_t ws_bytes,
                                      int* slabs, favae_stream_t stream) {
    FAVAE_REQUIRE(slabs);
    return conv_wgrad_impl(d, x, dy, scale, shift, x_absmax, dy_absmax, (float*)ws, 0, ws, ws_bytes, stream, slabs);
}

extern "C" int favae_conv_wgrad(const favae_conv_desc* d, const float* x, const float* dy, const float* scale,
                                const float* shift, const float* x_absmax, const float* dy_absmax, float* dw, int accumulate,
                                void* ws, size_t ws_bytes, favae_stream_t stream) {
    return conv_wgrad_impl(d, x, dy, scale, shift, x_absmax, dy_absmax, dw, accumulate, ws, ws_bytes, stream);
}

static bool wgrad_nine_geom_ok(const favae_conv_desc* d) {     // mirrors the `dense3` condition of conv_wgrad_impl
    if (thin_kind(d, false) || thin_kind(d, true) || desc_special(d) || force_generic() || force_nobuf() || !use_b6())
        return false;
    if (d->act != FAVAE_ACT_NONE && d->act != FAVAE_ACT_SILU) return false;
    int bco, bci;
    wgrad_tiles(d, &bco, &bci);
    const size_t xb = (size_t)d->N * d->Hin * d->Win * d->Cin * 4, yb = (size_t)d->N * d->Hout * d->Wout * d->Cout * 4;
    return bco == 128 && bci == 128 && d->Cin % 4 == 0 && d->Cout % 4 == 0 && d->gather == FAVAE_GATHER_PLAIN && d->stride == 1 &&
           d->Wout % 16 == 0 && xb < (1u << 31) && yb < (1u << 31) && d->KH == 3 && d->KW == 3 && d->pad == 1;
}

static int conv_wgrad_impl(const favae_conv_desc* d_in, const float* x, const float* dy, const float* scale, const float* shift,
                           const float* x_absmax, const float* dy_absmax, float* dw,
                           int accumulate, void* ws, size_t ws_bytes, favae_stream_t stream, int* slabs_out) {
    // bf16 activation storage (round 6): FAVAE_ACT_BF16IO on the descriptor's `act` = x and dy are bf16 tensors; only the nine-tap
    // kernel of scheme 4 has that instantiation (favae_conv_bf16io_ok(d, has_affine, 2))
    FAVAE_REQUIRE(d_in);
    favae_conv_desc d_local = *d_in;
    const bool bf16io = (d_local.act & FAVAE_ACT_BF16IO) != 0;
    d_local.act &= ~FAVAE_ACT_BF16IO;
    const favae_conv_desc* d = &d_local;
    FAVAE_REQUIRE(desc_ok(d) && x && dy && dw && ws);
    // fp16 planes need both operand maxima; without them the bf16 scheme (no range restrictions) runs
    const int cm = conv_mode();
    const int np = cm == 4 ? 4 : (((cm == 2 || cm == 1) && x_absmax && dy_absmax) ? cm : 3);
    FAVAE_REQUIRE((scale == nullptr) == (shift == nullptr));
    if (ws_bytes < favae_conv_wgrad_workspace(d)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    // roofline numerators: same FLOPs as the forward conv; one read of x and of dy, one write of dw
    FAVAE_PROF_NOTE(2.0 * d->N * d->Hout * d->Wout * d->Cout * d->KH * d->KW * d->Cin,
                    4.0 * ((double)d->N * d->Hin * d->Win * d->Cin + (double)d->N * d->Hout * d->Wout * d->Cout +
                           (double)d->Cout * d->KH * d->KW * d->Cin));
    {
        const int tk = thin_kind(d, scale != nullptr);
        const int xf0 = scale ? (d->act == FAVAE_ACT_SILU ? 2 : (d->act == FAVAE_ACT_NONE ? 1 : 3)) : 0;
        if (tk && bf16io) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
        if (tk && xf0 != 3 && al16(x) && al16(dy) && al16(scale) && al16(shift) && al16(ws)) {
            ThinArgs t{};
            t.x = x; t.dy = dy; t.scale = scale; t.shift = shift; t.part = (float*)ws;
            t.N = d->N; t.H = d->Hin; t.W = d->Win;
            t.aff_stride = d->affine_per_image ? d->Cin : 0;
            t.act = d->act;
            hipStream_t s = (hipStream_t)stream;
            int blocks;
            if (tk == 1) {
                t.Cw = d->Cout; t.xb = 64;
                const int items = d->N * d->Hin * cdiv(d->Win, t.xb);
                blocks = items < THIN_WGRAD_BLOCKS ? items : THIN_WGRAD_BLOCKS;
                size_t shm = (size_t)3 * 27 * t.Cw * sizeof(float);
                const size_t stage_b = (size_t)3 * (t.xb + 2) * 3 * sizeof(float);
                if (shm < stage_b) shm = stage_b;
                FAVAE_KLAUNCH((thin_in_kernel<3, true>), dim3(blocks), dim3(256), shm, s, t);
            } else {
                t.Cw = d->Cin;
                const int PL = 256 / (t.Cw / 4);
                int seg = cdiv(d->Win, PL);
                seg = seg < 4 ? 4 : (seg > 32 ? 32 : seg);
                t.xb = seg * PL;
                const int items = d->N * d->Hin * cdiv(d->Win, t.xb);
                blocks = items < THIN_WGRAD_BLOCKS ? items : THIN_WGRAD_BLOCKS;
                const size_t shm = (size_t)3 * 27 * t.Cw * sizeof(float);
                if (xf0 == 0) FAVAE_KLAUNCH((thin_out_kernel<3, 0, true>), dim3(blocks), dim3(256), shm, s, t);
                else if (xf0 == 1) FAVAE_KLAUNCH((thin_out_kernel<3, 1, true>), dim3(blocks), dim3(256), shm, s, t);
                else FAVAE_KLAUNCH((thin_out_kernel<3, 2, true>), dim3(blocks), dim3(256), shm, s, t);
            }
            FAVAE_CHECK_LAUNCH();
            if (slabs_out) { *slabs_out = blocks; return FAVAE_OK; }
            const size_t nw = (size_t)d->Cout * 9 * d->Cin;
            FAVAE_KLAUNCH(reduce_slabs_kernel, dim3(cdiv(nw, 256)), dim3(256), 0, s, (const float*)ws, dw, nw, blocks, accumulate);
            FAVAE_CHECK_LAUNCH();
            return FAVAE_OK;
        }
    }
    int bco, bci, chunk;
    wgrad_tiles(d, &bco, &bci);
    WgradArgs a;
    a.x = x; a.dy = dy; a.scale = scale; a.shift = shift; a.part = (float*)ws;
    a.x_amax = x_absmax; a.dy_amax = dy_absmax;
    const bool special = desc_special(d);
    a.pad_w = d->pad + d->pad_dw;
    a.dy_step = 1; a.dy_row = d->Wout; a.dy_img = d->Hout * d->Wout; a.dy_off = 0;
    if (d->lat_step == 2) {
        if (d->lat_side != 1) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);          // weight gradients: only the conv OUTPUT may be a sub-grid
        a.dy_step = 2; a.dy_row = 2 * d->Wout; a.dy_img = 4 * d->Hout * d->Wout; a.dy_off = d->lat_oh * a.dy_row + d->lat_ow;
    }
    a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.Cin = d->Cin; a.Hout = d->Hout; a.Wout = d->Wout; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.gather = d->gather; a.act = d->act;
    a.aff_stride = d->affine_per_image ? d->Cin : 0;
    a.M = d->N * d->Hout * d->Wout;
    a.tiles_co = cdiv(d->Cout, bco);
    a.tiles_ci = cdiv(d->Cin, bci);
    const int tiles = a.tiles_co * a.tiles_ci * d->KH * d->KW;
    a.splitk = wgrad_splitk(d, tiles, &chunk);
    a.chunk = chunk;
    a.vec_i = (d->Cin % 4 == 0);
    a.vec_o = (d->Cout % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(tiles, a.splitk);
#define FAVAE_LAUNCH_WGRAD(G)                                                                                      \
    do {                                                                                                           \
        if (bco == 128 && bci == 128) FAVAE_KLAUNCH((conv_wgrad_fast_kernel<128, 128, 2, 2, G>), grid, dim3(256), 0, s, a); \
        else if (bco == 32) FAVAE_KLAUNCH((conv_wgrad_fast_kernel<32, 128, 1, 4, G>), grid, dim3(256), 0, s, a);            \
        else FAVAE_KLAUNCH((conv_wgrad_fast_kernel<128, 32, 4, 1, G>), grid, dim3(256), 0, s, a);                           \
    } while (0)
    const size_t xb = (size_t)d->N * d->Hin * d->Win * d->Cin * 4, yb = (size_t)d->N * d->Hout * d->Wout * d->Cout * 4;
    const size_t ab = (size_t)(d->affine_per_image ? d->N : 1) * d->Cin * 4;
    const int xf = scale ? (d->act == FAVAE_ACT_SILU ? 2 : (d->act == FAVAE_ACT_NONE ? 1 : 3)) : 0;
    const bool ups_b6 = d->gather == FAVAE_GATHER_UPSAMPLE2 && xf == 0 && use_b6() && bco == 128 && bci == 128;
    // stride 2 (Downsample): only the per-tap split-precision kernel implements it
    const bool s2_sp = d->stride == 2 && use_b6() && bco == 128 && bci == 128 && d->gather == FAVAE_GATHER_PLAIN && !special;
    const bool buf_ok = !force_generic() && !force_nobuf() && a.vec_i && a.vec_o && (d->gather == FAVAE_GATHER_PLAIN || ups_b6) &&
                        (d->stride == 1 || s2_sp) && d->Wout % 16 == 0 && xb < (1u << 31) && yb < (1u << 31) && xf != 3;
    a.x_bytes = (unsigned)xb; a.aff_bytes = (unsigned)ab;
    if (special && !(buf_ok && use_b6() && bco == 128 && bci == 128 && d->gather == FAVAE_GATHER_PLAIN &&
                     (size_t)d->N * a.dy_img * d->Cout * 4 < ((size_t)1 << 32)))
        return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
#define FAVAE_LAUNCH_WBUF(X)                                                                                       \
    do {                                                                                                           \
        if (bco == 128 && bci == 128) FAVAE_KLAUNCH((conv_wgrad_buf_kernel<128, 128, 2, 2, X>), grid, dim3(256), 0, s, a); \
        else if (bco == 32) FAVAE_KLAUNCH((conv_wgrad_buf_kernel<32, 128, 1, 4, X>), grid, dim3(256), 0, s, a);            \
        else FAVAE_KLAUNCH((conv_wgrad_buf_kernel<128, 32, 4, 1, X>), grid, dim3(256), 0, s, a);                           \
    } while (0)
    const bool dense3 = !special && buf_ok && use_b6() && bco == 128 && bci == 128 && d->gather == FAVAE_GATHER_PLAIN &&
                      d->KH == 3 && d->KW == 3 && d->pad == 1 && d->stride == 1;
    const bool nine = dense3 && use_nine() && d->Hout == d->Hin && d->Wout == d->Win;
    if (bf16io && !(nine && np == 4 && nine_mode() == 1 && xf != 3)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    if (bf16io) a.x_bytes = (unsigned)(xb / 2);
    if (nine && bf16io) {
        a.tiles_co = cdiv(d->Cout, 128);
        a.tiles_ci = cdiv(d->Cin, 64);
        const int tiles9 = a.tiles_co * a.tiles_ci;
        int sps;
        a.splitk = nine_splitk(d, tiles9, a.splitk, 256, &sps);
        a.chunk = sps;
        const dim3 g9((unsigned)(tiles9 * ((a.splitk + 7) / 8) * 8));
        if (xf == 0) FAVAE_KLAUNCH((conv_wgrad_nine_sp_kernel<0, 4, 128, false, 1, bf16_t>), g9, dim3(512), 0, s, a);
        else if (xf == 1) FAVAE_KLAUNCH((conv_wgrad_nine_sp_kernel<1, 4, 128, false, 1, bf16_t>), g9, dim3(512), 0, s, a);
        else FAVAE_KLAUNCH((conv_wgrad_nine_sp_kernel<2, 4, 128, false, 1, bf16_t>), g9, dim3(512), 0, s, a);
    } else if (nine) {
        // all nine taps per workgroup (BCO co x 64 ci), split-K over whole 16-pixel column strips
        const int nm = nine_mode();
        const int bco9 = nm == 3 ? 64 : 128;
        a.tiles_co = cdiv(d->Cout, bco9);
        a.tiles_ci = cdiv(d->Cin, 64);
        const int tiles9 = a.tiles_co * a.tiles_ci;
        int sps;
        a.splitk = nine_splitk(d, tiles9, a.splitk, 256, &sps);
        a.chunk = sps;
        const dim3 g9((unsigned)(tiles9 * ((a.splitk + 7) / 8) * 8));
#define FAVAE_LAUNCH_NINE_M(X, P)                                                                                   \
    do {                                                                                                            \
        if (nm == 1) FAVAE_KLAUNCH((conv_wgrad_nine_sp_kernel<X, P, 128, false, 1>), g9, dim3(512), 0, s, a);        \
        else if (nm == 2) FAVAE_KLAUNCH((conv_wgrad_nine_sp_kernel<X, P, 128, false, 2>), g9, dim3(512), 0, s, a);   \
        else FAVAE_KLAUNCH((conv_wgrad_nine_sp_kernel<X, P, 64, true, 2>), g9, dim3(256), 0, s, a);             \
    } while (0)
#define FAVAE_LAUNCH_NINE(X)                                  \
    do {                                                      \
        if (np == 2) FAVAE_LAUNCH_NINE_M(X, 2);               \
        else if (np == 1) FAVAE_LAUNCH_NINE_M(X, 1);          \
        else if (np == 4) FAVAE_LAUNCH_NINE_M(X, 4);          \
        else FAVAE_LAUNCH_NINE_M(X, 3);                       \
    } while (0)
        if (xf == 0) FAVAE_LAUNCH_NINE(0);
        else if (xf == 1) FAVAE_LAUNCH_NINE(1);
        else FAVAE_LAUNCH_NINE(2);
#undef FAVAE_LAUNCH_NINE
#undef FAVAE_LAUNCH_NINE_M
    } else if (buf_ok && use_b6() && bco == 128 && bci == 128) {
#define FAVAE_LAUNCH_WSP(X, U)                                                                            \
    do {                                                                                                  \
        if (np == 2) FAVAE_KLAUNCH((conv_wgrad_sp_kernel<X, U, 2>), grid, dim3(256), 0, s, a);       \
        else if (np == 1) FAVAE_KLAUNCH((conv_wgrad_sp_kernel<X, U, 1>), grid, dim3(256), 0, s, a);  \
        else if (np == 4) FAVAE_KLAUNCH((conv_wgrad_sp_kernel<X, U, 4>), grid, dim3(256), 0, s, a);  \
        else FAVAE_KLAUNCH((conv_wgrad_sp_kernel<X, U, 3>), grid, dim3(256), 0, s, a);               \
    } while (0)
        if (d->gather == FAVAE_GATHER_UPSAMPLE2) FAVAE_LAUNCH_WSP(0, true);
        else if (xf == 0) FAVAE_LAUNCH_WSP(0, false);
        else if (xf == 1) FAVAE_LAUNCH_WSP(1, false);
        else FAVAE_LAUNCH_WSP(2, false);
#undef FAVAE_LAUNCH_WSP
    } else if (buf_ok && d->gather == FAVAE_GATHER_PLAIN) {
        if (xf == 0) FAVAE_LAUNCH_WBUF(0);
        else if (xf == 1) FAVAE_LAUNCH_WBUF(1);
        else FAVAE_LAUNCH_WBUF(2);
    } else if (a.vec_i && a.vec_o && !force_generic()) {
        if (d->gather == FAVAE_GATHER_PLAIN) FAVAE_LAUNCH_WGRAD(FAVAE_GATHER_PLAIN);
        else if (d->gather == FAVAE_GATHER_UPSAMPLE2) FAVAE_LAUNCH_WGRAD(FAVAE_GATHER_UPSAMPLE2);
        else FAVAE_LAUNCH_WGRAD(FAVAE_GATHER_DILATE2);
    } else if (bco == 128 && bci == 128)
        FAVAE_KLAUNCH((conv_wgrad_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, a);
    else if (bco == 32)
        FAVAE_KLAUNCH((conv_wgrad_kernel<32, 128, 1, 4>), grid, dim3(256), 0, s, a);
    else
        FAVAE_KLAUNCH((conv_wgrad_kernel<128, 32, 4, 1>), grid, dim3(256), 0, s, a);
#undef FAVAE_LAUNCH_WGRAD
#undef FAVAE_LAUNCH_WBUF
    FAVAE_CHECK_LAUNCH();
    if (slabs_out) { *slabs_out = a.splitk; return FAVAE_OK; }
    const size_t n = (size_t)d->Cout * d->KH * d->KW * d->Cin;
    FAVAE_KLAUNCH(reduce_slabs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, (const float*)ws, dw, n, a.splitk, accumulate);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

namespace {
// out_j[i] (+)= sum_z part_j[z][i] for up to FAVAE_REDUCE_JOBS_MAX jobs in ONE launch: block -> (job, 256-element chunk) through the
// prefix table of the jobs' block counts; the same four interleaved partial sums as reduce_slabs_kernel (bit-identical results)
struct ReduceTable {
    favae_reduce_job job[FAVAE_REDUCE_JOBS_MAX];
    unsigned first_block[FAVAE_REDUCE_JOBS_MAX + 1];
    int njobs;
};
// VEC: four consecutive outputs per thread, the slabs read as float4 (every job: n % 4 == 0, part 16-byte aligned) -- the same sums per
// output in the same order, a quarter of the load instructions and four times the bytes in flight per thread (round 5: the scalar
// kernel ran ~2 workgroups per CU through 8-16 dependent round trips: 195 us per flush of 12 jobs, 2.7 ms of the weight-gradient stream
// per step)
template <bool VEC>
__global__ __launch_bounds__(256) void reduce_slabs_grouped_kernel(ReduceTable t) {
    int j = 0;
    while (j + 1 < t.njobs && blockIdx.x >= t.first_block[j + 1]) ++j;
    const favae_reduce_job jb = t.job[j];
    const size_t n = (size_t)jb.n;
    if constexpr (VEC) {
        const size_t i = ((size_t)(blockIdx.x - t.first_block[j]) * 256 + threadIdx.x) * 4;
        if (i >= n) return;
        const float* part = jb.part + i;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        auto ld = [&](int z) { return *reinterpret_cast<const float4*>(part + (size_t)z * n); };
        auto acc = [](float4& s, const float4 v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
        int z = 0;
        for (; z + 7 < jb.slabs; z += 8) {
            const float4 v0 = ld(z), v1 = ld(z + 1), v2 = ld(z + 2), v3 = ld(z + 3), v4 = ld(z + 4), v5 = ld(z + 5), v6 = ld(z + 6), v7 = ld(z + 7);
            acc(s0, v0); acc(s1, v1); acc(s2, v2); acc(s3, v3);
            acc(s0, v4); acc(s1, v5); acc(s2, v6); acc(s3, v7);
        }
        for (; z + 3 < jb.slabs; z += 4) { acc(s0, ld(z)); acc(s1, ld(z + 1)); acc(s2, ld(z + 2)); acc(s3, ld(z + 3)); }
        for (; z < jb.slabs; ++z) acc(s0, ld(z));
        const float sum[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                              (s0.w + s1.w) + (s2.w + s3.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) jb.out[i + e] = jb.accumulate ? jb.out[i + e] + sum[e] : sum[e];     // out: any 4-byte alignment
        return;
    }
    const size_t i = (size_t)(blockIdx.x - t.first_block[j]) * 256 + threadIdx.x;
    if (i >= n) return;
    const float* part = jb.part;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 7 < jb.slabs; z += 8) {                  // eight loads in flight, summed in reduce_slabs_kernel's order
        const float v0 = part[(size_t)z * n + i], v1 = part[(size_t)(z + 1) * n + i], v2 = part[(size_t)(z + 2) * n + i],
                    v3 = part[(size_t)(z + 3) * n + i], v4 = part[(size_t)(z + 4) * n + i], v5 = part[(size_t)(z + 5) * n + i],
                    v6 = part[(size_t)(z + 6) * n + i], v7 = part[(size_t)(z + 7) * n + i];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        s0 += v4; s1 += v5; s2 += v6; s3 += v7;
    }
    for (; z + 3 < jb.slabs; z += 4) {
        s0 += part[(size_t)z * n + i];
        s1 += part[(size_t)(z + 1) * n + i];
        s2 += part[(size_t)(z + 2) * n + i];
        s3 += part[(size_t)(z + 3) * n + i];
    }
    for (; z < jb.slabs; ++z) s0 += part[(size_t)z * n + i];
    const float sum = (s0 + s1) + (s2 + s3);
    jb.out[i] = jb.accumulate ? jb.out[i] + sum : sum;
}
}  // namespace

// jobs: HOST array.  Jobs of one call must have distinct `out` ranges (two jobs accumulating into the same gradient would race).
extern "C" int favae_reduce_slabs_grouped(const favae_reduce_job* jobs, int njobs, favae_stream_t stream) {
    FAVAE_REQUIRE(jobs && njobs > 0);
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < njobs; base += FAVAE_REDUCE_JOBS_MAX) {
        ReduceTable t;
        t.njobs = njobs - base < FAVAE_REDUCE_JOBS_MAX ? njobs - base : FAVAE_REDUCE_JOBS_MAX;
        unsigned blocks = 0;
        double bytes = 0.0;
        bool vec = true;
        for (int j = 0; j < t.njobs; ++j) {
            const favae_reduce_job& jb = jobs[base + j];
            FAVAE_REQUIRE(jb.part && jb.out && jb.n > 0 && jb.slabs > 0);
            vec = vec && jb.n % 4 == 0 && (reinterpret_cast<uintptr_t>(jb.part) & 15) == 0;
        }
        for (int j = 0; j < t.njobs; ++j) {
            const favae_reduce_job& jb = jobs[base + j];
            t.job[j] = jb;
            t.first_block[j] = blocks;
            blocks += (unsigned)cdiv(vec ? jb.n / 4 : jb.n, 256);
            bytes += 4.0 * jb.n * (jb.slabs + (jb.accumulate ? 2 : 1));
        }
        t.first_block[t.njobs] = blocks;
        FAVAE_PROF_NOTE(0, bytes);
        if (vec) FAVAE_KLAUNCH(reduce_slabs_grouped_kernel<true>, dim3(blocks), dim3(256), 0, s, t);
        else FAVAE_KLAUNCH(reduce_slabs_grouped_kernel<false>, dim3(blocks), dim3(256), 0, s, t);
        FAVAE_CHECK_LAUNCH();
    }
    return FAVAE_OK;
}

namespace {
// Upsample phase weights (include/favae_hip.h): tap sets of the 3x3 kernel that land on the same low-resolution pixel
__device__ __forceinline__ void phase_taps(int par, int a, int& k0, int& k1) {      // taps k0..k1 (inclusive)
    if (par == 0) { k0 = a == 0 ? 0 : 1; k1 = a == 0 ? 0 : 2; }
    else { k0 = a == 0 ? 0 : 2; k1 = a == 0 ? 1 : 2; }
}
__global__ __launch_bounds__(256) void upsample_weights_kernel(const float* __restrict__ w, float* __restrict__ weff, int Cout,
                                                               int Cin) {
    const size_t n = (size_t)4 * Cout * 4 * Cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        size_t t = i / Cin;
        const int b = (int)(t & 1), a = (int)((t >> 1) & 1);
        t >>= 2;
        const int co = (int)(t % Cout), ph = (int)(t / Cout);
        int h0, h1, w0, w1;
        phase_taps(ph >> 1, a, h0, h1);
        phase_taps(ph & 1, b, w0, w1);
        float s = 0.f;
        for (int kh = h0; kh <= h1; ++kh)
            for (int kw = w0; kw <= w1; ++kw) s += w[(((size_t)co * 3 + kh) * 3 + kw) * Cin + ci];
        weff[i] = s;
    }
}
__global__ __launch_bounds__(256) void upsample_wgrad_fold_kernel(const float* __restrict__ dweff, float* __restrict__ dw, int Cout,
                                                                  int Cin, int accumulate) {
    const size_t n = (size_t)Cout * 9 * Cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        size_t t = i / Cin;
        const int kw = (int)(t % 3), kh = (int)((t / 3) % 3), co = (int)(t / 9);
        float s = 0.f;
        for (int py = 0; py < 2; ++py)
            for (int a = 0; a < 2; ++a) {
                int h0, h1;
                phase_taps(py, a, h0, h1);
                if (kh < h0 || kh > h1) continue;
                for (int px = 0; px < 2; ++px)
                    for (int b = 0; b < 2; ++b) {
                        int w0, w1;
                        phase_taps(px, b, w0, w1);
                        if (kw < w0 || kw > w1) continue;
                        s += dweff[((((size_t)(py * 2 + px) * Cout + co) * 2 + a) * 2 + b) * Cin + ci];
                    }
            }
        dw[i] = accumulate ? dw[i] + s : s;
    }
}
}  // namespace

extern "C" int favae_upsample_weights(const float* w, float* weff, int Cout, int Cin, favae_stream_t stream) {
    FAVAE_REQUIRE(w && weff && Cout > 0 && Cin > 0);
    const size_t n = (size_t)16 * Cout * Cin;
    FAVAE_KLAUNCH(upsample_weights_kernel, dim3((unsigned)(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256))), dim3(256), 0,
                       (hipStream_t)stream, w, weff, Cout, Cin);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_upsample_wgrad_fold(const float* dweff, float* dw, int Cout, int Cin, int accumulate, favae_stream_t stream) {
    FAVAE_REQUIRE(dweff && dw && Cout > 0 && Cin > 0);
    const size_t n = (size_t)9 * Cout * Cin;
    FAVAE_KLAUNCH(upsample_wgrad_fold_kernel, dim3((unsigned)(cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256))), dim3(256), 0,
                       (hipStream_t)stream, dweff, dw, Cout, Cin, accumulate);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_conv_subpixel_ok(int N, int H, int W, int Cin, int Cout) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("FAVAE_CONV_SUBPIXEL"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off || N <= 0 || H <= 0 || W <= 0) return 0;
    favae_conv_desc d{};
    d.N = N; d.Hin = H; d.Win = W; d.Cin = Cin; d.Hout = H; d.Wout = W; d.Cout = Cout; d.KH = 2; d.KW = 2; d.stride = 1; d.pad = 1;
    d.lat_step = 2; d.lat_side = 1;
    // forward / data-gradient phases (im2col split kernel) and the per-tap split weight gradient (128 x 128 tiles, W % 16 == 0)
    return sp_fwd_eligible(&d, false) && Cout % 16 == 0 && Cin > 64 && W % 16 == 0 && Cin % 4 == 0 && Cout % 4 == 0 &&
           (size_t)N * 4 * H * W * (Cout > Cin ? Cout : Cin) * 4 < ((size_t)1 << 31);
}

namespace {
// weight_flip_kernel + split_w_kernel in one pass: the flipped weights of the data-gradient convolution go straight into
// pre-split records.  The range of the flipped tensor is the range of w: `amax_src` is the header of the forward's record
// buffer (no second maximum reduction).  Cout % 4 == 0.
template <int SCH>
__global__ __launch_bounds__(256) void weight_flip_split_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                                int Cout, int KH, int KW, int Cin,
                                                                const float* __restrict__ amax_src) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int kh = tap / KW, kw = tap % KW;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        reinterpret_cast<float*>(out)[0] = amax_src ? amax_src[0] : 0.f;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * KH * KW + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
    constexpr int NP = sp::Scheme<SCH>::NPL;
    const float Sw = (sp::Scheme<SCH>::SCALED && amax_src) ? sp::pow2_scale(amax_src) : 1.f;
    const int tapf = (KH - 1 - kh) * KW + (KW - 1 - kw);
    // 32 ci x 8 co-quads = 256 records per tile: thread -> (ci = tid / 8, quad = tid % 8)
    const int r = threadIdx.x >> 3, qd = threadIdx.x & 7;
    const int ci = ci0 + r, co = co0 + 4 * qd;
    if (ci < Cin && co < Cout) {
        const float4 v = make_float4(tile[4 * qd][r], tile[4 * qd + 1][r], tile[4 * qd + 2][r], tile[4 * qd + 3][r]);
        uint2 p[NP];
        sp::Scheme<SCH>::split4(v, Sw, p);
        unsigned* o = reinterpret_cast<unsigned*>(out + sp::WHDR) + ((((size_t)ci * KH * KW + tapf) * Cout + co) / 4) * (2 * NP);
#pragma unroll
        for (int k = 0; k < NP; ++k) { o[2 * k] = p[k].x; o[2 * k + 1] = p[k].y; }
    }
}
}  // namespace

extern "C" int favae_weight_flip_split(const float* w, void* out, int Cout, int KH, int KW, int Cin, int planes,
                                       const float* absmax_src, favae_stream_t stream) {
    FAVAE_REQUIRE(w && out && Cout > 0 && KH > 0 && KW > 0 && Cin > 0 && Cout % 4 == 0 &&
                  (planes == 3 || planes == 4 || ((planes == 2 || planes == 1) && absmax_src)));
    FAVAE_REQUIRE((((uintptr_t)out) & 15) == 0);
    dim3 grid(cdiv(Cin, 32), cdiv(Cout, 32), KH * KW);
    if (planes == 2)
        FAVAE_KLAUNCH((weight_flip_split_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out, Cout, KH, KW,
                           Cin, absmax_src);
    else if (planes == 1)
        FAVAE_KLAUNCH((weight_flip_split_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out, Cout, KH, KW,
                           Cin, absmax_src);
    else if (planes == 4)
        FAVAE_KLAUNCH((weight_flip_split_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out, Cout, KH, KW,
                           Cin, absmax_src);
    else
        FAVAE_KLAUNCH((weight_flip_split_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out, Cout, KH, KW,
                           Cin, absmax_src);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

namespace {
// Data gradient of the stride-2 Downsample conv (3x3, zero padding on the bottom/right only: y[o] = sum_k x[2o + k] w[k]) by
// output parity instead of a x2 zero-dilated input: dx[2u] = dy[u] w[0] + dy[u-1] w[2], dx[2u+1] = dy[u] w[1] per axis, i.e.
// four convs over dy with 2x2 / 2x1 / 1x2 / 1x1 kernels -- 9 instead of 36 taps per 2x2 block of dx.  This kernel writes the
// four weight sets as pre-split records, one after the other: wph[ci][a][b][co] = w[co][kh(py,a)][kw(px,b)][ci] with
// kh(even,0) = 2, kh(even,1) = 0, kh(odd,0) = 1.  One thread per record (4 consecutive co).
template <int SCH>
__global__ __launch_bounds__(256) void downsample_dgrad_weights_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                                       int Cout, int Cin, const float* __restrict__ amax_src) {
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(out)[0] = amax_src ? amax_src[0] : 0.f;
    constexpr int NP = sp::Scheme<SCH>::NPL;
    const float Sw = (sp::Scheme<SCH>::SCALED && amax_src) ? sp::pow2_scale(amax_src) : 1.f;
    const size_t q = Cout / 4, per_tap = (size_t)Cin * q;          // records per (phase tap)
    const size_t total = 9 * per_tap;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        // phase order and sizes: (even,even) 2x2 | (even,odd) 2x1 | (odd,even) 1x2 | (odd,odd) 1x1 ; record layout [ci][a][b][co/4]
        size_t r = i;
        int ph = 0, kh_n = 2, kw_n = 2;
        const size_t sz[4] = {4 * per_tap, 2 * per_tap, 2 * per_tap, per_tap};
        while (r >= sz[ph]) { r -= sz[ph]; ++ph; }
        kh_n = ph < 2 ? 2 : 1;
        kw_n = (ph & 1) ? 1 : 2;
        const int cq = (int)(r % q);
        size_t t = r / q;
        const int b = (int)(t % kw_n); t /= kw_n;
        const int a = (int)(t % kh_n);
        const int ci = (int)(t / kh_n);
        const int kh = kh_n == 2 ? (a == 0 ? 2 : 0) : 1;
        const int kw = kw_n == 2 ? (b == 0 ? 2 : 0) : 1;
        const float* src = w + (((size_t)(4 * cq) * 3 + kh) * 3 + kw) * Cin + ci;
        const size_t cs = (size_t)9 * Cin;                      // stride between output channels of w
        const float4 v = make_float4(src[0], src[cs], src[2 * cs], src[3 * cs]);
        uint2 p[NP];
        sp::Scheme<SCH>::split4(v, Sw, p);
        unsigned* o = reinterpret_cast<unsigned*>(out + sp::WHDR) + i * (2 * NP);
#pragma unroll
        for (int k = 0; k < NP; ++k) { o[2 * k] = p[k].x; o[2 * k + 1] = p[k].y; }
    }
}
}  // namespace

extern "C" int favae_downsample_dgrad_weights(const float* w, void* out, int Cout, int Cin, int planes, const float* absmax_src,
                                              favae_stream_t stream) {
    FAVAE_REQUIRE(w && out && Cout > 0 && Cin > 0 && Cout % 4 == 0 && (planes == 3 || planes == 4 || ((planes == 2 || planes == 1) && absmax_src)));
    FAVAE_REQUIRE((((uintptr_t)out) & 15) == 0);
    const size_t total = (size_t)9 * Cin * (Cout / 4);
    const unsigned blocks = (unsigned)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256));
    if (planes == 2)
        FAVAE_KLAUNCH((downsample_dgrad_weights_kernel<2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out,
                           Cout, Cin, absmax_src);
    else if (planes == 1)
        FAVAE_KLAUNCH((downsample_dgrad_weights_kernel<1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out,
                           Cout, Cin, absmax_src);
    else if (planes == 4)
        FAVAE_KLAUNCH((downsample_dgrad_weights_kernel<4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out,
                           Cout, Cin, absmax_src);
    else
        FAVAE_KLAUNCH((downsample_dgrad_weights_kernel<3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out,
                           Cout, Cin, absmax_src);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_weight_flip(const float* w, float* wt, int Cout, int KH, int KW, int Cin, favae_stream_t stream) {
    FAVAE_REQUIRE(w && wt && Cout > 0 && KH > 0 && KW > 0 && Cin > 0);
    dim3 grid(cdiv(Cin, 32), cdiv(Cout, 32), KH * KW);
    FAVAE_KLAUNCH(weight_flip_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, Cout, KH, KW, Cin);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

static int colsum_blocks(int64_t M) {
    long b = (M + 255) / 256;
    if (b > 512) b = 512;              // (2048 blocks made this pass 1.4 x faster and the slab reduction behind it 1.5 x slower: net loss)
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" size_t favae_colsum_workspace(int64_t M, int C) { return (size_t)colsum_blocks(M) * C * sizeof(float); }

extern "C" int favae_colsum(const float* a, float* out, int64_t M, int C, int accumulate, float* absmax_out, void* ws,
                            size_t ws_bytes, favae_stream_t stream) {
    FAVAE_REQUIRE(a && out && ws && M > 0 && C > 0);
    if (ws_bytes < favae_colsum_workspace(M, C)) return favae_prof_fail_(FAVAE_ERR_WORKSPACE);
    const int nb = colsum_blocks(M);
    const long rpb = (M + nb - 1) / nb;
    hipStream_t s = (hipStream_t)stream;
    unsigned* amax = (unsigned*)absmax_out;
    if (amax && favae_zero_target(amax, sizeof(float), s) != hipSuccess) return favae_prof_fail_(FAVAE_ERR_LAUNCH);
    FAVAE_PROF_NOTE(0, 4.0 * M * C);
    if (C % 4 == 0 && ((((uintptr_t)a) & 15) == 0))
        FAVAE_KLAUNCH(colsum_partial_vec_kernel, dim3(1, nb), dim3(256), 0, s, a, (float*)ws, (long)M, C, rpb, amax);
    else
        FAVAE_KLAUNCH(colsum_partial_kernel, dim3(cdiv(C, 256), nb), dim3(256), 0, s, a, (float*)ws, (long)M, C, rpb, amax);
    FAVAE_CHECK_LAUNCH();
    FAVAE_KLAUNCH(reduce_slabs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, (const float*)ws, out, (size_t)C, nb, accumulate);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

// second stage of a column sum whose per-block partials part[blocks][C] came out of another pass (favae_gn_act_bwd_colsum):
// out[c] (+)= sum_b part[b][c], blocks in ascending order (deterministic)
extern "C" int favae_colsum_finish(const float* part, int blocks, int C, float* out, int accumulate, favae_stream_t stream) {
    FAVAE_REQUIRE(part && out && blocks > 0 && C > 0);
    FAVAE_KLAUNCH(reduce_slabs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, part, out, (size_t)C, blocks, accumulate);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_upsample2x_bwd(const float* du, float* dx, int N, int H, int W, int C, favae_stream_t stream) {
    FAVAE_REQUIRE(du && dx && N > 0 && H > 0 && W > 0 && C > 0);
    if (C % 4) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    size_t total = (size_t)N * H * W * (C / 4);
    FAVAE_KLAUNCH(upsample2x_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, du, dx, N, H, W,
                       C / 4);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}
