// fp32 convolutions on the 16-bit matrix pipe by exact operand splitting (included by conv.hip).  Default conv path;
// FAVAE_CONV_MODE=h3|b6|fp32|h1 selects the scheme (fp32 = the fp32-MFMA kernels of conv_buf.h / conv_fast.h / conv.hip).
//
// Scheme<3> "b6" -- three bf16 planes by truncation, a = a1 + a2 + a3 (8 + 8 + 8 significand bits, exact); products of two
//   bf16 are exact in fp32, and keeping the six terms with i + j <= 4 (a1b1, a1b2, a2b1, a2b2, a1b3, a3b1) leaves a relative
//   error of ~2^-24 per product.  6 x v_mfma_f32_32x32x16_bf16 per fp32 product block (6/16 of the fp32-MFMA pipe time).
//   No range restrictions (bf16 has the fp32 exponent).
// Scheme<2> "h3" -- two fp16 planes by round-to-nearest of the operand scaled by a power of two S (exact):
//   a S = a1 + a2 + e, |e| <= 2^-22 |a S| (or 2^-25 absolute once a2 is subnormal); a b ~ a1b1 + a1b2 + a2b1, the dropped
//   a2b2 is <= 2^-22 |a b|.  3 x v_mfma_f32_32x32x16_f16 -- half the matrix work of b6.  The per-product error (~2^-22.5,
//   random sign) stays below the fp32 accumulation rounding that every scheme shares (measured rms vs fp64: tools/
//   conv_accuracy.py).  fp16 has 5 exponent bits, so every operand tensor carries a device-side |max| (or upper bound)
//   from which the kernel derives S = 2^(14 - floor(log2 amax)): S amax in [2^14, 2^15) -- no overflow, and elements down to
//   2^-17 of the maximum keep full relative precision.  The accumulator is un-scaled in the epilogue (exact, powers of two).
//
// Kernels: conv_fwd_sp_kernel (implicit GEMM, any gather), conv3x3_halo_sp_kernel (3x3 s1, input halo staged once per
// K chunk), conv_wgrad_sp_kernel (per tap), conv_wgrad_nine_sp_kernel (all nine taps of a 3x3 conv per workgroup); split_w*_kernel
// pre-split the weights once per call into per-4-float records.
#pragma once

#include "split_planes.h"

namespace sp {
// one pre-split weight record (4 consecutive k) -> NP plane pieces
template <int NP, typename R>
__device__ __forceinline__ void load_wrec(R rw, unsigned voff, unsigned soff, uint2 (&p)[NP]) {
    if constexpr (NP == 1) {
        p[0] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rw, voff, soff, 0));
    } else {
        const float4 t = bload(rw, voff, soff);
        p[0] = make_uint2(__float_as_uint(t.x), __float_as_uint(t.y));
        p[1] = make_uint2(__float_as_uint(t.z), __float_as_uint(t.w));
        if constexpr (NP == 3) p[2] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rw, voff + 16u, soff, 0));
    }
}

}  // namespace sp

// ---------------------------------------------------------------------------------------------------------------
// implicit-GEMM forward / data gradient, 128 x 128 x 16 tiles.  WS: the weight operand arrives pre-split (favae_split_weights),
// so the B tile is a pure copy global -> LDS and only the activation tile is split in the K loop (WS = false: NP = 3 only).
// NW = waves per workgroup: 4 (2x2 waves of 64x64) or 8 (4x2 waves of 32x64: twice the resident waves for the same LDS
// footprint, which is what hides the load -> split -> ds_write -> barrier -> ds_read chain of this short-MFMA kernel).
// ---------------------------------------------------------------------------------------------------------------
template <int GATHER, int XFORM, bool WS, int NW, int SCH>
__global__ __launch_bounds__(64 * NW) void conv_fwd_sp_kernel(ConvArgs a) {
    using S = sp::Scheme<SCH>;
    constexpr int NP = S::NPL;                 // operand planes of the scheme
    static_assert(WS || SCH == 3, "in-kernel weight split exists for the bf16x3 scheme only");
    constexpr int BN = 128, WTM = (NW == 4 ? 64 : 32), WTN = 64, MI = WTM / 32, NI = 2;
    constexpr int R = 8 / NW;                      // staged rows per thread and operand (128 rows x 4 quads / threads)
    constexpr int RSTEP = 16 * NW;                 // row distance between a thread's staged rows
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * S::ROWB];
    unsigned char* As = lds;
    unsigned char* Bs = lds + 2 * BM * S::ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;          // NW/2 x 2 waves
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
    const int q4 = tid & 3, c4 = q4 * 4;
    const int taps = a.KH * a.KW;
    const float Sa = S::SCALED ? sp::pow2_scale(a.x_amax) : 1.f;

    const auto rx = make_rsrc(a.x, a.x_bytes);
    const auto rw = make_rsrc(a.w, a.w_bytes);
    const auto rsc_d = make_rsrc(XFORM ? a.scale : a.x, XFORM ? a.aff_bytes : 0u);
    const auto rsh_d = make_rsrc(XFORM ? a.shift : a.x, XFORM ? a.aff_bytes : 0u);

    int r_n[R], r_oh[R], r_ow[R];
    bool r_ok[R];
    {
        const int hw = a.Hout * a.Wout;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int m = m0 + (tid >> 2) + RSTEP * j;
            r_ok[j] = m < a.M;
            const int mm = r_ok[j] ? m : 0;
            r_n[j] = mm / hw;
            const int r = mm - r_n[j] * hw;
            r_oh[j] = r / a.Wout;
            r_ow[j] = r - r_oh[j] * a.Wout;
        }
    }
    constexpr int WB = WS ? S::WREC : 16;           // bytes per 4 weights in global memory
    unsigned vob[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int row = (tid >> 2) + RSTEP * j;
        vob[j] = (n0 + row < a.Cout) ? (unsigned)(((n0 + row) * taps * a.Cin + c4) / 4 * WB) : FAVAE_OOB;
    }
    unsigned voa[R], vos[R];
    int ld_tap = 0, ld_kc = 0;
    auto tap_state = [&](int tap) {
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            int sh, sw;
            // separate left padding: shift the column by the difference (pad_w == pad for every ordinary conv)
            const bool ok = r_ok[j] && gather_src_t<GATHER>(a.stride, a.pad, a.Hin, a.Win, r_oh[j], r_ow[j], kh, kw + (a.pad - a.pad_w), sh, sw);
            voa[j] = ok ? (unsigned)(((r_n[j] * a.in_img + sh * a.in_step * a.in_row + sw * a.in_step + a.in_off) * a.Cin + c4) * 4)
                        : FAVAE_OOB;
            if (XFORM) vos[j] = ok ? (unsigned)((r_n[j] * a.aff_stride + c4) * 4) : FAVAE_OOB;
        }
    };
    tap_state(0);

    float4 ra[R], rsc[R], rsh[R], rb[R];
    uint2 rbp[R][NP];
    auto load_tiles = [&]() {
        const unsigned sk = (unsigned)(ld_kc * BK * 4);
        const unsigned sw = (unsigned)((ld_tap * a.Cin + ld_kc * BK) / 4 * WB);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            ra[j] = bload(rx, voa[j], sk);
            if (XFORM) {
                rsc[j] = bload(rsc_d, vos[j], sk);
                rsh[j] = bload(rsh_d, vos[j], sk);
            }
            if constexpr (WS) sp::load_wrec<NP>(rw, vob[j], sw, rbp[j]);
            else rb[j] = bload(rw, vob[j], sw);
        }
        if (++ld_kc == a.kchunks) {
            ld_kc = 0;
            if (++ld_tap < taps) tap_state(ld_tap);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int row = (tid >> 2) + RSTEP * j;
            uint2 p[NP];
            S::split4(xform4_t<XFORM>(ra[j], rsc[j], rsh[j], a.act), Sa, p);
            sp::store_planes<NP>(As + (buf * BM + row) * S::ROWB + q4 * 8, 32, p);
            if constexpr (WS) {
                sp::store_planes<NP>(Bs + (buf * BN + row) * S::ROWB + q4 * 8, 32, rbp[j]);
            } else {
                S::split4(rb[j], 1.f, p);
                sp::store_planes<NP>(Bs + (buf * BN + row) * S::ROWB + q4 * 8, 32, p);
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int T = taps * a.kchunks;
    const int frow = lane & 31, fh = (lane >> 5) * 16;
    load_tiles();
    store_tiles(0);
    __syncthreads();
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) load_tiles();
        bf16x8_t af[MI][NP], bf[NI][NP];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                af[i][p] = *reinterpret_cast<const bf16x8_t*>(As + (cur * BM + wm * WTM + i * 32 + frow) * S::ROWB + p * 32 + fh);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                bf[j][p] = *reinterpret_cast<const bf16x8_t*>(Bs + (cur * BN + wn * WTN + j * 32 + frow) * S::ROWB + p * 32 + fh);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) S::mma(af[i], bf[j], acc[i][j]);
        if (it + 1 < T) store_tiles(cur ^ 1);
        __syncthreads();
    }

    float un_a = 1.f, un_w = 1.f;
    if constexpr (S::SCALED) { un_a = sp::pow2_inv(Sa); un_w = sp::pow2_inv(sp::pow2_scale(a.w_amax)); }
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
                    size_t o = (size_t)row * a.Cout + col;
                    if (a.out_step != 1) {                      // output on a sub-grid of a larger tensor
                        const int hw = a.Hout * a.Wout;
                        const int on = row / hw, rr = row - on * hw;
                        const int oh = rr / a.Wout, ow = rr - oh * a.Wout;
                        o = ((size_t)on * a.out_img + (size_t)oh * a.out_step * a.out_row + ow * a.out_step + a.out_off) * a.Cout + col;
                    }
                    float v = acc[i][j][r];
                    if constexpr (S::SCALED) v = v * un_a * un_w;
                    v += bv;
                    if (a.resid) v += a.resid[o];
                    a.y[o] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient.  K = pixels, but NHWC tiles are pixel-major: the planes are stored [plane][pixel][channel] exactly as
// they arrive (one ds_write_b64 per plane per float4) and the MFMA fragments (8 consecutive pixels of one channel per lane)
// are fetched with gfx950's transposing LDS read ds_read_b64_tr_b16: in every 16-lane group lane s supplies the address of
// 4 consecutive channels of pixel row (s>>2), and lane c receives element (c&3) of the chunks of lanes
// {c>>2, 4+(c>>2), 8+(c>>2), 12+(c>>2)}  (probed on hardware, tools/experiments/tr_b16.hip).
// Pixel rows are padded to 320 B so that the 4 rows x 2 channel blocks a 32-lane half touches fall on different banks.
// Preconditions as conv_wgrad_buf_kernel (plain gather, Wout % 16 == 0, channels % 4 == 0), 128x128 tiles; stride 1 or 2 (round 3:
// the stride-2 Downsample convs, models/codec.py:26-29, ran their weight gradient on the fp32-MFMA kernel at 72 TFLOP/s).
// UPS: the conv input is the nearest-x2 upsampling of x (Upsample, models/codec.py:17) -- source pixel = virtual >> 1
// ---------------------------------------------------------------------------------------------------------------
template <int XFORM, bool UPS, int SCH>
__global__ __launch_bounds__(256) void conv_wgrad_sp_kernel(WgradArgs a) {
    using S = sp::Scheme<SCH>;
    constexpr int NP = S::NPL;                 // operand planes of the scheme
    constexpr int BCO = 128, BCI = 128, BKP = 16, MI = 2, NI = 2;
    constexpr int OPB = NP * sp::PLB;          // bytes per operand buffer
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * OPB];
    unsigned char* Os = lds;                   // [2][NP planes][16 px][320 B]
    unsigned char* Is = lds + 2 * OPB;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wo = wid >> 1, wi = wid & 1;
    const int taps = a.KH * a.KW;
    // One workgroup per (tap, channel tile pair, pixel chunk z): the taps of a chunk read the SAME dy pixels and overlapping x pixels.
    // Workgroups are handed to the 8 XCDs (private L2s) round robin in dispatch order, so with tap = blockIdx % taps every tap fetched
    // its operands from HBM itself -- 4.2-4.8 x the algorithmic bytes at 6.1 TB/s: this kernel was HBM-bound on its own re-reads
    // (VERDICT r4 item 6).  The linearised grid is remapped so that an XCD gets a contiguous range of logical ids: the taps of a chunk sit
    // on one XCD in neighbouring dispatch slots and share its L2.  The slab a workgroup writes depends on the logical (z, tap, tiles)
    // only: results are bit-identical.
    int t = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    const int z = t / (int)gridDim.x;
    t -= z * (int)gridDim.x;
    const int tap = t % taps; t /= taps;
    const int ci0 = (t % a.tiles_ci) * BCI;
    const int co0 = (t / a.tiles_ci) * BCO;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int p_begin = z * a.chunk;
    const int p_end = min(a.M, p_begin + a.chunk);
    const int T = (p_end > p_begin) ? (p_end - p_begin + BKP - 1) / BKP : 0;
    const float So = S::SCALED ? sp::pow2_scale(a.dy_amax) : 1.f, Si = S::SCALED ? sp::pow2_scale(a.x_amax) : 1.f;

    const auto rx = make_rsrc(a.x, a.x_bytes);
    // dense dy: rows >= p_end read as zeros through the descriptor's range check; dy on a sub-grid (a.dy_step == 2, dispatcher
    // guarantees whole 16-pixel steps): the range is the whole tensor
    const auto rdy = make_rsrc(a.dy, a.dy_step == 1 ? (unsigned)p_end * (unsigned)a.Cout * 4u
                                                    : (unsigned)a.N * (unsigned)a.dy_img * (unsigned)a.Cout * 4u);
    const auto rsc_d = make_rsrc(XFORM ? a.scale : a.x, XFORM ? a.aff_bytes : 0u);
    const auto rsh_d = make_rsrc(XFORM ? a.shift : a.x, XFORM ? a.aff_bytes : 0u);

    // staging slots: float4 index i = tid + 256 j  ->  pixel i/32, channel quad i%32
    unsigned voo[2], vos[2];
    int s_p[2], s_c[2];
    bool o_ok[2], i_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + 256 * j;
        s_p[j] = i >> 5;
        s_c[j] = (i & 31) * 4;
        o_ok[j] = co0 + s_c[j] < a.Cout;
        i_ok[j] = ci0 + s_c[j] < a.Cin;
        voo[j] = o_ok[j] ? (unsigned)((s_p[j] * a.dy_step * a.Cout + co0 + s_c[j]) * 4) : FAVAE_OOB;
        vos[j] = (unsigned)((ci0 + s_c[j]) * 4);
    }
    int s_n, s_oh, s_ow;
    {
        const int hw = a.Hout * a.Wout;
        const int mb = min(p_begin, a.M - 1);
        s_n = mb / hw;
        const int r = mb - s_n * hw;
        s_oh = r / a.Wout;
        s_ow = r - s_oh * a.Wout;
    }
    int ld_pb = p_begin;

    float4 ro[2], ri[2], rsc[2], rsh[2];
    auto load_tiles = [&]() {
        const unsigned so = a.dy_step == 1
                                ? (unsigned)ld_pb * (unsigned)a.Cout * 4u
                                : (unsigned)(s_n * a.dy_img + s_oh * a.dy_step * a.dy_row + s_ow * a.dy_step + a.dy_off) * (unsigned)a.Cout * 4u;
#pragma unroll
        for (int j = 0; j < 2; ++j) ro[j] = bload(rdy, voo[j], so);
        const int vh = s_oh * a.stride + kh - a.pad;               // stride 2: the Downsample conv (pad 0, zeros beyond the far edge)
        const bool row_ok = (unsigned)vh < (unsigned)(UPS ? 2 * a.Hin : a.Hin);
        const int ih = UPS ? vh >> 1 : vh;
        const unsigned sx = row_ok ? (unsigned)(((s_n * a.Hin + ih) * a.Win) * a.Cin) * 4u : 0u;
        const unsigned ss = (unsigned)(s_n * a.aff_stride) * 4u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int vw = (s_ow + s_p[j]) * a.stride + kw - a.pad_w;
            const int iw = UPS ? vw >> 1 : vw;
            const bool ok = row_ok && i_ok[j] && (unsigned)vw < (unsigned)(UPS ? 2 * a.Win : a.Win) && ld_pb + s_p[j] < p_end;
            const unsigned vx = ok ? (unsigned)((iw * a.Cin + ci0 + s_c[j]) * 4) : FAVAE_OOB;
            ri[j] = bload(rx, vx, sx);
            if (XFORM) {
                const unsigned vs = ok ? vos[j] : FAVAE_OOB;
                rsc[j] = bload(rsc_d, vs, ss);
                rsh[j] = bload(rsh_d, vs, ss);
            }
        }
        ld_pb += BKP;
        s_ow += BKP;
        if (s_ow >= a.Wout) {
            s_ow = 0;
            if (++s_oh >= a.Hout) { s_oh = 0; ++s_n; }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint2 p[NP];
            const int off = buf * OPB + s_p[j] * sp::RSB + s_c[j] * 2;
            S::split4(ro[j], So, p);
            sp::store_planes<NP>(Os + off, sp::PLB, p);
            S::split4(xform4_t<XFORM>(ri[j], rsc[j], rsh[j], a.act), Si, p);
            sp::store_planes<NP>(Is + off, sp::PLB, p);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing fragment reads: lane = 16 g + s ; pixel row 8 (g>>1) + (s>>2) (+4 for the second read), channels 16 (g&1) + 4 (s&3)
    const int s16 = lane & 15, g = lane >> 4;
    const int frag_off = (8 * (g >> 1) + (s16 >> 2)) * sp::RSB + (16 * (g & 1) + 4 * (s16 & 3)) * 2;
    const unsigned char* Ofr = Os + frag_off + wo * 64 * 2;
    const unsigned char* Ifr = Is + frag_off + wi * 64 * 2;

    if (T > 0) {
        load_tiles();
        store_tiles(0);
    }
    __syncthreads();
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) load_tiles();
        bf16x8_t af[MI][NP], bf[NI][NP];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) af[i][p] = sp::tr_frag(Ofr + cur * OPB + p * sp::PLB + i * 64);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[j][p] = sp::tr_frag(Ifr + cur * OPB + p * sp::PLB + j * 64);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) S::mma(af[i], bf[j], acc[i][j]);
        if (it + 1 < T) store_tiles(cur ^ 1);
        __syncthreads();
    }
    float un_o = 1.f, un_i = 1.f;
    if constexpr (S::SCALED) { un_o = sp::pow2_inv(So); un_i = sp::pow2_inv(Si); }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int ci = ci0 + wi * 64 + j * 32 + (lane & 31);
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wo * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r];
                if constexpr (S::SCALED) v = v * un_o * un_i;
                if (co < a.Cout) a.part[(((size_t)z * a.Cout + co) * taps + tap) * a.Cin + ci] = v;
            }
        }
}

// ---- weight pre-split ---------------------------------------------------------------------------------------------------
// out: WREC-byte records {plane0[4], plane1[4] (, plane2[4])} per 4 consecutive floats of `in` (n % 4 == 0)
template <int SCH>
__global__ __launch_bounds__(256) void split_w_kernel(const float4* __restrict__ in, unsigned* __restrict__ out, size_t n4,
                                                      const float* __restrict__ amax, float* __restrict__ hdr_out = nullptr) {
    constexpr int NP = sp::Scheme<SCH>::NPL;
    const float Sw = sp::Scheme<SCH>::SCALED ? sp::pow2_scale(amax) : 1.f;
    if (hdr_out && blockIdx.x == 0 && threadIdx.x == 0) *hdr_out = *amax;      // a maximum supplied by the caller goes into the header
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        uint2 p[NP];
        sp::Scheme<SCH>::split4(in[i], Sw, p);
        unsigned* o = out + i * (2 * NP);
#pragma unroll
        for (int k = 0; k < NP; ++k) { o[2 * k] = p[k].x; o[2 * k + 1] = p[k].y; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution with an LDS-staged halo tile (the "LDS-staged 3x3 input tiles" of the north star).
// A workgroup owns an 8 x 16 pixel tile (= the 128 MFMA rows) x 128 output channels.  Per 16-channel K chunk the
// (8+2) x (16+2) = 180-pixel input halo is loaded, transformed (fused GroupNorm/SiLU) and split into planes ONCE and
// then serves all 9 filter taps as shifted views -- 6.4x fewer activation loads / transforms / splits / LDS stores than the
// tap-by-tap im2col staging of conv_fwd_sp_kernel (timing ablations: those were ~35 % of that kernel).  Weights arrive
// pre-split and are streamed tap by tap through a second, double-buffered LDS tile.
// Per-thread global offsets are constants of the launch: only the scalar offsets advance (K chunk, tap).
// Preconditions: KH = KW = 3, stride 1, pad 1, plain gather, H % 8 == 0, W % 16 == 0, Cin % 16 == 0, pre-split weights.
// ---------------------------------------------------------------------------------------------------------------
// GB: GroupNorm-backward partial sums in the epilogue (gb_*)
// SE: per-tile (sum y, sum y^2) of the output in the epilogue (gs_part): pass 1 of the GroupNorm that consumes this conv's output
// AT (round 6): storage type of the activation tensors x, resid, y (and the GroupNorm input of the GB epilogue) -- float, or bf16_t
// (common.h: bf16 activation storage, scheme 4 only): a thread's four channels are one 8-byte access, an epilogue lane's channel 2 bytes.
template <int XFORM, int SCH, int KS = 3, bool GB = false, bool SE = false, typename AT = float>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(SCH != 3 ? 6 : 4, 8))) void conv3x3_halo_sp_kernel(ConvArgs a) {
    static_assert(sizeof(AT) == 4 || SCH == 4, "bf16 activation storage: the one-bf16-plane scheme");
    constexpr unsigned EB = ActT<AT>::B;
    static_assert(!GB || (XFORM == 0 && KS == 3), "GroupNorm-backward sums: plain dense 3x3 data gradient");
    static_assert(!SE || (KS == 3 && !GB && SCH != 3), "output statistics: dense 3x3 forward conv, one or two planes");
    static_assert(!GB || SCH != 3, "GroupNorm-backward sums: one or two planes");
    using S = sp::Scheme<SCH>;
    constexpr int NP = S::NPL;                 // operand planes of the scheme
    // KS = 3: the 3x3 stride-1 pad-1 conv.  KS = 2: a 2x2 phase conv of an Upsample / of the Downsample data gradient (top / left
    // padding a.pad / a.pad_w in {0, 1}, ONE side of the conv on every second pixel of a tensor of twice the size: a.in_* / a.out_*).
    constexpr int TH = 8, TW = 16, HW = TW + KS - 1, HROWS = (TH + KS - 1) * HW, TAPS = KS * KS;   // 3x3: 180 halo pixels
    // halo row pitch = 18 rows rounded up to a multiple of 256 B: pitch % 256 == 0 puts the second tile row of a wave's 32 MFMA
    // rows on the same bank phase as pixels 16..31 of a contiguous run -> the ds_read_b128 fragment reads are conflict-free
    constexpr int HPITCH = (HW * S::ROWB + 255) / 256 * 256;
    static_assert((16 * S::ROWB) % 256 == 0, "row stride must keep 16-pixel runs bank-periodic");
    constexpr int HALO_B = (TH + KS - 1) * HPITCH, BT_B = 128 * S::ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HALO_B + 2 * BT_B];
    unsigned char* Hs = lds;                    // [2][TH + KS - 1][HPITCH]
    unsigned char* Bs = lds + 2 * HALO_B;       // [2][128][ROWB]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;      // 4 x 2 waves of 32 x 64
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % a.tiles_n;
    int spt = tile / a.tiles_n;                 // spatial tile index
    const int tiles_w = a.Wout / TW, tiles_h = a.Hout / TH;
    const int tx0 = (spt % tiles_w) * TW; spt /= tiles_w;
    const int ty0 = (spt % tiles_h) * TH;
    const int n = spt / tiles_h;
    const int n0 = tn * 128;
    const int q4 = tid & 3;
    const float Sa = S::SCALED ? sp::pow2_scale(a.x_amax) : 1.f;

    const auto rx = make_rsrc(a.x, a.x_bytes);
    const auto rw = make_rsrc(a.w, a.w_bytes);
    const auto rsc_d = make_rsrc(XFORM ? a.scale : a.x, XFORM ? a.aff_bytes : 0u);
    const auto rsh_d = make_rsrc(XFORM ? a.shift : a.x, XFORM ? a.aff_bytes : 0u);

    // halo staging slots of this thread (720 float4 over 512 threads): constant offsets.  The fused-transform operands
    // (scale, shift) depend on (image, channel quad) only: one load per K chunk serves both slots; padding pixels must
    // stay exactly zero after the transform, so they are masked with a select instead of zeroed operands.
    unsigned vh[2];
    int hoff[2];
    bool hok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + 512 * j;
        const int hrow = i >> 2;
        const int hy = hrow / HW, hx = hrow - hy * HW;
        hoff[j] = hy * HPITCH + hx * S::ROWB;
        const int y = ty0 - a.pad + hy, x = tx0 - a.pad_w + hx;
        hok[j] = hrow < HROWS && (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
        vh[j] = hok[j] ? (unsigned)(((n * a.in_img + y * a.in_step * a.in_row + x * a.in_step + a.in_off) * a.Cin + q4 * 4) * EB) : FAVAE_OOB;
    }
    const unsigned vs = (unsigned)((n * a.aff_stride + q4 * 4) * 4);
    const int brow = tid >> 2;                                           // weight row (output channel) staged by this thread
    const unsigned vb = (n0 + brow < a.Cout) ? (unsigned)(((n0 + brow) * TAPS * a.Cin + q4 * 4) / 4 * S::WREC) : FAVAE_OOB;

    float4 rh[2], rsc, rsh;
    uint2 rbp[NP];
    auto load_halo = [&](int kc) {
        const unsigned sk = (unsigned)(kc * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && tid >= HROWS * 4 - 512) continue;              // second slot exists for the first 208 threads only
            rh[j] = act_load4<AT>(rx, vh[j], (unsigned)(kc * 16) * EB);
        }
        if (XFORM) {
            rsc = bload(rsc_d, vs, sk);
            rsh = bload(rsh_d, vs, sk);
        }
    };
    auto store_halo = [&](int buf, int kc) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && tid >= HROWS * 4 - 512) continue;
            uint2 p[NP];
            float4 t = xform4_t<XFORM>(rh[j], rsc, rsh, a.act);
            if (XFORM && !hok[j]) t = make_float4(0.f, 0.f, 0.f, 0.f);
            S::split4(t, Sa, p);
            sp::store_planes<NP>(Hs + buf * HALO_B + hoff[j] + q4 * 8, 32, p);
        }
    };
    auto load_b = [&](int kc, int tap) { sp::load_wrec<NP>(rw, vb, (unsigned)((tap * a.Cin + kc * 16) / 4 * S::WREC), rbp); };
    auto store_b = [&](int buf) { sp::store_planes<NP>(Bs + buf * BT_B + brow * S::ROWB + q4 * 8, 32, rbp); };

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // fragment addressing: MFMA row = pixel p = wm*32 + (lane&31) of the 8x16 tile -> halo row (ty+kh), column tx+kw
    const int p = wm * 32 + (lane & 31);
    const int fh = (lane >> 5) * 16;
    const unsigned char* Afr = Hs + (p >> 4) * HPITCH + (p & 15) * S::ROWB + fh;
    const unsigned char* Bfr = Bs + (wn * 64 + (lane & 31)) * S::ROWB + fh;

    const int KC = a.Cin / 16;
    load_halo(0);
    load_b(0, 0);
    store_halo(0, 0);
    store_b(0);
    __syncthreads();
    int it = 0;
    for (int kc = 0; kc < KC; ++kc) {
        const int hb = kc & 1;
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap, ++it) {
            const int cur = it & 1;
            const bool last_tap = tap == TAPS - 1, more_kc = kc + 1 < KC;
            if (!last_tap) load_b(kc, tap + 1);
            else if (more_kc) load_b(kc + 1, 0);
            if (tap == TAPS / 2 && more_kc) load_halo(kc + 1);           // in flight over the second half of the taps
            const int kh = tap / KS, kw = tap - kh * KS;
            const unsigned char* Ab = Afr + hb * HALO_B + kh * HPITCH + kw * S::ROWB;
            const unsigned char* Bb = Bfr + cur * BT_B;
            bf16x8_t af[NP], bf[2][NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) af[pl] = *reinterpret_cast<const bf16x8_t*>(Ab + pl * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8_t*>(Bb + j * 32 * S::ROWB + pl * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j) S::mma(af, bf[j], acc[j]);
            if (!last_tap || more_kc) store_b(cur ^ 1);
            if (last_tap && more_kc) store_halo(hb ^ 1, kc + 1);
            __syncthreads();
        }
    }

    float un_a = 1.f, un_w = 1.f;
    if constexpr (S::SCALED) { un_a = sp::pow2_inv(Sa); un_w = sp::pow2_inv(sp::pow2_scale(a.w_amax)); }
    double gs1[2] = {0.0, 0.0}, gs2[2] = {0.0, 0.0};
    float se_amax = 0.f;                 // SE: max |y| of this thread's outputs (operand range of a plain conv that consumes y)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= a.Cout) continue;
        const float bv = a.bias ? a.bias[col] : 0.f;
        // output offsets of this lane's 16 rows: tile row pr >> 4 = 2 wm + (r >> 3), column (r & 3) + 8 ((r >> 2) & 1) + 4 (lane >> 5)
        const size_t obase = ((size_t)n * a.out_img + (size_t)ty0 * a.out_step * a.out_row + tx0 * a.out_step + a.out_off) * a.Cout + col;
        float xg[GB ? 16 : 1];
        if constexpr (GB) {              // the 16 x values first: one batch of independent loads in flight (32-bit offsets)
            const auto rgx = make_rsrc(a.gb_x, (unsigned)((size_t)a.N * a.out_img * a.Cout * EB));
            const unsigned ob32 = (unsigned)obase * EB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                xg[r] = act_load1<AT>(rgx, ob32 + (unsigned)(((pr >> 4) * a.out_row + (pr & 15)) * a.Cout) * EB, 0);
            }
        }
        float g_mu = 0.f, g_rs = 0.f, g_ga = 0.f, g_be = 0.f, f1 = 0.f, f2 = 0.f;
        if constexpr (GB) {
            const int grp = col / (a.Cout / a.gb_groups);
            g_mu = a.gb_mean[n * a.gb_groups + grp];
            g_rs = a.gb_rstd[n * a.gb_groups + grp];
            g_ga = a.gb_gamma[col];
            g_be = a.gb_beta[col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);          // pixel of the tile
            const size_t o = obase + ((size_t)(pr >> 4) * a.out_step * a.out_row + (pr & 15) * a.out_step) * a.Cout;
            float v = acc[j][r];
            if constexpr (S::SCALED) v = v * un_a * un_w;
            v += bv;
            if (a.resid) v += act_get<AT>(a.resid, o);
            if constexpr (!GB) act_put<AT>(a.y, o, v);
            if constexpr (GB) {          // v = da at (pixel, channel col); the conv input x has the same shape
                const float xh = (xg[r] - g_mu) * g_rs;
                const float dyv = v * favae_act_grad(fmaf(xh, g_ga, g_be), a.gb_act & 0xff);
                act_put<AT>(a.y, o, v);
                f1 += dyv;               // 16 terms in fp32, everything above that in fp64
                f2 = fmaf(dyv, xh, f2);
            }
            if constexpr (SE) {          // statistics feed E[y^2] - mean^2: fp64 from the first product on (norm.hip)
                gs1[j] += (double)v;
                gs2[j] += (double)v * (double)v;
                se_amax = fmaxf(se_amax, fabsf(v));
            }
        }
        if constexpr (GB) { gs1[j] = (double)f1; gs2[j] = (double)f2; }
    }
    if constexpr (GB || SE) {
        // fixed summation order: 16 rows per lane, the two half-waves, then the four pixel-row waves -> one (S1, S2) pair per
        // channel of this 8x16-pixel tile, reduced over the tiles of the image by gn_bwd_finalize_kernel (deterministic)
        double* red = reinterpret_cast<double*>(lds);                // [4 wm][128 channels][2]
        __syncthreads();                                             // the main loop's last LDS reads are done
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gs1[j] += __shfl_xor(gs1[j], 32, 64);
            gs2[j] += __shfl_xor(gs2[j], 32, 64);
            if (lane < 32) {
                red[(wm * 128 + wn * 64 + j * 32 + lane) * 2] = gs1[j];
                red[(wm * 128 + wn * 64 + j * 32 + lane) * 2 + 1] = gs2[j];
            }
        }
        __syncthreads();
        if constexpr (SE) {
            if (a.gs_amax) {             // one atomic per workgroup; non-negative floats order like their bit patterns
                float* wmx = reinterpret_cast<float*>(lds + 4 * 128 * 2 * sizeof(double));
                se_amax = wave_max(se_amax);
                if (lane == 0) wmx[wid] = se_amax;
                __syncthreads();
                if (tid == 0) {
                    float m = wmx[0];
#pragma unroll
                    for (int w = 1; w < 8; ++w) m = fmaxf(m, wmx[w]);
                    atomicMax(a.gs_amax, __float_as_uint(m));
                }
            }
        }
        if (tid < 128 && n0 + tid < a.Cout) {
            double u = 0.0, w2 = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { u += red[(q * 128 + tid) * 2]; w2 += red[(q * 128 + tid) * 2 + 1]; }
            const int tpi = tiles_w * tiles_h;
            const int ti = (ty0 / TH) * tiles_w + tx0 / TW;
            double* out = (GB ? a.gb_part : a.gs_part) + (((size_t)n * tpi + ti) * a.Cout + n0 + tid) * 2;
            out[0] = u;
            out[1] = w2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 weight gradient, ALL NINE taps per workgroup, every operand element loaded / transformed / split ONCE (round 3).
// A workgroup owns (128 co x 64 ci x 9 taps) and walks DOWN 16-pixel-wide column strips of the images: per step it stages one
// 16-pixel row segment of dy and one 18-pixel (halo) row segment of x; the x rows live in a 4-slot LDS ring, so the three filter
// rows kh = 0,1,2 of step h read the ring slots of rows h-1, h, h+1 and every x row is fetched from HBM / L2 once instead of by
// three filter-row workgroups (the round-1 row kernel, removed in round 6: 3.5x the algorithmic traffic, and GroupNorm+SiLU + plane split of every
// x element three times).  27 MFMA product blocks (x planes products) per wave and step against one dy float4 and 0.56 x
// float4 staged per thread: 2.75x less vector-ALU work per MFMA than the row3 kernel.
// 8 waves = 4 (co) x 2 (ci), each 32 co x 32 ci x 9 taps = 144 accumulator registers.
// Split-K over whole strips (slab z = strips [z sps, (z+1) sps)); the tiles of one slab sit on ONE XCD (blockIdx % 8) next to each
// other in dispatch order, so the dy rows shared by the ci tiles and the x rows shared by the co tiles are served by that L2.
// Preconditions: KH = KW = 3, stride 1, pad 1, plain gather, Wout % 16 == 0, channels % 4 == 0, operands < 2 GiB.
// ---------------------------------------------------------------------------------------------------------------
namespace sp {
constexpr int XPITCH = 192;                // bytes per pixel of an x ring plane: 64 ci x 2 B + 64 B pad (conflict-free tr reads:
                                           // the 4 pixel rows of a 32-lane half land on the four 64-byte quarters of the 256-byte bank line)
}
// BCO = output channels per workgroup: 128 (8 waves, 4 co x 2 ci) or 64 (4 waves, 2 x 2).  CAP1 = declare > 80 KB of LDS per
// workgroup: a residency cap of ONE workgroup per CU (160 KB), so that the HBM-bound kernels of the main stream find free wave slots and registers next
// to this register-heavy kernel (it runs on the weight-gradient stream, meant to overlap exactly those kernels).
// PF = global prefetch distance in steps: the loads of stage j + PF are issued before the products of step j and stored to LDS one
// step before they are used (PF register stages).  One step is 27 MFMA blocks per wave (~0.9k cycles), a loaded HBM round trip
// is longer: with PF = 1 the step time IS the load latency (measured: the 4-wave variant ran at half the rate of the 8-wave one).
namespace sp {
template <int I> struct IC { static constexpr int value = I; };
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<N, I + 1>(f);
    }
}
}  // namespace sp
// AT (round 6): storage type of x and dy -- float, or bf16_t (bf16 activation storage, scheme 4; common.h)
template <int XFORM, int SCH, int BCO = 128, bool CAP1 = false, int PF = 2, typename AT = float>
__global__ __launch_bounds__(BCO * 4) void conv_wgrad_nine_sp_kernel(WgradArgs a) {
    static_assert(sizeof(AT) == 4 || SCH == 4, "bf16 activation storage: the one-bf16-plane scheme");
    constexpr unsigned EB = ActT<AT>::B;
    using S = sp::Scheme<SCH>;
    constexpr int NP = S::NPL;                 // operand planes of the scheme
    static_assert(PF >= 1 && PF <= 3, "prefetch distance in steps");
    constexpr int T = BCO * 4;                                       // threads
    constexpr int OPITCH = BCO == 128 ? sp::RSB : sp::XPITCH;        // bytes per pixel of a dy plane
    constexpr int OPL = 16 * OPITCH, OB = NP * OPL;                  // dy: plane, buffer (16 px x BCO co)
    constexpr int IPL = 18 * sp::XPITCH, IROW = NP * IPL;            // x: plane of one ring row (18 px x 64 ci), ring row
    constexpr int NXS = (18 * 16 + T - 1) / T;                       // x staging slots per thread (288 float4 per row)
    constexpr int LUSE = 2 * OB + 4 * IROW;
    constexpr int LPAD = (CAP1 && LUSE < 82 * 1024) ? 82 * 1024 - LUSE : 0;      // > 80 KB per workgroup: one workgroup per CU
    __shared__ __attribute__((aligned(16))) unsigned char lds[LUSE + LPAD];
    unsigned char* Os = lds;                                         // [2][NP][16][OPITCH]
    unsigned char* Is = lds + 2 * OB;                                // [4][NP][18][XPITCH]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wo = wid >> 1, wi = wid & 1;
    // blockIdx -> (tile, slab): XCD = blockIdx % 8; inside an XCD consecutive workgroups are the tiles of one slab
    const int tiles = a.tiles_co * a.tiles_ci;
    const int q = blockIdx.x >> 3;
    const int tile = q % tiles;
    const int z = (q / tiles) * 8 + (blockIdx.x & 7);
    if (z >= a.splitk) return;
    const int ci0 = (tile % a.tiles_ci) * 64;
    const int co0 = (tile / a.tiles_ci) * BCO;
    const int strips_w = a.Wout >> 4;
    const int NS = a.N * strips_w;
    const int s_begin = z * a.chunk, s_end = min(NS, s_begin + a.chunk);       // chunk = strips per slab
    const int H = a.Hout;
    const float So = S::SCALED ? sp::pow2_scale(a.dy_amax) : 1.f, Si = S::SCALED ? sp::pow2_scale(a.x_amax) : 1.f;
    if (LPAD && a.M < 0) lds[LUSE + LPAD - 1 - tid] = 0;                      // never true: keeps the padding allocated

    const auto rx = make_rsrc(a.x, a.x_bytes);
    const auto rdy = make_rsrc(a.dy, (unsigned)a.M * (unsigned)a.Cout * EB);
    const auto rsc_d = make_rsrc(XFORM ? a.scale : a.x, XFORM ? a.aff_bytes : 0u);
    const auto rsh_d = make_rsrc(XFORM ? a.shift : a.x, XFORM ? a.aff_bytes : 0u);

    // staging slots: dy float4 i = tid -> pixel tid / (BCO/4), co quad tid % (BCO/4); x float4 i = tid + j T < 288 -> pixel i / 16, ci quad i % 16
    const int opx = tid / (BCO / 4), oq = (tid % (BCO / 4)) * 4;
    const unsigned voo = (co0 + oq < a.Cout) ? (unsigned)((opx * a.Cout + co0 + oq) * EB) : FAVAE_OOB;
    const int iq = (tid & 15) * 4;                                   // T % 16 == 0: the same channel quad in every slot
    const bool ci_ok = ci0 + iq < a.Cin;
    const unsigned vsc = ci_ok ? (unsigned)((ci0 + iq) * 4) : FAVAE_OOB;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int s16 = lane & 15, g = lane >> 4;
    const unsigned char* Ofr = Os + (8 * (g >> 1) + (s16 >> 2)) * OPITCH + (16 * (g & 1) + 4 * (s16 & 3)) * 2 + wo * 64;
    const unsigned char* Ifr = Is + (8 * (g >> 1) + (s16 >> 2)) * sp::XPITCH + (16 * (g & 1) + 4 * (s16 & 3)) * 2 + wi * 64;

    // virtual step stream: every strip contributes v = -1 (seed: x rows -1 and 0, no product) and v = 0 .. H-1
    // stage(v) brings dy row v (v >= 0) and x row v + 1 (zeros when v + 1 == H) -- and zeros for x row -1 when v == -1
    // Every stage issues the SAME loads, unconditionally (stages behind the end of the slab, absent dy rows and the lanes without a
    // second x slot use the out-of-range offset: no memory traffic): with branch-free load / store sequences the compiler's
    // s_waitcnt insertion can count -- behind conditional loads it waited for the stage issued one step earlier right before issuing
    // the next one, which turned the prefetch distance back into one exposed round trip per step.
    int ld_s = s_begin, ld_v = -1;            // the stage the NEXT load_stage() call fetches
    int ld_n = s_begin / strips_w, ld_w0 = (s_begin - (s_begin / strips_w) * strips_w) << 4;
    float4 ro[PF], ri[PF][NXS], rsc[PF], rsh[PF];     // PF register stages; stage k sits in stage registers k % PF
    bool x_ok[PF][NXS];
    int st_v[PF];
    auto load_stage = [&](auto SL) {
        constexpr int L = decltype(SL)::value;
        const bool live = ld_s < s_end;
        st_v[L] = ld_v;
        const unsigned img = (unsigned)(ld_n * H) * (unsigned)a.Wout;
        // dy row v: pixels (n, v, w0 .. w0 + 15)
        ro[L] = act_load4<AT>(rdy, (live && ld_v >= 0) ? voo : FAVAE_OOB, (img + (unsigned)(max(ld_v, 0) * a.Wout + ld_w0)) * (unsigned)a.Cout * EB);
        // x row v + 1: pixels (n, v + 1, w0 - 1 .. w0 + 16)
        const int xr = ld_v + 1;
        const unsigned sx = (img + (unsigned)(min(xr, H - 1) * a.Wout)) * (unsigned)a.Cin * EB;
#pragma unroll
        for (int j = 0; j < NXS; ++j) {
            const int ipx = (tid + j * T) >> 4, iw = ld_w0 - 1 + ipx;
            x_ok[L][j] = live && ipx < 18 && ci_ok && xr < H && (unsigned)iw < (unsigned)a.Win;
            ri[L][j] = act_load4<AT>(rx, x_ok[L][j] ? (unsigned)((iw * a.Cin + ci0 + iq) * EB) : FAVAE_OOB, sx);
        }
        if (XFORM) {                          // GroupNorm affine of (image, channel quad): an L1 hit after the first step of a strip
            const unsigned ss = (unsigned)(ld_n * a.aff_stride) * 4u;
            rsc[L] = bload(rsc_d, live ? vsc : FAVAE_OOB, ss);
            rsh[L] = bload(rsh_d, live ? vsc : FAVAE_OOB, ss);
        }
        if (++ld_v == H) {
            ld_v = -1;
            ++ld_s;
            ld_n = ld_s / strips_w;
            ld_w0 = (ld_s - ld_n * strips_w) << 4;
        }
    };
    auto store_stage = [&](auto SL) {
        constexpr int L = decltype(SL)::value;
        const int sv = st_v[L];
        uint2 p[NP];
        S::split4(ro[L], So, p);
        sp::store_planes<NP>(Os + (sv & 1) * OB + opx * OPITCH + oq * 2, OPL, p);   // sv == -1: zeros into the idle buffer 1
#pragma unroll
        for (int j = 0; j < NXS; ++j) {
            if (tid + j * T >= 18 * 16) continue;
            float4 t = xform4_t<XFORM>(ri[L][j], rsc[L], rsh[L], a.act);
            if (XFORM && !x_ok[L][j]) t = make_float4(0.f, 0.f, 0.f, 0.f);
            S::split4(t, Si, p);
            unsigned char* d = Is + ((tid + j * T) >> 4) * sp::XPITCH + iq * 2;
            sp::store_planes<NP>(d + ((sv + 2) & 3) * IROW, IPL, p);            // row v + 1 -> slot (v + 2) & 3
            if (sv == -1) {                                                     // row -1 of the strip -> slot 0: the top padding
#pragma unroll
                for (int k = 0; k < NP; ++k) p[k] = make_uint2(0u, 0u);
                sp::store_planes<NP>(d, IPL, p);
            }
        }
    };

    const int J = (s_end - s_begin) * (H + 1);
    // prologue: stage 0 into LDS, stages 1 .. PF-1 into their registers
    sp::static_for<PF>([&](auto U) {
        constexpr int u = decltype(U)::value;
        load_stage(sp::IC<u>{});
        if (u == 0) store_stage(sp::IC<0>{});
    });
    __syncthreads();
    int v = -1;
    for (int j0 = 0; j0 < J; j0 += PF) {
        sp::static_for<PF>([&](auto U) {
            constexpr int u = decltype(U)::value;
            if (j0 + u < J) {
                load_stage(sp::IC<u>{});                                        // stage j + PF -> the registers stage j just left
                if (v >= 0) {
                    // fragments of filter row kh + 1 are requested before the products of row kh are issued (two register sets)
                    bf16x8_t af[NP], bf[2][3][NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) af[p] = sp::tr_frag_p<OPITCH>(Ofr + (v & 1) * OB + p * OPL);
                    auto read_row = [&](int kh, bf16x8_t (&dst)[3][NP]) {
                        const unsigned char* row = Ifr + ((v + kh) & 3) * IROW; // x row v - 1 + kh
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int p = 0; p < NP; ++p) dst[kw][p] = sp::tr_frag_p<sp::XPITCH>(row + p * IPL + kw * sp::XPITCH);
                    };
                    read_row(0, bf[0]);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        if (kh < 2) read_row(kh + 1, bf[(kh + 1) & 1]);
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) S::mma(af, bf[kh & 1][kw], acc[kh * 3 + kw]);
                    }
                }
                // the seed stage of the next strip overwrites ring slots 0 and 1, which the last step of this strip may still read
                if (v == H - 1) __syncthreads();
                store_stage(sp::IC<(u + 1) % PF>{});
                __syncthreads();
                if (++v == H) v = -1;
            }
        });
    }

    float un_o = 1.f, un_i = 1.f;
    if constexpr (S::SCALED) { un_o = sp::pow2_inv(So); un_i = sp::pow2_inv(Si); }
    const int ci = ci0 + wi * 32 + (lane & 31);
    if (ci >= a.Cin) return;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float val = acc[t][r];
            if constexpr (S::SCALED) val = val * un_o * un_i;
            if (co < a.Cout) a.part[(((size_t)z * a.Cout + co) * 9 + t) * a.Cin + ci] = val;
        }
}
