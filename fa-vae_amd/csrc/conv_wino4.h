// Dense 3x3 stride-1 pad-1 convolution (forward and data gradient) as Winograd F(4x4, 3x3) on the split-precision matrix pipe
// (included by conv.hip behind conv_wino.h; scheme h3 = two scaled fp16 planes, 3 x v_mfma_f32_32x32x16_f16 per fp32 product block).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 4x4 output tile and (ci, co) pair: 36 multiplies instead of 144
//
// i.e. 2.25 positions per output pixel where F(2x2, 3x3) (conv_wino.h) has 4: 0.56 x the MFMAs and 0.56 x the values to split per output.
// PMC and phase traces of the F(2x2) kernel (profiles/r04_pmc_conv.md, r04_wino_phase_trace.txt) show a SIMD's matrix and vector
// instructions adding up, so only fewer instructions per output help -- this kernel is that lever.  The price is
// accuracy: the 6x6 transforms carry constants up to 8 and sums of 100 x the input range, measured 2.3e-6 rms / 1e-5 max of the output
// range per conv against 3.6e-7 for F(2x2) (tools/experiments/wino_f43_accuracy.py).  That rules it out wherever codebook indices depend
// on the result (the encoder forward); data gradients (bar 5e-3) and, optionally, decoder layers take it (conv.hip wino4_ok, FAVAE_WINO4).
//
// Workgroup = 32 x 16 output pixels (8 x 4 Winograd tiles = the 32 rows of ONE MFMA row block) x 64 output channels, 8 waves, two per SIMD.
// Matrix role: wave (g, c) owns the nine positions 9 g .. 9 g + 8 (position = 6 a + b) for the 32 output channels of half c: nine
// f32x16 accumulators (144 registers).  Per 16-channel K chunk and workgroup 216 MFMAs for 512 pixels (F(2x2): 384).
//   * the 18 x 34 input halo is loaded once (fused GroupNorm / activation applied, scaled) and staged as fp32 in LDS;
//   * thread (tile, channel pair, half h) reads five rows of its 6 x 6 patch (8-byte reads), forms rows 3 h .. 3 h + 2 of B^T d B
//     (column transform first: 6 vector instructions per column, then 12 per row), splits the 18 positions into (hi, lo) fp16 planes and
//     stores them position-major: V[position][plane][tile][16 k] -- the MFMA A fragments are contiguous 1 KB runs;
//   * the weights stream from L2 straight into B fragments (fragment-ordered records, wino4_weights_body) through a ring of four units
//     (position, plane pair) per wave, refilled one unit at a time: 9 units per chunk and 4 slots make the slot of unit j rotate with the
//     chunk index, so the K loop is unrolled by four (Cin % 64 == 0).
// V is single-buffered (73.7 KB) and so is the raw halo (49 KB): the matrix phase of chunk k and the transform phase of chunk k + 1
// alternate behind two barriers per chunk; the halo loads of chunk k + 1 are in flight during the matrix phase of chunk k.
// Epilogue: the 36 accumulators of a (tile, channel) pair live in four waves; they are exchanged raw through LDS in two passes (tile rows
// 0-1, 2-3: 147 KB each), thread (co = lane, tile pair) forms A^T M A for two tiles (100 vector instructions each) and finishes with the
// bias / residual / statistics epilogues of the F(2x2) kernel -- 64 consecutive channels per pixel, coalesced 256-byte stores.  The
// GroupNorm partial sums keep the 16 x 16-pixel grid of the F(2x2) kernel (two per workgroup), so the consumers do not change.
// Preconditions: KH = KW = 3, stride 1, pad 1, plain gather, dense tensors, H % 16 == 0, W % 32 == 0, Cin % 64 == 0, Cout % 64 == 0.
#pragma once

#include "conv_wino.h"

namespace wino4 {
constexpr int HPX = 34, HPY = 18, NPX = HPX * HPY;   // halo pixels: 612
constexpr int RAWP = 80;                      // bytes per staged halo pixel: 16 channels fp32 + 16 B (four consecutive tiles of a tile row, 320 B
                                              // apart, read the four 64-byte quarters of a 256-byte bank row)
constexpr int RAW_B = NPX * RAWP;             // 48960
constexpr int VPL = 32 * 32;                  // bytes per (position, plane): 32 tiles x 16 k fp16
constexpr int V_B = 36 * 2 * VPL;             // 73728 per K chunk
constexpr int AFF_C = 512;
constexpr int EX_B = 36 * 4 * 2 * 64 * 8;     // 147456: raw accumulators of one epilogue pass [position][tile quad][half][co][2 tiles]
constexpr int RED_B = 8 * 64 * 2 * 8 + 64;    // statistics reduction behind them
constexpr int LDS_B = EX_B + RED_B;           // 155712 of the 160 KB (K loop: V_B + RAW_B + 2 * AFF_C * 4 = 126784)
constexpr int UCH = 36 * 2 * 2 * 1024;        // weight bytes per (co tile of 64, K chunk): [position][co block][plane][lane][16 B]
constexpr float HEAD_D = 1.0f / 128.0f;       // |B^T d B| <= 100 max|d| (row sums of |B^T|: 10): seven more bits of fp16 head room
constexpr float HEAD_W = 1.0f;                // |G g G^T| <= max|g| (row sums of |G| <= 1)

__device__ __forceinline__ float2 f2(float x, float y) { return make_float2(x, y); }
__device__ __forceinline__ float2 add2(float2 p, float2 q) { return f2(p.x + q.x, p.y + q.y); }
__device__ __forceinline__ float2 sub2(float2 p, float2 q) { return f2(p.x - q.x, p.y - q.y); }
__device__ __forceinline__ float2 fma2(float c, float2 p, float2 q) { return f2(fmaf(c, p.x, q.x), fmaf(c, p.y, q.y)); }   // c p + q
// (hi, lo) fp16 planes of two values (already scaled), stored VPL apart: 4 vector instructions, two 4-byte stores
__device__ __forceinline__ void split_store2(unsigned char* dst, const float2 t) {
    const unsigned h = wino::cvt_pk_f16(t.x, t.y);
    const unsigned l = wino::cvt_pk_f16(wino::minus_lo_half(h, t.x), wino::minus_hi_half(h, t.y));
    *reinterpret_cast<unsigned*>(dst) = h;
    *reinterpret_cast<unsigned*>(dst + VPL) = l;
}
// B^T (6 x 6) of F(4, 3) on six values: rows 0-2 need d0..d4, rows 3-5 need d1..d5
__device__ __forceinline__ void bt_lo(const float2 d0, const float2 d1, const float2 d2, const float2 d3, const float2 d4, float2& t0,
                                      float2& t1, float2& t2) {
    const float2 p = fma2(-4.f, d2, d4), q = fma2(-4.f, d1, d3);
    t0 = fma2(4.f, d0, fma2(-5.f, d2, d4));
    t1 = add2(p, q);
    t2 = sub2(p, q);
}
__device__ __forceinline__ void bt_hi(const float2 d1, const float2 d2, const float2 d3, const float2 d4, const float2 d5, float2& t3,
                                      float2& t4, float2& t5) {
    const float2 u = sub2(d4, d2), w = sub2(d3, d1);
    t3 = fma2(2.f, w, u);
    t4 = fma2(-2.f, w, u);
    t5 = fma2(4.f, d1, fma2(-5.f, d3, d5));
}
// A^T (4 x 6) of F(4, 3) on six values
__device__ __forceinline__ void at4(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5,
                                    float& t0, float& t1, float& t2, float& t3) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    t0 = (m0 + s12) + s34;
    t1 = fmaf(2.f, d34, d12);
    t2 = fmaf(4.f, s34, s12);
    t3 = fmaf(8.f, d34, d12) + m5;
}
}  // namespace wino4

// w: OHWI fp32 [Cout][3][3][Cin]; FLIP as wino_weights_body.  out (behind the header): U = G g G^T (6 x 6 positions), scaled by
// S_U = HEAD_W * 2^(14 - floor(log2 max|w|)), split into (hi, lo) fp16 planes, in MFMA B-fragment order:
// [o / 64][i / 16][position 6 a + b][(o / 32) & 1][plane][lane = (o & 31) + 32 ((i / 8) & 1)][8 k].  One thread per (o, 8 i).
template <bool FLIP>
__device__ __forceinline__ void wino4_weights_body(const float* __restrict__ w, unsigned char* __restrict__ out, int Cout, int Cin,
                                                   const float* __restrict__ amax, float* __restrict__ hdr_out, int vec, int idx) {
    const int O = FLIP ? Cin : Cout, I = FLIP ? Cout : Cin;
    const int I8 = I / 8;
    if (hdr_out && idx == 0) *hdr_out = *amax;
    if (idx >= O * I8) return;
    const int o = FLIP ? idx % O : idx / I8, i8 = FLIP ? idx / O : idx % I8;
    const float S = sp::pow2_scale(amax) * wino4::HEAD_W;
    float g[3][3][8];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            if constexpr (FLIP) {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[kh][kw][e] = w[(((size_t)(i8 * 8 + e) * 3 + (2 - kh)) * 3 + (2 - kw)) * Cin + o] * S;
            } else if (!vec) {
                const float* p = w + (((size_t)o * 3 + kh) * 3 + kw) * Cin + i8 * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) g[kh][kw][e] = p[e] * S;
            } else {
                const float4* p = reinterpret_cast<const float4*>(w + (((size_t)o * 3 + kh) * 3 + kw) * Cin + i8 * 8);
                const float4 u = p[0], v = p[1];
                g[kh][kw][0] = u.x * S; g[kh][kw][1] = u.y * S; g[kh][kw][2] = u.z * S; g[kh][kw][3] = u.w * S;
                g[kh][kw][4] = v.x * S; g[kh][kw][5] = v.y * S; g[kh][kw][6] = v.z * S; g[kh][kw][7] = v.w * S;
            }
        }
    const int KC = I / 16;
    const int ct = o >> 6, cb = (o >> 5) & 1, ln = (o & 31) + 32 * (i8 & 1), kc = i8 >> 1;
    unsigned char* base = out + (size_t)(ct * KC + kc) * wino4::UCH + cb * 2048 + ln * 16;
    // G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
    auto grow = [](int r, float g0, float g1, float g2) {
        return r == 0 ? 0.25f * g0 : r == 5 ? g2
             : r == 1 ? (-1.f / 6.f) * ((g0 + g2) + g1) : r == 2 ? (-1.f / 6.f) * ((g0 + g2) - g1)
             : r == 3 ? fmaf(1.f / 6.f, g2, fmaf(1.f / 12.f, g1, (1.f / 24.f) * g0)) : fmaf(1.f / 6.f, g2, fmaf(-1.f / 12.f, g1, (1.f / 24.f) * g0));
    };
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        float t[3][8];                          // row a of G g
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int e = 0; e < 8; ++e) t[kw][e] = grow(a, g[0][kw][e], g[1][kw][e], g[2][kw][e]);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            _Float16 hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float u = grow(b, t[0][e], t[1][e], t[2][e]);
                // ONE value of u for both planes: left to itself the compiler evaluates the inlined expression twice with different fma
                // contraction, and where the two results straddle an fp16 rounding tie hi comes from one and lo = u' - hi' from the other
                // (measured: 10 of 295k records off by one fp16 ulp of hi, which dominated the weight error: 1.7e-6 instead of 1e-7 rms)
                asm volatile("" : "+v"(u));
                hi[e] = (_Float16)u;
                lo[e] = (_Float16)(u - (float)hi[e]);
            }
            unsigned char* d = base + (size_t)(a * 6 + b) * 4096;
            *reinterpret_cast<half8_t*>(d) = half8_t{hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
            *reinterpret_cast<half8_t*>(d + 1024) = half8_t{lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
        }
    }
}

template <bool FLIP>
__global__ __launch_bounds__(256) void wino4_weights_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, int Cout, int Cin,
                                                            const float* __restrict__ amax, float* __restrict__ hdr_out, int vec) {
    wino4_weights_body<FLIP>(w, out, Cout, Cin, amax, hdr_out, vec, blockIdx.x * 256 + threadIdx.x);
}

// GB / SE as conv3x3_wino_sp_kernel
template <int XFORM, bool GB, bool SE>
__global__ __launch_bounds__(512) void conv3x3_wino4_sp_kernel(ConvArgs a) {
    static_assert(!GB || XFORM == 0, "GroupNorm-backward sums: plain data gradient");
    static_assert(!(GB && SE), "one statistics epilogue at a time");
    using namespace wino4;
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];
    unsigned char* Vs = wlds;                      // [36 positions][2 planes][32 tiles][32 B]
    unsigned char* Rs = wlds + V_B;                // [612 halo pixels][RAWP]
    float* Aff = reinterpret_cast<float*>(wlds + V_B + RAW_B);      // [2][AFF_C]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = wid & 3, wc = wid >> 2;                                // matrix role: positions 9 wg .. 9 wg + 8, output-channel half wc
    const unsigned tile = (unsigned)xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_w = a.Wout / 32, tiles_h = a.Hout / 16;
    auto udiv = [](unsigned v, unsigned rcp) { return rcp ? __umulhi(v, rcp) : v; };      // rcp = 0: divisor 1
    unsigned spt = udiv(tile, a.wino_rcp_n);
    const int tn = (int)(tile - spt * (unsigned)a.tiles_n);
    unsigned sp2 = udiv(spt, a.wino_rcp_w);
    const int tx0 = (int)(spt - sp2 * (unsigned)tiles_w) * 32;
    const unsigned sp3 = udiv(sp2, a.wino_rcp_h);
    const int ty0 = (int)(sp2 - sp3 * (unsigned)tiles_h) * 16;
    const int n = (int)sp3;
    const int n0 = tn * 64;
    const int q4 = tid & 3;
    const float Sa = sp::pow2_scale(a.x_amax) * HEAD_D;

    const auto rx = make_rsrc(a.x, a.x_bytes);
    const auto rw = make_rsrc(a.w, a.w_bytes);
    float aff_sc = 0.f, aff_sh = 0.f;
    if (XFORM) {
        if (tid < a.Cin) {
            aff_sc = a.scale[n * a.aff_stride + tid];
            aff_sh = a.shift[n * a.aff_stride + tid];
        }
    }

    // halo staging slots of this thread (612 pixels x 4 channel quads = 2448 float4 over 512 threads: four full rounds + 400 threads)
    unsigned vh[5];
    bool hok[5];
    const int ro0 = (tid >> 2) * RAWP + q4 * 16;   // slot j: + j * 128 * RAWP
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int hrow = (tid >> 2) + 128 * j;
        const int hy = hrow / HPX, hx = hrow - hy * HPX;
        const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
        hok[j] = hrow < NPX && (unsigned)y < (unsigned)a.Hin && (unsigned)x < (unsigned)a.Win;
        vh[j] = hok[j] ? (unsigned)(((n * a.in_img + y * a.in_row + x) * a.Cin + q4 * 4) * 4) : FAVAE_OOB;
    }
    float4 rg[5];
    auto load_raw = [&](int kc) {
        const unsigned sk = (unsigned)(kc * 64);
#pragma unroll
        for (int j = 0; j < 5; ++j) rg[j] = bload(rx, vh[j], sk);
    };
    auto store_raw = [&](int kc) {                   // kc = the chunk the registers hold
        float4 rsc = make_float4(0.f, 0.f, 0.f, 0.f), rsh = rsc;
        if (XFORM) {
            rsc = *reinterpret_cast<const float4*>(Aff + kc * 16 + q4 * 4);
            rsh = *reinterpret_cast<const float4*>(Aff + AFF_C + kc * 16 + q4 * 4);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float4 t0 = xform4_t<XFORM>(rg[j], rsc, rsh, a.act);
            const float sj = (XFORM && !hok[j]) ? 0.f : Sa;                  // padding stays exactly zero behind the transform
            const float4 t = make_float4(t0.x * sj, t0.y * sj, t0.z * sj, t0.w * sj);
            if (j < 4 || tid < 400) *reinterpret_cast<float4*>(Rs + ro0 + j * 128 * RAWP) = t;
        }
    };

    // transform item of this thread: tile (tty, ttx) of the 4 x 8, channel pair cp, half ah = rows 3 ah .. 3 ah + 2 of B^T d B.  The 32
    // lanes of one ds_read_b64 service group are 8 channel pairs x 4 consecutive tiles of a tile row: 4 x 64 B on the four quarters of
    // the 256-byte bank row (pixel pitch 80 B, tile pitch 320 B); their V stores are 4 x 32 consecutive bytes.
    const int cp = lane & 7;
    const int ttx = ((lane >> 3) & 3) + 4 * (lane >> 5), tty = wid & 3, ah = wid >> 2;
    const int tt = tty * 8 + ttx;
    const unsigned char* rbase = Rs + ((4 * tty + ah) * HPX + 4 * ttx) * RAWP + cp * 8;       // patch rows ah .. ah + 4
    // a tile's 32-byte row holds k 0..7 | k 8..15; tiles 16..31 keep the two halves swapped (conflict-free fragment reads, as conv_wino.h)
    unsigned char* vbase = Vs + tt * 32 + (((cp >> 2) ^ ((tt >> 4) & 1)) * 16) + (cp & 3) * 4 + ah * (18 * 2 * VPL);
    // straight-line code per half (the half is wave-uniform: one branch around the whole transform, none inside)
    auto transform_h = [&](auto ah_c) __attribute__((always_inline)) {
        constexpr int AH = decltype(ah_c)::value;
        float2 W[3][6];                              // rows 3 ah .. 3 ah + 2 of B^T d, six columns
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float2 d[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) d[r] = *reinterpret_cast<const float2*>(rbase + (r * HPX + c) * RAWP);
            if constexpr (AH != 0) bt_hi(d[0], d[1], d[2], d[3], d[4], W[0][c], W[1][c], W[2][c]);
            else bt_lo(d[0], d[1], d[2], d[3], d[4], W[0][c], W[1][c], W[2][c]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float2 t[6];
            bt_lo(W[r][0], W[r][1], W[r][2], W[r][3], W[r][4], t[0], t[1], t[2]);
            bt_hi(W[r][1], W[r][2], W[r][3], W[r][4], W[r][5], t[3], t[4], t[5]);
#pragma unroll
            for (int b = 0; b < 6; ++b) split_store2(vbase + (r * 6 + b) * 2 * VPL, t[b]);
        }
    };
    auto transform = [&]() __attribute__((always_inline)) {
        if (ah) transform_h(sp::IC<1>{});
        else transform_h(sp::IC<0>{});
    };

    // fragments: A = V[position][plane][32 tiles][16 k], B = this wave's weight records through a ring of four (position) units
    const unsigned char* Afr = Vs + (9 * wg) * 2 * VPL + (lane & 31) * 32 + (((lane >> 5) ^ ((lane >> 4) & 1)) * 16);
    const unsigned vw = (unsigned)(lane * 16);
    const int KC = a.Cin / 16;
    half8_t bfr[4][2];
    const unsigned wbase = (unsigned)(tn * KC) * (unsigned)UCH + (unsigned)((9 * wg * 2 + wc) * 2048);
    auto load_b = [&](int slot, int kc, int u) {     // unit u (position 9 wg + u) of chunk kc
        const unsigned so = wbase + (unsigned)kc * (unsigned)UCH + (unsigned)(u * 4096);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) bfr[slot][pl] = __builtin_bit_cast(half8_t, bload(rw, vw, so + (unsigned)(pl * 1024)));
    };
    f32x16 acc[9];
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    // prologue: every first load before the first wait (as conv_wino.h): halo of chunk 0, ring units 0..3, (scale, shift)
    WTRACE(0, 0);
    load_raw(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) load_b(u, 0, u);
    if (XFORM) {
        if (tid < a.Cin) {
            Aff[tid] = aff_sc;
            Aff[AFF_C + tid] = aff_sh;
        }
        __syncthreads();
    }
    store_raw(0);
    __syncthreads();
    transform();
    if (KC > 1) load_raw(1);                        // behind the transform: its registers are free again
    load_b(2, 0, 2);
    load_b(3, 0, 3);
    __syncthreads();
    WTRACE(0, 1);

    // One K chunk, ROT = kc & 3: at the top V = chunk kc, the raw tile is free, rg holds the halo loads of chunk kc + 1 (in flight), ring slot
    // (ROT + j) & 3 holds unit j for j = 0..3 (units 2, 3 requested behind the previous transform, still in flight).  Matrix phase: 27 MFMAs, each unit's slot refilled with the unit four ahead (the last
    // five from the next chunk); then the halo of chunk kc + 1 is staged; barrier; transform phase (chunk kc + 1 -> V); barrier.
    auto chunk = [&](int kc, auto rot_c) {
        constexpr int ROT = decltype(rot_c)::value;
        const bool more = kc + 1 < KC;
        const int kn = more ? kc + 1 : kc;
        WTRACE(kc + 1, 0);
        half8_t af[2][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[0][pl] = *reinterpret_cast<const half8_t*>(Afr + pl * VPL);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int s = (ROT + j) & 3;
            // the A fragments of unit j + 1 are requested BEFORE the three MFMAs of unit j (the scheduling barriers keep them there: left
            // alone the compiler sinks every fragment read to just in front of its MFMA and the LDS latency is exposed 18 times per chunk)
            if (j + 1 < 9) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) af[(j + 1) & 1][pl] = *reinterpret_cast<const half8_t*>(Afr + ((j + 1) * 2 + pl) * VPL);
            }
            __builtin_amdgcn_sched_barrier(0);
            // smallest terms first: lo x hi, hi x lo, hi x hi
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j & 1][1], bfr[s][0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j & 1][0], bfr[s][1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j & 1][0], bfr[s][0], acc[j], 0, 0, 0);
            if (j + 4 < 9) load_b(s, kc, j + 4);
            else if (j + 4 - 9 < 2) load_b(s, kn, j + 4 - 9);      // units 0, 1 of the next chunk; 2, 3 behind the transform (registers)
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_raw(kc + 1);
        WTRACE(kc + 1, 1);
        __syncthreads();                            // every wave is through with V; the raw tile holds chunk kc + 1
        WTRACE(kc + 1, 2);
        if (more) {
            transform();
            __builtin_amdgcn_sched_barrier(0);
            if (kc + 2 < KC) load_raw(kc + 2);      // in flight through the barrier and the next matrix phase
            load_b((ROT + 1 + 2) & 3, kc + 1, 2);   // ring units 2, 3 of the next chunk (its rotation is ROT + 1)
            load_b((ROT + 1 + 3) & 3, kc + 1, 3);
            WTRACE(kc + 1, 3);
            __syncthreads();
            WTRACE(kc + 1, 4);
        }
    };
    // Epilogue operand (residual / GroupNorm input of the GB sums: 512 pixels x 256 B = 1024 cache lines): each thread touches two lines
    // in front of the last four chunks, so that the epilogue's loads -- which can only be requested late, see load_pre -- hit the L2
    // instead of paying the HBM latency with nothing to overlap it (one workgroup per CU).
    unsigned pf0 = 0, pf1 = 0;
    const bool pre_any = GB || a.resid != nullptr;
    const auto rpf = make_rsrc(GB ? (const void*)a.gb_x : (a.resid ? (const void*)a.resid : (const void*)a.y),
                               pre_any ? (unsigned)((size_t)a.N * a.out_img * a.Cout * 4) : 0u);
    const unsigned pfo = (unsigned)(((n * a.out_img + (ty0 + (tid >> 5)) * a.out_row + tx0 + (tid & 31)) * a.Cout + n0) * 4);
    for (int kc = 0; kc < KC; kc += 4) {
        if (pre_any && kc + 4 >= KC) {
            pf0 = __builtin_amdgcn_raw_buffer_load_b32(rpf, pfo, 0, 0);
            pf1 = __builtin_amdgcn_raw_buffer_load_b32(rpf, pfo, 128, 0);
        }
        chunk(kc, sp::IC<0>{});
        chunk(kc + 1, sp::IC<1>{});
        chunk(kc + 2, sp::IC<2>{});
        chunk(kc + 3, sp::IC<3>{});
    }
    WTRACE(47, 0);

    // ---- epilogue.  Reader role of wave wid: tile quad tq = wid >> 1 (tile row jr = tq >> 1 of the pass, tile columns 4 (tq & 1) ..),
    // half hf = wid & 1 -> the two tiles (column 4 (tq & 1) + 2 hf + t, t = 0, 1); lane = output channel.
    const float un = sp::pow2_inv(Sa) * sp::pow2_inv(sp::pow2_scale(a.w_amax) * HEAD_W);
    const int col = n0 + lane;
    const unsigned ybytes = (unsigned)((size_t)a.N * a.out_img * a.Cout * 4);
    const auto ry = make_rsrc(a.y, ybytes);
    const unsigned vcol = (unsigned)(col * 4);
    const int tq = wid >> 1, hf = wid & 1;
    const unsigned rowb = (unsigned)(a.out_row * a.Cout * 4), pxb = (unsigned)(a.Cout * 4);
    // byte offset of pixel (i, jj) of tile t in pass ps
    auto pix_off = [&](int ps, int t, int i, int jj) {
        const unsigned pix0 = (unsigned)(n * a.out_img + (ty0 + 4 * (2 * ps + (tq >> 1))) * a.out_row + tx0 + 4 * (4 * (tq & 1) + 2 * hf));
        return pix0 * pxb + (unsigned)i * rowb + (unsigned)(4 * t + jj) * pxb;
    };
    const bool has_res = a.resid != nullptr;
    const auto rpre = make_rsrc(GB ? (const void*)a.gb_x : (has_res ? (const void*)a.resid : (const void*)a.y), (GB || has_res) ? ybytes : 0u);
    const float bv = a.bias ? a.bias[col] : 0.f;
    float g_mu = 0.f, g_rs = 0.f, g_ga = 0.f, g_be = 0.f;
    if constexpr (GB) {
        const int grp = col / (a.Cout / a.gb_groups);
        g_mu = a.gb_mean[n * a.gb_groups + grp];
        g_rs = a.gb_rstd[n * a.gb_groups + grp];
        g_ga = a.gb_gamma[col];
        g_be = a.gb_beta[col];
    }
    double gs1 = 0.0, gs2 = 0.0;
    float se_amax = 0.f;
    float* Ms = reinterpret_cast<float*>(wlds);
    // writer: accumulator register r of a lane = tile 8 (r / 4) + 4 (lane >> 5) + (r & 3) = tile row r / 4, tile column 4 (lane >> 5) + (r & 3);
    // [position][tile quad = 2 (row in pass) + (lane >> 5)][half][co][2]: 8-byte stores, consecutive lanes consecutive addresses
    float* mw = Ms + ((9 * wg * 4 + (lane >> 5)) * 2 * 64 + wc * 32 + (lane & 31)) * 2;
    const float* mr = Ms + ((tq * 2 + hf) * 64 + lane) * 2;
    // The residual (forward) or the GroupNorm input x (GB data gradient) of a pass's 32 outputs is requested behind that pass's exchange
    // stores (half of the accumulators are dead by then: 64 values beside 144 accumulator registers spill, and the scratch traffic of a
    // spill waits for the whole load queue -- 13800 cycles in front of the first exchange store, tools/wino4_trace.py) and in front of
    // its global stores (vmcnt counts stores too).  The lines were touched during the last K chunk (pf0 / pf1 below): L2 hits.
    float pre[2][2][4][4];
    auto load_pre = [&](int ps) __attribute__((always_inline)) {
        if (GB || has_res) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        pre[ps][t][i][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rpre, vcol, pix_off(ps, t, i, jj), 0));
        }
    };
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        if (ps) __syncthreads();                    // the readers of pass 0 are through
#pragma unroll
        for (int u = 0; u < 9; ++u)
#pragma unroll
            for (int jr = 0; jr < 2; ++jr) {
                const int r0 = 4 * (2 * ps + jr);
                float* p = mw + (u * 4 + jr * 2) * 2 * 64 * 2;
                *reinterpret_cast<float2*>(p) = make_float2(acc[u][r0], acc[u][r0 + 1]);
                *reinterpret_cast<float2*>(p + 64 * 2) = make_float2(acc[u][r0 + 2], acc[u][r0 + 3]);
            }
        __builtin_amdgcn_sched_barrier(0);
        load_pre(ps);
        WTRACE(47, 1 + 3 * ps);
        __syncthreads();
        WTRACE(47, 2 + 3 * ps);
        float2 m[36];
#pragma unroll
        for (int p = 0; p < 36; ++p) m[p] = *reinterpret_cast<const float2*>(mr + p * (4 * 2 * 64 * 2));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float T[4][6];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                auto M = [&](int aa) { return t ? m[aa * 6 + b].y : m[aa * 6 + b].x; };
                at4(M(0), M(1), M(2), M(3), M(4), M(5), T[0][b], T[1][b], T[2][b], T[3][b]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float f1 = 0.f, f2 = 0.f;           // four outputs in fp32, then fp64 (the F(2x2) kernel's granularity)
                float v[4];
                at4(T[i][0], T[i][1], T[i][2], T[i][3], T[i][4], T[i][5], v[0], v[1], v[2], v[3]);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float y = fmaf(v[jj], un, bv);
                    if constexpr (!GB) {
                        if (has_res) y += pre[ps][t][i][jj];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), ry, vcol, pix_off(ps, t, i, jj), 0);
                    }
                    if constexpr (GB) {
                        const float xh = (pre[ps][t][i][jj] - g_mu) * g_rs;
                        const float dyv = y * favae_act_grad(fmaf(xh, g_ga, g_be), a.gb_act & 0xff);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), ry, vcol, pix_off(ps, t, i, jj), 0);
                        f1 += dyv;
                        f2 = fmaf(dyv, xh, f2);
                    }
                    if constexpr (SE) {
                        gs1 += (double)y;
                        gs2 += (double)y * (double)y;
                        se_amax = fmaxf(se_amax, fabsf(y));
                    }
                }
                if constexpr (GB) { gs1 += (double)f1; gs2 += (double)f2; }
            }
        }
    }
    WTRACE(47, 6);
    if constexpr (GB || SE) {
        // fixed summation order: 64 outputs per thread, then the four waves of a 16-pixel-wide half -> one (S1, S2) pair per channel and
        // 16 x 16-pixel tile (the F(2x2) kernel's grid: two tiles per workgroup)
        double* red = reinterpret_cast<double*>(wlds + EX_B);             // [8 waves][64 channels][2], behind the exchange
        red[(wid * 64 + lane) * 2] = gs1;
        red[(wid * 64 + lane) * 2 + 1] = gs2;
        if constexpr (SE) {
            if (a.gs_amax) {
                float* wmx = reinterpret_cast<float*>(wlds + EX_B + 8 * 64 * 2 * sizeof(double));
                se_amax = wave_max(se_amax);
                if (lane == 0) wmx[wid] = se_amax;
            }
        }
        __syncthreads();
        if constexpr (SE) {
            if (a.gs_amax && tid == 0) {
                const float* wmx = reinterpret_cast<const float*>(wlds + EX_B + 8 * 64 * 2 * sizeof(double));
                float mx = wmx[0];
#pragma unroll
                for (int w = 1; w < 8; ++w) mx = fmaxf(mx, wmx[w]);
                atomicMax(a.gs_amax, __float_as_uint(mx));
            }
        }
        if (tid < 128) {
            const int c = tid & 63, side = tid >> 6;                    // waves of the left / right 16 columns: tq & 1 == side
            double u = 0.0, w2 = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int w = 2 * side + (q & 1) + 4 * (q >> 1);
                u += red[(w * 64 + c) * 2];
                w2 += red[(w * 64 + c) * 2 + 1];
            }
            const int tw16 = a.Wout / 16;
            const int tpi = tw16 * tiles_h;
            const int ti = (ty0 / 16) * tw16 + tx0 / 16 + side;
            double* out = (GB ? a.gb_part : a.gs_part) + (((size_t)n * tpi + ti) * a.Cout + n0 + c) * 2;
            out[0] = u;
            out[1] = w2;
        }
    }
    if ((pf0 & pf1) == 0x7fedcba9u && tid == 0x7fffffff) a.y[0] = 0.f;     // never true: keeps the two prefetch loads alive
    WTRACE(47, 7);
}
