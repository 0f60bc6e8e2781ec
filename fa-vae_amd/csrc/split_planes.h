// Operand planes of the split-precision matrix path (shared by the conv kernels, conv_split.h, and the batched GEMM of the
// attention core, gemm.hip): the fp16 / bf16 splitting schemes, the power-of-two operand scaling, LDS plane stores and the
// transposing fragment read.  See conv_split.h for the arithmetic.
#pragma once

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

namespace sp {
constexpr unsigned TOP = 0xFFFF0000u;
constexpr int WHDR = 256;                  // bytes of header in front of the weight records (float[0] = |max| of the weights)

// S = 2^(14 - floor(log2 amax)) from a device-side maximum (1 for zero / non-finite maxima)
__device__ __forceinline__ float pow2_scale(const float* amax) {
    const int e = (int)((__float_as_uint(*amax) & 0x7FFFFFFFu) >> 23);
    if (e == 0 || e == 255) return 1.f;
    int se = 268 - e;                      // biased exponent of S
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ float pow2_inv(float S) {
    return __uint_as_float((254u - (__float_as_uint(S) >> 23)) << 23);
}

template <int NP> struct Scheme;

template <> struct Scheme<3> {
    static constexpr int NPL = 3;          // operand planes
    static constexpr bool SCALED = false;  // bf16 keeps the fp32 exponent: no operand ranges needed
    static constexpr int ROWB = 112;       // LDS row: 3 planes x 32 B (16 k) + 16 B pad -> conflict-free ds_read_b128
    static constexpr int WREC = 24;        // bytes per pre-split 4-float weight record {plane0[4], plane1[4], plane2[4]}
    // split four consecutive-k floats into three planes of 4 bf16 (2 dwords each), exact by truncation
    static __device__ __forceinline__ void split4(const float4 v, float, uint2 (&p)[3]) {
        const float a[4] = {v.x, v.y, v.z, v.w};
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = __float_as_uint(a[e]);
            const float r1 = a[e] - __uint_as_float(h[e] & TOP);
            m[e] = __float_as_uint(r1);
            const float r2 = r1 - __uint_as_float(m[e] & TOP);
            l[e] = __float_as_uint(r2);
        }
        // pack the upper halves of two words: low 16 bits <- even k, high 16 bits <- odd k
        p[0] = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
        p[1] = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
        p[2] = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
    }
    // smallest terms first
    static __device__ __forceinline__ void mma(const bf16x8_t (&a)[3], const bf16x8_t (&b)[3], f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    }
};

template <> struct Scheme<2> {
    static constexpr int NPL = 2;
    static constexpr bool SCALED = true;
    static constexpr int ROWB = 80;        // 2 planes x 32 B + 16 B pad (20-bank row stride: conflict-free ds_read_b128)
    static constexpr int WREC = 16;        // {plane0[4 fp16], plane1[4]}
    // hi = rne(a) as a packed pair (v_cvt_pk_f16_f32); the residual a - hi from one mixed-precision FMA per value (v_fma_mix_f32: the
    // f16 half of the pair x -1 + the fp32 value: exact, the same bits as a - (float)hi) -- 8 vector instructions per four values
    // behind the scaling; the compiler's own lowering of the C expression converts every hi twice (16).  A SIMD hides about five vector
    // instructions per MFMA (profiles/r06_mfma_valu_sweep.txt); the direct kernels issue 3-12: beyond the fifth every one costs ~4 cycles.
    static __device__ __forceinline__ void split4(const float4 v, float S, uint2 (&p)[2]) {
        const float a0 = v.x * S, a1 = v.y * S, a2 = v.z * S, a3 = v.w * S;
        unsigned h01, h23, l01, l23;
        float r0, r1, r2, r3;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(a0), "v"(a1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(a2), "v"(a3));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "v"(a0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "v"(a1));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "v"(a2));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "v"(a3));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l01) : "v"(r0), "v"(r1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l23) : "v"(r2), "v"(r3));
        p[0] = make_uint2(h01, h23);
        p[1] = make_uint2(l01, l23);
    }
    static __device__ __forceinline__ void mma(const bf16x8_t (&a)[2], const bf16x8_t (&b)[2], f32x16& c) {
        const half8_t a0 = __builtin_bit_cast(half8_t, a[0]), a1 = __builtin_bit_cast(half8_t, a[1]);
        const half8_t b0 = __builtin_bit_cast(half8_t, b[0]), b1 = __builtin_bit_cast(half8_t, b[1]);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
    }
};

// "h1": ONE scaled fp16 plane, one MFMA per product block -- the mixed-precision mode (fp16 operands with an 11-bit
// significand, fp32 accumulation; BASELINE config 5 asks for 16-bit compute).  Same power-of-two scaling as h3, so the fp16
// exponent range is never the limit.  NOT fp32-grade: per-product relative error ~2^-12 (FAVAE_CONV_MODE=h1 / favae_set_conv_mode(1)).
template <> struct Scheme<1> {
    static constexpr int NPL = 1;
    static constexpr bool SCALED = true;
    static constexpr int ROWB = 48;        // 32 B + 16 B pad (12-bank row stride: 16 consecutive rows hit 16 distinct bank quads)
    static constexpr int WREC = 8;         // {plane0[4 fp16]}
    static __device__ __forceinline__ void split4(const float4 v, float S, uint2 (&p)[1]) {
        const half2_t h01 = {(_Float16)(v.x * S), (_Float16)(v.y * S)}, h23 = {(_Float16)(v.z * S), (_Float16)(v.w * S)};
        p[0] = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    }
    static __device__ __forceinline__ void mma(const bf16x8_t (&a)[1], const bf16x8_t (&b)[1], f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a[0]), __builtin_bit_cast(half8_t, b[0]), c, 0, 0, 0);
    }
};

// "b1": ONE bf16 plane (round to nearest even), one MFMA per product block -- the bf16 mixed-precision mode BASELINE configs[4] names
// (what `accelerate --mixed_precision bf16` gives the reference's convs: bf16 operands, fp32 accumulation).  bf16 carries the fp32
// exponent: no operand ranges, no scaling.  8-bit significand: per-product relative error ~2^-9 (h1's fp16 plane: ~2^-12).
// Scheme id 4 (FAVAE_CONV_MODE=b1 / favae_set_conv_mode(4)); one plane: NPL = 1.
template <> struct Scheme<4> {
    static constexpr int NPL = 1;
    static constexpr bool SCALED = false;
    static constexpr int ROWB = 48;        // as Scheme<1>
    static constexpr int WREC = 8;         // {plane0[4 bf16]}
    static __device__ __forceinline__ unsigned rne_hi(float f) {             // fp32 -> bf16 bits in the upper half, round to nearest even
        const unsigned u = __float_as_uint(f);
        return (u & 0x7F800000u) == 0x7F800000u ? u : u + 0x7FFFu + ((u >> 16) & 1u);   // inf / nan pass through
    }
    static __device__ __forceinline__ void split4(const float4 v, float, uint2 (&p)[1]) {
        const unsigned a = rne_hi(v.x), b = rne_hi(v.y), c = rne_hi(v.z), d = rne_hi(v.w);
        p[0] = make_uint2(__builtin_amdgcn_perm(b, a, 0x07060302u), __builtin_amdgcn_perm(d, c, 0x07060302u));
    }
    static __device__ __forceinline__ void mma(const bf16x8_t (&a)[1], const bf16x8_t (&b)[1], f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    }
};

template <int NP>
__device__ __forceinline__ void store_planes(unsigned char* d, int plane_stride, const uint2 (&p)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<uint2*>(d + i * plane_stride) = p[i];
}

// ---- transposing fragment reads of the weight-gradient kernels --------------------------------------------------------
constexpr int RSB = 320;                   // bytes per pixel row of a plane (128 ch x 2 B + 64 B pad: conflict-free tr reads)
constexpr int PLB = 16 * RSB;              // bytes per plane (16 pixels)

// PITCH = bytes per pixel row of the plane the fragment is read from
template <int PITCH>
__device__ __forceinline__ bf16x8_t tr_frag_p(const unsigned char* p) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 4 * PITCH));
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* p) { return tr_frag_p<RSB>(p); }
}  // namespace sp
