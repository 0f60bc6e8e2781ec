// Shared device/host helpers for libfavae_hip (gfx950 / CDNA4 only: wave = 64 lanes, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/favae_hip.h"

#define FAVAE_CHECK_LAUNCH()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return favae_prof_fail_(FAVAE_ERR_LAUNCH); \
    } while (0)

// ---- launch profiler (prof.hip): FAVAE_KLAUNCH == hipLaunchKernelGGL, plus two events on the launch stream when enabled --------
extern int favae_prof_level_;
void* favae_prof_begin_(const void* host_fn, hipStream_t s);
void favae_prof_end_(void* rec, hipStream_t s);
void favae_prof_note_(double flops, double bytes);      // algorithmic work of the NEXT launch of this thread (roofline numerator)
int favae_prof_fail_(int code);                         // error return of an entry point: drops a pending note (it must not reach a later launch)
#define FAVAE_PROF_NOTE(flops, bytes)                          \
    do {                                                       \
        if (favae_prof_level_) favae_prof_note_((double)(flops), (double)(bytes)); \
    } while (0)
#define FAVAE_KLAUNCH(kern, grid, block, shm, s, ...)                                                   \
    do {                                                                                                \
        void* pr__ = favae_prof_level_ ? favae_prof_begin_((const void*)(kern), (s)) : nullptr;         \
        hipLaunchKernelGGL(kern, grid, block, shm, s, __VA_ARGS__);                                     \
        if (pr__) favae_prof_end_(pr__, (s));                                                           \
    } while (0)

#define FAVAE_REQUIRE(cond)                                    \
    do {                                                       \
        if (!(cond)) return favae_prof_fail_(FAVAE_ERR_BAD_ARG); \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 doubles of LDS. Result valid in every thread.
__device__ __forceinline__ double block_sum_d256(double v, double* red) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Sum over aligned groups of LP = 16, 32 or 64 lanes, every lane gets its group's total: the xor butterfly v += shfl_xor(v, off),
// off = LP/2 .. 1, with the same operands in the same order (bit-identical to that loop), but the four steps inside a 16-lane row are
// DPP moves on the vector ALU instead of ds_bpermute round trips through the LDS crossbar (15 per pixel and lane in thin_out_kernel:
// what bounded it, 1.03 -> 0.77 ms): xor 8 = row_ror:8, xor 4 = row_shl:4 for the lanes with bit 2 clear (banks 0, 2) merged with
// row_shr:4 for the others (banks 1, 3), xor 2 / xor 1 = quad permutations.  Only the steps that cross a row (16, 32) stay shuffles.
template <int LP>
__device__ __forceinline__ float group_sum_xor(float acc) {
    static_assert(LP == 16 || LP == 32 || LP == 64, "group of 16, 32 or 64 lanes");
    if (LP == 64) acc += __shfl_xor(acc, 32, 64);
    if (LP >= 32) acc += __shfl_xor(acc, 16, 64);
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x128, 0xF, 0xF, true));      // row_ror:8
    {
        const int v = __builtin_bit_cast(int, acc);
        int u = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);               // row_shl:4, banks 0 and 2: lane i <- lane i + 4
        u = __builtin_amdgcn_update_dpp(u, v, 0x114, 0xF, 0xA, false);                    // row_shr:4, banks 1 and 3: lane i <- lane i - 4
        acc += __builtin_bit_cast(float, u);
    }
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x4E, 0xF, 0xF, true));       // xor 2
    acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0xB1, 0xF, 0xF, true));       // xor 1
    return acc;
}

// ---- buffer addressing: wave-uniform descriptor + per-lane 32-bit byte offset (VGPR) + scalar byte offset (SGPR); offsets at or
// beyond the descriptor's size (FAVAE_OOB) load zeros / drop the store in hardware (no exec-mask predication, no 64-bit address math)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define FAVAE_OOB 0x80000000u

template <typename R>
__device__ __forceinline__ float4 bload(R rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}
template <typename R>
__device__ __forceinline__ void bstore(R rsrc, unsigned voff, unsigned soff, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, voff, soff, 0);
#ifndef FAVAE_NO_STORE_NOP
    // A vector instruction that overwrites the first data register of a 128-bit buffer store in the very next issue slot changes what
    // the store writes (gfx950, measured: tools/experiments/apply_race.py).  The ISA manuals list this hazard (one wait state between a
    // VMEM store of more than 64 bits and a VALU write of its data registers); the compiler's hazard recogniser skips it when the store
    // takes its soffset from an SGPR -- as every store through this helper does.  The hazard only shows when the wave gets consecutive
    // issue slots, e.g. next to a kernel of another stream whose waves sit in long MFMA sequences.
    asm volatile("s_nop 0" ::: "memory");
#endif
}
__device__ __forceinline__ auto make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}

// ---- bf16 activation STORAGE (round 6; BASELINE configs[4] "bf16") ---------------------------------------------------------------
// The 16-bit operand mode b1 rounds conv operands to one bf16 plane but, through round 5, kept every activation fp32 in HBM -- where
// its big layers sit at the HBM roofline (2.1 GB per launch at 5.4 TB/s on 128 -> 128 @256^2).  With the storage type AT = bf16_t the
// kernels on the ResnetBlock chain read and write activations / activation gradients as bf16 (round to nearest even on store, exact
// widening on load) and keep every accumulation, statistic and reduction in fp32 / fp64 as before.  Thread -> channel mapping, LDS
// layouts and arithmetic are those of the fp32 instantiation (AT = float, the default: bit-identical code): a thread's four channels
// are one 8-byte instead of one 16-byte access.  act_off<AT>(e) = byte offset of element e.
typedef unsigned short bf16_t;
template <typename AT> struct ActT { static constexpr unsigned B = 4; };
template <> struct ActT<bf16_t> { static constexpr unsigned B = 2; };
__device__ __forceinline__ float bf16_up(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ unsigned bf16_pack_rne(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ unsigned short bf16_rne(float v) { return (unsigned short)(bf16_pack_rne(v, 0.f) & 0xffffu); }
// four consecutive channels at byte offset voff + soff
template <typename AT, typename R>
__device__ __forceinline__ float4 act_load4(R rsrc, unsigned voff, unsigned soff) {
    if constexpr (sizeof(AT) == 4) {
        return bload(rsrc, voff, soff);
    } else {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t u = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
        return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                           __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
    }
}
template <typename AT, typename R>
__device__ __forceinline__ void act_store4(R rsrc, unsigned voff, unsigned soff, float4 v) {
    if constexpr (sizeof(AT) == 4) {
        bstore(rsrc, voff, soff, v);
    } else {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t u = {bf16_pack_rne(v.x, v.y), bf16_pack_rne(v.z, v.w)};
        __builtin_amdgcn_raw_buffer_store_b64(u, rsrc, voff, soff, 0);
    }
}
// one element through a buffer descriptor (epilogues: a lane owns one channel of a pixel)
template <typename AT, typename R>
__device__ __forceinline__ float act_load1(R rsrc, unsigned voff, unsigned soff) {
    if constexpr (sizeof(AT) == 4) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    else return bf16_up((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, soff, 0));
}
template <typename AT, typename R>
__device__ __forceinline__ void act_store1(R rsrc, unsigned voff, unsigned soff, float v) {
    if constexpr (sizeof(AT) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b16((short)bf16_rne(v), rsrc, voff, soff, 0);
}
// plain pointers (epilogue of the direct kernels)
template <typename AT> __device__ __forceinline__ float act_get(const void* p, size_t i) {
    if constexpr (sizeof(AT) == 4) return reinterpret_cast<const float*>(p)[i];
    else return bf16_up(reinterpret_cast<const unsigned short*>(p)[i]);
}
// four consecutive channels from a plain pointer (element index i, a multiple of 4: 16-byte / 8-byte aligned)
template <typename AT> __device__ __forceinline__ float4 act_get4(const void* p, size_t i) {
    if constexpr (sizeof(AT) == 4) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
    else {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + i);
        return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                           __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
    }
}
template <typename AT> __device__ __forceinline__ void act_put(void* p, size_t i, float v) {
    if constexpr (sizeof(AT) == 4) reinterpret_cast<float*>(p)[i] = v;
    else reinterpret_cast<unsigned short*>(p)[i] = bf16_rne(v);
}

// Zero arena (favae_set_zero_arena): a device range the caller guarantees to be all zero when an entry point receives a pointer into it
// as a reduction target (max |x| scalars: atomicMax on the bit pattern needs a zeroed start).  Such targets are not memset again -- one
// hipMemsetAsync per training step over the arena instead of one 4-byte memset launch per conv call (117-156 per step, 5 us + a queue
// gap each).  Pointers outside the arena are zeroed here as before.
inline const char* g_zero_lo = nullptr;
inline const char* g_zero_hi = nullptr;
inline bool favae_prezeroed(const void* p, size_t bytes) {
    const char* c = (const char*)p;
    return g_zero_lo && c >= g_zero_lo && c + bytes <= g_zero_hi;
}
inline hipError_t favae_zero_target(void* p, size_t bytes, hipStream_t s) {
    return favae_prezeroed(p, bytes) ? hipSuccess : hipMemsetAsync(p, 0, bytes, s);
}

__device__ __forceinline__ float silu_f(float y) { return y / (1.0f + __expf(-y)); }

// d act(y) / dy of the fused input activations (FAVAE_ACT_*: 1 SiLU, 2 LeakyReLU(0.2), 3 ReLU)
__device__ __forceinline__ float favae_act_grad(float y, int act) {
    if (act == 1) {
        const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
        return s * (1.0f + y * (1.0f - s));
    }
    if (act == 2) return y > 0.f ? 1.0f : 0.2f;
    if (act == 3) return y > 0.f ? 1.0f : 0.0f;
    return 1.0f;
}

// XCD-aware bijective remap of a 1-D block id: each of the 8 XCDs (private L2) gets a contiguous chunk of logical
// tile ids so that neighbouring tiles (shared halo rows / shared A panels) hit the same L2.  (guide T1, bijective form)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    const int q = nwg / nx, r = nwg % nx;
    const int xcd = bid % nx, k = bid / nx;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
