// How many single-issue vector instructions does one SIMD hide in the shadow of a v_mfma_f32_32x32x16_f16 (32 pipe cycles)?
// round 4/5 rejected every interleaved variant of the Winograd kernel on a 12-fillers-per-MFMA test
// (mfma_valu_roles.hip, removed in round 6: git history), which cannot tell "times add" (32 + c*NV) from "up to 5 hidden, the rest exposed".  This sweeps NV.
//
// One workgroup per CU (150 KB of LDS), 256 threads (ONE wave per SIMD) or 512 threads (TWO waves per SIMD).  Every SIMD executes
// the same totals in every arrangement: MT MFMAs on four independent accumulators and NV*MT fillers on independent registers.
//   arrangement I1: one wave per SIMD, the stream  MFMA, NV fillers, MFMA, NV fillers ...
//   arrangement I2: two waves per SIMD, each runs that stream for MT/2 MFMAs
//   arrangement S2: two waves per SIMD, wave A runs the MT MFMAs back to back, wave B the NV*MT fillers (specialised roles)
// Filler kinds: v_fma_f32 | the operand split of conv_split.h as it compiles (v_cvt_f16_f32, v_cvt_f32_f16, v_sub_f32, v_cvt_f16_f32,
// v_pack_b32_f16 pattern approximated by cvt/sub pairs) | ds_write_b64 | ds_read_b64.
// Cycles are s_memtime ticks (= shader cycles) of the longest wave of SIMD 0..3 of workgroup 0, divided by MT; wall time beside it.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/sweep tools/experiments/mfma_valu_sweep.hip ; run: /tmp/sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(v, c1, c2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2))
#define CVTH(h, v) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(v))
#define CVTF(f, h) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f) : "v"(h))
#define SUB(v, a, b) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v) : "v"(a), "v"(b))
#define DSW(addr, lo, hi) asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(lo) : "memory")
#define DSR(dst, addr) asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(addr) : "memory")

constexpr int NM = 8;   // MFMAs per loop iteration
enum { K_FMA = 0, K_SPLIT = 1, K_DSW = 2, K_DSR = 3, K_VMEM = 4 };
__device__ const float4* g_stream;          // K_VMEM: 2 MB that stay in the L2; every load a fresh 1 KB per wave (no L1 reuse)

template <int KIND, int NV> struct Fill {
    float v[NV > 0 ? NV : 1];
    uint32_t h[2];
    uint32_t addr;
    float c1, c2;
    double wr;
    float4 ld[4];
    unsigned goff;
    __device__ __forceinline__ void init(const float* out) {
        goff = (blockIdx.x * 8 + (threadIdx.x >> 6)) * 4096 + (threadIdx.x & 63);      // float4 index: a wave's own 64 KB window
        for (int e = 0; e < 4; ++e) ld[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = 0; e < (NV > 0 ? NV : 1); ++e) v[e] = threadIdx.x + e;
        c1 = out[0] + 1.0f; c2 = out[1];
        addr = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 1024;   // conflict-free 64-bit accesses, a private 512 B per wave
        wr = 0.0; h[0] = h[1] = 0;
    }
    // NV single-issue instructions on independent registers
    __device__ __forceinline__ void run() {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            if (KIND == K_FMA) FMA(v[e], c1, c2);
            if (KIND == K_SPLIT) {     // rotate through the split's instruction kinds
                if ((e & 3) == 0) CVTH(h[0], v[e]);
                if ((e & 3) == 1) CVTF(v[e], h[0]);
                if ((e & 3) == 2) SUB(v[e], v[e], c1);
                if ((e & 3) == 3) CVTH(h[1], v[e]);
            }
            if (KIND == K_DSW) DSW(addr, wr, wr);
            if (KIND == K_DSR) DSR(wr, addr);
            if (KIND == K_VMEM) {                 // 1 KB per wave-instruction, four in flight per wave, addresses walk the window
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[e & 3]) : "v"(g_stream + ((goff + 64 * (cnt++ & 63)) & 131071)) : "memory");
            }
        }
    }
    unsigned cnt = 0;
    __device__ __forceinline__ float fold() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        float s = (float)wr + h[0] + h[1] + ld[0].x + ld[1].y + ld[2].z + ld[3].w;
        for (int e = 0; e < (NV > 0 ? NV : 1); ++e) s += v[e];
        return s;
    }
};

// ARR: 0 = interleaved stream in every wave; 1 = specialised (waves 0-3 MFMA only, waves 4-7 fillers only; needs 512 threads)
// NACC: accumulators the MFMAs rotate over = distance between two MFMAs on the same accumulator
template <int KIND, int NV, int ARR, int NACC = 4>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int mt_per_wave) {
    extern __shared__ float lds[];
    const int wid = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); }
    Fill<KIND, NV> f; f.init(out);
    const int iters = mt_per_wave / NM;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (ARR == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) { MFMA(acc[m & (NACC - 1)], a, b); f.run(); }
        }
    } else if (wid < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) MFMA(acc[m & (NACC - 1)], a, b);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) f.run();
        }
    }
    float s = f.fold();
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    asm volatile("s_nop 0" :: "v"(s));
    const long long t1 = __builtin_readcyclecounter();
    if (s == 1234.5f) out[4] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wid] = t1 - t0;
}

struct Res { float ms; double cyc; };
template <int KIND, int NV, int ARR, int NACC = 4> Res run(float* d, long long* dc, int threads, int mt_per_wave) {
    const size_t lds = 150 * 1024;
    hipFuncSetAttribute((const void*)k<KIND, NV, ARR, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<KIND, NV, ARR, NACC><<<256, threads, lds>>>(d, dc, 64);
    hipEventRecord(s);
    k<KIND, NV, ARR, NACC><<<256, threads, lds>>>(d, dc, mt_per_wave);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    long long h[8]; hipMemcpy(h, dc, sizeof h, hipMemcpyDeviceToHost);
    long long mx = 0; for (int w = 0; w < threads / 64; ++w) if (h[w] > mx) mx = h[w];
    return {ms, (double)mx};
}

template <int KIND, int NV> void row(float* d, long long* dc, const char* kind) {
    const int MT = 160000;            // MFMAs per SIMD in every arrangement
    const Res i1 = run<KIND, NV, 0>(d, dc, 256, MT);
    const Res i2 = run<KIND, NV, 0>(d, dc, 512, MT / 2);
    const Res s2 = run<KIND, NV, 1>(d, dc, 512, MT);
    printf("%-6s NV=%2d | I1 %7.2f cyc/MFMA %6.3f ms | I2 %7.2f cyc/MFMA %6.3f ms | S2 %7.2f cyc/MFMA %6.3f ms\n", kind, NV,
           i1.cyc / MT, i1.ms, i2.cyc / MT, i2.ms, s2.cyc / MT, s2.ms);
}
template <int KIND> void kind_rows(float* d, long long* dc, const char* kind) {
    row<KIND, 0>(d, dc, kind); row<KIND, 1>(d, dc, kind); row<KIND, 2>(d, dc, kind); row<KIND, 3>(d, dc, kind);
    row<KIND, 4>(d, dc, kind); row<KIND, 5>(d, dc, kind); row<KIND, 6>(d, dc, kind); row<KIND, 8>(d, dc, kind);
    row<KIND, 10>(d, dc, kind); row<KIND, 12>(d, dc, kind); row<KIND, 16>(d, dc, kind);
}
template <int NACC> void dep_rows(float* d, long long* dc) {
    const int MT = 160000;
    const Res a = run<K_FMA, 0, 0, NACC>(d, dc, 256, MT), b = run<K_FMA, 3, 0, NACC>(d, dc, 256, MT), c = run<K_FMA, 0, 1, NACC>(d, dc, 512, MT),
              e = run<K_FMA, 4, 1, NACC>(d, dc, 512, MT), f = run<K_FMA, 0, 0, NACC>(d, dc, 512, MT / 2);
    printf("MFMAs rotate over %d accumulator(s): I1 NV=0 %6.2f  NV=3 %6.2f | S2 NV=0 %6.2f  NV=4 %6.2f | I2 NV=0 %6.2f cyc/MFMA\n", NACC,
           a.cyc / MT, b.cyc / MT, c.cyc / MT, e.cyc / MT, f.cyc / MT);
}
int main(int argc, char** argv) {
    float* d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
    long long* dc; hipMalloc(&dc, 64); hipMemset(dc, 0, 64);
    printf("# distance between MFMAs on the same accumulator (the Winograd kernel's M phase alternates TWO per position)\n");
    dep_rows<1>(d, dc); dep_rows<2>(d, dc); dep_rows<4>(d, dc);
    {   // weight-fragment streaming: 1 KB loads from the L2 beside the MFMAs (the Winograd kernel's wide tiling asks for 0.67 per MFMA and wave)
        float4* st; hipMalloc(&st, 131072 * sizeof(float4)); hipMemset(st, 0, 131072 * sizeof(float4));
        hipMemcpyToSymbol(HIP_SYMBOL(g_stream), &st, sizeof(st));
        printf("# L2 -> register streaming beside MFMAs: NV x 1 KB global_load_dwordx4 per MFMA slot; B/clk/CU = 4 SIMDs x NV x 1024 / (cyc per MFMA)\n");
        row<K_VMEM, 0>(d, dc, "vmem"); row<K_VMEM, 1>(d, dc, "vmem"); row<K_VMEM, 2>(d, dc, "vmem"); row<K_VMEM, 3>(d, dc, "vmem"); row<K_VMEM, 4>(d, dc, "vmem");
    }
    if (argc > 1) return 0;
    printf("# cycles (s_memtime ticks of the longest wave of workgroup 0) per MFMA slot of a SIMD; every arrangement runs the same totals per SIMD\n");
    printf("# I1 = one wave/SIMD interleaved; I2 = two waves/SIMD, both interleaved; S2 = two waves/SIMD, one MFMA-only + one filler-only\n");
    kind_rows<K_FMA>(d, dc, "fma");
    kind_rows<K_SPLIT>(d, dc, "split");
    kind_rows<K_DSW>(d, dc, "dsw64");
    kind_rows<K_DSR>(d, dc, "dsr64");
    return 0;
}
