// Do an MFMA-only wave and a vector-ALU-only wave that share a SIMD overlap?  (mfma_valu_overlap.hip asked the question for ONE instruction
// stream that interleaves both: there the times add.)  512-thread workgroups, one per CU (150 KB of LDS): waves 0-3 land on the four SIMDs,
// waves 4-7 on the same four again.  Roles by wave id:
//   mode 0: waves 0-3 MFMA loop, waves 4-7 FMA loop          (specialised pair on every SIMD)
//   mode 1: waves 0-3 MFMA loop, waves 4-7 exit              (matrix alone)
//   mode 2: waves 0-3 exit,      waves 4-7 FMA loop          (vector alone)
//   mode 3: all eight waves: MFMA + FMA interleaved 1:NV, half the iterations each (same totals per SIMD as mode 0)
//   mode 4: waves 0-3 MFMA loop, waves 4-7 LDS-read loop (ds_read_b128) -- the known-good overlap, as a control
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/roles tools/experiments/mfma_valu_roles.hip ; run: /tmp/roles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// exact instruction streams: volatile asm keeps program order, the compiler neither packs the FMAs nor moves them across the MFMAs
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(v, c1, c2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2))

constexpr int NM = 8;        // MFMAs per iteration
constexpr int NV = 12;       // FMAs per MFMA slot (NM * NV per iteration): 12 x 2..4 clk against 32 clk of matrix pipe

__device__ __forceinline__ void mfma_loop(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) MFMA(acc[m & 3], a, b);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 1234.5f) out[2] = s;
}
__device__ __forceinline__ void fma_loop(float* out, int iters) {
    float v[NV];
    for (int e = 0; e < NV; ++e) v[e] = threadIdx.x + e;
    const float c1 = out[0] + 1.0f, c2 = out[1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
#pragma unroll
            for (int e = 0; e < NV; ++e) FMA(v[e], c1, c2);
        }
    }
    float s = 0.f;
    for (int e = 0; e < NV; ++e) s += v[e];
    if (s == 1234.5f) out[3] = s;
}
__device__ __forceinline__ void both_loop(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); }
    float v[NV];
    for (int e = 0; e < NV; ++e) v[e] = threadIdx.x + e;
    const float c1 = out[0] + 1.0f, c2 = out[1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            MFMA(acc[m & 3], a, b);
#pragma unroll
            for (int e = 0; e < NV; ++e) FMA(v[e], c1, c2);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int e = 0; e < NV; ++e) s += v[e];
    if (s == 1234.5f) out[4] = s;
}
__device__ __forceinline__ void lds_loop(float* out, int iters, const float* lds) {
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = reinterpret_cast<const float4*>(lds) + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const float4 t = p[m * 64];
            s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
        }
    }
    if (s4.x + s4.y + s4.z + s4.w == 1234.5f) out[5] = s4.x;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ float lds[];
    const int wid = threadIdx.x >> 6;
    if (MODE == 3) { both_loop(out, iters / 2); return; }
    if (wid < 4) {
        if (MODE == 0 || MODE == 1 || MODE == 4) mfma_loop(out, iters);
    } else {
        if (MODE == 0 || MODE == 2) fma_loop(out, iters);
        if (MODE == 4) lds_loop(out, iters, lds);
    }
}

template <int MODE> float run(float* d, int iters) {
    const size_t lds = 150 * 1024;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<MODE><<<256, 512, lds>>>(d, 10);
    hipEventRecord(s);
    k<MODE><<<256, 512, lds>>>(d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}
int main() {
    float* d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
    const int iters = 20000;
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters);
    const double nm = (double)NM * iters, nv = (double)NM * NV * iters;
    printf("specialised pair %.3f ms | MFMA waves alone %.3f ms (%.1f clk/MFMA at 2.4 GHz) | FMA waves alone %.3f ms (%.2f clk/FMA) | sum %.3f max %.3f\n",
           t0, t1, t1 * 1e-3 * 2.4e9 / nm, t2, t2 * 1e-3 * 2.4e9 / nv, t1 + t2, t1 > t2 ? t1 : t2);
    printf("interleaved in every wave (same totals per SIMD) %.3f ms | MFMA waves + LDS-read waves %.3f ms\n", t3, t4);
    return 0;
}
