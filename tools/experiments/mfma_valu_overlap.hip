// Does one wave per SIMD overlap its own MFMAs with its own vector-ALU instructions?  Three loops, one wave per SIMD (150 KB of LDS per
// workgroup keeps a second workgroup off the CU): MODE 0 = 12 MFMA + 84 v_fma interleaved 1:7, MODE 1 = the MFMAs alone, MODE 2 = the FMAs alone.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ov tools/experiments/mfma_valu_overlap.hip ; run: /tmp/ov [waves per SIMD: 1|2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ float lds[];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); }
    float v[14];
    for (int e = 0; e < 14; ++e) v[e] = threadIdx.x + e;
    const float c1 = out[0], c2 = out[1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (MODE != 2) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            if (MODE != 1) {
#pragma unroll
                for (int e = 0; e < 7; ++e) v[(m & 1) * 7 + e] = __builtin_fmaf(v[(m & 1) * 7 + e], c1, c2);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int e = 0; e < 14; ++e) s += v[e];
    if (s == 1234.5f) out[2] = s + lds[threadIdx.x];
}

template <int MODE> float run(float* d, int iters, int blocks, size_t lds) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<MODE><<<blocks, 256, lds>>>(d, 10);
    hipEventRecord(s);
    k<MODE><<<blocks, 256, lds>>>(d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms;
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    float* d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
    const int iters = 20000, blocks = 256 * wps;
    const size_t lds = wps == 1 ? 150 * 1024 : 70 * 1024;
    const float t0 = run<0>(d, iters, blocks, lds), t1 = run<1>(d, iters, blocks, lds), t2 = run<2>(d, iters, blocks, lds);
    const double mf = 12.0 * iters;     // MFMAs per wave
    printf("waves/SIMD %d: interleaved %.3f ms, MFMA alone %.3f ms (%.1f clk/MFMA at 2.4 GHz), FMA alone %.3f ms (%.2f clk/FMA); sum %.3f\n", wps, t0, t1,
           t1 * 1e-3 * 2.4e9 / mf, t2, t2 * 1e-3 * 2.4e9 / (84.0 * iters), t1 + t2);
    return 0;
}
