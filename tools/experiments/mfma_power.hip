// Power wall probe: how fast can gfx950 issue v_mfma_f32_32x32x16_f16 from registers (no LDS, no global traffic in the loop) on
// (a) random fp16 operands and (b) all-zero operands?  Prints TFLOP/s of fp16 MFMA work for both.  The conv kernels of this
// repository sustain ~1.0-1.05 PFLOP/s of MFMA work on random data.
// build: hipcc -O3 --offload-arch=gfx950 tools/experiments/mfma_power.hip -o /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const half8* __restrict__ ab, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    half8 a[2], b[2];
    a[0] = ab[(size_t)t * 4 + 0]; a[1] = ab[(size_t)t * 4 + 1];
    b[0] = ab[(size_t)t * 4 + 2]; b[1] = ab[(size_t)t * 4 + 3];
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
}

int main(int argc, char** argv) {
    const int blocks = 256 * 8, threads = 256;
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;          // 4000: 3 ms per launch; 400000: 0.3 s per launch (sustained clocks)      // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    const size_t n = (size_t)blocks * threads;
    std::vector<_Float16> h(n * 32);
    half8* d_ab; float* d_out;
    hipMalloc(&d_ab, n * 32 * sizeof(_Float16));
    hipMalloc(&d_out, n * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {       // 0 random, 1 zeros, 2 random again
        srand(1);
        for (auto& v : h) v = mode == 1 ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
        hipMemcpy(d_ab, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        for (int nacc = 4; nacc <= 4; ++nacc) {
            hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(threads), 0, 0, d_ab, d_out, iters);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(threads), 0, 0, d_ab, d_out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 5.0 * (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 32768.0;
            printf("%s operands: %.1f TFLOP/s of fp16 MFMA work (%.2f ms)\n", mode == 1 ? "zero  " : "random", flops / (ms * 1e-3) * 1e-12, ms / 5);
        }
    }
    return 0;
}
