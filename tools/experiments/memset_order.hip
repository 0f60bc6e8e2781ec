// Is hipMemsetAsync(p, 0, 4, s) followed by a kernel that atomicMax'es into p (same stream) always ordered -- also while another process
// or another stream keeps the GPU busy?  (tools/race_probe.py: spectra / gradients of the training step differ between repetitions only
// when two processes share the GPU; every atomicMax target of the library was zeroed by a 4-byte hipMemsetAsync.)
// Iteration i: memset, 256 workgroups atomicMax values <= V_i = 1000 + i, a 1-thread kernel copies the result to out[i].  Expected V_i.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void amax(unsigned* m, unsigned v) {
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    atomicMax(m, v - (g % 7u));
}
__global__ void zero_k(unsigned* m) { *m = 0; }
__global__ void copy1(const unsigned* m, unsigned* out) { *out = *m; }
__global__ void busy(float* x, int n, int it) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = x[i % n];
    for (int k = 0; k < it; ++k) v = v * 1.0001f + 0.5f;
    x[i % n] = v;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;       // 0: hipMemsetAsync, 1: own zeroing kernel
    const int bg = argc > 3 ? atoi(argv[3]) : 1;         // background kernels on a second stream
    unsigned *m, *out; float* x;
    hipMalloc(&m, 4096); hipMalloc(&out, iters * 4); hipMalloc(&x, 1 << 24);
    hipMemset(out, 0xff, iters * 4); hipMemset(x, 0, 1 << 24);
    hipStream_t s2; hipStreamCreate(&s2);
    hipStream_t s = 0;                                   // the legacy default stream, as torch's current stream on ROCm
    for (int i = 0; i < iters; ++i) {
        if (bg && (i % 8) == 0) hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, s2, x, 1 << 22, 2000);
        if (mode == 0) hipMemsetAsync(m, 0, 4, s); else hipLaunchKernelGGL(zero_k, dim3(1), dim3(1), 0, s, m);
        hipLaunchKernelGGL(amax, dim3(256), dim3(256), 0, s, m, 1000u + (unsigned)i);
        hipLaunchKernelGGL(copy1, dim3(1), dim3(1), 0, s, m, out + i);
    }
    hipDeviceSynchronize();
    std::vector<unsigned> h(iters);
    hipMemcpy(h.data(), out, iters * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < iters; ++i)
        if (h[i] != 1000u + i) { if (bad < 10) printf("  iteration %d: got %u, expected %u\n", i, h[i], 1000u + i); ++bad; }
    printf("mode %s, background %d: %d of %d iterations wrong\n", mode ? "zero kernel" : "hipMemsetAsync", bg, bad, iters);
    return 0;
}
