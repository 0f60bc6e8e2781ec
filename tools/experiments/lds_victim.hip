// Minimal victim for the two-process irreproducibility seen in fft_lines_kernel (tools/experiments/ffl_race2.py): a kernel with the same
// skeleton -- float2 lines global -> LDS (bit-reversed rows), barrier, log2(L) in-place butterfly passes (sums / differences only, no
// twiddles) with a barrier each, LDS -> global -- launched again and again on one input; every output is compared on the device with the
// first.  Run it alone and next to another process that trains (tools/r04_victim.sh).  mode 1: no LDS at all (plain copy * 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int L = 64, IC = 32, NT = 256;
__global__ __launch_bounds__(512) void victim(const float2* in, float2* out, long inner, int mode) {
    extern __shared__ __attribute__((aligned(16))) float2 data[];
    const long chunks = inner / IC;
    const long o = blockIdx.x / chunks, i0 = (blockIdx.x % chunks) * IC;
    const int ic = threadIdx.x % IC, bl = threadIdx.x / IC, BL = NT / IC;
    const size_t base = (size_t)o * L * inner + i0 + ic;
    if (mode == 1) {
        for (int l = bl; l < L; l += BL) { float2 v = in[base + (size_t)l * inner]; out[base + (size_t)l * inner] = make_float2(2 * v.x, 2 * v.y); }
        return;
    }
    for (int l = bl; l < L; l += BL) data[(int)(__brev((unsigned)l) >> 26) * IC + ic] = in[base + (size_t)l * inner];
    __syncthreads();
    for (int half = 1; half < L; half <<= 1) {
        for (int b = bl; b < L / 2; b += BL) {
            const int grp = b / half, j = b % half, p0 = grp * 2 * half + j;
            float2 x0 = data[p0 * IC + ic], x1 = data[(p0 + half) * IC + ic];
            data[p0 * IC + ic] = make_float2(x0.x + x1.x, x0.y + x1.y);
            data[(p0 + half) * IC + ic] = make_float2(x0.x - x1.x, x0.y - x1.y);
        }
        __syncthreads();
    }
    for (int l = bl; l < L; l += BL) out[base + (size_t)l * inner] = data[l * IC + ic];
}
__global__ void compare(const float2* a, const float2* b, size_t n, unsigned* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (a[i].x != b[i].x || a[i].y != b[i].y)) atomicAdd(bad, 1u);
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const long outer = 4, inner = 33 * 128;                       // the (4, 128, 64, 64) half spectrum
    const size_t n = (size_t)outer * L * inner;
    float2 *in, *ref, *out; unsigned* bad;
    hipMalloc(&in, n * 8); hipMalloc(&ref, n * 8); hipMalloc(&out, n * 8); hipMalloc(&bad, 4 * (size_t)iters);
    std::vector<float2> h(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) h[i] = make_float2((float)rand() / RAND_MAX - 0.5f, (float)rand() / RAND_MAX - 0.5f);
    hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 4 * (size_t)iters);
    const int blocks = (int)(outer * (inner / IC));
    const size_t shm = (size_t)L * IC * 8;
    hipLaunchKernelGGL(victim, dim3(blocks), dim3(NT), shm, 0, in, ref, inner, mode);
    for (int i = 0; i < iters; ++i) {
        hipLaunchKernelGGL(victim, dim3(blocks), dim3(NT), shm, 0, in, out, inner, mode);
        hipLaunchKernelGGL(compare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ref, out, n, bad + i);
    }
    hipDeviceSynchronize();
    std::vector<unsigned> hb(iters);
    hipMemcpy(hb.data(), bad, 4 * (size_t)iters, hipMemcpyDeviceToHost);
    int nb = 0; unsigned long long tot = 0;
    for (int i = 0; i < iters; ++i) if (hb[i]) { if (nb < 6) printf("  launch %d: %u elements differ\n", i, hb[i]); ++nb; tot += hb[i]; }
    printf("victim mode %d: %d of %d launches differ from the first (%llu elements)\n", mode, nb, iters, tot);
    return 0;
}
