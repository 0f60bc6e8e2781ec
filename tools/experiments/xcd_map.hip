// Which XCD runs workgroup b of a 1-D grid?  (conv_wino.h's xcd_remap assumes b % 8; one whole-CU workgroup per CU as in the Winograd
// kernel: 161 KB of dynamic LDS, 512 threads.)  Prints the share of workgroups whose XCC_ID equals b % 8 and the first 32 ids.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void probe(int* xcc, long long* t0, int spin) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xf);
        t0[blockIdx.x] = (long long)wall_clock64();
    }
    lds[threadIdx.x] = (char)threadIdx.x;
    long long s = clock64();
    while (clock64() - s < spin) {}
    __syncthreads();
}
int main() {
    const int n = 4096;
    int* d; long long* t;
    hipMalloc(&d, n * 4); hipMalloc(&t, n * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 161088);
    for (int spin : {20000, 200000}) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(512), 161088, 0, d, t, spin);
        hipDeviceSynchronize();
        std::vector<int> h(n); std::vector<long long> ht(n);
        hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, n * 8, hipMemcpyDeviceToHost);
        int ok = 0; for (int i = 0; i < n; ++i) ok += (h[i] == i % 8);
        printf("spin %d: %d of %d workgroups on XCD b %% 8; first 32:", spin, ok, n);
        for (int i = 0; i < 32; ++i) printf(" %d", h[i]);
        // are blocks b and b+8 (a pair of channel tiles in xcd_remap order) started close in time?
        double dsum = 0; long long mx = 0;
        for (int i = 0; i + 8 < n; i += 16) { long long dd = llabs(ht[i + 8] - ht[i]); dsum += dd; if (dd > mx) mx = dd; }
        printf("\n   start-time distance of b and b+8: mean %.0f ticks (100 MHz), max %lld\n", dsum / (n / 16), mx);
    }
    return 0;
}
