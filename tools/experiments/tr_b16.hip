// probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 value = its own index; every lane reads with a chosen address
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = l * 8;                          // lane l -> bytes [8l, 8l+8)
    else if (mode == 1) addr = (l & 15) * 32 + (l >> 4) * 8;   // 16 rows of 32 B (16 u16), lane group g takes 8-byte column block g
    else addr = (l & 15) * 256 + (l >> 4) * 8;            // rows of 256 B
    addr += (unsigned)(size_t)lds;                        // LDS base is 0 for the only shared array, keep generic
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("l%2d: %5d %5d %5d %5d   ", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3) printf("\n"); }
    }
    return 0;
}
