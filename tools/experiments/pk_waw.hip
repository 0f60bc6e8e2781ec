// Is a packed-fp32 result half that is overwritten right afterwards by an unpacked vector instruction safe when the wave shares its SIMD
// with waves that sit in MFMA sequences?  (Hypothesis behind the FFT finding of round 4)  Victim: one wave per SIMD, explicit
// registers:   v_pk_fma_f32 v[20:21], v[22:23], v[24:25], v[26:27]   ;   <GAP unrelated instructions>   ;   v_sub_f32 v20, v28, v29
// then v20 must be v28 - v29 and v21 the packed result's high half.  Aggressor: an MFMA loop with a 208-register allocation, two waves per
// SIMD, on a second stream.  Prints mismatches per variant (GAP = 0, 1, 2; overwrite of the low / high half).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void aggressor(const half8* __restrict__ ab, float* __restrict__ out, int iters) {
    asm volatile("v_mov_b32 v207, 0" ::: "v207");                 // allocation of 208: room for one 96-register wave beside two of these
    const int t = blockIdx.x * 256 + threadIdx.x;
    half8 a = ab[(size_t)t * 2], b = ab[(size_t)t * 2 + 1];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
}

template <int GAP, int HALF>
__global__ __launch_bounds__(64) void victim(const float* __restrict__ in, unsigned* __restrict__ bad, int iters) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    float a0 = in[t * 8 + 0], a1 = in[t * 8 + 1], b0 = in[t * 8 + 2], b1 = in[t * 8 + 3], c0 = in[t * 8 + 4], c1 = in[t * 8 + 5];
    float x = in[t * 8 + 6], y = in[t * 8 + 7];
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        float lo, hi;
        asm volatile(
            "v_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\tv_mov_b32 v24, %4\n\tv_mov_b32 v25, %5\n\tv_mov_b32 v26, %6\n\tv_mov_b32 v27, %7\n\t"
            "v_mov_b32 v28, %8\n\tv_mov_b32 v29, %9\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"
            "s_nop 4\n\t"
            "v_pk_fma_f32 v[20:21], v[22:23], v[24:25], v[26:27]\n\t"
            ".if %10 >= 1\n\tv_add_f32 v30, v28, v29\n\t.endif\n\t"
            ".if %10 >= 2\n\tv_add_f32 v31, v28, v29\n\t.endif\n\t"
            ".if %11 == 0\n\tv_sub_f32 v20, v28, v29\n\t.else\n\tv_sub_f32 v21, v28, v29\n\t.endif\n\t"
            "s_nop 4\n\t"
            "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21\n\t"
            : "=v"(lo), "=v"(hi)
            : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(x), "v"(y), "n"(GAP), "n"(HALF)
            : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        const float want_lo = HALF == 0 ? x - y : fmaf(a0, b0, c0);
        const float want_hi = HALF == 0 ? fmaf(a1, b1, c1) : x - y;
        if (lo != want_lo || hi != want_hi) ++nbad;
        x += 1.0f;                                            // keep the loop from being hoisted
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int GAP, int HALF>
void run(const float* in, unsigned* bad, const half8* ab, float* aout, hipStream_t s1, hipStream_t s2, const char* what) {
    hipMemsetAsync(bad, 0, 4, s1);
    hipStreamSynchronize(s1);
    for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(aggressor, dim3(512), dim3(256), 0, s2, ab, aout, 20000);            // 2 blocks of 4 waves per CU: 2 waves per SIMD
        hipLaunchKernelGGL((victim<GAP, HALF>), dim3(1024), dim3(64), 0, s1, in, bad, 20000);   // 4 single-wave blocks per CU
    }
    hipDeviceSynchronize();
    unsigned h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("%-48s mismatches %u of %.3g checks\n", what, h, 20.0 * 1024 * 64 * 20000);
}

int main() {
    const int nv = 1024 * 64;
    std::vector<float> h(nv * 8);
    srand(3);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 4.f - 2.f;
    float* in; unsigned* bad; half8* ab; float* aout;
    hipMalloc(&in, h.size() * 4); hipMalloc(&bad, 4); hipMalloc(&ab, 512 * 256 * 2 * sizeof(half8)); hipMalloc(&aout, 512 * 256 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(ab, 0x3c, 512 * 256 * 2 * sizeof(half8));
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    run<0, 0>(in, bad, ab, aout, s1, s2, "v_pk_fma ; v_sub on the LOW half (gap 0)");
    run<1, 0>(in, bad, ab, aout, s1, s2, "v_pk_fma ; 1 instr ; v_sub on the LOW half");
    run<2, 0>(in, bad, ab, aout, s1, s2, "v_pk_fma ; 2 instr ; v_sub on the LOW half");
    run<0, 1>(in, bad, ab, aout, s1, s2, "v_pk_fma ; v_sub on the HIGH half (gap 0)");
    run<1, 1>(in, bad, ab, aout, s1, s2, "v_pk_fma ; 1 instr ; v_sub on the HIGH half");
    return 0;
}
