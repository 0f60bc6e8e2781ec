// Batched fp32 GEMM on v_mfma_f32_32x32x2_f32 + row softmax: the single-head attention core of AttnBlock
// (models/codec.py:87-102: nn.MultiheadAttention(C, num_heads=1)):
//     S = (Q K^T)/sqrt(C)   -> favae_bgemm(ta=0,tb=0)      P = softmax(S)   -> favae_softmax_rows
//     O = P V               -> favae_bgemm(ta=0,tb=1)
//     backward: dV = P^T dO (ta=1,tb=1), dP = dO V^T (0,0), dS = P*(dP - rowsum(dP*P))/sqrt(C), dQ = dS K (0,1),
//               dK = dS^T Q (1,1)
// Operand convention: C[m][n] = alpha * sum_k A(m,k) * B(n,k).  t? = 0: operand stored [rows][k] (k contiguous,
// leading dimension ld = row stride); t? = 1: stored [k][rows] (rows contiguous, ld = k stride).
// Tile 128x128x16, 4 waves of 64x64, register-staged double-buffered LDS like the conv kernels.
#include "common.h"
#include "split_planes.h"

namespace {

constexpr int GBM = 128, GBN = 128, GBK = 16, GLDK = GBK + 4;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    long lda, ldb, ldc, sA, sB, sC;
    int M, N, K;
    float alpha;
    int accumulate, vecA, vecB;
};

// loads one 128 x 16 operand tile into registers (2 float4 per thread)
template <int T>
__device__ __forceinline__ void load_operand(const float* base, long ld, int rows, int K, int r0, int k0, int vec, int tid,
                                             float4 (&r)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        r[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (T == 0) {
            const int row = r0 + (tid >> 2) + 64 * j, k = k0 + (tid & 3) * 4;
            if (row < rows && k < K) {
                const float* p = base + (size_t)row * ld + k;
                if (vec) r[j] = *reinterpret_cast<const float4*>(p);
                else {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = (k + e < K) ? p[e] : 0.f;
                    r[j] = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
        } else {
            const int i = tid + 256 * j;
            const int k = k0 + (i >> 5), row = r0 + (i & 31) * 4;
            if (k < K && row < rows) {
                const float* p = base + (size_t)k * ld + row;
                if (vec) r[j] = *reinterpret_cast<const float4*>(p);
                else {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = (row + e < rows) ? p[e] : 0.f;
                    r[j] = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
        }
    }
}

template <int T>
__device__ __forceinline__ void store_operand(float* lds, int tid, const float4 (&r)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (T == 0) {
            *reinterpret_cast<float4*>(&lds[((tid >> 2) + 64 * j) * GLDK + (tid & 3) * 4]) = r[j];
        } else {
            const int i = tid + 256 * j;
            *reinterpret_cast<float4*>(&lds[(i >> 5) * GBM + (i & 31) * 4]) = r[j];
        }
    }
}

// fragment for k-group kk (8 k values): returns the 4 values this lane feeds to the 4 MFMAs of the group
template <int T>
__device__ __forceinline__ float4 read_frag(const float* lds, int row, int kk, int lane) {
    const int kb = kk * 8 + (lane >> 5) * 4;
    if (T == 0) return *reinterpret_cast<const float4*>(&lds[row * GLDK + kb]);
    return make_float4(lds[(kb + 0) * GBM + row], lds[(kb + 1) * GBM + row], lds[(kb + 2) * GBM + row],
                       lds[(kb + 3) * GBM + row]);
}

constexpr int lds_elems(int T) { return T == 0 ? GBM * GLDK : GBK * GBM; }

template <int TA, int TB>
__global__ __launch_bounds__(256) void bgemm_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (lds_elems(TA) + lds_elems(TB))];
    float* As = lds;
    float* Bs = lds + 2 * lds_elems(TA);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = (a.N + GBN - 1) / GBN;
    const int m0 = (blockIdx.x / tiles_n) * GBM, n0 = (blockIdx.x % tiles_n) * GBN;
    const int b = blockIdx.y;
    const float* A = a.A + (size_t)b * a.sA;
    const float* B = a.B + (size_t)b * a.sB;
    float* C = a.C + (size_t)b * a.sC;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[2], rb[2];
    const int T = (a.K + GBK - 1) / GBK;
    load_operand<TA>(A, a.lda, a.M, a.K, m0, 0, a.vecA, tid, ra);
    load_operand<TB>(B, a.ldb, a.N, a.K, n0, 0, a.vecB, tid, rb);
    store_operand<TA>(As, tid, ra);
    store_operand<TB>(Bs, tid, rb);
    __syncthreads();
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) {
            load_operand<TA>(A, a.lda, a.M, a.K, m0, (it + 1) * GBK, a.vecA, tid, ra);
            load_operand<TB>(B, a.ldb, a.N, a.K, n0, (it + 1) * GBK, a.vecB, tid, rb);
        }
        const float* Ab = As + cur * lds_elems(TA);
        const float* Bb = Bs + cur * lds_elems(TB);
#pragma unroll
        for (int kk = 0; kk < GBK / 8; ++kk) {
            float4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = read_frag<TA>(Ab, wm * 64 + i * 32 + (lane & 31), kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = read_frag<TB>(Bb, wn * 64 + j * 32 + (lane & 31), kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (it + 1 < T) {
            store_operand<TA>(As + (cur ^ 1) * lds_elems(TA), tid, ra);
            store_operand<TB>(Bs + (cur ^ 1) * lds_elems(TB), tid, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < a.M) {
                    float* o = C + (size_t)row * a.ldc + col;
                    float v = a.alpha * acc[i][j][r];
                    if (a.accumulate) v += *o;
                    *o = v;
                }
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same batched GEMM on the 16-bit matrix pipe with fp32-grade products (two scaled fp16 planes per operand, 3
// v_mfma_f32_32x32x16_f16 per fp32 product block: conv_split.h / DESIGN.md section 3).  128 x 128 x 16 tiles, 8 waves of 32 x 64.
// Operands are split on their way to LDS:
//   t = 0 (stored [rows][k]): plane rows of 80 B per matrix row, fragments by ds_read_b128 (as conv_fwd_sp_kernel's operands);
//   t = 1 (stored [k][rows]): planes [k][rows] exactly as they arrive, fragments by the transposing ds_read_b64_tr_b16
//   (as the weight-gradient kernels' operands).  Both give lane (row = lane & 31, group = lane >> 5) the k values 8 group .. + 7,
//   so any combination of the two layouts feeds one MFMA.
// amaxA / amaxB: device scalars >= max|A|, max|B| (the power-of-two operand scales are derived from them).
// Requirements (else the caller uses favae_bgemm): 16-byte aligned operands, ld / batch strides % 4 == 0, K % 4 == 0 for t = 0
// operands, rows % 4 == 0 for t = 1 operands.
// ---------------------------------------------------------------------------------------------------------------------------
struct GemmSpArgs {
    const float* A;
    const float* B;
    float* C;
    const float* amaxA;
    const float* amaxB;
    long lda, ldb, ldc, sA, sB, sC;
    int M, N, K;
    float alpha;
    int accumulate;
};

constexpr int SP_OPB = 10240;                      // bytes of one staged operand tile: 128 x 80 (t = 0) = 2 x 16 x 320 (t = 1)

template <int T>
__device__ __forceinline__ float4 sp_load(const float* base, long ld, int rows, int K, int r0, int k0, int tid) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (T == 0) {
        const int row = r0 + (tid >> 2), k = k0 + (tid & 3) * 4;
        if (row < rows && k < K) v = *reinterpret_cast<const float4*>(base + (size_t)row * ld + k);
    } else {
        const int k = k0 + (tid >> 5), row = r0 + (tid & 31) * 4;
        if (k < K && row < rows) v = *reinterpret_cast<const float4*>(base + (size_t)k * ld + row);
    }
    return v;
}

template <int T>
__device__ __forceinline__ void sp_store(unsigned char* lds, int tid, const float4 v, float S) {
    uint2 p[2];
    sp::Scheme<2>::split4(v, S, p);
    if (T == 0) sp::store_planes<2>(lds + (tid >> 2) * 80 + (tid & 3) * 8, 32, p);
    else sp::store_planes<2>(lds + (tid >> 5) * sp::RSB + (tid & 31) * 8, sp::PLB, p);
}

// fragment (8 k values of matrix row `row` of the tile, plane pl) of this lane
template <int T>
__device__ __forceinline__ bf16x8_t sp_frag(const unsigned char* lds, int row0, int pl, int lane) {
    if (T == 0) return *reinterpret_cast<const bf16x8_t*>(lds + (row0 + (lane & 31)) * 80 + pl * 32 + (lane >> 5) * 16);
    const int s16 = lane & 15, g = lane >> 4;
    return sp::tr_frag(lds + pl * sp::PLB + (8 * (g >> 1) + (s16 >> 2)) * sp::RSB + (row0 + 16 * (g & 1) + 4 * (s16 & 3)) * 2);
}

template <int TA, int TB>
__global__ __launch_bounds__(512) void bgemm_sp_kernel(GemmSpArgs a) {
    using S = sp::Scheme<2>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * SP_OPB];
    unsigned char* As = lds;
    unsigned char* Bs = lds + 2 * SP_OPB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;          // 4 x 2 waves of 32 x 64
    const int tiles_n = (a.N + 127) / 128;
    const int m0 = (blockIdx.x / tiles_n) * 128, n0 = (blockIdx.x % tiles_n) * 128;
    const int b = blockIdx.y;
    const float* A = a.A + (size_t)b * a.sA;
    const float* B = a.B + (size_t)b * a.sB;
    float* C = a.C + (size_t)b * a.sC;
    const float Sa = sp::pow2_scale(a.amaxA), Sb = sp::pow2_scale(a.amaxB);

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int T = (a.K + 15) / 16;
    float4 ra = sp_load<TA>(A, a.lda, a.M, a.K, m0, 0, tid);
    float4 rb = sp_load<TB>(B, a.ldb, a.N, a.K, n0, 0, tid);
    sp_store<TA>(As, tid, ra, Sa);
    sp_store<TB>(Bs, tid, rb, Sb);
    __syncthreads();
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) {
            ra = sp_load<TA>(A, a.lda, a.M, a.K, m0, (it + 1) * 16, tid);
            rb = sp_load<TB>(B, a.ldb, a.N, a.K, n0, (it + 1) * 16, tid);
        }
        bf16x8_t af[2], bf[2][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[pl] = sp_frag<TA>(As + cur * SP_OPB, wm * 32, pl, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bf[j][pl] = sp_frag<TB>(Bs + cur * SP_OPB, wn * 64 + j * 32, pl, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) S::mma(af, bf[j], acc[j]);
        if (it + 1 < T) {
            sp_store<TA>(As + (cur ^ 1) * SP_OPB, tid, ra, Sa);
            sp_store<TB>(Bs + (cur ^ 1) * SP_OPB, tid, rb, Sb);
        }
        __syncthreads();
    }
    const float un = sp::pow2_inv(Sa) * sp::pow2_inv(Sb);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= a.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < a.M) {
                float* o = C + (size_t)row * a.ldc + col;
                float v = a.alpha * (acc[j][r] * un);
                if (a.accumulate) v += *o;
                *o = v;
            }
        }
    }
}

// one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, float* p, long rows, int L) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* sr = s + row * L;
    float* pr = p + row * L;
    float mx = -INFINITY;
    for (int i = lane; i < L; i += 64) mx = fmaxf(mx, sr[i]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < L; i += 64) {
        const float e = expf(sr[i] - mx);
        pr[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < L; i += 64) pr[i] *= inv;
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* p, const float* dp, float* ds, long rows, int L,
                                                               float alpha) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pr = p + row * L;
    const float* dr = dp + row * L;
    float* o = ds + row * L;
    float dot = 0.f;
    for (int i = lane; i < L; i += 64) dot = fmaf(pr[i], dr[i], dot);
    dot = wave_sum(dot);
    for (int i = lane; i < L; i += 64) o[i] = alpha * pr[i] * (dr[i] - dot);
}


// softmax over the rows of a chunk of scores, in place, plus the row log-sum-exp (saved for the backward pass, which recomputes
// the probabilities as exp(s - lse) instead of keeping the N x L x L matrix).  Row `r` of batch element `n` of the chunk is
// query r0 + r of that element: lse[n * Ltot + r0 + r].
__global__ __launch_bounds__(256) void softmax_rows_lse_kernel(float* s, float* lse, long rows, int L, int rq, int r0, int Ltot) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* sr = s + row * L;
    float mx = -INFINITY;
    for (int i = lane; i < L; i += 64) mx = fmaxf(mx, sr[i]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < L; i += 64) {
        const float e = expf(sr[i] - mx);
        sr[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < L; i += 64) sr[i] *= inv;
    if (lane == 0) lse[(row / rq) * (long)Ltot + r0 + (row % rq)] = mx + logf(sum);
}

// backward point-wise stage of one query chunk: s holds the (scaled) scores, dp = dO V^T.  In place:
//   s  <- p  = exp(s - lse)                              (the forward probabilities, recomputed)
//   dp <- ds = alpha * p * (dp - delta),  delta = rowsum(dO * O)     (softmax backward with the 1/sqrt(C) of the scores folded in)
// and max|ds| (bit-pattern atomicMax: order-independent) for the operand scaling of the GEMMs that consume ds.
// One wave per row, rows strided over the grid; 16-byte accesses when L % 4 == 0 (VEC); ONE atomicMax per block (the round-2 version
// issued one same-address atomic per row: 8192 of them serialised at ~10 ns each and held the pass at 4 % of the HBM rate).
template <bool VEC>
__global__ __launch_bounds__(256) void attn_bwd_point_kernel(float* s, float* dp, const float* lse, const float* delta, long rows,
                                                             int L, int rq, int r0, int Ltot, float alpha, unsigned* ds_amax) {
    const int lane = threadIdx.x & 63;
    float mx = 0.f;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
        const long qi = (row / rq) * (long)Ltot + r0 + (row % rq);
        const float l = lse[qi], dl = delta[qi];
        float* sr = s + row * L;
        float* dr = dp + row * L;
        if constexpr (VEC) {
            for (int i = lane * 4; i < L; i += 256) {
                const float4 sv = *reinterpret_cast<const float4*>(sr + i), dv = *reinterpret_cast<const float4*>(dr + i);
                float4 p, ds;
                p.x = expf(sv.x - l); p.y = expf(sv.y - l); p.z = expf(sv.z - l); p.w = expf(sv.w - l);
                ds.x = alpha * p.x * (dv.x - dl); ds.y = alpha * p.y * (dv.y - dl);
                ds.z = alpha * p.z * (dv.z - dl); ds.w = alpha * p.w * (dv.w - dl);
                *reinterpret_cast<float4*>(sr + i) = p;
                *reinterpret_cast<float4*>(dr + i) = ds;
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(ds.x), fabsf(ds.y))), fmaxf(fabsf(ds.z), fabsf(ds.w)));
            }
        } else {
            for (int i = lane; i < L; i += 64) {
                const float p = expf(sr[i] - l);
                const float ds = alpha * p * (dr[i] - dl);
                sr[i] = p;
                dr[i] = ds;
                mx = fmaxf(mx, fabsf(ds));
            }
        }
    }
    mx = wave_max(mx);
    __shared__ float wm[4];
    if (lane == 0) wm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(ds_amax, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// out[row] = sum_c a[row][c] * b[row][c]   (delta = rowsum(dO * O)); one wave per row
__global__ __launch_bounds__(256) void rowdot_kernel(const float* a, const float* b, float* out, long rows, int C) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* ar = a + row * C;
    const float* br = b + row * C;
    float acc = 0.f;
    for (int i = lane; i < C; i += 64) acc = fmaf(ar[i], br[i], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc;
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int favae_bgemm(int ta, int tb, int M, int N, int K, float alpha, const float* A, int64_t lda, int64_t strideA,
                           const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc, int64_t strideC, int batch,
                           int accumulate, favae_stream_t stream) {
    FAVAE_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && (ta == 0 || ta == 1) && (tb == 0 || tb == 1));
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.sA = strideA; a.sB = strideB; a.sC = strideC;
    a.M = M; a.N = N; a.K = K; a.alpha = alpha; a.accumulate = accumulate;
    a.vecA = aligned16(A) && lda % 4 == 0 && strideA % 4 == 0 && (ta == 0 ? K % 4 == 0 : M % 4 == 0);
    a.vecB = aligned16(B) && ldb % 4 == 0 && strideB % 4 == 0 && (tb == 0 ? K % 4 == 0 : N % 4 == 0);
    dim3 grid(cdiv(M, GBM) * cdiv(N, GBN), batch);
    hipStream_t s = (hipStream_t)stream;
    if (ta == 0 && tb == 0) FAVAE_KLAUNCH((bgemm_kernel<0, 0>), grid, dim3(256), 0, s, a);
    else if (ta == 0 && tb == 1) FAVAE_KLAUNCH((bgemm_kernel<0, 1>), grid, dim3(256), 0, s, a);
    else if (ta == 1 && tb == 0) FAVAE_KLAUNCH((bgemm_kernel<1, 0>), grid, dim3(256), 0, s, a);
    else FAVAE_KLAUNCH((bgemm_kernel<1, 1>), grid, dim3(256), 0, s, a);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_softmax_rows(const float* s, float* p, int64_t rows, int L, favae_stream_t stream) {
    FAVAE_REQUIRE(s && p && rows > 0 && L > 0);
    FAVAE_KLAUNCH(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, s, p, (long)rows, L);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_softmax_rows_bwd(const float* p, const float* dp, float* ds, int64_t rows, int L, float alpha,
                                      favae_stream_t stream) {
    FAVAE_REQUIRE(p && dp && ds && rows > 0 && L > 0);
    FAVAE_KLAUNCH(softmax_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, p, dp, ds, (long)rows,
                       L, alpha);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_bgemm_sp(int ta, int tb, int M, int N, int K, float alpha, const float* A, int64_t lda, int64_t strideA,
                              const float* amaxA, const float* B, int64_t ldb, int64_t strideB, const float* amaxB, float* C,
                              int64_t ldc, int64_t strideC, int batch, int accumulate, favae_stream_t stream) {
    FAVAE_REQUIRE(A && B && C && amaxA && amaxB && M > 0 && N > 0 && K > 0 && batch > 0 && (ta == 0 || ta == 1) && (tb == 0 || tb == 1));
    const bool okA = aligned16(A) && lda % 4 == 0 && strideA % 4 == 0 && (ta == 0 ? K % 4 == 0 : M % 4 == 0);
    const bool okB = aligned16(B) && ldb % 4 == 0 && strideB % 4 == 0 && (tb == 0 ? K % 4 == 0 : N % 4 == 0);
    if (!okA || !okB || (ta == 1 && tb == 0)) return favae_prof_fail_(FAVAE_ERR_UNSUPPORTED);
    GemmSpArgs a;
    a.A = A; a.B = B; a.C = C; a.amaxA = amaxA; a.amaxB = amaxB;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.sA = strideA; a.sB = strideB; a.sC = strideC;
    a.M = M; a.N = N; a.K = K; a.alpha = alpha; a.accumulate = accumulate;
    dim3 grid(cdiv(M, 128) * cdiv(N, 128), batch);
    hipStream_t s = (hipStream_t)stream;
    FAVAE_PROF_NOTE(2.0 * M * N * K * batch, 4.0 * batch * ((double)M * K + (double)N * K + (double)M * N));
    if (ta == 0 && tb == 0) FAVAE_KLAUNCH((bgemm_sp_kernel<0, 0>), grid, dim3(512), 0, s, a);
    else if (ta == 0 && tb == 1) FAVAE_KLAUNCH((bgemm_sp_kernel<0, 1>), grid, dim3(512), 0, s, a);
    else FAVAE_KLAUNCH((bgemm_sp_kernel<1, 1>), grid, dim3(512), 0, s, a);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_softmax_rows_lse(float* s, float* lse, int64_t rows, int L, int rows_per_batch, int row0, int Ltot,
                                      favae_stream_t stream) {
    FAVAE_REQUIRE(s && lse && rows > 0 && L > 0 && rows_per_batch > 0 && row0 >= 0 && row0 + rows_per_batch <= Ltot);
    FAVAE_PROF_NOTE(0, 8.0 * rows * L);
    FAVAE_KLAUNCH(softmax_rows_lse_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, s, lse, (long)rows, L,
                  rows_per_batch, row0, Ltot);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_attn_bwd_point(float* s, float* dp, const float* lse, const float* delta, int64_t rows, int L,
                                    int rows_per_batch, int row0, int Ltot, float alpha, float* ds_absmax, favae_stream_t stream) {
    FAVAE_REQUIRE(s && dp && lse && delta && ds_absmax && rows > 0 && L > 0 && rows_per_batch > 0 && row0 >= 0 &&
                  row0 + rows_per_batch <= Ltot);
    hipStream_t st = (hipStream_t)stream;
    if (favae_zero_target(ds_absmax, sizeof(float), st) != hipSuccess) return favae_prof_fail_(FAVAE_ERR_LAUNCH);
    FAVAE_PROF_NOTE(0, 16.0 * rows * L);
    long blocks = cdiv(rows, 4);
    if (blocks > 2048) blocks = 2048;
    if (L % 4 == 0 && aligned16(s) && aligned16(dp))
        FAVAE_KLAUNCH((attn_bwd_point_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, s, dp, lse, delta, (long)rows, L,
                      rows_per_batch, row0, Ltot, alpha, (unsigned*)ds_absmax);
    else
        FAVAE_KLAUNCH((attn_bwd_point_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, s, dp, lse, delta, (long)rows, L,
                      rows_per_batch, row0, Ltot, alpha, (unsigned*)ds_absmax);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}

extern "C" int favae_rowdot(const float* a, const float* b, float* out, int64_t rows, int C, favae_stream_t stream) {
    FAVAE_REQUIRE(a && b && out && rows > 0 && C > 0);
    FAVAE_KLAUNCH(rowdot_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)rows, C);
    FAVAE_CHECK_LAUNCH();
    return FAVAE_OK;
}
