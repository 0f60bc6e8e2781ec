// Buffer-addressed versions of the implicit-GEMM kernels (included by conv.hip).  Same tiling and accumulation order as
// the generic kernels (bit-identical results); the difference is how operands reach the LDS tiles:
//   * all global reads are `buffer_load_dwordx4` through wave-uniform descriptors: per-lane 32-bit byte offset (VGPR, changes
//     only when the filter tap changes) + scalar byte offset (SGPR, advances with the K loop).  No 64-bit address arithmetic,
//     no exec-mask predication and no zero-fill in the K loop: padding / dilation holes / out-of-tile rows use the offset
//     0x80000000, which the hardware range check turns into zeros;
//   * invalid rows also read scale = shift = 0, and act(0*0+0) = 0 for SiLU and LeakyReLU, so the fused GroupNorm transform
//     needs no select either;
//   * the input transform is a template parameter (no branches in the loop).
// PMC motivation: profiles/r01_pmc_conv.md (VALU work per K step was not hidden behind the MFMAs).
// Preconditions (dispatcher): Cin % 16 == 0, every operand < 2 GiB; wgrad additionally: plain gather, stride 1,
// Wout % 16 == 0, Cout % 4 == 0.
#pragma once

// buffer addressing helpers (make_rsrc, bload, bstore, FAVAE_OOB): common.h

// XFORM: 0 none | 1 affine | 2 affine + SiLU | 3 affine + LeakyReLU(0.2) or ReLU (act = FAVAE_ACT_RELU: slope 0)
__device__ __forceinline__ float leaky_slope(int act) { return act == FAVAE_ACT_RELU ? 0.0f : 0.2f; }

template <int XFORM>
__device__ __forceinline__ float xform1(float v, float sc, float sh, int act) {
    if (XFORM == 0) return v;
    v = fmaf(v, sc, sh);
    if (XFORM == 2) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    if (XFORM == 3) v = v > 0.f ? v : leaky_slope(act) * v;
    return v;
}

template <int XFORM>
__device__ __forceinline__ float4 xform4_t(float4 v, float4 sc, float4 sh, int act) {
    v.x = xform1<XFORM>(v.x, sc.x, sh.x, act);
    v.y = xform1<XFORM>(v.y, sc.y, sh.y, act);
    v.z = xform1<XFORM>(v.z, sc.z, sh.z, act);
    v.w = xform1<XFORM>(v.w, sc.w, sh.w, act);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// forward / data gradient
// ---------------------------------------------------------------------------------------------------------------
template <int BN, int WAVES_M, int WAVES_N, int GATHER, int XFORM>
__global__ __launch_bounds__(256) void conv_fwd_buf_kernel(ConvArgs a) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int B_LD = (BN * BK / 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDK];
    float* As = lds;
    float* Bs = lds + 2 * BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
    const int c4 = (tid & 3) * 4;
    const int taps = a.KH * a.KW;

    const auto rx = make_rsrc(a.x, a.x_bytes);
    const auto rw = make_rsrc(a.w, a.w_bytes);
    const auto rsc_d = make_rsrc(XFORM ? a.scale : a.x, XFORM ? a.aff_bytes : 0u);
    const auto rsh_d = make_rsrc(XFORM ? a.shift : a.x, XFORM ? a.aff_bytes : 0u);

    int r_n[2], r_oh[2], r_ow[2];
    bool r_ok[2];
    {
        const int hw = a.Hout * a.Wout;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + (tid >> 2) + 64 * j;
            r_ok[j] = m < a.M;
            const int mm = r_ok[j] ? m : 0;
            r_n[j] = mm / hw;
            const int r = mm - r_n[j] * hw;
            r_oh[j] = r / a.Wout;
            r_ow[j] = r - r_oh[j] * a.Wout;
        }
    }
    unsigned vob[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int row = (tid >> 2) + 64 * j;
        const bool ok = row < BN && n0 + row < a.Cout;
        vob[j] = ok ? (unsigned)(((n0 + row) * taps * a.Cin + c4) * 4) : FAVAE_OOB;
    }
    unsigned voa[2], vos[2];
    int ld_tap = 0, ld_kc = 0;
    auto tap_state = [&](int tap) {
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int sh, sw;
            const bool ok = r_ok[j] && gather_src_t<GATHER>(a.stride, a.pad, a.Hin, a.Win, r_oh[j], r_ow[j], kh, kw, sh, sw);
            voa[j] = ok ? (unsigned)((((r_n[j] * a.Hin + sh) * a.Win + sw) * a.Cin + c4) * 4) : FAVAE_OOB;
            if (XFORM) vos[j] = ok ? (unsigned)((r_n[j] * a.aff_stride + c4) * 4) : FAVAE_OOB;
        }
    };
    tap_state(0);

    float4 ra[2], rsc[2], rsh[2], rb[B_LD];
    auto load_tiles = [&]() {
        const unsigned sk = (unsigned)(ld_kc * BK * 4);
        const unsigned sw = (unsigned)((ld_tap * a.Cin + ld_kc * BK) * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ra[j] = bload(rx, voa[j], sk);
            if (XFORM) {
                rsc[j] = bload(rsc_d, vos[j], sk);
                rsh[j] = bload(rsh_d, vos[j], sk);
            }
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) rb[j] = bload(rw, vob[j], sw);
        if (++ld_kc == a.kchunks) {
            ld_kc = 0;
            if (++ld_tap < taps) tap_state(ld_tap);
        }
    };
    float* a_st[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) a_st[j] = As + ((tid >> 2) + 64 * j) * LDK + c4;
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<float4*>(a_st[j] + buf * BM * LDK) = xform4_t<XFORM>(ra[j], rsc[j], rsh[j], a.act);
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int row = (tid >> 2) + 64 * j;
            if (row < BN) *reinterpret_cast<float4*>(&Bs[(buf * BN + row) * LDK + c4]) = rb[j];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Schedule (2 LDS buffers, ONE barrier per K step, placed in the MIDDLE of the step's MFMA stream):
    //   top   : ds_write tile i+1 (its global loads were issued a whole step earlier) ; issue global loads of tile i+2
    //   first : fragment reads of k-group 1 (tile i) -> 16 MFMAs of k-group 0
    //   middle: lgkmcnt(0) + s_barrier  (tile i+1 is now complete in LDS, tile i-1's buffer is free)
    //   second: fragment reads of k-group 0 of tile i+1 -> 16 MFMAs of k-group 1
    // so the LDS write -> barrier -> read latency of the classic end-of-step barrier is covered by the second MFMA group
    // (PMC: with VALU work already minimal the pipe was still idle ~17 % of the time, profiles/r01_pmc_conv.md).
    const int T = taps * a.kchunks;
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    const float* Ab0 = As + (wm * WTM + frow) * LDK + fk;
    const float* Bb0 = Bs + (wn * WTN + frow) * LDK + fk;
    float4 af0[MI], bf0[NI], af1[MI], bf1[NI];
    auto read_frags = [&](int buf, int kk, float4 (&af)[MI], float4 (&bf)[NI]) {
        const float* Ab = Ab0 + buf * BM * LDK + kk * 8;
        const float* Bb = Bb0 + buf * BN * LDK + kk * 8;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK);
    };
    auto mfma_group = [&](const float4 (&af)[MI], const float4 (&bf)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
            }
    };
    load_tiles();
    store_tiles(0);
    if (T > 1) load_tiles();                               // tile 1 in flight
    __syncthreads();
    read_frags(0, 0, af0, bf0);
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) store_tiles(cur ^ 1);              // tile it+1: registers -> LDS (other buffer)
        if (it + 2 < T) load_tiles();                      // tile it+2: global -> registers
        read_frags(cur, 1, af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (it + 1 < T) read_frags(cur ^ 1, 0, af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
    }

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * WTN + j * 32 + (lane & 31);
            if (col >= a.Cout) continue;
            const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < a.M) {
                    const size_t o = (size_t)row * a.Cout + col;
                    float v = acc[i][j][r] + bv;
                    if (a.resid) v += a.resid[o];
                    a.y[o] = v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient (plain gather, stride 1, Wout % 16 == 0: every 16-pixel step is one output-row segment, so the
// step's (image, row, first column) live in SGPRs and only the column validity is per lane)
// ---------------------------------------------------------------------------------------------------------------
template <int BCO, int BCI, int WAVES_O, int WAVES_I, int XFORM>
__global__ __launch_bounds__(256) void conv_wgrad_buf_kernel(WgradArgs a) {
    constexpr int BKP = 16;
    constexpr int WTO = BCO / WAVES_O, WTI = BCI / WAVES_I;
    constexpr int MI = WTO / 32, NI = WTI / 32;
    constexpr int O_LD = (BKP * BCO / 4 + 255) / 256;
    constexpr int I_LD = (BKP * BCI / 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[2 * BKP * (BCO + BCI)];
    float* Os = lds;
    float* Is = lds + 2 * BKP * BCO;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wo = wid / WAVES_I, wi = wid % WAVES_I;
    const int taps = a.KH * a.KW;
    int t = blockIdx.x;
    const int tap = t % taps; t /= taps;
    const int ci0 = (t % a.tiles_ci) * BCI;
    const int co0 = (t / a.tiles_ci) * BCO;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int z = blockIdx.y;
    const int p_begin = z * a.chunk;
    const int p_end = min(a.M, p_begin + a.chunk);
    const int T = (p_end > p_begin) ? (p_end - p_begin + BKP - 1) / BKP : 0;

    const auto rx = make_rsrc(a.x, a.x_bytes);
    const auto rdy = make_rsrc(a.dy, (unsigned)p_end * (unsigned)a.Cout * 4u);       // rows >= p_end read as zeros
    const auto rsc_d = make_rsrc(XFORM ? a.scale : a.x, XFORM ? a.aff_bytes : 0u);
    const auto rsh_d = make_rsrc(XFORM ? a.shift : a.x, XFORM ? a.aff_bytes : 0u);

    unsigned voo[O_LD];
    int o_p[O_LD], o_c[O_LD];
#pragma unroll
    for (int j = 0; j < O_LD; ++j) {
        const int i = tid + 256 * j;
        o_p[j] = i / (BCO / 4);
        o_c[j] = (i % (BCO / 4)) * 4;
        const bool ok = o_p[j] < BKP && co0 + o_c[j] < a.Cout;
        voo[j] = ok ? (unsigned)((o_p[j] * a.Cout + co0 + o_c[j]) * 4) : FAVAE_OOB;
    }
    int i_p[I_LD], i_c[I_LD];
    bool i_ok[I_LD];
    unsigned vos[I_LD];
#pragma unroll
    for (int j = 0; j < I_LD; ++j) {
        const int i = tid + 256 * j;
        i_p[j] = i / (BCI / 4);
        i_c[j] = (i % (BCI / 4)) * 4;
        i_ok[j] = i_p[j] < BKP && ci0 + i_c[j] < a.Cin;
        vos[j] = (unsigned)((ci0 + i_c[j]) * 4);
    }
    // scalar position of the next step's first pixel
    int s_n, s_oh, s_ow;
    {
        const int hw = a.Hout * a.Wout;
        const int mb = min(p_begin, a.M - 1);
        s_n = mb / hw;
        const int r = mb - s_n * hw;
        s_oh = r / a.Wout;
        s_ow = r - s_oh * a.Wout;
    }
    int ld_pb = p_begin;

    float4 ro[O_LD], ri[I_LD], rsc[I_LD], rsh[I_LD];
    auto load_tiles = [&]() {
        const unsigned so = (unsigned)ld_pb * (unsigned)a.Cout * 4u;
#pragma unroll
        for (int j = 0; j < O_LD; ++j) ro[j] = bload(rdy, voo[j], so);
        const int ih = s_oh + kh - a.pad;
        const bool row_ok = (unsigned)ih < (unsigned)a.Hin;
        const unsigned sx = row_ok ? (unsigned)(((s_n * a.Hin + ih) * a.Win) * a.Cin) * 4u : 0u;
        const unsigned ss = (unsigned)(s_n * a.aff_stride) * 4u;
#pragma unroll
        for (int j = 0; j < I_LD; ++j) {
            const int iw = s_ow + i_p[j] + kw - a.pad;
            const bool ok = row_ok && i_ok[j] && (unsigned)iw < (unsigned)a.Win && ld_pb + i_p[j] < p_end;
            const unsigned vx = ok ? (unsigned)((iw * a.Cin + ci0 + i_c[j]) * 4) : FAVAE_OOB;
            ri[j] = bload(rx, vx, sx);
            if (XFORM) {
                const unsigned vs = ok ? vos[j] : FAVAE_OOB;
                rsc[j] = bload(rsc_d, vs, ss);
                rsh[j] = bload(rsh_d, vs, ss);
            }
        }
        ld_pb += BKP;
        s_ow += BKP;
        if (s_ow >= a.Wout) {
            s_ow = 0;
            if (++s_oh >= a.Hout) { s_oh = 0; ++s_n; }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < O_LD; ++j)
            if (o_p[j] < BKP) *reinterpret_cast<float4*>(&Os[(buf * BKP + o_p[j]) * BCO + o_c[j]]) = ro[j];
#pragma unroll
        for (int j = 0; j < I_LD; ++j)
            if (i_p[j] < BKP)
                *reinterpret_cast<float4*>(&Is[(buf * BKP + i_p[j]) * BCI + i_c[j]]) = xform4_t<XFORM>(ri[j], rsc[j], rsh[j], a.act);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // same mid-step barrier schedule as conv_fwd_buf_kernel: k-pairs 0..3 | barrier | k-pairs 4..7
    const int frow = lane & 31, fk = lane >> 5;
    constexpr int HALF = BKP / 4;                          // k-pairs per half step
    float af0[HALF][MI], bf0[HALF][NI], af1[HALF][MI], bf1[HALF][NI];
    auto read_frags = [&](int buf, int half, float (&af)[HALF][MI], float (&bf)[HALF][NI]) {
        const float* Ob = Os + (buf * BKP + fk + 2 * HALF * half) * BCO + wo * WTO + frow;
        const float* Ib = Is + (buf * BKP + fk + 2 * HALF * half) * BCI + wi * WTI + frow;
#pragma unroll
        for (int kk = 0; kk < HALF; ++kk) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[kk][i] = Ob[kk * 2 * BCO + i * 32];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[kk][j] = Ib[kk * 2 * BCI + j * 32];
        }
    };
    auto mfma_group = [&](const float (&af)[HALF][MI], const float (&bf)[HALF][NI]) {
#pragma unroll
        for (int kk = 0; kk < HALF; ++kk)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
    };
    if (T > 0) {
        load_tiles();
        store_tiles(0);
        if (T > 1) load_tiles();
    }
    __syncthreads();
    if (T > 0) read_frags(0, 0, af0, bf0);
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) store_tiles(cur ^ 1);
        if (it + 2 < T) load_tiles();
        read_frags(cur, 1, af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (it + 1 < T) read_frags(cur ^ 1, 0, af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int ci = ci0 + wi * WTI + j * 32 + (lane & 31);
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wo * WTO + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < a.Cout) a.part[(((size_t)z * a.Cout + co) * taps + tap) * a.Cin + ci] = acc[i][j][r];
            }
        }
}
