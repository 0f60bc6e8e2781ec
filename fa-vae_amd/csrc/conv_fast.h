// Specialised (16-byte vector path, compile-time gather mode) versions of the implicit-GEMM kernels of conv.hip.
// Same tiling and numerics (bit-identical accumulation order) as the generic kernels; what changes is the instruction
// stream around the MFMAs:
//   * Cin % 4 == 0 (and Cout % 4 == 0 for wgrad) is a precondition -> no scalar tail paths, no per-element predicates;
//   * the gather mode is a template parameter, the input transform a wave-uniform switch -> scalar branches only;
//   * per-tap source offsets / validity are computed once per tap (fwd) or updated incrementally per step (wgrad)
//     instead of re-deriving (n, oh, ow) with integer divisions every K step;
//   * SiLU uses v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the IEEE division sequence.
#pragma once

template <int GATHER>
__device__ __forceinline__ bool gather_src_t(int stride, int pad, int Hin, int Win, int oh, int ow, int kh, int kw, int& sh,
                                             int& sw) {
    const int vh = oh * stride + kh - pad;
    const int vw = ow * stride + kw - pad;
    if (GATHER == FAVAE_GATHER_PLAIN) {
        sh = vh; sw = vw;
        return (unsigned)vh < (unsigned)Hin && (unsigned)vw < (unsigned)Win;
    } else if (GATHER == FAVAE_GATHER_UPSAMPLE2) {
        sh = vh >> 1; sw = vw >> 1;
        return (unsigned)vh < (unsigned)(2 * Hin) && (unsigned)vw < (unsigned)(2 * Win);
    } else {
        sh = vh >> 1; sw = vw >> 1;
        return vh >= 0 && vw >= 0 && !((vh | vw) & 1) && sh < Hin && sw < Win;
    }
}

__device__ __forceinline__ float silu_fast(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }

__device__ __forceinline__ float4 xform4(float4 v, float4 sc, float4 sh, int xform, int act) {
    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
    if (xform == 2) {
        v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w);
    } else if (xform == 3) {
        const float sl = act == FAVAE_ACT_RELU ? 0.0f : 0.2f;
        v.x = v.x > 0.f ? v.x : sl * v.x; v.y = v.y > 0.f ? v.y : sl * v.y;
        v.z = v.z > 0.f ? v.z : sl * v.z; v.w = v.w > 0.f ? v.w : sl * v.w;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// forward / data gradient
// ---------------------------------------------------------------------------------------------------------------
template <int BN, int WAVES_M, int WAVES_N, int GATHER>
__global__ __launch_bounds__(256) void conv_fwd_fast_kernel(ConvArgs a) {
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
    // xform: 0 none, 1 affine, 2 affine+SiLU, 3 affine+LeakyReLU  (wave-uniform)
    const int xform = a.scale ? (a.act == FAVAE_ACT_SILU ? 2 : (a.act == FAVAE_ACT_NONE ? 1 : 3)) : 0;

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
    // weight rows of this thread
    const float* wrow[B_LD];
    bool w_ok[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int row = (tid >> 2) + 64 * j;
        w_ok[j] = row < BN && n0 + row < a.Cout;
        wrow[j] = a.w + (size_t)(w_ok[j] ? n0 + row : 0) * taps * a.Cin + c4;
    }

    // per-tap state of the prefetcher
    const float* src[2];
    bool sok[2];
    int ld_tap = 0, ld_kc = 0;
    auto tap_state = [&](int tap) {
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int sh, sw;
            sok[j] = r_ok[j] && gather_src_t<GATHER>(a.stride, a.pad, a.Hin, a.Win, r_oh[j], r_ow[j], kh, kw, sh, sw);
            src[j] = a.x + ((size_t)(r_n[j] * a.Hin + (sok[j] ? sh : 0)) * a.Win + (sok[j] ? sw : 0)) * a.Cin + c4;
        }
    };
    tap_state(0);

    float4 ra[2], rsc[2], rsh[2], rb[B_LD];
    bool rav[2];
    auto load_tiles = [&]() {
        const int k0 = ld_kc * BK;
        const bool kin = k0 + c4 < a.Cin;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rav[j] = sok[j] && kin;
            ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rav[j]) {
                ra[j] = *reinterpret_cast<const float4*>(src[j] + k0);
                if (xform) {
                    const size_t so = (size_t)r_n[j] * a.aff_stride + k0 + c4;
                    rsc[j] = *reinterpret_cast<const float4*>(a.scale + so);
                    rsh[j] = *reinterpret_cast<const float4*>(a.shift + so);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w_ok[j] && kin) rb[j] = *reinterpret_cast<const float4*>(wrow[j] + (size_t)ld_tap * a.Cin + k0);
        }
        if (++ld_kc == a.kchunks) {
            ld_kc = 0;
            if (++ld_tap < taps) tap_state(ld_tap);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 v = ra[j];
            if (xform) {
                const float4 t = xform4(v, rsc[j], rsh[j], xform, a.act);
                v = rav[j] ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            *reinterpret_cast<float4*>(&As[(buf * BM + (tid >> 2) + 64 * j) * LDK + c4]) = v;
        }
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

    const int T = taps * a.kchunks;
    load_tiles();
    store_tiles(0);
    __syncthreads();
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) load_tiles();
        const float* Ab = As + (cur * BM + wm * WTM + frow) * LDK + fk;
        const float* Bb = Bs + (cur * BN + wn * WTN + frow) * LDK + fk;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK + kk * 8);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (it + 1 < T) store_tiles(cur ^ 1);
        __syncthreads();
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
// weight gradient
// ---------------------------------------------------------------------------------------------------------------
template <int BCO, int BCI, int WAVES_O, int WAVES_I, int GATHER>
__global__ __launch_bounds__(256) void conv_wgrad_fast_kernel(WgradArgs a) {
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
    const int xform = a.scale ? (a.act == FAVAE_ACT_SILU ? 2 : (a.act == FAVAE_ACT_NONE ? 1 : 3)) : 0;

    // fixed (pixel slot, channel quad) of this thread for both tiles; pixel coordinates advance by BKP per step
    int o_p[O_LD], o_c[O_LD];
    bool o_ok[O_LD];
#pragma unroll
    for (int j = 0; j < O_LD; ++j) {
        const int i = tid + 256 * j;
        o_p[j] = i / (BCO / 4);
        o_c[j] = (i % (BCO / 4)) * 4;
        o_ok[j] = o_p[j] < BKP && co0 + o_c[j] < a.Cout;
    }
    int i_p[I_LD], i_c[I_LD], i_n[I_LD], i_oh[I_LD], i_ow[I_LD];
    bool i_ok[I_LD];
    {
        const int hw = a.Hout * a.Wout;
#pragma unroll
        for (int j = 0; j < I_LD; ++j) {
            const int i = tid + 256 * j;
            i_p[j] = i / (BCI / 4);
            i_c[j] = (i % (BCI / 4)) * 4;
            i_ok[j] = i_p[j] < BKP && ci0 + i_c[j] < a.Cin;
            const int m = min(p_begin + i_p[j], a.M - 1);
            i_n[j] = m / hw;
            const int r = m - i_n[j] * hw;
            i_oh[j] = r / a.Wout;
            i_ow[j] = r - i_oh[j] * a.Wout;
        }
    }
    int ld_pb = p_begin;

    float4 ro[O_LD], ri[I_LD], rsc[I_LD], rsh[I_LD];
    bool riv[I_LD];
    auto load_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < O_LD; ++j) {
            const int m = ld_pb + o_p[j];
            ro[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o_ok[j] && m < p_end) ro[j] = *reinterpret_cast<const float4*>(a.dy + (size_t)m * a.Cout + co0 + o_c[j]);
        }
#pragma unroll
        for (int j = 0; j < I_LD; ++j) {
            const int m = ld_pb + i_p[j];
            int sh, sw;
            const bool ok = i_ok[j] && m < p_end &&
                            gather_src_t<GATHER>(a.stride, a.pad, a.Hin, a.Win, i_oh[j], i_ow[j], kh, kw, sh, sw);
            riv[j] = ok;
            ri[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                ri[j] = *reinterpret_cast<const float4*>(a.x + ((size_t)(i_n[j] * a.Hin + sh) * a.Win + sw) * a.Cin + ci0 + i_c[j]);
                if (xform) {
                    const size_t so = (size_t)i_n[j] * a.aff_stride + ci0 + i_c[j];
                    rsc[j] = *reinterpret_cast<const float4*>(a.scale + so);
                    rsh[j] = *reinterpret_cast<const float4*>(a.shift + so);
                }
            }
            // advance this slot's pixel by BKP
            i_ow[j] += BKP;
            while (i_ow[j] >= a.Wout) { i_ow[j] -= a.Wout; ++i_oh[j]; }
            while (i_oh[j] >= a.Hout) { i_oh[j] -= a.Hout; ++i_n[j]; }
        }
        ld_pb += BKP;
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < O_LD; ++j)
            if (o_p[j] < BKP) *reinterpret_cast<float4*>(&Os[(buf * BKP + o_p[j]) * BCO + o_c[j]]) = ro[j];
#pragma unroll
        for (int j = 0; j < I_LD; ++j) {
            float4 v = ri[j];
            if (xform) {
                const float4 tt = xform4(v, rsc[j], rsh[j], xform, a.act);
                v = riv[j] ? tt : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (i_p[j] < BKP) *reinterpret_cast<float4*>(&Is[(buf * BKP + i_p[j]) * BCI + i_c[j]]) = v;
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (T > 0) {
        load_tiles();
        store_tiles(0);
    }
    __syncthreads();
    const int frow = lane & 31, fk = lane >> 5;
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) load_tiles();
        const float* Ob = Os + (cur * BKP + fk) * BCO + wo * WTO + frow;
        const float* Ib = Is + (cur * BKP + fk) * BCI + wi * WTI + frow;
#pragma unroll
        for (int kk = 0; kk < BKP / 2; ++kk) {
            float af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = Ob[kk * 2 * BCO + i * 32];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = Ib[kk * 2 * BCI + j * 32];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < T) store_tiles(cur ^ 1);
        __syncthreads();
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
