// EXPERIMENT (opt-in, FAVAE_CONV_B6=1): fp32 convolution on the bf16 matrix pipe with an exact 3-way operand split.
//
// Every fp32 operand is cut (by truncation, exactly) into three bf16 pieces  a = a1 + a2 + a3  (8 + 8 + 8 significand bits);
// products of two bf16 are exact in fp32, so  a*b = sum_{i,j} ai*bj  and keeping the six terms with i + j <= 4
// (a1b1, a1b2, a2b1, a2b2, a1b3, a3b1) leaves a relative error of ~2^-24 per product -- the same class as the rounding of an
// fp32 FMA -- while v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32: 6/16 of the matrix-pipe time.
// Same tiling as conv_fwd_buf_kernel (128x128x16, 4 waves of 64x64); LDS holds the three bf16 planes of each row
// ([row][plane][16 k] + 16 B pad = 112 B rows -> conflict-free ds_read_b128 fragment reads).
#pragma once

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

namespace b6 {
constexpr int ROWB = 112;                 // bytes per LDS row: 3 planes x 32 B + 16 B pad
constexpr unsigned TOP = 0xFFFF0000u;

// split four consecutive-k floats into three planes of 4 bf16 (2 dwords each), exact by truncation
__device__ __forceinline__ void split4(const float4 v, uint2& p0, uint2& p1, uint2& p2) {
    const float a[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = __float_as_uint(a[e]);
        const float r1 = a[e] - __uint_as_float(h[e] & TOP);
        m[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(m[e] & TOP);
        l[e] = __float_as_uint(r2);
    }
    // pack the upper halves of two words: low 16 bits <- even k, high 16 bits <- odd k
    p0 = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    p1 = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    p2 = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}
}  // namespace b6

template <int GATHER, int XFORM>
__global__ __launch_bounds__(256) void conv_fwd_b6_kernel(ConvArgs a) {
    constexpr int BN = 128, WTM = 64, WTN = 64, MI = 2, NI = 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * b6::ROWB];
    unsigned char* As = lds;
    unsigned char* Bs = lds + 2 * BM * b6::ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / a.tiles_n) * BM, n0 = (tile % a.tiles_n) * BN;
    const int q4 = tid & 3, c4 = q4 * 4;
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
    unsigned vob[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (tid >> 2) + 64 * j;
        vob[j] = (n0 + row < a.Cout) ? (unsigned)(((n0 + row) * taps * a.Cin + c4) * 4) : FAVAE_OOB;
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

    float4 ra[2], rsc[2], rsh[2], rb[2];
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
            rb[j] = bload(rw, vob[j], sw);
        }
        if (++ld_kc == a.kchunks) {
            ld_kc = 0;
            if (++ld_tap < taps) tap_state(ld_tap);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (tid >> 2) + 64 * j;
            uint2 p0, p1, p2;
            b6::split4(xform4_t<XFORM>(ra[j], rsc[j], rsh[j]), p0, p1, p2);
            unsigned char* d = As + (buf * BM + row) * b6::ROWB + q4 * 8;
            *reinterpret_cast<uint2*>(d) = p0;
            *reinterpret_cast<uint2*>(d + 32) = p1;
            *reinterpret_cast<uint2*>(d + 64) = p2;
            b6::split4(rb[j], p0, p1, p2);
            d = Bs + (buf * BN + row) * b6::ROWB + q4 * 8;
            *reinterpret_cast<uint2*>(d) = p0;
            *reinterpret_cast<uint2*>(d + 32) = p1;
            *reinterpret_cast<uint2*>(d + 64) = p2;
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
    const int frow = lane & 31, fh = (lane >> 5) * 16;
    load_tiles();
    store_tiles(0);
    __syncthreads();
    for (int it = 0; it < T; ++it) {
        const int cur = it & 1;
        if (it + 1 < T) load_tiles();
        bf16x8_t af[MI][3], bf[NI][3];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                af[i][p] = *reinterpret_cast<const bf16x8_t*>(As + (cur * BM + wm * WTM + i * 32 + frow) * b6::ROWB + p * 32 + fh);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bf[j][p] = *reinterpret_cast<const bf16x8_t*>(Bs + (cur * BN + wn * WTN + j * 32 + frow) * b6::ROWB + p * 32 + fh);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                // smallest terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
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
