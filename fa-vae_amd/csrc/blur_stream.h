// Row-streaming 9-tap Gaussian blur (included by blur.hip): the fast path for the FCM taps of every FA-VAE configuration
// (kernel 9, C % 32 == 0).  Same mathematics as blur_sep_kernel (header of blur.hip); different data movement:
//
//   * a workgroup owns (image n, a strip of COLS-8 output columns, 32 channels, a segment of RS rows) and MARCHES down the
//     rows.  Thread = (column, channel quad): 8 consecutive threads cover the 128-byte channel segment of one pixel, so every
//     global access is a full cache line and every LDS access a conflict-free ds_read/write_b128.
//   * per step ONE input row (COLS pixels incl. the 4+4 halo columns, reflect indexing folded into the thread's constant
//     column) goes global -> registers (prefetched one step ahead) -> LDS; each thread reads its 9 horizontal neighbours
//     (9 ds_read_b128 for 36 FMAs -- the tile kernel did one ds_read_b32 per FMA) and keeps the 9-row vertical window of
//     row-blurred values in REGISTERS: no second LDS tile, no k^2 halo, x is read ~1.14x (strip halo) instead of 2.25x.
//   * backward: the same march also carries a 9-row register window of dy.  With Dwin = dy[i-4..i+4]:
//       V'[i]   = sum_t wrow_i[t] Dwin[t]        vertical adjoint INCLUDING the reflect fold (the fold only re-weights rows
//                                                inside the window: wrow_i[t] = sum_{r in pre(i)} g[r - i + 8 - t])
//       dx[i,j] = sum_t wcol_j[t] V'[i, j-4+t]   horizontal adjoint + fold (V' row exchanged through LDS; wcol per thread)
//       dg_a   += <dy[i-4], U[i-4+a-4 .. ]>  +  <V'[i], Xe[i, j+a-4]>     (both operands already in registers)
//     dg partials are reduced per workgroup and summed over workgroups in a fixed order by favae_colsum (deterministic).
#pragma once

namespace {

struct StreamArgs {
    const float* x;
    const float* dy;
    const float* sigma;
    float* y;         // forward output
    float* dx;        // backward: may be null
    float* part;      // backward: [blocks][9] partials of dg, may be null
    const float* dx_add;  // backward: optional tensor added to dx
    int N, H, W, C, RS, strips, segs, cchunks;
};

__device__ __forceinline__ float4 f4_fma(float s, float4 v, float4 a) {
    return make_float4(fmaf(s, v.x, a.x), fmaf(s, v.y, a.y), fmaf(s, v.z, a.z), fmaf(s, v.w, a.w));
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b, float acc) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, fmaf(a.w, b.w, acc))));
}

// sum_{r in pre(i)} gz[r - i + 8 - t]   (gz = g inside [0,8], 0 outside) -- weight of window slot t for output index i
__device__ __forceinline__ float fold_weight(const float* gs, int i, int n, int t) {
    float w = gs[8 - t];
    if (i >= 1 && i <= 4) {
        const int m = 8 - t - 2 * i;
        if (m >= 0) w += gs[m];
    }
    if (i <= n - 2 && i >= n - 5) {
        const int m = 2 * (n - 1) - 2 * i + 8 - t;
        if (m <= 8) w += gs[m];
    }
    return w;
}

template <int MODE, int COLS>   // MODE 0 forward, 1 backward; COLS thread columns (COLS - 8 output columns per strip)
__global__ __launch_bounds__(COLS * 8) void blur9_stream_kernel(StreamArgs a) {
    constexpr int K = 9, P = 4, NT = COLS * 8, OUTC = COLS - 2 * P;
    __shared__ float4 Xs[2][COLS * 8];
    __shared__ float4 Vs[MODE == 1 ? 2 : 1][MODE == 1 ? COLS * 8 : 1];
    __shared__ float gs[32];
    __shared__ float red[K][NT / 64];

    int b = blockIdx.x;
    const int cchunk = b % a.cchunks; b /= a.cchunks;
    const int strip = b % a.strips; b /= a.strips;
    const int seg = b % a.segs;
    const int n = b / a.segs;
    const int tid = threadIdx.x, quad = tid & 7, col = tid >> 3;
    const int ys = seg * a.RS, ye = min(a.H, ys + a.RS);
    const int x0 = strip * OUTC;
    const int gj = x0 - P + col;                                   // image column of this thread (may lie outside)
    const bool out_col = col >= P && col < COLS - P && gj < a.W;   // columns this thread produces output for
    const int xr = reflect_idx(gj, a.W);                           // reflect-extended source column
    const bool d_col = gj >= 0 && gj < a.W;
    const size_t img = (size_t)n * a.H * a.W;
    const int coff = cchunk * 32 + quad * 4;

    make_kernel1d(a.sigma, K, gs);
    __syncthreads();
    float g[K];
#pragma unroll
    for (int t = 0; t < K; ++t) g[t] = gs[t];
    float wcol[K];
    if (MODE == 1) {
#pragma unroll
        for (int t = 0; t < K; ++t) wcol[t] = fold_weight(gs, gj, a.W, t);
    }

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 Uw[K], Dw[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { Uw[t] = zero4; Dw[t] = zero4; }
    float accA[K], accB[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { accA[t] = 0.f; accB[t] = 0.f; }

    auto load_x = [&](int yp) -> float4 {
        const int r = reflect_idx(yp, a.H);
        return *reinterpret_cast<const float4*>(a.x + ((img + (size_t)r * a.W + xr) * a.C + coff));
    };
    auto load_d = [&](int q) -> float4 {
        if (q < 0 || q >= a.H || !d_col) return zero4;
        return *reinterpret_cast<const float4*>(a.dy + ((img + (size_t)q * a.W + gj) * a.C + coff));
    };

    // march over extended rows yp; backward starts 4 rows earlier to fill the dy window (rows yp+4 are loaded at step yp)
    const int y_first = MODE == 1 ? ys - 2 * P : ys - P, y_last = ye - 1 + P;
    float4 nx = zero4, nd = zero4;
    if (y_first >= ys - P) nx = load_x(y_first);
    if (MODE == 1) nd = load_d(y_first + P);
    for (int yp = y_first; yp <= y_last; ++yp) {
        const float4 cx = nx, cd = nd;
        if (yp + 1 <= y_last) {                                    // prefetch the next step's rows
            if (yp + 1 >= ys - P) nx = load_x(yp + 1);
            if (MODE == 1) nd = load_d(yp + 1 + P);
        }
        if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < K - 1; ++t) Dw[t] = Dw[t + 1];
            Dw[K - 1] = cd;                                        // Dw[t] = dy[yp - 4 + t]
        }
        if (yp < ys - P) continue;                                 // (uniform) dy-window warm-up steps
        const int buf = yp & 1;
        Xs[buf][tid] = cx;
        __syncthreads();
        float4 xs[K];
        float4 h = zero4;
        if (col >= P && col < COLS - P) {
#pragma unroll
            for (int t = 0; t < K; ++t) {
                xs[t] = Xs[buf][tid + (t - P) * 8];
                h = f4_fma(g[t], xs[t], h);
            }
        } else {
#pragma unroll
            for (int t = 0; t < K; ++t) xs[t] = zero4;
        }
#pragma unroll
        for (int t = 0; t < K - 1; ++t) Uw[t] = Uw[t + 1];
        Uw[K - 1] = h;                                             // Uw[t] = U[yp - 8 + t]
        const int yo = yp - P;                                     // output row completed by this step
        const bool row_out = yo >= ys && yo < ye;
        if (MODE == 0) {
            if (row_out && out_col) {
                float4 o = zero4;
#pragma unroll
                for (int t = 0; t < K; ++t) o = f4_fma(g[t], Uw[t], o);
                *reinterpret_cast<float4*>(a.y + ((img + (size_t)yo * a.W + gj) * a.C + coff)) = o;
            }
            continue;
        }
        // ---- backward ----------------------------------------------------------------------------------------------
        if (a.part && row_out && out_col) {                        // dg_a += <dy[yo], U[yo + a - 4]>
#pragma unroll
            for (int t = 0; t < K; ++t) accA[t] = f4_dot(Dw[0], Uw[t], accA[t]);
        }
        const bool v_row = yp >= ys && yp < ye;                    // (uniform) image row i = yp of this segment
        if (!v_row) continue;
        float4 v = zero4;
        if (yp >= P + 1 && yp <= a.H - 2 - P) {                    // interior row: no fold
#pragma unroll
            for (int t = 0; t < K; ++t) v = f4_fma(g[K - 1 - t], Dw[t], v);
        } else {
#pragma unroll
            for (int t = 0; t < K; ++t) v = f4_fma(fold_weight(gs, yp, a.H, t), Dw[t], v);
        }
        if (a.part && out_col) {                                   // dg_a += <V'[i], Xe[i, j + a - 4]>
#pragma unroll
            for (int t = 0; t < K; ++t) accB[t] = f4_dot(v, xs[t], accB[t]);
        }
        if (a.dx) {
            Vs[buf][tid] = v;
            __syncthreads();
            if (out_col) {
                float4 o = zero4;
#pragma unroll
                for (int t = 0; t < K; ++t) o = f4_fma(wcol[t], Vs[buf][tid + (t - P) * 8], o);
                const size_t oo = (img + (size_t)yp * a.W + gj) * a.C + coff;
                if (a.dx_add) {
                    const float4 q = *reinterpret_cast<const float4*>(a.dx_add + oo);
                    o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
                }
                *reinterpret_cast<float4*>(a.dx + oo) = o;
            }
        }
    }
    if (MODE == 1 && a.part) {
        const int lane = tid & 63, wid = tid >> 6;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const float s = wave_sum(accA[t] + accB[t]);
            if (lane == 0) red[t][wid] = s;
        }
        __syncthreads();
        if (tid < K) {
            float s = 0.f;
            for (int w = 0; w < NT / 64; ++w) s += red[tid][w];
            a.part[(size_t)blockIdx.x * K + tid] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, round 3: the same march with the sigma gradient formed DIRECTLY instead of through nine tap gradients.
// With c_a = d g_a / d sigma (sum_a c_a = 0; g = normalised Gaussian taps):
//   d sigma = sum_a c_a dg_a = < C_v^T dY , G_h Xe >  +  < G_v^T dY , C_h Xe >
//   first term, per extended row r (the row-blurred row h = (G_h Xe)[r] this step produces anyway):
//       w1[r] = sum_t c[8 - t] dy[r - 4 + t]      over the dy rows that belong to THIS workgroup's segment
//   second term, per image row i: b = sum_t c[t] Xe[i, j - 4 + t]  (next to h, from the same LDS reads), v = V'[i] as before
// Both need one extra 9-tap filter and ONE dot product per element where the round-2 kernel kept nine dot-product accumulators
// against a 9-row register window of U (2 x 9 accumulators + 36 window registers + the 9 staged x quads): 200 -> ~120 registers,
// two workgroups per CU instead of one; one partial per workgroup instead of nine.
// ---------------------------------------------------------------------------------------------------------------
template <int COLS, int WPE>      // WPE = waves per SIMD the register allocation is held to (4: two workgroups per CU; 2: one)
__global__ __launch_bounds__(COLS * 8, WPE) void blur9_stream_bwd_kernel(StreamArgs a) {
    constexpr int K = 9, P = 4, NT = COLS * 8, OUTC = COLS - 2 * P;
    __shared__ float4 Xs[2][COLS * 8];
    __shared__ float4 Vs[2][COLS * 8];
    __shared__ float gs[32];          // [0..8] taps g, [16..24] their sigma derivative c
    __shared__ float red[NT / 64];

    int b = blockIdx.x;
    const int cchunk = b % a.cchunks; b /= a.cchunks;
    const int strip = b % a.strips; b /= a.strips;
    const int seg = b % a.segs;
    const int n = b / a.segs;
    const int tid = threadIdx.x, quad = tid & 7, col = tid >> 3;
    const int ys = seg * a.RS, ye = min(a.H, ys + a.RS);
    const int x0 = strip * OUTC;
    const int gj = x0 - P + col;
    const bool in_strip = col >= P && col < COLS - P;
    const bool out_col = in_strip && gj < a.W;
    const int xr = reflect_idx(gj, a.W);
    const bool d_col = gj >= 0 && gj < a.W;
    const size_t img = (size_t)n * a.H * a.W;
    const int coff = cchunk * 32 + quad * 4;

    make_kernel1d(a.sigma, K, gs);
    if (tid == 0) {                   // c_a = g_a (t_a^2 - sum_j g_j t_j^2) / sigma^3, in double from the double taps
        const double sg = a.sigma[0];
        double p[K], S = 0.0, m2 = 0.0;
        for (int j = 0; j < K; ++j) { const double t = j - 4.0; p[j] = exp(-0.5 * (t / sg) * (t / sg)); S += p[j]; }
        for (int j = 0; j < K; ++j) { const double t = j - 4.0; m2 += p[j] / S * t * t; }
        for (int j = 0; j < K; ++j) { const double t = j - 4.0; gs[16 + j] = (float)(p[j] / S * (t * t - m2) / (sg * sg * sg)); }
    }
    __syncthreads();
    float g[K], c[K], wcol[K];        // g, c: wave-uniform -> scalar registers
#pragma unroll
    for (int t = 0; t < K; ++t) {
        g[t] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, gs[t])));
        c[t] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, gs[16 + t])));
        wcol[t] = fold_weight(gs, gj, a.W, t);
    }

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 Dw[K];
#pragma unroll
    for (int t = 0; t < K; ++t) Dw[t] = zero4;
    float acc = 0.f;

    auto load_x = [&](int yp) -> float4 {
        const int r = reflect_idx(yp, a.H);
        return *reinterpret_cast<const float4*>(a.x + ((img + (size_t)r * a.W + xr) * a.C + coff));
    };
    auto load_d = [&](int q) -> float4 {
        if (q < 0 || q >= a.H || !d_col) return zero4;
        return *reinterpret_cast<const float4*>(a.dy + ((img + (size_t)q * a.W + gj) * a.C + coff));
    };

    const int y_first = ys - 2 * P, y_last = ye - 1 + P;
    float4 nx = zero4, nd = load_d(y_first + P);
    for (int yp = y_first; yp <= y_last; ++yp) {
        const float4 cx = nx, cd = nd;
        if (yp + 1 <= y_last) {
            if (yp + 1 >= ys - P) nx = load_x(yp + 1);
            nd = load_d(yp + 1 + P);
        }
#pragma unroll
        for (int t = 0; t < K - 1; ++t) Dw[t] = Dw[t + 1];
        Dw[K - 1] = cd;                                            // Dw[t] = dy[yp - 4 + t]
        if (yp < ys - P) continue;                                 // (uniform) dy-window warm-up steps
        const int buf = yp & 1;
        Xs[buf][tid] = cx;
        __syncthreads();
        const bool v_row = yp >= ys && yp < ye;                    // (uniform) image row i = yp of this segment
        float4 h = zero4, bq = zero4;
        if (in_strip) {
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const float4 xv = Xs[buf][tid + (t - P) * 8];
                h = f4_fma(g[t], xv, h);
                bq = f4_fma(c[t], xv, bq);
            }
        }
        if (a.part && out_col) {
            // first term: dy rows yp - 4 + t of this segment only (the window also holds the neighbours' rows for V')
            float4 w1 = zero4;
            if (yp - P >= ys && yp + P < ye) {                     // (uniform) interior step: the whole window belongs to the segment
#pragma unroll
                for (int t = 0; t < K; ++t) w1 = f4_fma(c[K - 1 - t], Dw[t], w1);
            } else {
#pragma unroll
                for (int t = 0; t < K; ++t) {
                    const int q = yp - P + t;
                    w1 = f4_fma((q >= ys && q < ye) ? c[K - 1 - t] : 0.f, Dw[t], w1);
                }
            }
            acc = f4_dot(w1, h, acc);
        }
        if (!v_row) continue;
        float4 v = zero4;
        if (yp >= P + 1 && yp <= a.H - 2 - P) {                    // interior row: no fold
#pragma unroll
            for (int t = 0; t < K; ++t) v = f4_fma(g[K - 1 - t], Dw[t], v);
        } else {
#pragma unroll
            for (int t = 0; t < K; ++t) v = f4_fma(fold_weight(gs, yp, a.H, t), Dw[t], v);
        }
        if (a.part && out_col) acc = f4_dot(v, bq, acc);           // second term
        if (a.dx) {
            Vs[buf][tid] = v;
            __syncthreads();
            if (out_col) {
                float4 o = zero4;
#pragma unroll
                for (int t = 0; t < K; ++t) o = f4_fma(wcol[t], Vs[buf][tid + (t - P) * 8], o);
                const size_t oo = (img + (size_t)yp * a.W + gj) * a.C + coff;
                if (a.dx_add) {
                    const float4 q = *reinterpret_cast<const float4*>(a.dx_add + oo);
                    o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
                }
                *reinterpret_cast<float4*>(a.dx + oo) = o;
            }
        }
    }
    if (a.part) {
        const int lane = tid & 63, wid = tid >> 6;
        const float sw = wave_sum(acc);
        __syncthreads();
        if (lane == 0) red[wid] = sw;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < NT / 64; ++w) tot += red[w];
            a.part[blockIdx.x] = tot;
        }
    }
}

constexpr int STREAM_COLS = 64;

bool stream_ok(int ksize, int N, int H, int W, int C) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("FAVAE_BLUR_STREAM"); off = (e && e[0] == '0') ? 1 : 0; }
    return !off && ksize == 9 && C % 32 == 0 && H >= 6 && W >= 6 && (size_t)N * H * W * C < ((size_t)1 << 40);
}

void stream_plan(int N, int H, int W, int C, StreamArgs& a) {
    a.N = N; a.H = H; a.W = W; a.C = C;
    a.RS = H >= 128 ? 64 : (H >= 32 ? 32 : 16);
    if (a.RS > H) a.RS = H;
    a.segs = (H + a.RS - 1) / a.RS;
    a.strips = (W + (STREAM_COLS - 8) - 1) / (STREAM_COLS - 8);
    a.cchunks = C / 32;
}

}  // namespace
