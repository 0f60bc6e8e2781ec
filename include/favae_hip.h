/*
 * libfavae_hip -- C ABI of the MI355X (gfx950) FA-VAE training-step hot path.
 *
 * The reference (oppo-us-research/FA-VAE) is 100 % Python and has no FFI of its own; its "kernels" are the
 * ATen calls its modules issue.  Each entry point below replaces one such call pattern; the comment above each
 * declaration cites the reference site (paths relative to the reference repository root).  INTEGRATION.md shows
 * the ctypes stubs a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise; the caller owns all memory (PyTorch's
 *     allocator in practice); the library never allocates device memory and is re-entrant per stream.  Process-global state is
 *     limited to two explicit switches: the conv arithmetic mode (favae_set_conv_mode, default from FAVAE_CONV_MODE) and the
 *     launch profiler (favae_prof_*, off by default);
 *   - activations are NHWC ("channels last": N, H, W, C with C fastest); conv weights are OHWI
 *     ([Cout][KH][KW][Cin], i.e. a torch (Cout,Cin,KH,KW) tensor in channels_last memory format);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - workspaces are caller-allocated; every *_workspace() query returns the bytes needed;
 *   - return value: FAVAE_OK (0) or an error code; nothing is printed.
 */
#ifndef FAVAE_HIP_H
#define FAVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* favae_stream_t;

enum {
    FAVAE_OK = 0,
    FAVAE_ERR_BAD_ARG = 1,
    FAVAE_ERR_LAUNCH = 2,
    FAVAE_ERR_UNSUPPORTED = 3,
    FAVAE_ERR_WORKSPACE = 4
};

/* library/ABI version, bumped on any signature change */
int favae_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Convolution family: implicit GEMM, fp32-grade.  Default arithmetic: the operands are split into two scaled fp16 planes and the
 * fp32 product is formed from 3 v_mfma_f32_32x32x16_f16 products with fp32 accumulation ("h3");
 * favae_set_conv_mode selects three bf16 planes / 6 products ("b6"), the exact fp32 MFMA v_mfma_f32_32x32x2_f32 ("fp32", also the
 * fallback for shapes the split tiles do not cover), or ONE fp16 plane ("h1": 16-bit mixed precision, not fp32-grade).
 * Replaces: nn.Conv2d 3x3/1x1 inside ResnetBlock / NonResnetBlock (models/codec.py:38-46,50,65-73,77), conv_in /
 * final (:140,170-175,447-450), Downsample = F.pad(0,1,0,1)+conv s2 (:26-29), Upsample = nearest x2 + conv (:17-18),
 * the packed in/out projections of nn.MultiheadAttention (:92), the 4x4 convs of the Discriminator
 * (models/discriminator.py:198-213) -- with the preceding GroupNorm(+SiLU) / BatchNorm+LeakyReLU applied on the
 * fly to the input tile, and bias + residual add fused in the epilogue.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N, Hin, Win, Cin;       /* input  (N,Hin,Win,Cin)  NHWC  */
    int32_t Hout, Wout, Cout;       /* output (N,Hout,Wout,Cout) NHWC */
    int32_t KH, KW, stride, pad;    /* pad = top/left zero padding; bottom/right padding is implied by Hout/Wout */
    int32_t gather;                 /* 0 plain | 1 input is nearest-upsampled x2 on the fly | 2 input is zero-dilated x2
                                       (transposed conv = data gradient of a stride-2 conv) */
    int32_t act;                    /* activation applied after the affine input transform: 0 none | 1 SiLU | 2 LeakyReLU(0.2) */
    int32_t affine_per_image;       /* 1: scale/shift are [N][Cin] (GroupNorm); 0: [Cin] shared by the batch (BatchNorm) */
    /* Sub-grid addressing for the phase decomposition of the Upsample convs (all zero = an ordinary conv).  nearest x2 followed
     * by a 3x3 conv equals, for each output parity (py,px), a 2x2 conv of the low-resolution input with summed weights
     * (favae_upsample_weights): 16 instead of 36 multiply-adds per low-resolution pixel.  lat_step = 2 puts ONE side of the
     * conv on every second pixel (offset lat_oh, lat_ow) of a tensor of twice the stated height/width: lat_side = 1 the output
     * (also resid, and dy in favae_conv_wgrad), lat_side = 2 the input.  Only the split-precision kernels implement it. */
    int32_t lat_step, lat_side, lat_oh, lat_ow;
    int32_t pad_dw;                 /* left zero padding = pad + pad_dw (top padding = pad) */
    int32_t w_rec_offset;           /* favae_conv_fwd_split: byte offset of this conv's records behind the header of wsplit */
} favae_conv_desc;

#define FAVAE_GATHER_PLAIN 0
#define FAVAE_GATHER_UPSAMPLE2 1
#define FAVAE_GATHER_DILATE2 2
#define FAVAE_ACT_NONE 0
#define FAVAE_ACT_SILU 1
#define FAVAE_ACT_LEAKY02 2
#define FAVAE_ACT_RELU 3      /* VGG16 feature stack of LPIPS (losses/lpips.py:74-110) */

/* y = conv(T(x), w) + bias + resid, T(x) = act(x*scale+shift) (scale==NULL -> T = identity).  bias, resid, scale,
 * shift may be NULL.  The data gradient of a stride-1 conv is this same call on dy with favae_weight_flip()'d weights. */
int favae_conv_fwd(const favae_conv_desc* d, const float* x, const float* w, const float* bias, const float* resid,
                   const float* scale, const float* shift, float* y, favae_stream_t stream);

/* Split-precision path (fp32 convolution on the 16-bit matrix pipe).
 * favae_conv_wants_split_weights() returns the number of planes the library would run this conv with: 0 (fp32-MFMA kernels,
 * use favae_conv_fwd), 3 (three bf16 planes, six products) or 2 (two scaled fp16 planes, three products).  For 2 or 3 the
 * caller splits the OHWI weights once with favae_split_weights() into a buffer of favae_split_weights_bytes() and calls
 * favae_conv_fwd_split() -- same semantics as favae_conv_fwd.  With planes == 2 the kernels also need `x_absmax`: a device
 * float holding an upper bound of |T(x)| (favae_absmax() of x for T = identity; the bound favae_gn_stats() emits otherwise);
 * planes == 3 has no range restriction and ignores it.  Both schemes keep fp32-grade products (tools/conv_accuracy.py).
 * planes == 1 ("h1") is the 16-bit mixed-precision mode: ONE scaled fp16 plane per operand (11-bit significand), fp32
 * accumulation, same range arguments as planes == 2 -- what `accelerate launch --mixed_precision fp16|bf16` turns the reference's
 * F.conv2d calls into (favae_scripts/train_favae.py:240 builds the Accelerator; BASELINE config 5).  Not fp32-grade.
 * planes == 4 ("b1") is the bf16 form of it: ONE bf16 plane per operand (round to nearest even, 8-bit significand, the fp32
 * exponent: no range arguments, x_absmax may be NULL), v_mfma_f32_32x32x16_bf16, fp32 accumulation -- the arithmetic BASELINE
 * configs[4] names ("bf16").  4 is a scheme id, not a plane count: records are 8 bytes per 4 weights as for planes == 1.
 * favae_set_conv_mode(planes) selects the scheme for subsequent calls (0 = fp32-MFMA kernels, 1, 2 = default, 3, 4); it overrides
 * the FAVAE_CONV_MODE environment variable (fp32 | h1 | h3 | b6 | b1) read on first use.  Process-wide, not thread-safe against
 * concurrent launches. */
int favae_set_conv_mode(int planes);
int favae_get_conv_mode(void);
int favae_conv_wants_split_weights(const favae_conv_desc* d, int has_affine);
size_t favae_split_weights_bytes(int64_t n, int planes);
int favae_split_weights(const float* in, void* out, int64_t n, int planes, favae_stream_t stream);
/* The same with max |in| supplied by the caller (device float), so that no reduction pass runs per conv call; and the pass that
 * produces such maxima for EVERY weight tensor of a model at once: out[s] = max |x[seg_off[s] .. seg_off[s+1])| over a flat parameter
 * buffer (favae_step.TrainStep refreshes them once per optimizer step).  seg_off [nseg + 1], chunk_seg / chunk_first [nchunks]: device
 * arrays; chunk c covers elements chunk_first[c] .. +4096 (clipped to its segment chunk_seg[c]); chunks never straddle segments. */
int favae_split_weights_amax(const float* in, void* out, int64_t n, int planes, const float* amax, favae_stream_t stream);
int favae_segment_absmax(const float* x, const int64_t* seg_off, int nseg, const int* chunk_seg, const int64_t* chunk_first, int nchunks,
                         float* out, favae_stream_t stream);
/* Winograd F(2x2, 3x3) for the dense 3x3 stride-1 convs of the h3 scheme (csrc/conv_wino.h): the matrix pipe does 4/9 of the multiplies
 * of the direct kernel; transforms in fp32, products on two scaled fp16 planes, fp32 accumulation and output transform.
 * favae_conv_wino_ok(d, has_affine) = 1 when favae_conv_fwd_split / _stats / favae_conv_dgrad_gnbwd run d on that kernel: the caller
 * then passes records made by favae_wino_weights (favae_wino_weights_bytes(Cout, Cin) bytes) and planes = 2 | FAVAE_PLANES_WINO; the
 * tile counts of favae_conv_stats_tiles / favae_conv_gnbwd_tiles are that kernel's (16 x 16 pixels; 16 x 8 where the output channels
 * tile by 128: the kernel then runs 16 x 8 pixels x 128 channels per workgroup -- each tile's GroupNorm / SiLU staging, input transform
 * and operand split are done once for 128 output channels instead of once per 64 -- same result bits, same records;
 * FAVAE_WINO_WIDE=0 / favae_set_wino_wide(0) keeps the 16 x 16 x 64 tiling everywhere).  favae_wino_weights: w = OHWI fp32
 * [Cout][3][3][Cin]; flip = 0 -> records of the forward conv, flip = 1 -> of its data gradient (Cin outputs, taps flipped);
 * amax = device float max|w| or NULL (computed).  FAVAE_WINO=0 in the environment keeps every conv on the direct kernels. */
#define FAVAE_PLANES_WINO 0x100
int favae_conv_wino_ok(const favae_conv_desc* d, int has_affine);
int favae_set_wino(int on);            /* run-time override of FAVAE_WINO; returns the previous setting */
int favae_set_wino_wide(int on);       /* run-time override of FAVAE_WINO_WIDE; returns the previous setting */
int favae_get_wino(void);              /* the current setting, no side effect */
/* One-plane Winograd (ABI 20): in the 16-bit mixed-precision modes (favae_set_conv_mode 1 = h1: ONE scaled fp16 plane; 4 = b1: ONE bf16
 * plane; BASELINE configs[4], train_favae.py:239-240) favae_conv_wino_ok is 1 as well -- h1: wherever the kernel tiles (Cout % 64 == 0),
 * b1: where the 16 x 8 x 128 tiling applies (Cout % 128 == 0); FAVAE_WINO1=0: never -- the same kernel with one plane and ONE product
 * per block.  The caller then passes
 * planes = 1 | FAVAE_PLANES_WINO with the ordinary records (their head plane is read), or 4 | FAVAE_PLANES_WINO with records made by
 * favae_wino_weights with bit 2 of `flip` set (4 forward, 5 data gradient: bf16 head plane, unscaled; no operand bound is asked for).
 * In these two modes the caller may ALSO keep a conv with more than 64 output channels on the direct one-plane kernel (planes = 1 or 4
 * without the flag) call by call: B^T d B before the rounding costs 1.5-1.7 x the direct kernel's error, and the Python host keeps the b1 forward convs direct.
 * The statistics / GroupNorm-backward variants (favae_conv_fwd_split_stats, favae_conv_dgrad_gnbwd) accept that direct call only where
 * the direct kernel's 16 x 8 tile grid IS the grid favae_conv_stats_tiles / favae_conv_gnbwd_tiles report, i.e. where the wide Winograd
 * tiling applies (Cout % 128 == 0 and FAVAE_WINO_WIDE on); elsewhere they return FAVAE_ERR_UNSUPPORTED instead of writing partial sums
 * past the buffer the caller sized from the tile count. */
/* Winograd F(4x4, 3x3) (csrc/conv_wino4.h, ABI 18): 36 instead of 64 multiplies per 16 outputs -- 0.56 x the matrix work and operand
 * splitting of the F(2x2) kernel, at 2.3e-6 rms (F(2x2): 3.6e-7) of the output range per conv.  Meant for results no codebook index
 * depends on: the data gradients (autograd of models/codec.py:38-46) and, by the caller's choice, decoder layers.
 * favae_conv_wino4_ok(d, has_affine) = 1 when d ALSO tiles into that kernel (favae_conv_wino_ok plus W % 32 == 0, Cin % 64 == 0); the
 * caller may then pass records made by favae_wino_weights with bit 1 of `flip` set (flip = 2 forward, 3 data gradient;
 * favae_wino4_weights_bytes(Cout, Cin) bytes; favae_wino_job.flip likewise) and planes = 2 | FAVAE_PLANES_WINO | FAVAE_PLANES_WINO4 to
 * favae_conv_fwd_split / _stats / favae_conv_dgrad_gnbwd.  The tile grid of the partial sums stays the F(2x2) kernel's (16 x 16 pixels).
 * FAVAE_WINO4=0 in the environment (favae_set_wino4(0)) makes favae_conv_wino4_ok return 0 for every shape. */
#define FAVAE_PLANES_WINO4 0x200
/* bf16 activation STORAGE (ABI 21): with this flag on `planes` (scheme 4 = b1 only) favae_conv_fwd_split / _stats / favae_conv_dgrad_gnbwd /
 * favae_conv_wgrad(_slabs) (flag on the descriptor's `act` there) take x, resid (and the GroupNorm input of the data-gradient epilogue,
 * and dy of the weight gradient) and write
 * y as bf16 tensors: the same element counts at two bytes each, rounded to nearest even on store, widened exactly on load; products,
 * accumulation, statistics and partial sums stay fp32 / fp64.  favae_conv_bf16io_ok(d, has_affine, kind) says whether the kernel that
 * would run `d` has the bf16 instantiation (kind 0 forward, 1 data gradient with the GroupNorm-backward epilogue, 2 weight gradient);
 * otherwise the caller converts with favae_cast_bf16 / favae_cast_f32 and uses the fp32 call. */
#define FAVAE_PLANES_BF16IO 0x400
int favae_conv_bf16io_ok(const favae_conv_desc* d, int has_affine, int kind);
/* element-wise conversion passes between the two storage types (n elements, 8-byte aligned) */
int favae_cast_bf16(const float* in, void* out, int64_t n, favae_stream_t stream);
int favae_cast_f32(const void* in, float* out, int64_t n, favae_stream_t stream);
int favae_conv_wino4_ok(const favae_conv_desc* d, int has_affine);
int favae_set_wino4(int on);           /* returns the previous setting */
size_t favae_wino4_weights_bytes(int Cout, int Cin);
/* Zero arena: the max|x| outputs of favae_absmax / favae_colsum / favae_conv_fwd_split_stats / favae_attn_bwd_point are atomicMax targets
 * that must start at zero; each call zeroes its own (one 4-byte memset launch) UNLESS the pointer lies inside [p, p + bytes), a range the
 * caller keeps zero for such targets (favae_step.TrainStep: one memset per step).  bytes = 0 removes the arena. */
int favae_set_zero_arena(const void* p, size_t bytes);
/* Records of many weight tensors in ONE launch (favae_step.TrainStep: every dense 3x3 conv weight of the model, both directions, after each
 * optimizer step).  jobs / block_job are DEVICE arrays: job j = {w, out = header + records (16-byte aligned), amax = device float max|w|,
 * Cout, Cin, flip, block0}; block_job[b] = the job of block b; job j owns the blocks block0 .. block0 + ceil(Cout Cin / 2048) - 1. */
typedef struct favae_wino_job {
    const float* w;
    void* out;
    const float* amax;
    int32_t Cout, Cin, flip, block0;
} favae_wino_job;
int favae_wino_weights_grouped(const void* jobs, const int* block_job, int nblocks, favae_stream_t stream);
size_t favae_wino_weights_bytes(int Cout, int Cin);
int favae_wino_weights(const float* w, void* out, int Cout, int Cin, int flip, const float* amax, favae_stream_t stream);
int favae_conv_fwd_split(const favae_conv_desc* d, const float* x, const void* wsplit, int planes, const float* x_absmax,
                         const float* bias, const float* resid, const float* scale, const float* shift, float* y,
                         favae_stream_t stream);
/* out[0] = max |x[i]| (device scalar) */
int favae_absmax(const float* x, int64_t n, float* out, favae_stream_t stream);

/* dw[co][kh][kw][ci] = sum_{n,oh,ow} dy[n,oh,ow,co] * T(x)[gathered (n,oh,ow,kh,kw), ci]   (split-K, deterministic:
 * partial slabs in `ws`, summed in a fixed order).  autograd's convolution_backward weight path. */
size_t favae_conv_wgrad_workspace(const favae_conv_desc* d);
/* x_absmax / dy_absmax: device bounds of |T(x)| and |dy| as for favae_conv_fwd_split (both or neither; NULL -> the bf16
 * planes or the fp32-MFMA kernels run). */
int favae_conv_wgrad(const favae_conv_desc* d, const float* x, const float* dy, const float* scale, const float* shift,
                     const float* x_absmax, const float* dy_absmax, float* dw, int accumulate, void* ws, size_t ws_bytes,
                     favae_stream_t stream);

/* The same weight gradient with the slab reduction left to the caller: the kernel's partial slabs stay in `ws`
 * ([*slabs][Cout][KH][KW][Cin] floats; *slabs is written on the host) until favae_reduce_slabs_grouped sums the slabs of MANY layers
 * in one launch -- per layer the reduction is a latency-bound 40-60 us launch (214 of them per training step), grouped it is one
 * bandwidth-bound pass at the end of backward.  Same summation order as favae_conv_wgrad: bit-identical gradients. */
int favae_conv_wgrad_slabs(const favae_conv_desc* d, const float* x, const float* dy, const float* scale, const float* shift,
                           const float* x_absmax, const float* dy_absmax, void* ws, size_t ws_bytes, int* slabs,
                           favae_stream_t stream);
typedef struct favae_reduce_job {
    const float* part;              /* [slabs][n] */
    float* out;                     /* [n]: out[i] (+)= sum_z part[z][i], z ascending in four interleaved partial sums */
    int64_t n;
    int32_t slabs;
    int32_t accumulate;
} favae_reduce_job;
#define FAVAE_REDUCE_JOBS_MAX 96
/* jobs: host array (any length; launched in groups of FAVAE_REDUCE_JOBS_MAX).  The `out` ranges of one call must be disjoint. */
int favae_reduce_slabs_grouped(const favae_reduce_job* jobs, int njobs, favae_stream_t stream);

/* wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci]  (weights of the data-gradient convolution) */
int favae_weight_flip(const float* w, float* wt, int Cout, int KH, int KW, int Cin, favae_stream_t stream);
/* the same, written directly as pre-split records for favae_conv_fwd_split (favae_split_weights_bytes(Cout*KH*KW*Cin, planes)
 * bytes).  absmax_src: device float holding max|w| -- the first float of the forward's record buffer -- so that the data
 * gradient needs neither a flipped fp32 copy nor a second maximum reduction (required for planes == 2).  Cout % 4 == 0. */
int favae_weight_flip_split(const float* w, void* out, int Cout, int KH, int KW, int Cin, int planes, const float* absmax_src,
                            favae_stream_t stream);

/* Data gradient of the Downsample conv (3x3, stride 2, padding bottom/right only, models/codec.py:21-31) by output parity:
 * dx[2u] = dy[u] w[0] + dy[u-1] w[2], dx[2u+1] = dy[u] w[1] along each axis -> four convs over dy with 2x2, 2x1, 1x2 and 1x1
 * kernels whose outputs interleave (favae_conv_desc.lat_side = 1), 9 instead of 36 taps per 2x2 block of dx.  Writes the four
 * weight sets [Cin][a][b][Cout] as pre-split records one after the other behind one header (record offsets: 0, 4, 6, 8 times
 * Cin*Cout/4 records); out: favae_split_weights_bytes(9*Cin*Cout, planes) bytes; absmax_src as in favae_weight_flip_split. */
int favae_downsample_dgrad_weights(const float* w, void* out, int Cout, int Cin, int planes, const float* absmax_src,
                                   favae_stream_t stream);

/* Upsample (nearest x2 + 3x3 conv, models/codec.py:11-18) as four phase convs: weff[py*2+px][co][a][b][ci] (a,b in {0,1}) =
 * sum of the w[co][kh][kw][ci] whose tap lands on the same low-resolution pixel (rows: py=0: {0},{1,2}; py=1: {0,1},{2}).
 * favae_upsample_wgrad_fold is the adjoint: dw[co][kh][kw][ci] (+)= sum of the dweff entries that contain that tap. */
int favae_upsample_weights(const float* w, float* weff, int Cout, int Cin, favae_stream_t stream);
int favae_upsample_wgrad_fold(const float* dweff, float* dw, int Cout, int Cin, int accumulate, favae_stream_t stream);
/* 1 if the four phase convs of an Upsample with these sizes run on the split-precision kernels (else use gather = 1) */
int favae_conv_subpixel_ok(int N, int H, int W, int Cin, int Cout);

/* out[c] (+)= sum_m a[m][c]  (bias gradient; M rows of C) ; deterministic two-stage.  `accumulate` != 0 adds to `out`
 * (gradients written straight into a pre-zeroed flat gradient buffer, as favae_conv_wgrad / favae_gn_act_bwd do). */
size_t favae_colsum_workspace(int64_t M, int C);
/* absmax_out (optional device float) receives max |a| -- the range of dy the fp16 split-precision conv kernels need, read
 * off the pass that already streams dy for the bias gradient. */
int favae_colsum(const float* a, float* out, int64_t M, int C, int accumulate, float* absmax_out, void* ws, size_t ws_bytes,
                 favae_stream_t stream);

/* adjoint of nearest x2 upsampling: dx[n,h,w,c] = sum of the 2x2 children of du (N,2H,2W,C) */
int favae_upsample2x_bwd(const float* du, float* dx, int N, int H, int W, int C, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * GroupNorm (+SiLU) statistics and backward.  Replaces nn.GroupNorm(32,C)/nn.SiLU (models/codec.py:39-40,42-43,
 * 66-67,69-70,90,170-171,447-448) and, with G == C and N folded into HW, the batch statistics of nn.BatchNorm2d
 * (models/discriminator.py:207).
 * ---------------------------------------------------------------------------------------------------------- */
/* mean/rstd: [N][G] (biased variance, eps inside the sqrt); scale[n][c] = rstd*gamma[c], shift[n][c] = beta[c] -
 * mean*rstd*gamma[c] -- the per-(image,channel) affine the conv kernels apply on load. */
/* absmax_out (optional device float): upper bound of |act(x*scale+shift)| over the whole tensor, max_c |gamma_c| sqrt(group
 * size) + |beta_c| -- the operand range the fp16 split-precision conv kernels need (favae_conv_fwd_split). */
size_t favae_gn_workspace(int N, int64_t HW, int C);
int favae_gn_stats(const float* x, const float* gamma, const float* beta, int N, int64_t HW, int C, int G, float eps,
                   float* mean, float* rstd, float* scale, float* shift, float* absmax_out, void* ws, size_t ws_bytes,
                   favae_stream_t stream);
/* the same statistics of a tensor STORED as bf16 (x: bf16 elements, C % 4 == 0, 8-byte aligned): bf16 activation storage, ABI 21 */
int favae_gn_stats_bf16(const void* x, const float* gamma, const float* beta, int N, int64_t HW, int C, int G, float eps,
                        float* mean, float* rstd, float* scale, float* shift, float* absmax_out, void* ws, size_t ws_bytes,
                        favae_stream_t stream);

/* Given da = dL/d act(GN(x)): dx, dgamma[C], dbeta[C].  act as in favae_conv_desc.  If `dx_add` != NULL it is
 * added to dx (fused skip-connection gradient).  dx may alias da. */
int favae_gn_act_bwd(const float* da, const float* x, const float* gamma, const float* beta, const float* mean,
                     const float* rstd, int N, int64_t HW, int C, int G, int act, const float* dx_add, float* dx,
                     float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, favae_stream_t stream);

/* The same backward (tiles == 0: as favae_gn_act_bwd; tiles > 0: as favae_gn_act_bwd_tiles, pass 1 taken from the data-gradient
 * conv's epilogue) whose apply pass ALSO emits two by-products of the tensor dx it writes: per-block column sums
 * cs_part[favae_gn_bwd_colsum_blocks(N, HW, C)][C] and max |dx| (cs_absmax, one device float).  dx is the output gradient `dy` of the
 * conv in front of this GroupNorm (models/codec.py:38-46: conv -> GroupNorm -> SiLU -> conv), whose bias gradient is
 * favae_colsum_finish(cs_part) and whose fp16 operand range is cs_absmax -- favae_colsum's extra read of the tensor goes away.
 * favae_gn_bwd_colsum_blocks returns 0 when the shape does not run the row-organised apply pass (C % 4, C > 1024, image >= 2 GiB). */
int favae_gn_bwd_colsum_blocks(int N, int64_t HW, int C);
int favae_gn_act_bwd_colsum(const float* da, const float* x, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, int N, int64_t HW, int C, int G, int act, const float* dx_add, float* dx,
                            float* dgamma, float* dbeta, int accumulate, int tiles, void* ws, size_t ws_bytes, float* cs_part,
                            float* cs_absmax, favae_stream_t stream);
/* out[c] (+)= sum_b part[b][c], b ascending (second stage of favae_colsum for partials produced by another pass) */
int favae_colsum_finish(const float* part, int blocks, int C, float* out, int accumulate, favae_stream_t stream);

/* BatchNorm2d running-stat update (momentum m, unbiased variance), models/discriminator.py:207 in train mode */
int favae_bn_update_running(const float* mean, const float* rstd, int C, int64_t count, float eps, float momentum,
                            float* running_mean, float* running_var, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Single-head attention core.  Replaces nn.MultiheadAttention(C, 1 head) SDPA (models/codec.py:92,99).
 * ---------------------------------------------------------------------------------------------------------- */
/* C[b] = alpha * op(A[b]) * op(B[b]) (+ C[b] if accumulate).  ta/tb = 0: operand stored [rows][k] (k contiguous);
 * 1: stored [k][rows].  A is M x K, B is N x K in the "0" convention (i.e. C = A * B^T). */
int favae_bgemm(int ta, int tb, int M, int N, int K, float alpha, const float* A, int64_t lda, int64_t strideA,
                const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc, int64_t strideC, int batch,
                int accumulate, favae_stream_t stream);
int favae_softmax_rows(const float* s, float* p, int64_t rows, int L, favae_stream_t stream);
/* ds = alpha * p * (dp - rowsum(dp*p)) */
int favae_softmax_rows_bwd(const float* p, const float* dp, float* ds, int64_t rows, int L, float alpha,
                           favae_stream_t stream);

/* The attention core without the N x L x L probability matrix (flash-style: row log-sum-exp saved, probabilities recomputed in the
 * backward pass), in query chunks whose score tile stays in the Infinity Cache, on the split-precision matrix path:
 *   favae_bgemm_sp        favae_bgemm with fp32-grade products on the 16-bit matrix pipe (two scaled fp16 planes per operand,
 *                         DESIGN.md section 3); amaxA / amaxB: device scalars >= max|A|, max|B|.  Returns FAVAE_ERR_UNSUPPORTED
 *                         for (ta, tb) = (1, 0), unaligned operands, ld / stride % 4 != 0 (the caller then uses favae_bgemm).
 *   favae_softmax_rows_lse  in-place softmax of `rows` = batch * rows_per_batch score rows (row r of batch element n = query
 *                         row0 + r) + lse[n * Ltot + row0 + r] = log-sum-exp of the row
 *   favae_attn_bwd_point  in place: s <- p = exp(s - lse); dp <- ds = alpha * p * (dp - delta); *ds_absmax = max|ds|
 *   favae_rowdot          out[r] = sum_c a[r][c] * b[r][c]          (delta = rowsum(dO * O))
 * Replaces F.scaled_dot_product_attention inside nn.MultiheadAttention (models/codec.py:92,99) and its autograd. */
int favae_bgemm_sp(int ta, int tb, int M, int N, int K, float alpha, const float* A, int64_t lda, int64_t strideA,
                   const float* amaxA, const float* B, int64_t ldb, int64_t strideB, const float* amaxB, float* C, int64_t ldc,
                   int64_t strideC, int batch, int accumulate, favae_stream_t stream);
int favae_softmax_rows_lse(float* s, float* lse, int64_t rows, int L, int rows_per_batch, int row0, int Ltot, favae_stream_t stream);
int favae_attn_bwd_point(float* s, float* dp, const float* lse, const float* delta, int64_t rows, int L, int rows_per_batch,
                         int row0, int Ltot, float alpha, float* ds_absmax, favae_stream_t stream);
int favae_rowdot(const float* a, const float* b, float* out, int64_t rows, int C, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Learnable-sigma Gaussian blur (depthwise, reflect padding).  Replaces _get_gaussian_kernel1d/2d + F.pad(reflect)
 * + F.conv2d(groups=C)  (models/codec.py:255-277, models/vqgan_fcm.py:20-41).  `sigma` is a device scalar.
 * ---------------------------------------------------------------------------------------------------------- */
int favae_blur_fwd(const float* x, const float* sigma, int ksize, int N, int H, int W, int C, float* y,
                   favae_stream_t stream);
size_t favae_blur_bwd_workspace(int ksize, int N, int H, int W, int C);
/* dx = adjoint(reflect-pad o blur)(dy); dsigma (1 float, overwritten) = dL/dsigma */
int favae_blur_bwd(const float* x, const float* dy, const float* sigma, int ksize, int N, int H, int W, int C,
                   float* dx, float* dsigma, void* ws, size_t ws_bytes, favae_stream_t stream);
/* the same with dx = adjoint-blur(dy) + dx_add (both required): the gradient of the blurred tensor's other consumer (the codec's trunk,
 * models/codec.py:209-215) folded into the store instead of a separate accumulation pass over two tensors */
int favae_blur_bwd_add(const float* x, const float* dy, const float* sigma, int ksize, int N, int H, int W, int C, const float* dx_add,
                       float* dx, float* dsigma, void* ws, size_t ws_bytes, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Focal-frequency / dynamic-spectrum loss.  Replaces focal_frequency_loss.FocalFrequencyLoss(loss_weight, alpha=1)
 * (pip 0.3.0; call sites favae_scripts/train_favae.py:313,318,326, losses/vqgan_losses.py:14,25-26).
 * 1 <= H, W <= 1024.  Power-of-two lengths (every shipped FA-VAE configuration: 256, 64, 16) run the in-LDS radix-2 FFT; any other
 * length runs a direct O(L^2) DFT per line in the same kernel (same passes, same half-spectrum layout) -- a correct fallback, not a fast path.
 *   loss = loss_weight * mean( w * |F|^2 ),  F = fft2_ortho(pred - target),  w = clamp(|F| / max_plane|F|, 0, 1), NaN -> 0
 * `spec` (N*H*(W/2+1)*C*2 floats: the input is real, only the bins k <= W/2 of the Hermitian spectrum are kept; the loss counts
 * the others through their mirror images) receives (2*loss_weight/M) * w * F, from which favae_ffl_bwd forms
 *   dL/dpred = gloss * Re ifft2_ortho(spec),  dL/dtarget = -dL/dpred.
 * ---------------------------------------------------------------------------------------------------------- */
size_t favae_ffl_spec_floats(int N, int H, int W, int C);     /* floats `spec` must hold */
size_t favae_ffl_workspace(int N, int H, int W, int C);
int favae_ffl_fwd(const float* pred, const float* target, int N, int H, int W, int C, float loss_weight, float* loss,
                  float* spec, void* ws, size_t ws_bytes, favae_stream_t stream);
int favae_ffl_bwd(const float* spec, const float* gloss, int N, int H, int W, int C, float* gpred, float* gtarget,
                  void* ws, size_t ws_bytes, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Cosine-similarity vector quantiser.  Replaces CosineSimCodebook.forward / VectorQuantize.forward
 * (models/l2_quantize.py:391-444, 533-596).
 * ---------------------------------------------------------------------------------------------------------- */
/* tokens z (T,d), codebook embed (C,d).  zn = l2norm(z), en = l2norm(embed), idx[t] = argmax_c <zn[t],en[c]> (first max;
 * rows whose top-2 gap < tie_eps are re-evaluated in fp64), zq[t] = embed[idx[t]] (un-normalised row).  idx is int64. */
size_t favae_vq_workspace(int T, int d, int C);
int favae_vq_lookup(const float* z, const float* embed, int T, int d, int C, float tie_eps, int64_t* idx, float* zq,
                    float* zn, float* en, void* ws, size_t ws_bytes, favae_stream_t stream);
/* bins[c] = #tokens with idx==c ; embed_sum[c] = sum of zn rows with idx==c (deterministic: stable counting sort of the
 * tokens by code, rows added in ascending token order) */
size_t favae_vq_segment_workspace(int T, int C);
int favae_vq_segment_sum(const float* zn, const int64_t* idx, int T, int d, int C, float* bins, float* embed_sum, void* ws,
                         size_t ws_bytes, favae_stream_t stream);
/* EMA (l2_quantize.py:421-438, ema_inplace :45-46 = moving_avg.mul_(decay).add_(new, alpha=1 - decay)):
 * cluster_size = decay*cluster_size + (1-decay)*bins; embed = decay*embed + (1-decay)*(bins==0 ? en : l2norm(embed_sum/bins)).
 * `decay` is a double like the Python float of the reference: the two fp32 factors are (float)decay and (float)(1.0 - decay), the
 * values torch rounds its scalar arguments to (1.f - (float)decay differs from that by one ulp for decay = 0.8). */
int favae_vq_ema_update(float* embed, float* cluster_size, const float* en, const float* bins, const float* embed_sum,
                        int C, int d, double decay, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Discriminator training terms (BASELINE config 5): hinge losses (losses/hinge.py:5-14) and the backward of an
 * activation applied without normalisation (first LeakyReLU of models/discriminator.py:198-201).
 * ---------------------------------------------------------------------------------------------------------- */
/* loss = mean_i h(x_i): mode 0: -x (hinge_g_loss), 1: relu(1 - x) (real half of hinge_d_loss), 2: relu(1 + x) (fake half).
 * ws: favae_reduce_workspace().  Backward: dx_i = g * h'(x_i) / n  (g: device scalar). */
int favae_hinge_mean(const float* x, int64_t n, int mode, float* loss, void* ws, size_t ws_bytes, favae_stream_t stream);
int favae_hinge_mean_bwd(const float* x, const float* g, int64_t n, int mode, float* dx, favae_stream_t stream);
/* dx = da * act'(x), act as in favae_conv_desc */
int favae_act_bwd(const float* da, const float* x, int act, int64_t n, float* dx, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * LPIPS perceptual distance (losses/lpips.py:17-110; `lpips(x, x_recon)` of favae_scripts/train_favae.py:77).  The VGG16
 * 3x3 convs run through favae_conv_fwd* with act = FAVAE_ACT_RELU applied on the operand load, so feature tensors hold
 * PRE-activation values; the entry points below apply the ReLU themselves.  NHWC fp32.
 * ---------------------------------------------------------------------------------------------------------- */
/* One level of LPIPS.forward (losses/lpips.py:44-48): val[n] (+)= mean_hw sum_c w_c (a^ - b^)^2 with
 * a^ = relu(a) / max(|relu(a)|_2, 1e-12) over channels (F.normalize).  C in {64,128,256,512}; w = the level's 1x1 `lin`
 * weights [C]; ws: favae_lpips_level_workspace(N).  Backward: db = d val[n]/d b * g[n]  (b = pre-activation features of the
 * image that carries the gradient, i.e. the SECOND argument of lpips(); a gets no gradient in the training step). */
size_t favae_lpips_level_workspace(int N);
int favae_lpips_level(const float* a, const float* b, const float* w, int N, int HW, int C, float* val, int accumulate, void* ws,
                      size_t ws_bytes, favae_stream_t stream);
int favae_lpips_level_bwd(const float* a, const float* b, const float* w, const float* g, int N, int HW, int C, float* db,
                          favae_stream_t stream);
/* nn.MaxPool2d(2, 2) of torchvision's vgg16 features (losses/lpips.py:88-96): H, W even, C % 4 == 0; first maximum in
 * scan order wins, NaN propagates (aten).  Backward routes dy to that element and writes zeros elsewhere (dx fully written). */
int favae_maxpool2(const float* x, int N, int H, int W, int C, float* y, favae_stream_t stream);
int favae_maxpool2_bwd(const float* x, const float* dy, int N, int H, int W, int C, float* dx, favae_stream_t stream);
/* ScalingLayer (losses/lpips.py:55-62): y = (x - shift[c]) / scale[c], c = i mod C; shift == NULL: y = x / scale[c] (its gradient) */
int favae_channel_affine(const float* x, const float* shift, const float* scale, int64_t n, int C, float* y, favae_stream_t stream);

/* straight-through value exactly as the reference forms it: out = x + (q - x)   (models/l2_quantize.py:554) */
int favae_vq_ste(const float* x, const float* q, float* out, int64_t n, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Reductions / element-wise glue.
 * ---------------------------------------------------------------------------------------------------------- */
size_t favae_reduce_workspace(int64_t n);
/* loss[0] = scale * sum |a-b|       (L1: favae_scripts/train_favae.py:76 with scale = 1/n) */
int favae_absdiff_sum(const float* a, const float* b, int64_t n, float scale, float* loss, void* ws, size_t ws_bytes,
                      favae_stream_t stream);
/* loss[0] = scale * sum (a-b)^2     (commit loss: models/l2_quantize.py:560 with scale = weight/n) */
int favae_sqdiff_sum(const float* a, const float* b, int64_t n, float scale, float* loss, void* ws, size_t ws_bytes,
                     favae_stream_t stream);
/* out = (out_add ? out_add : 0) + g[0]*scale*sign(a-b)   (dL1/da; pass -scale for dL1/db) */
int favae_absdiff_bwd(const float* a, const float* b, const float* g, float scale, int64_t n, const float* out_add,
                      float* out, favae_stream_t stream);
/* out = (out_add ? out_add : 0) + g[0]*scale*(a-b)       (commit-loss gradient, straight-through add) */
int favae_sqdiff_bwd(const float* a, const float* b, const float* g, float scale, int64_t n, const float* out_add,
                     float* out, favae_stream_t stream);
/* y = alpha*x + beta*y */
int favae_axpby(const float* x, float alpha, float* y, float beta, int64_t n, favae_stream_t stream);
/* ------------------------------------------------------------------------------------------------------------
 * Attention FCM (--use_gauss_attn): TransEncoderBlock = GroupNorm(32) + nn.TransformerEncoderLayer(C, nhead 8, FFN 2048, ReLU,
 * dropout 0.1, post-norm, batch_first), models/codec.py:108-122; DecoderFcmAttnGauss :1011-1129 (its fcm_4 is a
 * ResnetBlock(dropout=0.1)).  The linear layers are 1x1 favae_conv_fwd calls on NHWC tokens, the heads are favae_bgemm /
 * favae_softmax_rows calls on strided q/k/v views; what is new are the token-wise passes below.
 * ---------------------------------------------------------------------------------------------------------- */
/* y[n,p,c] = act(x[n,p,c] * scale[n,c] + shift[n,c]): GroupNorm materialised from favae_gn_stats()' per-(image, channel) affine
 * (the block's residual branch is the normalised tensor itself).  C % 4 == 0. */
int favae_affine_rows(const float* x, const float* scale, const float* shift, float* y, int N, int64_t HW, int C, int act,
                      favae_stream_t stream);
/* nn.LayerNorm(C) over `rows` tokens of C contiguous floats (norm1 / norm2 of the encoder layer): y = (x - mean) * rstd * gamma +
 * beta, biased variance, statistics in fp64; mean / rstd (rows floats each) are kept for the backward.  C % 4 == 0, C <= 2048. */
int favae_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int C,
                        float eps, favae_stream_t stream);
/* dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)), g = dy * gamma; dy_xhat[r][c] = dy * xhat, whose favae_colsum() is
 * dgamma (dbeta = favae_colsum(dy)) -- both column sums stay deterministic and accumulate into the flat gradient buffer. */
int favae_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                        float* dy_xhat, int64_t rows, int C, favae_stream_t stream);
/* nn.Dropout(p) in training mode (and, with `gate`, ReLU followed by dropout): y[i] = keep(seed, i) && (gate == NULL ||
 * gate[i] > 0) ? x[i] / (1 - p) : 0.  keep() is a counter-based hash of (i, seed) -- no generator state; the backward is the
 * same call on dy with the same seed (and gate = the forward's pre-activation).  p == 0 keeps everything (pure ReLU with a
 * gate).  x may alias y.  n < 2^32. */
int favae_dropout(const float* x, const float* gate, float* y, int64_t n, float p, uint32_t seed, favae_stream_t stream);

/* Input pipeline tail on the device: uint8 HWC pixels (PIL decode + resize stay on the host workers) -> fp32 NHWC,
 * y = (u/255 - mean[c]) / std[c] -- T.ToTensor() + T.Normalize(mean, std) of datasets/general_dataloader.py:33-38 with the same
 * fp32 operations (bit-identical), written in the layout the convs read.  `mean`, `std`: HOST arrays of C floats (C <= 4).
 * The batch crosses PCIe as bytes (4x less than the reference's float batch). */
int favae_u8_to_float_nhwc(const unsigned char* in, float* out, int64_t pixels, int C, const float* mean, const float* std,
                           favae_stream_t stream);
/* layout converters: NCHW <-> NHWC */
int favae_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, favae_stream_t stream);
int favae_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, favae_stream_t stream);
/* torch.optim.Adam (no weight decay / amsgrad) over one flat buffer: favae_scripts/train_favae.py:297-305 */
int favae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    int step, float grad_scale, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * GroupNorm statistics pass inside the conv that PRODUCES the normalised tensor.  nn.GroupNorm (models/codec.py:38,42,92,171)
 * needs per (image, group) the mean and variance of its input, which is the output of the previous conv: the dense 3x3 forward
 * kernel sums y and y^2 per channel over its 8x16-pixel tile in the epilogue (fp64 from the first product, fixed order:
 * deterministic) into part[N][tiles][Cout][2], and favae_gn_stats_tiles finishes the statistics without reading the tensor.
 *   favae_conv_stats_tiles(d, has_affine, planes)   tiles per image when favae_conv_fwd_split(d, ...) runs that kernel, else 0;
 *     planes = the planes word the conv call will be given (only FAVAE_PLANES_WINO4 is looked at: the Winograd kernels' grids differ)
 * ---------------------------------------------------------------------------------------------------------- */
int favae_conv_stats_tiles(const favae_conv_desc* d, int has_affine, int planes);
/* y_absmax (optional device float): max |y| of the output as a further by-product -- the fp16 operand range of a conv that consumes y
 * WITHOUT a normalisation in front (Downsample, Upsample, nin_shortcut: models/codec.py:17,26-29,50), which otherwise costs one
 * favae_absmax pass over y. */
int favae_conv_fwd_split_stats(const favae_conv_desc* d, const float* x, const void* wsplit, int planes, const float* x_absmax,
                               const float* bias, const float* resid, const float* scale, const float* shift, float* y,
                               void* part, size_t part_bytes, float* y_absmax, favae_stream_t stream);
int favae_gn_stats_tiles(const void* part, int tiles, const float* gamma, const float* beta, int N, int64_t HW, int C, int G,
                         float eps, float* mean, float* rstd, float* scale, float* shift, float* absmax_out, void* ws,
                         size_t ws_bytes, favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * GroupNorm-backward pass 1 inside the data-gradient conv.  The backward of act(GroupNorm(x)) needs, per (image, channel), S1 = sum
 * dy and S2 = sum dy*xhat with dy = da * act'(y), da = the data gradient of the conv that consumed it.  The dense 3x3 data-gradient
 * kernel has da in its accumulators: favae_conv_dgrad_gnbwd = favae_conv_fwd_split(d, dy, flipped weights) -> da, whose epilogue
 * also reads the matching tile of x and writes per-tile partial sums part[N][tiles][C][2] (double, fixed summation order:
 * deterministic); favae_gn_act_bwd_tiles = favae_gn_act_bwd without its streaming pass 1 (two tensor reads less per GroupNorm).
 *   favae_conv_gnbwd_tiles(d, planes)   tiles per image when the data gradient `d` runs that kernel, else 0 (planes: as above)
 *   favae_gn_bwd_tiles_workspace    bytes of the workspace shared by the two calls: it STARTS with `part`
 * Reference: autograd of GroupNorm + SiLU in ResnetBlock / NonResnetBlock / final (models/codec.py:38-46,65-73,170-175).
 * ---------------------------------------------------------------------------------------------------------- */
/* bf16 activation STORAGE (ABI 21, round 6; BASELINE configs[4] "bf16": accelerate's autocast keeps conv outputs in bf16,
 * favae_scripts/train_favae.py:239-240).  With the flag on `act`, favae_gn_act_bwd / _tiles / _colsum take da, x, dx_add and write dx as
 * bf16 tensors (same element counts; round to nearest even on store, exact widening on load); statistics, sums and the arithmetic stay
 * fp32 / fp64.  favae_gn_stats_bf16 = favae_gn_stats of a tensor stored as bf16.  Conv entry points: FAVAE_PLANES_BF16IO on `planes`. */
#define FAVAE_ACT_BF16IO 0x200
int favae_conv_gnbwd_tiles(const favae_conv_desc* d, int planes);
size_t favae_gn_bwd_tiles_workspace(int N, int tiles, int C);
int favae_conv_dgrad_gnbwd(const favae_conv_desc* d, const float* dy, const void* wsplit, int planes, const float* dy_absmax,
                           float* da, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           int groups, int act, void* part, size_t part_bytes, favae_stream_t stream);
int favae_gn_act_bwd_tiles(const float* da, const float* x, const float* gamma, const float* beta, const float* mean,
                           const float* rstd, int N, int64_t HW, int C, int G, int act, const float* dx_add, float* dx,
                           float* dgamma, float* dbeta, int accumulate, int tiles, void* ws, size_t ws_bytes,
                           favae_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Launch profiler (measurement only; no reference counterpart -- the reference is timed from outside by its caller).
 * level 0: off (default).  level 1: every launch that carries >= 1 GFLOP of algorithmic work (the matrix-bound conv forward /
 * data-gradient / weight-gradient kernels) is bracketed by two HIP events recorded on ITS launch stream; level 2: every launch of
 * the library.  favae_prof_report() waits for the recorded launches and writes one line per kernel instantiation:
 *   name \t launches \t total_us \t min_us \t max_us \t algorithmic FLOPs \t algorithmic bytes      (sums over the launches)
 * into buf (NUL-terminated, truncated to cap) and returns the size needed; favae_prof_reset() drops the records.
 * bench.py computes its `roofline` object from this, inside the timed region.
 * ---------------------------------------------------------------------------------------------------------- */
int favae_prof_enable(int level);
int favae_prof_reset(void);
int64_t favae_prof_report(char* buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* FAVAE_HIP_H */
