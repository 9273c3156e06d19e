/*
 * librac_hip.so -- C ABI of the MI355X (gfx950) hot path of robot_aware_control:
 * the conv-SVG dynamics model (train step + frozen rollouts) and the CEM cost tail.
 *
 * The reference has no FFI layer: the path sits behind Python/PyTorch operators
 * (SURVEY.md section 8b).  Each entry point below replaces the ATen operator(s)
 * the reference calls at the cited file:line (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to caller-owned memory; the library never
 *     allocates, frees or synchronises; kernels are launched on `stream`
 *     (a hipStream_t passed as void*).
 *   - activations are fp32 NHWC ("channels last": [B][H][W][C]); conv weights are
 *     fp32 [Cout][kh][kw][Cin] (the channels_last memory of a (Cout,Cin,kh,kw)
 *     tensor); frames at the model boundary are NCHW planes as the reference
 *     dataloader produces them (src/dataset/robonet/robonet_dataset.py:434-451).
 *   - return value 0 on success, negative RAC_E* otherwise; rac_last_error()
 *     returns a thread-local message for the last failure.
 */
#ifndef RAC_HIP_H
#define RAC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAC_OK 0
#define RAC_EINVAL (-1)   /* bad argument (shape / alignment / null pointer) */
#define RAC_ELAUNCH (-2)  /* hipLaunch failed */

#define RAC_ABI_VERSION 11

int rac_version(void);
const char* rac_device_arch(void); /* "gfx950" */
const char* rac_last_error(void);

/* ------------------------------------------------------------------------ *
 * Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * stride 1, "same" padding (pad = ksize/2), ksize odd.
 * ------------------------------------------------------------------------ */
enum {
  RAC_CONV_FWD = 0,   /* out[p][co] = sum_{tap,ci} in[p+tap][ci] * w[co][tap][ci]          */
  RAC_CONV_DGRAD = 1, /* out[p][ci] = sum_{tap,co} in[p-tap][co] * w[co][tap][ci]          */
  RAC_CONV_WGRAD = 2  /* dw[co][tap][ci] += sum_p dy[p][co] * x[p+tap][ci]                 */
};
enum { RAC_ACT_NONE = 0, RAC_ACT_LEAKY02 = 1, RAC_ACT_SIGMOID = 2 };

typedef struct rac_conv_args {
  int32_t mode;       /* RAC_CONV_* */
  int32_t B, H, W;    /* batch and spatial size (input == output) */
  int32_t ksize;      /* 3 or 5 */
  int32_t Cin;        /* weight inner dim  (channels of the conv input)  */
  int32_t Cout;       /* weight outer dim  (channels of the conv output) */
  int32_t act;        /* RAC_ACT_*  (FWD / DGRAD epilogue) */
  int32_t split_k;    /* >=1.  FWD/DGRAD: slab s written at out + s*slab_stride (no epilogue);
                         WGRAD: partial sums combined with float atomics */
  int32_t accumulate; /* WGRAD: 1 -> dw += result (dw holds a valid gradient), 0 -> dw = result */
  int32_t a_split;    /* A-side virtual concat: channels [0,a_split) come from a0, the rest from a1.
                         FWD: the conv input (Cin = a_split + rest); DGRAD: unused (0).
                         WGRAD: the saved conv input x. 0 or == channel count -> single source */
  int32_t o_split;    /* DGRAD: output channels [0,o_split) go to out0, the rest to out1 (0 -> single) */
  int64_t slab_stride;/* elements between split-K slabs (FWD/DGRAD with split_k > 1) */
  const float* a0;    /* FWD: input NHWC (first a_split channels); DGRAD: dy NHWC [.,Cout]; WGRAD: x part 0 */
  const float* a1;    /* second source of the virtual concat, or NULL */
  const float* w;     /* FWD/DGRAD: weights [Cout][k][k][Cin]; WGRAD: dy NHWC [.,Cout] */
  float* out0;        /* FWD: [.,Cout]; DGRAD: [.,Cin or o_split]; WGRAD: dw [Cout][k][k][Cin] */
  float* out1;        /* DGRAD second destination or NULL.  rac_conv2d_fwd_split: optional [B][H/2][W/2][Cout] -- the 2 x 2
                         max pool of the activated output, written by the same epilogue (MaxPool2d(2) behind a vgg_layer,
                         vgg_64.py:104-129); only where rac_conv2d_fwd_split_pool_ok says so */
  const float* bias;  /* [N] added first, or NULL */
  const float* scale; /* [N] v = v*scale + shift (folded eval BatchNorm), or NULL */
  const float* shift;
  double* stats;      /* [G][2][N] fp64 sum / sum of squares of the raw conv output (train BatchNorm), or NULL */
  int64_t stats_rows; /* rows (pixels) per statistics group: G = B*H*W / stats_rows groups (time steps batched along
                         the batch axis keep per-step statistics); 0 = one group.  Multiple of 128. */
  int32_t a0_up;      /* rac_conv2d_fwd_split, maps larger than a 128-pixel tile: a0 is the [B][H/2][W/2][.] tensor whose
                         nearest-neighbour 2x upsampling is the conv's first source (vgg_64.py:205-216, nn.UpsamplingNearest2d
                         before upc3 / upc4 / upc5): pixel (y, x) reads a0 at (y / 2, x / 2); the upsampled tensor is never
                         materialised.  0 elsewhere. */
  int32_t amax_per_image; /* rac_conv2d_fwd_split: a_amax0 / a_amax1 / out_amax are arrays of B slots, one per image, and
                         every image is scaled by its OWN maximum: an image's result then cannot depend on what else is in
                         the batch (the frozen model's rollouts: candidates_batch_size and the shard a candidate lands in
                         must not change its cost, trajectory_sampler.py:123-174).  Needs split_k 1 and H*W % 16 == 0. */
} rac_conv_args;

/* Replaces aten::conv2d / conv_transpose2d and their backward on the hot path:
 *   src/prediction/models/vgg_64.py:8-18 (vgg_layer), :218-220 (ConvTranspose2d head),
 *   src/prediction/models/lstm.py:129-149 (ConvLSTMCell gates), :273-274 (mu/logvar heads),
 *   src/prediction/models/dynamics.py:496-513 (input convs). */
int rac_conv2d(const rac_conv_args* a, void* stream);


/* ------------------------------------------------------------------------ *
 * Split-precision convolutions on the fp16 matrix pipe (v_mfma_f32_16x16x32_f16, fp32 accumulation): the same
 * aten::conv2d call sites as rac_conv2d, for the layers that carry the FLOPs (ConvLSTM gate convs lstm.py:129-149,
 * the >= 64-channel vgg layers vgg_64.py:8-18, the input convs dynamics.py:496-513).
 *
 * Every fp32 operand tensor X is scaled by a power of two s_X (max |x| s_X in [2^14, 2^15)) and written as two fp16
 * parts, x s_X = h1 + h2 (22 significant bits); a product keeps h1 g1 + h1 g2 + h2 g1 (the dropped term is <= 2^-22
 * relative) with fp32 accumulation, and the exact scales are removed in the epilogue.  Against an fp64 reference the
 * results are as close as rac_conv2d's exact-fp32 MFMA path (the fp32 accumulation dominates both; tests hold every
 * kernel to <= 4x that path's error) at 3 fp16 MFMA products per fp32 product: 2500 / 3 = 833 TFLOP/s algorithmic peak.
 * Scales come from device-side maxima (`amax` slots: the IEEE bit pattern of max |x|, which orders like an unsigned
 * integer), so nothing synchronises with the host.
 * ------------------------------------------------------------------------ */
/* *amax = max(*amax, bits(max |x|)) over x0[0..n0) and x1[0..n1) (x1 may be NULL, n1 = 0).  The slot must hold 0 or
 * an earlier maximum on entry; several calls may accumulate into one slot.  n % 4 == 0, 16-byte aligned. */
int rac_absmax(const float* x0, int64_t n0, const float* x1, int64_t n1, uint32_t* amax, void* stream);
/* amax[r] = bits(max |x[r][0..row_len)|) for every row r < rows (one slot per image: `amax_per_image` operands of
 * rac_conv2d_fwd_split).  Plain stores: the slots need no initialisation.  row_len % 4 == 0, 16-byte aligned. */
int rac_absmax_rows(const float* x, int64_t rows, int64_t row_len, uint32_t* amax, void* stream);
/* Conv weight (fp32, [Cout][k][k][Cin] memory) -> the two fp16 parts of w * s_W in MFMA fragment order
 *   parts[part][R/32][K/32][k*k][nb 2][lane 64][8],  lane = 16 q + (r mod 16), row r = 32 tile + 16 nb + lane mod 16,
 *   k = 32 chunk + 8 q + j   (the B operand of v_mfma_f32_16x16x32_f16: one coalesced 1 KB load per MFMA operand).
 * transposed = 0: rows = Cout, K = Cin (forward).  transposed = 1: rows = Cin, K = Cout, taps flipped: the weight of
 * the forward conv that IS the data gradient (dgrad(dy, W) == fwd(dy, Wt)).  w_amax = rac_absmax of w.
 * Channel counts % 32 == 0; part_stride in elements. */
int rac_weight_frag_split(const float* w, const uint32_t* w_amax, uint16_t* parts, int32_t Cout, int32_t Cin,
                          int32_t ksize, int32_t transposed, int64_t part_stride, void* stream);
/* The two calls above for MANY tensors in one launch each (every conv weight of a model after an optimiser step).
 * `jobs` is an array in DEVICE memory; job j owns the workgroups [block_begin_j, block_begin_{j+1}) of the launch, with
 * block_begin the running sum of rac_absmax_blocks(n) / rac_weight_frag_blocks(Cout, Cin, ksize); total_blocks is the
 * final sum.  Per-job requirements as for the single calls (amax slots zero or an earlier maximum on entry). */
typedef struct rac_absmax_job {
  const float* x;
  int64_t n;
  uint32_t* amax;
  int64_t block_begin;
} rac_absmax_job;
typedef struct rac_frag_job {
  const float* w;
  const uint32_t* w_amax;
  uint16_t* parts;
  int64_t part_stride;
  int32_t Cout, Cin, ksize, transposed;
  int64_t block_begin;
} rac_frag_job;
int64_t rac_absmax_blocks(int64_t n);
int64_t rac_weight_frag_blocks(int32_t Cout, int32_t Cin, int32_t ksize);
int rac_absmax_multi(const rac_absmax_job* jobs, int32_t n_jobs, int64_t total_blocks, void* stream);
int rac_weight_frag_split_multi(const rac_frag_job* jobs, int32_t n_jobs, int64_t total_blocks, void* stream);
/* 1 if rac_conv2d_fwd_split takes this shape: k 3 or 5, Cin % 32 == Cout % 32 == 0 (a_split % 32 == 0), and either
 * H*W <= 128 with a whole number of images per tile that is a multiple of 16 rows (8x8, 6x8 latent maps), or
 * W <= 128 with R | H image rows per tile, R*W <= 128 a multiple of 16, halo included <= 256 staged rows. */
int rac_conv2d_split_supported(int32_t H, int32_t W, int32_t ksize, int32_t Cin, int32_t Cout, int32_t a_split);
/* FWD conv (as rac_conv2d mode RAC_CONV_FWD; epilogue fields bias, scale/shift, act, stats, split_k slabs behave the
 * same) with a->a0 / a->a1 the fp32 NHWC activations (split into parts on the way into LDS), a->w the fragment-order
 * parts of rac_weight_frag_split, a_amax0 / a_amax1 the maxima of a0 / a1 (a_amax1 may be NULL), w_amax as given to
 * rac_weight_frag_split.  The data gradient of a conv is this call on dy with the transposed parts.
 * a->Cout need not be a multiple of 32 (the 4-channel output head): the parts must then come from the weight with its
 * rows zero-padded to the next multiple of 32; only Cout columns are computed into the output (row stride Cout). */
int rac_conv2d_fwd_split(const rac_conv_args* a, const uint32_t* a_amax0, const uint32_t* a_amax1,
                         int64_t w_part_stride, int32_t w_cin, const uint32_t* w_amax, uint32_t* out_amax, void* stream);
/* 1 if rac_conv2d_fwd_split with these arguments can also write a->out1, the 2 x 2 max pool of its activated output
 * (ConvEncoder's `mp` behind the last vgg_layer of c1 / c2 / c3, vgg_64.py:104-129, frozen model): the unrolled 3x3 kernels
 * on maps larger than a tile, full 128-pixel tiles and column blocks, split_k 1, no statistics, an output below 4 GiB, a
 * 16-row block and the block under it held by one wave (W = 16, 32, or 2-D tiles).  0 otherwise: call rac_maxpool2_fwd.
 * Nothing is launched.  The pooled tensor's maximum is bounded by the output's (out_amax): no separate slot. */
int rac_conv2d_fwd_split_pool_ok(const rac_conv_args* a, int32_t w_cin);
/* The FROZEN model's ConvLSTM cell in one launch (lstm.py:129-149 without a tape): the gate conv of
 * rac_conv2d_fwd_split with the cell arithmetic in its epilogue,
 *   (i, f, o, g) = conv([a0 | a1]) + bias;  c = sig(f) c_prev + sig(i) tanh(g);  h = sig(o) tanh(c),
 * so the [M][4g] gate tensor is never written.  a->w: parts of the gate weight with its 4g ROWS GATE-INTERLEAVED in groups
 * of 16 channels -- row 64 (c / 16) + 16 gate + c % 16 holds gate `gate` (0 i, 1 f, 2 o, 3 g) of channel c; a->bias: the
 * gate bias in the ORIGINAL order [4][g]; a->Cout = 4g (% 64 == 0), split_k 1, a map that fits a tile (H*W <= 128);
 * a->out0 unused.  c_prev / h_out / c_out: [M][g].  Transcendentals to ~1e-7 absolute (v_exp_f32 with an FMA-recovered
 * argument error), not libm's: results agree with rac_lstm_cell_fwd to that level, not to the bit. */
int rac_convlstm_cell_fwd_split(const rac_conv_args* a, const uint32_t* a_amax0, const uint32_t* a_amax1,
                                int64_t w_part_stride, int32_t w_cin, const uint32_t* w_amax, const float* c_prev,
                                float* h_out, float* c_out, void* stream);
/* w_cin (0 = a->Cin): input channels the weight parts were built with.  a->Cin < w_cin runs the conv over the first
 * a->Cin input channels only (a channel prefix is a prefix of every tile's weight stream): the ConvLSTM cells' first
 * step, whose hidden state is all zeros, skips the hidden half of K (and of the data gradient's N, through a->Cout). */
/* `*_amax` OUTPUT arguments (here and on rac_affine_act, rac_bn_bwd_apply, rac_tilecat_fwd, rac_slab_reduce,
 * rac_lstm_cell_bwd; nullable): the kernel folds max |v| of the tensor it writes into the slot as rac_absmax would,
 * which saves the consumer conv a reduction pass over it (split_k > 1 writes raw slabs: no out_amax there). */

/* Weight gradient dw[co][ky][kx][ci] (+)= sum_{steps, pixels} dy[p][co] x[p+tap][ci] on the same pipe (the backward of
 * the conv2d call sites above; aten::conv2d backward w.r.t. the weight).  Operands are the fp32 NHWC tensors as the
 * forward pass saw them (x = virtual concat [x0 | x1] at a_split), one triple per time step: T steps are summed in
 * ONE launch (one read-modify-write of dw).  *_amax = rac_absmax slots of the operands (one common scale per
 * operand kind is taken from the maximum over the steps).  nsplit > 1 splits the pixel range: part 0 goes to dw, parts
 * 1.. to `slabs` (nsplit - 1 arrays of slab_stride elements) which the caller then adds with rac_slab_accumulate --
 * deterministic, no atomics.  Needs ksize 3 or 5, Cout % 16 == 0, channel counts % 8 == 0, a_split % 64 == 0. */
#define RAC_WGRAD_MAX_STEPS 16
typedef struct rac_wgrad_args {
  int32_t B, H, W;       /* images per step and their size */
  int32_t ksize, Cin, Cout, a_split;
  int32_t T;             /* time steps, 1 .. RAC_WGRAD_MAX_STEPS */
  int32_t nsplit;        /* K splits, <= T * ceil(B*H / 32) */
  int32_t accumulate;    /* 1: dw += result, 0: dw = result */
  const float* dy[RAC_WGRAD_MAX_STEPS];
  const float* x0[RAC_WGRAD_MAX_STEPS];
  const float* x1[RAC_WGRAD_MAX_STEPS];       /* NULL: single source */
  const uint32_t* dy_amax[RAC_WGRAD_MAX_STEPS];
  const uint32_t* x0_amax[RAC_WGRAD_MAX_STEPS];
  const uint32_t* x1_amax[RAC_WGRAD_MAX_STEPS];
  float* dw;             /* [Cout][k][k][Cin] */
  float* slabs;          /* workspace for nsplit > 1 */
  int64_t slab_stride;
  int32_t x1_zero_steps; /* the x1 tensors of steps [0, x1_zero_steps) are all zeros (a ConvLSTM's initial hidden state):
                            their share of the x1 half of dw is skipped, not computed */
  int32_t presplit;      /* 1: dy / x0 / x1 point to the fp16 part pairs rac_split_steps wrote ([2][elements], split under
                            the SAME slot lists: every dy_amax for dy, every x0_amax and x1_amax for x0 and x1) */
  int32_t all_ky;        /* 1 (ksize 3, Cout <= 128, H % 32 == 0, no presplit, no x1_zero_steps): one workgroup keeps all nine
                            taps of its 64 x 64 (co, ci) tile, so dy and x leave HBM once per tile instead of once per kernel
                            row -- the thin layers on 32x32 / 64x64 maps; nsplit then counts per tile, not per (tile, ky) */
  int32_t col_segments;  /* all_ky: K is split into (32-row group, column segment) units, col_segments (| W) per image row,
                            so that nsplit may reach T * groups * col_segments workgroups per tile (0 / 1: whole rows) */
} rac_wgrad_args;
int rac_conv2d_wgrad_split(const rac_wgrad_args* a, void* stream);
/* parts[t] ([2][n] halves) = the two fp16 parts of xs[t] (n floats, n % 8 == 0) under ONE power-of-two scale taken from the
 * maximum over the n_amax slots: the operands of rac_conv2d_wgrad_split split once instead of in every workgroup that
 * stages them (a tile of dy is read by one workgroup per input-channel tile and kernel row).  xs / parts / amax are
 * HOST arrays of device pointers. */
int rac_split_steps(const float* const* xs, uint16_t* const* parts, int32_t T, int64_t n, const uint32_t* const* amax,
                    int32_t n_amax, void* stream);
/* out[i] += sum_s slabs[s*slab_stride + i]  (fixed order) */
int rac_slab_accumulate(const float* slabs, int32_t n_slabs, int64_t slab_stride, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------ *
 * BatchNorm2d (training statistics) + LeakyReLU(0.2)
 *   src/prediction/models/vgg_64.py:12-14 (nn.BatchNorm2d, nn.LeakyReLU(0.2))
 * ------------------------------------------------------------------------ */
/* `groups` (G >= 1) in this section: the M rows are G equal row ranges, each ONE BatchNorm call of the reference
 * (time steps batched along the row axis keep their per-call batch statistics).  stats / sums are [G][2][C],
 * scale / shift / mean / invstd are [G][C].
 * stats = fp64 [G][2][C] (sum, sum of squares over `count` = B*H*W values per group, from rac_conv2d).
 * Writes mean/invstd (biased variance, eps), scale = gamma*invstd, shift = beta - mean*scale,
 * and applies the groups' running-stat momentum updates in order, `n_updates` times each (the reference
 * encodes the same frame twice per train step: dynamics.py:584,619). */
int rac_bn_finalize(const double* stats, int64_t count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, int32_t n_updates, float* scale, float* shift,
                    float* mean, float* invstd, int32_t C, int32_t groups, void* stream);
/* y[m][c] = act(x[m][c]*scale[c] + shift[c]) */
int rac_affine_act(const float* x, const float* scale, const float* shift, int32_t act, float* y, int64_t M,
                   int32_t C, int32_t groups, uint32_t* y_amax, void* stream);
/* rac_bn_finalize followed by rac_affine_act on x in ONE launch (C = 4 * 2^k <= 1024, 16-byte aligned maps): the same
 * arithmetic, value for value -- scale / shift / mean / invstd are still written (the backward pass reads them). */
int rac_bn_apply_act(const double* stats, int64_t count, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, int32_t n_updates, const float* x, int32_t act, float* y,
                     float* scale, float* shift, float* mean, float* invstd, int64_t M, int32_t C, int32_t groups,
                     uint32_t* y_amax, void* stream);
/* Small train-mode layers (ONE statistics group, rac_bn_small_ok(M, C): a few MB -- the 8x8 / 16x16 vgg layers of one time
 * step): a workgroup owns a slice of channels and all rows, so the per-channel sums need no atomics and no second launch.
 * rac_bn_small_fwd = split-K combine (raw = sum of the slabs) + batch statistics + running-stat updates + affine +
 * activation (what rac_slab_reduce_stats + rac_bn_apply_act do, vgg_64.py:8-18 in training mode); rac_bn_small_bwd =
 * rac_bn_bwd_reduce + rac_bn_bwd_apply.  The same arithmetic per element; the sums in a fixed order. */
int rac_bn_small_ok(int64_t M, int32_t C);
int rac_bn_small_fwd(const float* slabs, int32_t n_slabs, int64_t slab_stride, float* raw, float* y, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum, float eps, int32_t n_updates,
                     float* scale, float* shift, float* mean, float* invstd, int64_t M, int32_t C, int32_t act, uint32_t* y_amax,
                     void* stream);
int rac_bn_small_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                     const float* invstd, float* dx, float* dgamma, float* dbeta, int64_t M, int32_t C, uint32_t* dx_amax,
                     void* stream);
/* sums = fp64 [2][C]: sum dz, sum dz*xhat  with z = x*scale+shift, dz = dy*(z>0?1:0.2), xhat=(x-mean)*invstd.
 * `sums` must be zero on entry. */
int rac_bn_bwd_reduce(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                      const float* invstd, double* sums, int64_t M, int32_t C, int32_t groups, void* stream);
/* dx = scale*(dz - sum_dz/M - xhat*sum_dzx/M); dgamma += sum_dzx; dbeta += sum_dz */
int rac_bn_bwd_apply(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                     const float* invstd, const double* sums, float* dx, float* dgamma, float* dbeta, int64_t M,
                     int32_t C, int32_t groups, uint32_t* dx_amax, void* stream);

/* MaxPool2d(2,2) / nearest x2 upsample on NHWC maps: vgg_64.py:120,126-128 / :221,235-240 */
int rac_maxpool2_fwd(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int rac_maxpool2_bwd(const float* x, const float* dy, float* dx, int32_t B, int32_t H, int32_t W, int32_t C,
                     void* stream);
int rac_upsample2_fwd(const float* x, float* y, int32_t B, int32_t h, int32_t w, int32_t C, void* stream);
int rac_upsample2_bwd(const float* dy, float* dx, int32_t B, int32_t h, int32_t w, int32_t C, void* stream);

/* out[b][p][:] = [v0[b] | v1[b] | v2[b] | m0[b][p] | m1[b][p]]   (dynamics.py:591-607,634-640:
 * action / robot-state tiling + channel concat in front of the three input convs) */
int rac_tilecat_fwd(const float* v0, int32_t n0, const float* v1, int32_t n1, const float* v2, int32_t n2,
                    const float* m0, int32_t c0, const float* m1, int32_t c1, int32_t pad, float* out, int32_t B,
                    int32_t HW, uint32_t* out_amax, int32_t amax_per_image, void* stream);
/* `pad` trailing zero channels; amax_per_image: out_amax is an array of B slots (plain stores), one per image */
/* dst[r][0:C] = src[r][0:C], dst[r][C:Cpad] = 0  -- 16-byte aligned rows for the vector-load conv path
 * (weights of the convs whose channel count is not a multiple of 4: first encoder layer, the three input convs) */
int rac_pad_rows(const float* src, int32_t C, float* dst, int32_t Cpad, int64_t R, void* stream);
/* dst[r][0:C] += src[r][0:C]   (src rows are Cpad wide): folds a padded weight gradient back */
int rac_unpad_add(const float* src, int32_t Cpad, float* dst, int32_t C, int64_t R, void* stream);
/* dst[m][0:n] = src[m][off:off+n]  (row strides Csrc / n) */
int rac_slice_channels(const float* src, int32_t Csrc, int32_t off, int32_t n, float* dst, int64_t M, void* stream);
/* dst[m] = [a[m][0:Ca] | b[m][0:Cb]]  (a or b NULL: zeros) -- the gradient of the split above */
int rac_cat2_channels(const float* a, int32_t Ca, const float* b, int32_t Cb, float* dst, int64_t M, uint32_t* out_amax,
                      void* stream);
/* out[c] += sum_m x[m][c]   (bias gradients).  `parts`: scratch of rac_colsum_blocks(M, C) * C floats -- every workgroup
 * stores its row block's sums there and a second launch adds the blocks in a fixed order (a bit-reproducible gradient);
 * NULL: one fp32 atomic per (workgroup, column), whose order is not. */
int rac_colsum_acc(const float* x, float* out, float* parts, int64_t M, int32_t C, void* stream);
int64_t rac_colsum_blocks(int64_t M, int32_t C);

/* out[c] += sum_{t < T} sum_m xs[t][m][c]: the bias gradient of a conv applied at T time steps (T <= RAC_WGRAD_MAX_STEPS,
 * `xs` a HOST array of T device pointers to [M][C] tensors) in one launch: rac_colsum_acc with every workgroup walking
 * its rows in all T tensors (`parts` as above). */
int rac_colsum_steps(const float* const* xs, int32_t T, float* out, float* parts, int64_t M, int32_t C, void* stream);

/* out[i] = sum_s slabs[s*slab_stride + i] + bias[i % N]   (deterministic split-K combine; bias may be NULL) */
int rac_slab_reduce(const float* slabs, int32_t n_slabs, int64_t slab_stride, const float* bias, float* out,
                    int64_t n, int32_t N, uint32_t* out_amax, void* stream);
/* rac_slab_reduce (no bias) + rac_col_stats in one pass: out[M][C] = sum of slabs, and the BatchNorm batch statistics of
 * `groups` row groups (vgg_64.py:8-18 in training mode: one nn.BatchNorm2d call per time step batched along M):
 * stats[g][0][c] += sum, stats[g][1][c] += sum of squares (fp64, zeroed by the caller).  C = 4 * 2^k <= 1024. */
int rac_slab_reduce_stats(const float* slabs, int32_t n_slabs, int64_t slab_stride, float* out, double* stats, int64_t M,
                          int32_t C, int32_t groups, uint32_t* out_amax, void* stream);
/* as rac_slab_reduce for [M][N] slabs, but columns [0,o_split) go to out0 (row stride o_split) and the rest to
 * out1 (row stride N - o_split): the split-K combine of a DGRAD whose input was a virtual concat, and of the merged
 * mu | logvar head conv (lstm.py:273-274; n_slabs = 1 splits a finished [M][N] tensor) */
int rac_slab_reduce2(const float* slabs, int32_t n_slabs, int64_t slab_stride, const float* bias, float* out0,
                     float* out1, int64_t M, int32_t N, int32_t o_split, uint32_t* out0_amax, uint32_t* out1_amax,
                     void* stream);
/* stats[c] += sum_m x[m][c]; stats[C+c] += sum_m x[m][c]^2  (fp64; BatchNorm statistics after a split-K combine) */
int rac_col_stats(const float* x, double* stats, int64_t M, int32_t C, int32_t groups, void* stream);
/* dx = dy * act'(.) expressed through the activation OUTPUT y (sigmoid: y(1-y); leaky: y>0 ? 1 : 0.2) */
int rac_act_bwd(const float* dy, const float* y, int32_t act, float* dx, int64_t n, void* stream);

/* ConvLSTM cell pointwise part (lstm.py:136-149); gate order i, f, o, g.
 * pre = sum_s slabs[s] + bias;  c = sig(f)*c_prev + sig(i)*tanh(g);  h = sig(o)*tanh(c).
 * act_out (nullable) receives the four activated gates [M][4g] for the backward pass. */
int rac_lstm_cell_fwd(const float* gate_slabs, int32_t n_slabs, int64_t slab_stride, const float* bias,
                      const float* c_prev, float* h_out, float* c_out, float* act_out, int64_t M, int32_t g,
                      void* stream);
/* dgates[M][4g] (pre-activation grads), dc_prev[M][g]; dc_next may be NULL (zero). */
int rac_lstm_cell_bwd(const float* dh, const float* dc_next, const float* act, const float* c_prev,
                      const float* c_new, float* dgates, float* dc_prev, int64_t M, int32_t g, uint32_t* dgates_amax,
                      void* stream);

/* ------------------------------------------------------------------------ *
 * NormConvLSTMCell (--lstm_group_norm True, lstm.py:151-198): GroupNorm(16, C) on NHWC maps and the
 * cell arithmetic split around the cell-state normalisation.
 * ------------------------------------------------------------------------ */
/* The whole cell behind its two gate convs, without a tape (the frozen model: planner rollouts, evaluation), in ONE launch:
 *   gates = GroupNorm(16, 4g)(g_ih) + GroupNorm(16, 4g)(g_hh);  i, f, o = sigmoid, g~ = tanh  (chunk order i, f, o, g~)
 *   c = GroupNorm(16, g)(f * c_prev + i * g~);  h = o * tanh(c)                                (lstm.py:174-198)
 * g_ih / g_hh = the convs' outputs incl. bias, [B][HW][4g]; c_prev, h, c = [B][HW][g]; g = 16 * 2^k; 16-byte aligned.
 * One workgroup per (image, quarter of the channels): an image's result does not depend on the batch.
 * Training (all five or none): act [B][HW][4g] = the activated gates, c_raw [B][HW][g] = the cell before its norm, stat_ih /
 * stat_hh / stat_c [2][B][16] = mean and 1 / std of every (image, group) -- the operands of rac_lstm_out_bwd,
 * rac_groupnorm_bwd and rac_lstm_core_bwd. */
int rac_norm_lstm_cell_fwd(const float* g_ih, const float* g_hh, const float* c_prev, const float* gamma_ih,
                           const float* beta_ih, const float* gamma_hh, const float* beta_hh, const float* gamma_c,
                           const float* beta_c, float* h, float* c, float* act, float* c_raw, float* stat_ih, float* stat_hh,
                           float* stat_c, int32_t B, int32_t HW, int32_t g, float eps, void* stream);
/* The adjoint of rac_norm_lstm_cell_fwd in ONE launch: from dh and dc (gradients of h and of the normalised cell; either
 * may be NULL) and the forward pass's act / c / c_raw / stats to dg_ih, dg_hh (gradients of the two gate convs' outputs,
 * their max |.| folded into the two slots), dc_prev, and -- all six or none -- the three norms' affine gradients (+=). */
int rac_norm_lstm_cell_bwd(const float* dh, const float* dc, const float* act, const float* c, const float* c_raw,
                           const float* c_prev, const float* g_ih, const float* g_hh, const float* stat_ih,
                           const float* stat_hh, const float* stat_c, const float* gamma_ih, const float* gamma_hh,
                           const float* gamma_c, float* dg_ih, float* dg_hh, float* dc_prev, float* dgamma_ih, float* dbeta_ih,
                           float* dgamma_hh, float* dbeta_hh, float* dgamma_c, float* dbeta_c, uint32_t* dg_ih_amax,
                           uint32_t* dg_hh_amax, int32_t B, int32_t HW, int32_t g, void* stream);
/* y = (x - mean_{b,g}) * rstd_{b,g} * gamma_c + beta_c over groups of C/G channels x HW pixels (biased variance,
 * eps); mean/rstd = fp32 [B][G] saved for the backward pass. */
int rac_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int32_t B, int32_t HW, int32_t C, int32_t G, float eps, void* stream);
/* dx; dgamma += sum dy*xhat; dbeta += sum dy  (dgamma/dbeta may be NULL together) */
int rac_groupnorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                      float* dx, float* dgamma, float* dbeta, int32_t B, int32_t HW, int32_t C, int32_t G,
                      void* stream);
/* h = o * tanh(c) with o = act[m][2g + c]  (lstm.py:196); backward: d_act [M][4g] (only the o slot non-zero), dc */
int rac_lstm_out_fwd(const float* act, const float* c, float* h, int64_t M, int32_t g, void* stream);
int rac_lstm_out_bwd(const float* dh, const float* act, const float* c, float* d_act, float* dc, int64_t M, int32_t g,
                     void* stream);
/* gate pre-activation gradients of the split cell: dc_raw (gradient of the un-normalised cell f*c_prev + i*g) and
 * d_act (gradient of the activated gates [M][4g], may be NULL) -> dgates [M][4g], dc_prev [M][g] */
int rac_lstm_core_bwd(const float* dc_raw, const float* d_act, const float* act, const float* c_prev, float* dgates,
                      float* dc_prev, int64_t M, int32_t g, void* stream);

/* Gradient maps given as SUMS OF SOURCES (the hand-scheduled backward of the recurrent core, ops.RecurrentCore; replaces
 * what autograd's accumulation adds and the split-K combine passes did between the ConvLSTM cells' backward kernels,
 * lstm.py:129-149 / 252-257 unrolled over time).  A source is a stack of n_slabs K-split slabs of a (possibly wider)
 * tensor, read through a column window: value[m][c] = sum_s p[s * slab_stride + m * row_stride + col_off + c].  A plain
 * [M][C] tensor is n_slabs = 1, row_stride = C, col_off = 0.  C % 4 == 0; a window may start at any column (sources are
 * read with 4-byte alignment).  The array is HOST memory (at most 3 entries, copied into the launch). */
typedef struct rac_grad_src {
  const float* p;
  int64_t slab_stride;
  int32_t n_slabs, row_stride, col_off, reserved;
} rac_grad_src;
/* out[m][c] = sum over the sources (+ max |out| folded into out_amax, nullable) */
int rac_grad_sum(const rac_grad_src* srcs, int32_t n_srcs, float* out, int64_t M, int32_t C, uint32_t* out_amax,
                 void* stream);
/* rac_lstm_cell_bwd with dh = sum over dh_srcs (n_srcs = 0: dh = 0) */
int rac_lstm_cell_bwd_srcs(const rac_grad_src* dh_srcs, int32_t n_srcs, const float* dc_next, const float* act,
                           const float* c_prev, const float* c_new, float* dgates, float* dc_prev, int64_t M, int32_t g,
                           uint32_t* dgates_amax, void* stream);
/* backward of the reparameterisation into the merged mu | logvar head's output gradient (lstm.py:273-279):
 * dy[m] = [dz + dmu_add | dz * eps * 0.5 * exp(0.5 * logvar) + dlogvar_add], dz = sum over dz_srcs; the addends (the
 * KL term's gradients, [M][z]) may be NULL; dy is [M][2z]. */
int rac_reparam_head_bwd(const rac_grad_src* dz_srcs, int32_t n_srcs, const float* logvar, const float* eps,
                         const float* dmu_add, const float* dlogvar_add, float* dy, int64_t M, int32_t z,
                         uint32_t* dy_amax, void* stream);

/* z = eps*exp(0.5*logvar) + mu (lstm.py:276-279); dlogvar = dz*eps*0.5*exp(0.5*logvar) */
int rac_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, int64_t n, void* stream);
int rac_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dlogvar, int64_t n, void* stream);

/* ------------------------------------------------------------------------ *
 * Frame-level elementwise ops (NCHW planes at the model boundary)
 * ------------------------------------------------------------------------ */
/* packed[b][p][:] = [img[b][0..2][p] * (zmask ? 1-zmask[b][p] : 1) | mask[b][0..Cm-1][p]]
 * = zero_robot_region (src/utils/image.py:5-19) + cat([img, mask]) (dynamics.py:578-582), NCHW -> NHWC */
int rac_pack_input(const float* img, const float* zmask, const float* mask, int32_t Cm, int32_t pad, float* packed,
                   int32_t B, int32_t HW, void* stream);
/* The frozen model's first encoder layer straight from the planes rac_pack_input would pack (no packed tensor):
 *   out[b][y][x][co] = act(scale[co] * conv3x3([img * (zmask == 0) | mask])[co] + shift[co])     NHWC, Cout = 64
 * w: the layer's weight, memory [Cout][3][3][3 + Cm]; scale / shift: eval BatchNorm folded (or NULL); act none / leaky;
 * out_amax as the other `*_amax` outputs.  H, W multiples of 16.  vgg_64.py:8-18 (c1[0]) on dynamics.py:578-582. */
int rac_first_layer_fwd(const float* img, const float* zmask, const float* mask, int32_t Cm, const float* w,
                        const float* scale, const float* shift, int32_t act, float* out, uint32_t* out_amax,
                        int32_t amax_per_image, int32_t B, int32_t H, int32_t W, int32_t Cout, void* stream);
/* amax_per_image: out_amax is an array of B zeroed slots, one per image */
/* The same layer on the split-precision matrix pipe (same arguments): the weights are the MFMA's A operand (M = 64
 * channels), 16 pixels the B operand gathered from the halo planes; every pixel is scaled by the maximum of its own
 * 9 x (3 + Cm) inputs and every channel by the maximum of its own weights (a column of B and a row of A may each carry
 * their own power of two), so no operand maxima are needed and an image's result depends on nothing but the image. */
int rac_first_layer_fwd_split(const float* img, const float* zmask, const float* mask, int32_t Cm, const float* w,
                              const float* scale, const float* shift, int32_t act, float* out, uint32_t* out_amax,
                              int32_t amax_per_image, int32_t B, int32_t H, int32_t W, int32_t Cout, void* stream);
/* The output head, ConvTranspose2d(64 -> 4, 3, 1, 1) + bias + Sigmoid (vgg_64.py:218-220), forward, as exact-fp32 FMAs
 * (a GEMM with N = 4 wastes the matrix pipe):
 *   y[b][p][c] = sigmoid(bias[c] + sum_{ky, kx, ci} x[b][p - (ky - 1, kx - 1)][ci] * w_taps[ky][kx][ci][c])
 * x NHWC [B][H][W][64], y NHWC [B][H][W][4]; w_taps = the (64, 4, 3, 3) ConvTranspose parameter re-ordered to
 * [3][3][64][4].  H % 8 == 0, W % 32 == 0.  A fixed order of sums per pixel: the result does not depend on the batch. */
int rac_head_fwd(const float* x, const float* w_taps, const float* bias, float* y, int32_t B, int32_t H, int32_t W,
                 void* stream);
/* Data gradient of that head w.r.t. its 64-channel input (the autograd of vgg_64.py:218-220; the gradient through the
 * Sigmoid is rac_act_bwd's):  dx[b][y][x][ci] = sum_{ky, kx, co} d[b][y + ky - 1][x + kx - 1][co] * w[ci][ky][kx][co],
 * d NHWC [B][H][W][4], w = the (64, 4, 3, 3) ConvTranspose parameter in its memory order [64][3][3][4], dx NHWC
 * [B][H][W][64] (every element written).  Exact-fp32 FMAs in a fixed order.  W % 16 == 0. */
int rac_head_dgrad(const float* d, const float* w, float* dx, int32_t B, int32_t H, int32_t W, void* stream);
/* The same head on the split-precision matrix pipe, as a tap-stacked 1 x 1 conv Z[tap][q][c] = sum_ci w[tap][ci][c] x[q][ci]
 * (M = 36 rows: weights = the MFMA's A operand, 16 pixels = B, loaded straight from HBM and split in registers) followed by
 * the shifted sum y[p] = sigmoid(bias + sum_tap Z[tap][p + d_tap]) through LDS: every input element is read, converted and
 * multiplied once.  x_amax = the rac_absmax slot of x (amax_per_image: B slots, one per image -- the frozen model); the
 * weight's scale is found in the kernel.  H % 16 == 0, W % 16 == 0. */
int rac_head_fwd_split(const float* x, const uint32_t* x_amax, int32_t amax_per_image, const float* w_taps,
                       const float* bias, float* y, int32_t B, int32_t H, int32_t W, void* stream);
/* parts[i][w][ky][kx][t] = sum over the pixels of workgroup i's 16 x 16 tiles of wide[p][w] * thin[p + (ky - 1, kx - 1)][t]:
 * the weight gradient of a 3x3 conv between a 64-channel NHWC tensor and a thin one (Ct <= 8 channels, row stride
 * thin_stride >= Ct: the packed frame carries pad channels), as n_parts partial sums of 64 * 9 * Ct floats each that
 * rac_slab_accumulate(parts, n_parts, 576 * Ct, dw, 576 * Ct) adds into the gradient (deterministic).  First encoder
 * layer: wide = dy, thin = packed frame, dw = the (64, Ct, 3, 3) weight's gradient (vgg_64.py:8-18 backward); output
 * head: wide = x, thin = d(sigmoid output), dw = the ConvTranspose (64, Ct, 3, 3) parameter's gradient
 * (vgg_64.py:218-220 backward).  H, W multiples of 16; n_parts <= number of tiles. */
int rac_thin_wgrad(const float* wide, const float* thin, int32_t thin_stride, int32_t Ct, float* parts, int32_t n_parts,
                   int32_t B, int32_t H, int32_t W, int32_t Cw, void* stream); /* `pad` trailing zero channels */
/* dimg[b][c][p] = dpacked[b][p][c] * (zmask ? 1-zmask : 1), c < 3 */
int rac_unpack_grad(const float* dpacked, int32_t C, const float* zmask, float* dimg, int32_t B, int32_t HW,
                    void* stream);
/* out = img * (1 - mask)   (also its own backward) */
int rac_zero_region(const float* img, const float* mask, float* out, int32_t B, int32_t HW, void* stream);
/* out[b][c][p] = (1-m)*prev[b][c][p] + m*x4[b][p][c],  m = x4[b][p][3]   (trainer.py:406-407) */
int rac_composite_fwd(const float* x4, const float* prev, float* out, int32_t B, int32_t HW, void* stream);
int rac_composite_bwd(const float* dout, const float* x4, const float* prev, float* dx4, float* dprev, int32_t B,
                      int32_t HW, void* stream);

/* Reconstruction losses + logging metrics in one pass (losses.py:11-78, trainer.py:149-161,426-452).
 * kind: 0 mse, 1 l1, 2 dontcare_mse, 3 dontcare_l1.  out[0] = loss, out[1] = robot_mse, out[2] = world_mse
 * (out[1..2] only when mask != NULL).  per_sample = workspace fp32 [B][8]. */
enum { RAC_LOSS_MSE = 0, RAC_LOSS_L1 = 1, RAC_LOSS_DONTCARE_MSE = 2, RAC_LOSS_DONTCARE_L1 = 3 };
int rac_recon_loss_fwd(int32_t kind, const float* pred, const float* target, const float* mask, float robot_weight,
                       const float* batch_weight, float* per_sample, float* out, int32_t B, int32_t HW, void* stream);
/* dpred = gout[0] * dloss/dpred ; per_sample from the forward call */
int rac_recon_loss_bwd(int32_t kind, const float* pred, const float* target, const float* mask, float robot_weight,
                       const float* batch_weight, const float* per_sample, const float* gout, float* dpred, int32_t B,
                       int32_t HW, void* stream);
/* kl_criterion (losses.py:97-106): out[0] = sum(...) / bs ; partial = workspace fp64 [1] (zeroed by the call) */
int rac_kl_fwd(const float* mu1, const float* lv1, const float* mu2, const float* lv2, int64_t n, int32_t bs,
               double* partial, float* out, void* stream);
int rac_kl_bwd(const float* mu1, const float* lv1, const float* mu2, const float* lv2, const float* gout, int64_t n,
               int32_t bs, float* dmu1, float* dlv1, float* dmu2, float* dlv2, void* stream);

/* Evaluation metrics of the eval path (src/utils/metrics.py:13-78, trainer.py:681-693), one launch:
 *   a' = zero_robot_region(mask, a), b' likewise (mask may be NULL);
 *   sq_err[n] += sum_{c,h,w} (clamp01(a') - clamp01(b'))^2 / 4          -> PSNR = 10 log10(1 / (sq_err / (3HW)))
 *                                                                         (the reference maps both to (x+1)/2 first)
 *   ssim_sum[n] += sum of the SSIM map of (a', b'): 11x11 Gaussian window sigma 1.5, zero padding, C1 = 0.01^2,
 *   C2 = 0.03^2; the map itself is written to ssim_map [N][3][H][W] when non-NULL.
 * sq_err / ssim_sum are fp32 [N] and must be zero on entry. */
int rac_psnr_ssim(const float* a, const float* b, const float* mask, float* sq_err, float* ssim_sum, float* ssim_map,
                  int32_t N, int32_t H, int32_t W, void* stream);

/* CEM step tail (trajectory_sampler.py:149-169 + losses.py:224-263), fused:
 *   next = (1-m)*curr + m*rgb; next *= (1-next_mask) if next_mask;
 *   cost = -sqrt(sum (255*(next-goal))^2) [dontcare: robot|goal-mask pixels dropped, / #world pixels]
 *   sum_cost[n] += (double)(weight * cost)   (fp32 product as the reference's `w * cost`, losses.py:313-316; fp64
 *                                             accumulate, kept on the device)
 * kind: 0 ImgL2Cost, 1 ImgDontcareCost.  cost_mask = thick robot mask of the next step (dontcare only). */
int rac_cem_step_tail(const float* x4, const float* curr, const float* next_mask, const float* goal_img,
                      const float* cost_mask, const uint8_t* goal_mask, int32_t kind, float weight, int32_t add_cost,
                      float* next_out, double* sum_cost, int32_t N, int32_t HW, void* stream);

/* Robot-aware CEM inputs for ALL candidates on the device (replaces the per-candidate CPU loop of
 * `robot_model.predict_batch(start_data, thick=True)`, src/cem/trajectory_sampler.py:86-109 over
 * src/dataset/wx250s/wx250s_model.py:121-163 / locobot_model.py:100-140):
 *   states[t][n] = normalised (x, y, push_height, 0, 0) with (x, y) = start + sum_{s<t} actions[s][n][0:2]   (t >= 1;
 *                  t = 0: the start state), computed in the robot's own frame (`diff` = offset of that frame) with the
 *                  reference's float32 / float64 promotion;
 *   masks[t][n]  = atlas[nearest grid node of (x, y)]: robot masks rendered ONCE by the analytical model on a regular
 *                  grid of end-effector positions (node (i, j) at (x0 + i dx, y0 + j dy); uint8 [ny][nx][HW]).
 * actions [T][N][A] time-first; start_state / low / high [5], or [N][5] with per_sample != 0 (the windows of a training
 * batch: every sample has its own start and bounds); outputs states [T+1][N][5], masks [T+1][N][HW] fp32. */
int rac_cem_robot_inputs(const float* actions, const float* start_state, const float* low, const float* high,
                         const uint8_t* atlas, int32_t nx, int32_t ny, float x0, float y0, float dx, float dy,
                         float diff_x, float diff_y, float push_height, float* states, float* masks, int32_t T,
                         int32_t N, int32_t A, int32_t HW, int32_t per_sample, void* stream);

/* torch.optim.Adam step (trainer.py:109-110,461), fused over one flat buffer. step >= 1. */
int rac_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int32_t step, void* stream);

/* The optimiser step of the split-precision conv weights fused with the refresh of their operand parts: per job (one
 * conv weight, [Cout][k][k][Cin] slices of the flat parameter / gradient / moment buffers) Adam as rac_adam_step, the
 * fp16 fragment parts of the UPDATED weight (forward and / or transposed, as rac_weight_frag_split writes them) and the
 * exact max |p_new| folded into amax_out (zero on entry).  The parts are scaled by the power of two that *scale_slot
 * implies, which must be an upper bound of max |p_new|: the caller derives it from the previous exact maximum and the
 * largest step Adam can take (rac_amax_bound), and hands the same slot to the convs as w_amax.  Replaces rac_adam_step
 * + rac_absmax_multi + rac_weight_frag_split_multi on these weights: one pass over p instead of four.
 * Jobs in DEVICE memory; job j owns the workgroups [block_begin_j, block_begin_{j+1}), rac_weight_frag_blocks each. */
typedef struct rac_adam_frag_job {
  float* p;
  const float* g;
  float* m;
  float* v;
  const uint32_t* scale_slot;
  uint32_t* amax_out;
  uint16_t* parts_fwd; /* or NULL */
  uint16_t* parts_t;   /* or NULL */
  int64_t part_stride;
  int32_t Cout, Cin, ksize, reserved;
  int64_t block_begin;
} rac_adam_frag_job;
int rac_adam_frag_multi(const rac_adam_frag_job* jobs, int32_t n_jobs, int64_t total_blocks, float lr, float beta1,
                        float beta2, float eps, int32_t step, void* stream);
/* The same pass on at most `max_workgroups` workgroups, each walking every max_workgroups-th block of work: an update that
 * runs beside other streams' kernels (optim.FusedAdam's late group under the next step's encoder) leaves them the CUs'
 * registers and LDS; 2 per CU (512) keep ~64 KB per CU in flight.  Same bits as rac_adam_frag_multi. */
int rac_adam_frag_multi_bounded(const rac_adam_frag_job* jobs, int32_t n_jobs, int64_t total_blocks, int32_t max_workgroups,
                                float lr, float beta1, float beta2, float eps, int32_t step, void* stream);
/* bound[idx[j]] = bits(float(exact[idx[j]]) + margin); exact[idx[j]] = 0   (j < n; idx in device memory) */
int rac_amax_bound(uint32_t* exact, uint32_t* bound, const int32_t* idx, int32_t n, float margin, void* stream);
/* rac_adam_step over a list of float4 ranges [begin4, begin4 + n4) of the flat buffers (device table; range r owns the
 * workgroups [block_begin_r, block_begin_{r+1}), ceil(n4 / 1024) each): everything the fused jobs do not cover. */
typedef struct rac_adam_range {
  int64_t begin4, n4, block_begin;
} rac_adam_range;
int rac_adam_ranges(float* p, const float* g, float* m, float* v, const rac_adam_range* ranges, int32_t n_ranges,
                    int64_t total_blocks, float lr, float beta1, float beta2, float eps, int32_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAC_HIP_H */
