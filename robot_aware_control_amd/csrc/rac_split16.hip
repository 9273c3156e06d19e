// Split-precision convolutions on the fp16 matrix pipe of gfx950 (v_mfma_f32_16x16x32_f16, fp32 accumulation).
//
// Every fp32 operand tensor X is brought to fp16 range by a power-of-two scale s_X = 2^k (k from the tensor's
// max |x|, so that max |x| s_X lies in [2^14, 2^15)) and written as the sum of two fp16 parts,
//     x s_X = h1 + h2 (+ r),   h1 = fp16(x s_X),  h2 = fp16(x s_X - h1),   |r| <= 2^-22 |x s_X|,
// and a product keeps the three part-products h1 g1 + h1 g2 + h2 g1 (the dropped h2 g2 is <= 2^-22 relative):
// 22 bits of every operand enter the MFMA, products are exact in the fp32 accumulator's input, sums are fp32.
// Against an fp64 reference the result is as close as the exact-fp32 MFMA kernels of rac_igemm.hip (the rounding of
// the fp32 accumulation dominates both: tests/test_gpu_ops.py holds every kernel here to <= 4x the fp32 kernel's
// error), at 3 fp16 MFMA products per fp32 product: algorithmic peak 2500 / 3 = 833 TFLOP/s.
// The scale is exact (power of two) and removed in the epilogue; tensors' max |x| are produced on the device
// (rac_absmax, an atomic max over the value's bit pattern) so nothing synchronises with the host.
//
// Kernels:
//   conv16_tile_kernel      forward conv (= data gradient with the transposed weight) for maps that fit a tile
//                           (H*W <= 128: the 8x8 / 6x8 latent maps): activations staged once per 32-channel chunk and
//                           reused by all k*k taps through a wave-uniform row shift, weights in MFMA fragment order
//                           straight from L2 into registers.
//   conv16_rows_kernel      the same for larger maps: a tile = R whole image rows plus a halo.
//   wgrad16_kernel          weight gradient: both operands in their natural NHWC layout, transposed on the way out of
//                           LDS by ds_read_b64_tr_b16; K walks pixel COLUMNS so that one staged input column serves
//                           every horizontal tap and the border taps are skipped instead of masked.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "rac_common.h"

namespace rac {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr unsigned OOB = 0xFFFFFFF0u;  // buffer offset past every range: the load returns zeros
constexpr int SBN = 128, SBK = 32;

__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  unsigned long long v = reinterpret_cast<unsigned long long>(p);
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(p)), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 load16(rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}

// power-of-two scale of a tensor from the bit pattern of its max |x|: max |x| * scale in [2^14, 2^15)
// (k clamped to +-80; an all-zero tensor gets scale 1)
__host__ __device__ __forceinline__ int scale_exp(unsigned amax_bits) {
  const int e = (int)((amax_bits >> 23) & 0xFF);
  if (e == 0) return 0;
  int k = 14 - (e - 127);
  return k > 80 ? 80 : (k < -80 ? -80 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __builtin_bit_cast(float, (unsigned)(k + 127) << 23); }

#ifndef RAC_EPILOGUE_FAST  // the conv kernels' predicate-free epilogue with buffer stores (0: the general path always)
#define RAC_EPILOGUE_FAST 1
#endif
// timing builds of the persistent rows kernel (tools/build_variant.sh ... -DRAC_EXP_PERSIST=<bits>; results are WRONG):
// 1 epilogue without its stores, 2 no epilogue, 4 staging with one conversion instead of the split, 8 fragment reads at tap 0
// only, 16 no weight loads inside the loop, 32 the next tile = this tile (no per-tile index arithmetic, staging loads hit L2), 64 the arithmetic done but this tile's addresses used
#ifndef RAC_EXP_PERSIST
#define RAC_EXP_PERSIST 0
#endif

// eight fp32 values (two 16-byte vectors) * scale -> two fp16 parts, packed as 16-byte MFMA operand vectors
__device__ __forceinline__ void split8h(u32x4 lo, u32x4 hi, float s, u32x4 (&q)[2]) {
  const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  f16x8 h1, h2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = __builtin_bit_cast(float, w[j]) * s;
    const _Float16 a = (_Float16)v;
    h1[j] = a;
    h2[j] = (RAC_EXP_PERSIST & 4) ? a : (_Float16)(v - (float)a);
  }
  q[0] = __builtin_bit_cast(u32x4, h1);
  q[1] = __builtin_bit_cast(u32x4, h2);
}

// acc += a * b with a = a1 + a2, b = b1 + b2 (fp16 parts): smallest terms first
__device__ __forceinline__ f32x4 mma3(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], c, 0, 0, 0);
  return c;
}

struct Conv16P {
  int B, H, W, ks, pad, Cin, Cout, act, split_k, a_split;
  long slab_stride;
  const float *a0, *a1;       // fp32 NHWC activations (virtual concat at a_split)
  const unsigned short* w;    // fp16 parts, fragment order [part][Cout/32][Cin/32][tap][nb 2][lane 64][8]
  long w_ps;                  // part stride in elements
  const unsigned *a_amax0, *a_amax1, *w_amax;  // bit patterns of max |x| of the operands (a_amax1 may be null)
  unsigned* out_amax;         // max |out| is folded in here (fused epilogue only; may be null)
  float* out0;
  const float *bias, *scale, *shift;
  double* stats;
  long stats_rows;
  int M, N, HW, P, taps, cchunks, nchunks, cps;
  int w_nchunks;  // chunk-taps per 32-column tile in the weight parts (>= nchunks: the conv may use a channel prefix)
  int xcd_group;  // remap workgroup ids so that the M-tiles sharing one weight slab run on one XCD (one L2).  (Only when
                  // the (N-tile, K-split) groups are a multiple of 8; giving the narrow layers' 1 / 2 / 4 groups 8 / groups
                  // XCDs each measured +0.5 % on a planner iteration: their slabs stream from the MALL just as well.)
  int tile_m;     // output rows per workgroup (whole images / whole image rows, a multiple of 16, <= 128)
  int n_store;    // columns stored and row stride of the output: N, or fewer when the weight rows are zero-padded to 32
  int a0_up;      // rows kernel: a0 is the half-resolution tensor, read at (y / 2, x / 2) (nearest 2x upsampling)
  // frozen ConvLSTM cell in the epilogue (tile kernel only): the weight rows are gate-interleaved in groups of 16 channels
  // (column 64 (c / 16) + 16 gate + c % 16), so a lane holds the four gates of one channel; out0 is not written
  const float* lstm_c_prev;
  float *lstm_h, *lstm_c;
  int lstm_g;
  int per_image;  // a_amax0 / a_amax1 / out_amax are arrays of B slots, one per image: every image is scaled by its OWN
                  // maximum, so its result cannot depend on what else is in the batch (the frozen model's rollouts)
  // 2-D tiles of the unrolled 3x3 rows kernels (0: tiles of whole image rows): a tile is seg_h image rows x seg_w pixels
  // (= 128 pixels; seg_w a multiple of 16, so a 16-row block lies inside one tile row) with a one-pixel halo all round:
  // (seg_h + 2)(seg_w + 2) staged pixels instead of (rows + 2) W -- 180 instead of 264 on a 64-wide map -- and maps wider
  // than a whole-row tile's halo allows (128x128) get a tile at all.  seg_tx tiles per image row, seg_tpi per image.
  int seg_w, seg_h, seg_tx, seg_tpi;
  // 2 x 2 max pooling of the activated output in the same epilogue (unrolled 3x3 rows kernels, full tiles only: the
  // launcher's conditions): pool_out[B][H/2][W/2][n_store]; pool_bpr = 16-row blocks per tile row (1 or 2): the block
  // under block b is b + pool_bpr, both held by one wave; a lane's four rows are four consecutive pixels of one image row.
  float* pool_out;
  int pool_bpr;
};

// origin of 2-D tile `bx`: image, first row, first column, and the pixel index of (y0, x0)
struct SegOrigin {
  int img, y0, x0, m0;
};
__device__ __forceinline__ SegOrigin seg_origin(const Conv16P& p, int bx) {
  SegOrigin o;
  o.img = bx / p.seg_tpi;
  const int r = bx - o.img * p.seg_tpi;
  const int ty = r / p.seg_tx;
  o.y0 = ty * p.seg_h;
  o.x0 = (r - ty * p.seg_tx) * p.seg_w;
  o.m0 = (o.img * p.H + o.y0) * p.W + o.x0;
  return o;
}
// pixel index of tile row `r16` (a multiple of 16) of the tile at m0: consecutive pixels from there up to the block's end
__device__ __forceinline__ int seg_row_pixel(const Conv16P& p, int m0, int r16) {
  return p.seg_w ? m0 + (r16 / p.seg_w) * p.W + r16 % p.seg_w : m0 + r16;
}

// LDS image of the tile kernel: CHUNK-major [part 2][8-channel group 4][row 144][16 B]; rows 128..143 are zeros.
// The 16 lanes of a ds_read_b128 group touch 16 distinct rows mod 16 = 16 distinct bank slots; a tap shift is one
// wave-uniform byte offset.
constexpr int T16_CP = 144 * 16, T16_PP = 4 * T16_CP, T16_ABUF = 2 * T16_PP;
// The y-major form (W = 8, two images per tile) keeps zero slots beside every image-row segment -- image row y of image i
// sits in LDS rows 36 y + (2 | 26) + x, two zero rows either side of each segment: a horizontal tap that leaves the
// image reads the padding, so a fragment address is the lane's constant + one wave-uniform tap offset + an immediate per
// block: no per-lane validity mask, compare and select (12 of the ~38 VALU instructions per 48 MFMAs).  The second
// image's segment starts 24 rows after the first's (= 8 mod 16): the 16 lanes of a read group still touch 16 distinct
// 16-byte slots mod 256 B (a 16-row pitch put both segments on the same slots: 2-way bank conflicts on every read,
// SQ_LDS_BANK_CONFLICT = half of the active LDS cycles).  288 rows per plane, 36 KB per buffer.
constexpr int T16Y_PITCH = 36, T16Y_S0 = 2, T16Y_S1 = 26;
constexpr int T16Y_CP = 8 * T16Y_PITCH * 16, T16Y_PP = 4 * T16Y_CP, T16Y_ABUF = 2 * T16Y_PP;

// Epilogue of both kernels: undo the two operand scales, then as rac_conv2d FWD (bias, fp64 BatchNorm statistics of the
// biased value, folded eval-BatchNorm scale / shift, activation, max |v|) -- or the raw partial sums of a K split.
// The accumulator block (mb, nb) holds rows m0 + (mb0 + mb) * 16 + 4 (lane >> 4) + reg, column ncol0 + 16 nb + (lane & 15).
// Written without per-element branches: a taken branch costs more than the arithmetic it would skip.
// `ia_rows` (LDS, per-image scales): 1 / scale of every tile row instead of the one `ia`.  mxb[mb] = max |v| of the
// lane's values in 16-row block mb (the caller folds them into one slot, or into one slot per image).
// `ym_hw` (0: off): the tile kernel's y-major row order -- tile row 16 y + 8 i + x holds pixel (y, x) of the tile's image
// i (W = 8, two images of ym_hw pixels per tile): block mb is image row mb of BOTH images.
template <int SIG, int MBLK, int NBLK>
__device__ __forceinline__ void conv16_epilogue_body(const Conv16P& p, const f32x4 (&acc)[MBLK][NBLK], int m0, int mb0,
                                                     int nmb, int ncol0, float ia, float iw, const float (*pre)[3],
                                                     const float* ia_rows, unsigned (&mxb)[MBLK], int ym_hw = 0) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int NS = p.n_store;
  const float slope = p.act == RAC_ACT_LEAKY02 ? 0.2f : 1.f;
  float iav[MBLK][4];
#pragma unroll
  for (int mb = 0; mb < MBLK; ++mb) {
    mxb[mb] = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) iav[mb][r] = ia_rows ? ia_rows[min((mb0 + mb) * 16 + 4 * lq + r, 127)] : ia;
  }
  // FAST PATH (wave-uniform; every tile but a ragged last one, every layer whose columns fill the wave's blocks; not the
  // training pass's statistics): no per-value row / column predicates, and the stores are buffer stores -- one base per
  // 16-row block (scalar), four lane offsets for the lane's four rows computed once, the column block as the instruction's
  // immediate -- instead of a 64-bit address per value.  Same arithmetic on every value: the same bits.  (The per-tile fixed
  // costs of the narrow layers' kernels add up, profiles/r04_persist_decomposition.md: this is the largest of them.)
  const int blk_last = min(mb0 + MBLK, nmb) - 1;  // the wave's last live block
  const int m_last = ym_hw ? m0 + ym_hw + blk_last * 8 + 7 : seg_row_pixel(p, m0, blk_last * 16) + 15;
  if (RAC_EPILOGUE_FAST && !p.stats && blk_last >= mb0 && m_last < p.M && ncol0 + NBLK * 16 <= NS &&
      (long)p.M * NS * 4 < (1l << 32)) {
    unsigned voff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = 4 * lq + r;
      voff[r] = (unsigned)(((ym_hw ? (rl >> 3) * ym_hw + (rl & 7) : rl) * NS + lr) * 4);
    }
    if (p.pool_out && !ym_hw) {
      // with the 2 x 2 max pool: per column block the values of all row blocks first, then every (block, block below) pair
      // gives the lane two pooled pixels (its four rows are pixels x .. x + 3 of one image row)
      const int Hp = p.H >> 1, Wp = p.W >> 1;
      rsrc_t dst[MBLK], pdst[MBLK];
#pragma unroll
      for (int mb = 0; mb < MBLK; ++mb) {
        const int mblk = __builtin_amdgcn_readfirstlane(seg_row_pixel(p, m0, (mb0 + mb) * 16));
        dst[mb] = make_rsrc(p.out0 + (long)mblk * NS + ncol0, (unsigned)(((long)(p.M - mblk) * NS - ncol0) * 4));
        const int img = mblk / p.HW, rem = mblk - img * p.HW;
        const int y = rem / p.W, x0 = rem - y * p.W;
        const int pf = (img * Hp + (y >> 1)) * Wp + (x0 >> 1);  // (used for the upper block of a pair: y even)
        pdst[mb] = make_rsrc(p.pool_out + (long)pf * NS + ncol0, (unsigned)(((long)((p.M >> 2) - pf) * NS - ncol0) * 4));
      }
      const unsigned pvoff = (unsigned)((2 * lq * NS + lr) * 4);
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb) {
        const float bias = pre ? pre[nb][0] : (p.bias ? p.bias[ncol0 + nb * 16 + lr] : 0.f);
        const float sc = pre ? pre[nb][1] : (p.scale ? p.scale[ncol0 + nb * 16 + lr] : 1.f);
        const float sh = pre ? pre[nb][2] : (p.scale ? p.shift[ncol0 + nb * 16 + lr] : 0.f);
        float vv[MBLK][4];
#pragma unroll
        for (int mb = 0; mb < MBLK; ++mb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[mb][nb][r] * iav[mb][r] * iw + bias;
            v = v * sc + sh;
            v = v > 0.f ? v : slope * v;
            if (SIG) v = sigmoid_acc(v);
            vv[mb][r] = v;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), dst[mb], (int)voff[r] + nb * 64, 0, 0);
            mxb[mb] = max(mxb[mb], absbits(v));
          }
#pragma unroll
        for (int mb = 0; mb < MBLK; ++mb) {
          // upper blocks of the pairs: tile rows 0, 2, ... = blocks with (mb / bpr) even
          const bool up = p.pool_bpr == 2 ? ((mb >> 1) & 1) == 0 : (mb & 1) == 0;
          if (!up) continue;  // (wave-uniform)
#pragma unroll
          for (int j = 0; j < 2; ++j) {  // as rac_maxpool2_fwd: max(max(top pair), max(bottom pair))
            const float a = fmaxf(vv[mb][2 * j], vv[mb][2 * j + 1]);
            const float b = p.pool_bpr == 2 ? fmaxf(vv[(mb + 2) % MBLK][2 * j], vv[(mb + 2) % MBLK][2 * j + 1])
                                            : fmaxf(vv[(mb + 1) % MBLK][2 * j], vv[(mb + 1) % MBLK][2 * j + 1]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(a, b)), pdst[mb],
                                                  (int)pvoff + j * NS * 4 + nb * 64, 0, 0);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb) {
      if (mb0 + mb >= nmb) continue;  // wave-uniform: past the tile's rows
      const int mblk = __builtin_amdgcn_readfirstlane(ym_hw ? m0 + (mb0 + mb) * 8 : seg_row_pixel(p, m0, (mb0 + mb) * 16));
      const rsrc_t dst = make_rsrc(p.out0 + (long)mblk * NS + ncol0, (unsigned)(((long)(p.M - mblk) * NS - ncol0) * 4));
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb) {
        const float bias = pre ? pre[nb][0] : (p.bias ? p.bias[ncol0 + nb * 16 + lr] : 0.f);
        const float sc = pre ? pre[nb][1] : (p.scale ? p.scale[ncol0 + nb * 16 + lr] : 1.f);
        const float sh = pre ? pre[nb][2] : (p.scale ? p.shift[ncol0 + nb * 16 + lr] : 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[mb][nb][r] * iav[mb][r] * iw + bias;
          v = v * sc + sh;
          v = v > 0.f ? v : slope * v;
          if (SIG) v = sigmoid_acc(v);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), dst, (int)voff[r] + nb * 64, 0, 0);
          mxb[mb] = max(mxb[mb], absbits(v));
        }
      }
    }
    return;
  }
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb) {
    const int n = ncol0 + nb * 16 + lr;
    const bool nok = n < NS;
    const int nc = nok ? n : 0;
    // (bias, scale, shift) of the column: fetched here, or by the caller before its main loop -- at this point a
    // global load is one more exposed round trip
    const float bias = pre ? pre[nb][0] : (p.bias ? p.bias[nc] : 0.f);
    const float sc = pre ? pre[nb][1] : (p.scale ? p.scale[nc] : 1.f);
    const float sh = pre ? pre[nb][2] : (p.scale ? p.shift[nc] : 0.f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb) {
      if (mb0 + mb >= nmb) continue;  // wave-uniform: past the tile's rows
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = 4 * lq + r;
        const int m = ym_hw ? m0 + (rl >> 3) * ym_hw + (mb0 + mb) * 8 + (rl & 7) : seg_row_pixel(p, m0, (mb0 + mb) * 16) + rl;
        const bool ok = nok & (m < p.M);
        float v = acc[mb][nb][r] * iav[mb][r] * iw + bias;  // two exact steps: ia * iw alone may underflow
        const float vs = ok ? v : 0.f;
        s1 += vs;
        s2 += vs * vs;
        v = v * sc + sh;
        v = v > 0.f ? v : slope * v;
        if (SIG) v = sigmoid_acc(v);
        if (ok) {
          if (!(RAC_EXP_PERSIST & 1)) p.out0[(long)m * NS + n] = v;
          mxb[mb] = max(mxb[mb], absbits(v));
        }
      }
    }
    if (p.stats) {
      s1 += __shfl_xor(s1, 16);
      s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lq == 0 && nok) {
        double* sg = p.stats + (p.stats_rows ? (long)(m0 / p.stats_rows) * 2 * NS : 0L);
        atomicAdd(sg + n, (double)s1);
        atomicAdd(sg + NS + n, (double)s2);
      }
    }
  }
}

// `out_amax`: the slot of the tile's output maximum -- p.out_amax, or with per-image scales the slot of the tile's first
// image (ia_rows == nullptr: the whole tile lies in ONE image, the rows kernel) / of image 0 (ia_rows given: the tile
// holds whole images of HW rows, HW % 16 == 0, the tile kernel; each image's blocks go to that image's slot).
template <int MBLK, int NBLK>
__device__ __forceinline__ void conv16_epilogue(const Conv16P& p, const f32x4 (&acc)[MBLK][NBLK], int m0, int mb0, int nmb,
                                                int ncol0, int bz, float ia, float iw, const float (*pre)[3] = nullptr,
                                                const float* ia_rows = nullptr, unsigned* out_amax = nullptr,
                                                int ym_hw = 0) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int NS = p.n_store;
  if (p.split_k > 1) {  // raw partial sums: the ConvLSTM cell kernel / rac_slab_reduce adds the slabs
    float* dst = p.out0 + (long)bz * p.slab_stride;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
      const int n = ncol0 + nb * 16 + lr;
#pragma unroll
      for (int mb = 0; mb < MBLK; ++mb) {
        if (mb0 + mb >= nmb) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = 4 * lq + r;
          const int m = ym_hw ? m0 + (rl >> 3) * ym_hw + (mb0 + mb) * 8 + (rl & 7) : seg_row_pixel(p, m0, (mb0 + mb) * 16) + rl;
          if ((n < NS) & (m < p.M)) dst[(long)m * NS + n] = acc[mb][nb][r] * ia * iw;
        }
      }
    }
    return;
  }
  unsigned mxb[MBLK];
  if (p.act == RAC_ACT_SIGMOID)
    conv16_epilogue_body<1>(p, acc, m0, mb0, nmb, ncol0, ia, iw, pre, ia_rows, mxb, ym_hw);
  else
    conv16_epilogue_body<0>(p, acc, m0, mb0, nmb, ncol0, ia, iw, pre, ia_rows, mxb, ym_hw);
  if (!out_amax) return;
  if (ia_rows && ym_hw) {  // y-major tile, one slot per image: lanes with lq < 2 hold image 0's rows, the others image 1's
    unsigned mx = 0;
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb) mx = max(mx, mxb[mb]);
    const int img0 = m0 / ym_hw;
    amax_commit(lq < 2 ? mx : 0u, out_amax + img0);
    if (m0 + ym_hw < p.M) amax_commit(lq >= 2 ? mx : 0u, out_amax + img0 + 1);
    return;
  }
  if (!ia_rows) {  // one slot for everything this wave wrote
    unsigned mx = 0;
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb) mx = max(mx, mxb[mb]);
    amax_commit(mx, out_amax);
    return;
  }
  int cur = -1;  // (wave-uniform) image of the blocks folded into mx so far
  unsigned mx = 0;
#pragma unroll
  for (int mb = 0; mb < MBLK; ++mb) {
    const int row = m0 + (mb0 + mb) * 16;
    if (mb0 + mb >= nmb || row >= p.M) continue;
    const int img = row / p.HW;
    if (img != cur) {
      if (cur >= 0) amax_commit(mx, out_amax + cur);
      cur = img, mx = 0;
    }
    mx = max(mx, mxb[mb]);
  }
  if (cur >= 0) amax_commit(mx, out_amax + cur);
}

// e^x to ~1e-7 relative for |x| < 80 in 6 instructions: v_exp_f32 on x log2(e) with the product's rounding error (an
// FMA recovers it exactly) and log2(e)'s own carried into a first-order correction.  libm's expf / tanhf cost an order of
// magnitude more -- enough to lose 2 % of a planner iteration when the cell sat in the GEMM epilogue (round 2).
__device__ __forceinline__ float exp_fast(float x) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;
  const float t = x * L2E_HI;
  const float e = __builtin_fmaf(x, L2E_HI, -t) + x * L2E_LO;
  return __builtin_amdgcn_exp2f(t) * __builtin_fmaf(e, 0.693147182464599609375f, 1.0f);
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_fast(-x)); }
// 1 - 2 / (1 + e^2x): absolute error ~1e-7 everywhere (what a gate activation needs: it multiplies O(1) states)
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + exp_fast(2.0f * x)); }

// ConvLSTM cell of the FROZEN model in the gate conv's epilogue (lstm.py:136-149): the 4g-wide gate tensor (524 MB per
// planner step at 1000 candidates) is never written and read back, and the separate cell launch disappears.  The wave's
// four 16-column blocks are the i, f, o, g gates of 16 channels (gate-interleaved weight rows).
template <int MBLK>
__device__ __forceinline__ void conv16_lstm_epilogue(const Conv16P& p, const f32x4 (&acc)[MBLK][4], int m0, int mb0, int nmb,
                                                     int ncol0, float ia, float iw, const float* ia_rows, int ym_hw) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int g = p.lstm_g;
  const int c = (ncol0 >> 6) * 16 + lr;
  if (c >= g) return;
  const float bi = p.bias[c], bf = p.bias[g + c], bo = p.bias[2 * g + c], bg = p.bias[3 * g + c];
#pragma unroll
  for (int mb = 0; mb < MBLK; ++mb) {
    if (mb0 + mb >= nmb) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = 4 * lq + r;
      const int m = ym_hw ? m0 + (rl >> 3) * ym_hw + (mb0 + mb) * 8 + (rl & 7) : m0 + (mb0 + mb) * 16 + rl;
      if (m >= p.M) continue;
      const float s = ia_rows ? ia_rows[(mb0 + mb) * 16 + rl] : ia;
      const float gi = sigmoid_fast(acc[mb][0][r] * s * iw + bi), gf = sigmoid_fast(acc[mb][1][r] * s * iw + bf);
      const float go = sigmoid_fast(acc[mb][2][r] * s * iw + bo), gg = tanh_fast(acc[mb][3][r] * s * iw + bg);
      const long i = (long)m * g + c;
      const float cn = gf * p.lstm_c_prev[i] + gi * gg;
      p.lstm_c[i] = cn;
      p.lstm_h[i] = go * tanh_fast(cn);
    }
  }
}

// WM = waves along the rows: 1 -> waves 1 x 4, each all 128 rows x 32 columns; 2 -> waves 2 x 2, each 64 rows x 64
// columns (half the LDS fragment reads and tap-shift selects per MFMA, twice the weight loads per wave).
// FULL: the tile is 128 rows (8x8 / 4x8 / ... maps): no per-block liveness tests inside the loop.
// YM: y-major tile rows (W = 8, two images per tile): LDS / accumulator row 16 y + 8 i + x = pixel (y, x) of image i, so a
// 16-row block is ONE image row of both images and a vertical tap that leaves the image leaves it for the whole block:
// the block's fragment reads and MFMAs are skipped -- 15 % of a 5x5 conv's products on 8-row maps (image rows 0, 1, 6, 7
// lose 2, 1, 1, 2 of their 5 kernel rows), 8 % of a 3x3 conv's -- instead of multiplying zero rows.  A tap's shift
// stays one wave-uniform offset (16 rows per image row); the epilogue maps rows back to the tensor's (image, y, x) order.
#ifndef RAC_TILE_READ_ALL
#define RAC_TILE_READ_ALL 0
#endif
#ifndef RAC_EXP_ROWS_NOZERO
#define RAC_EXP_ROWS_NOZERO 0
#endif
#ifndef RAC_WGRAD_ROLL  // weight-gradient kernel: input fragments rolled through the taps in halves (0: read per tap, then wait)
#define RAC_WGRAD_ROLL 1
#endif
#ifndef RAC_WGRAD_LATE_BLOCKS  // 5x5 128-co weight gradient on operand parts: MFMA blocks in front of the last one that take store pieces
#define RAC_WGRAD_LATE_BLOCKS 2
#endif
#ifndef RAC_EXP_WGRAD  // timing builds of the weight-gradient kernel (wrong results): 1 = no barrier inside the loop, 2 = no global loads inside the loop, 4 = no LDS stores inside the loop
#define RAC_EXP_WGRAD 0
#endif
#ifndef RAC_EXP_TILE  // timing builds of the tile kernel (wrong results): 1 = activations requested for the first chunk only, 2 = every chunk requests the FIRST chunk's addresses (L2 hits)
#define RAC_EXP_TILE 0
#endif
#ifndef RAC_EXP_ROWS_NOSTORE
#define RAC_EXP_ROWS_NOSTORE 0
#endif
#ifndef RAC_EXP_ROWS_NOB
#define RAC_EXP_ROWS_NOB 0
#endif
#ifndef RAC_TILE_A_BEHIND  // tile kernel's unrolled instances: activations requested behind a chunk's first weight request
#define RAC_TILE_A_BEHIND 1
#endif
#ifndef RAC_A_REQUEST_TAP  // the tap of a chunk behind whose weight request the next chunk's activations are requested
#define RAC_A_REQUEST_TAP 0
#endif
#ifndef RAC_ROWS_GENERIC_SEG
#define RAC_ROWS_GENERIC_SEG 1
#endif
#ifndef RAC_ROWS_REFILL  // the rows kernel's weight sets refilled in place, quarter by quarter (0: one set kept free for the requests)
#define RAC_ROWS_REFILL 1
#endif
#ifndef RAC_ROWS_REFILL_PIN  // scheduling barriers around a quarter's refill request: 1 = before, 2 = behind
#define RAC_ROWS_REFILL_PIN 3
#endif
#ifndef RAC_ROWS_RING3  // three weight register sets in the 128-column rows kernel too (needs its registers freed elsewhere)
#define RAC_ROWS_RING3 0
#endif
// KU = 3: the 9 taps of a 3x3 conv's chunk unrolled (K ranges are whole chunks): tap constants, and every wait exact.
template <int WM, bool FULL, bool YM = false, int KU = 0>
__global__ __launch_bounds__(256, 2) void conv16_tile_kernel(Conv16P p) {
  constexpr int WN = 4 / WM;   // waves along the columns
  constexpr int RB = 8 / WM;   // 16-row blocks per wave
  constexpr int NT = WM;       // 32-column weight tiles per wave
  constexpr int NB = 2 * NT;   // 16-column blocks per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int lr = lane & 15, lq = lane >> 4;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_group == 3) {
    // many M-tiles, K unsplit (the planner's shapes): an XCD owns the column tiles x, x + 8, ... and walks the M-tiles with
    // the column tile as the FAST index -- the workgroups that read one activation tile run back to back on one L2 (one HBM
    // fetch per line instead of one per column tile of the XCD), the XCD's weight slabs stream side by side
    const int mt = gridDim.x, nt = gridDim.y, npx = nt >> 3;
    const int lin = bx + mt * by;
    const int xcd = lin & 7, s = lin >> 3;
    bx = s / npx;
    by = xcd + 8 * (s - bx * npx);
  } else if (p.xcd_group) {
    const int mt = gridDim.x, nt = gridDim.y;
    const int lin = bx + mt * (by + nt * bz);
    const int xcd = lin & 7, s = lin >> 3;
    const int grp = (s / mt) * 8 + xcd;
    bx = s % mt;
    by = grp % nt;
    bz = grp / nt;
  }
  const int TM = FULL ? 128 : p.tile_m;
  const int nmb = FULL ? 8 : TM >> 4;  // live 16-row blocks of the tile's 8
  // A short tile's live blocks are dealt out EVENLY over the wave rows (round 6): RBL blocks per wave row, wave row wm
  // holding blocks wm RBL ... -- a 6x8 map's tile (two images, 6 live blocks) gave wave row 0 four blocks and wave row 1
  // two, so the tile took the time of a full one for 96 rows; now 3 + 3: a quarter of the MFMA time of every conv on the
  // reference's default 48x64 frame's latent maps.  A full tile: RBL = RB (compile time, nothing changes).
  const int RBL = FULL ? RB : (nmb + WM - 1) / WM;
  const int mbw = FULL ? wm * RB : wm * RBL;  // the wave's first block
  const int m0 = bx * TM, n0 = by * SBN;
  const int kc_begin = bz * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);
  constexpr int CP = YM ? T16Y_CP : T16_CP, PP = 4 * CP, ABUF = 2 * PP;
  const int srow = tid & 127;  // (the tile row whose per-image scale this thread publishes)
  // the accumulator row of the staged pixel: its tile row, or (YM) 16 y + 8 image + x ...
  const int lrow = (YM && srow < TM) ? ((srow % p.HW) >> 3) * 16 + (srow / p.HW) * 8 + (srow & 7) : srow;
  __shared__ float ia_sh[128];  // per-image scales: 1 / scale of every tile row's image
  unsigned am;
  if (p.per_image) {  // the scale of the image this thread's staged row belongs to
    const int img = min((m0 + srow) / p.HW, p.B - 1);
    am = p.a_amax0[img];
    if (p.a_amax1) am = max(am, p.a_amax1[img]);
    ia_sh[lrow] = pow2f(-scale_exp(am));
  } else {
    am = *p.a_amax0;
    if (p.a_amax1) am = max(am, *p.a_amax1);
  }
  const int ka = scale_exp(am), kw = scale_exp(*p.w_amax);
  // Staging roles: a thread requests 32 contiguous bytes (8 channels) of two tile rows 64 apart; a wave covers 16 rows x 4
  // groups: four of its lanes cover a pixel's 128-byte chunk row, its request touches 16 cache lines.  (Rounds 1-3 gave a wave 64 consecutive ROWS of
  // one 8-channel group: 64 lines per request, each line fetched again by the three other waves' requests behind 64 KB of
  // weight traffic per step -- the vector memory pipe, not HBM, paid for it: see profiles/r04_tile_staging.md.)
  // Lane order inside a wave: 8 consecutive rows of ONE group per 8 lanes -- an LDS store retires 8 lanes per clock into 32
  // banks (profiles/r04_lds_conflicts.md), and the planes of the four groups start on the same bank: 8 rows of one plane
  // are 128 contiguous bytes (conflict free), one row's four groups would be 4 lanes on one 16-byte slot.  The request
  // still touches 16 lines: which lanes of the instruction share a line does not matter.  (Skewing the planes by 32 bytes
  // instead made every fragment READ conflict: SQ_LDS_BANK_CONFLICT 49 % of the LDS-active cycles.)
  const int agrp = (tid >> 3) & 3;
  const int arow0 = 16 * (tid >> 6) + (tid & 7) + 8 * ((tid >> 5) & 1);
  int arow_lds[2];
  bool a_ok2[2], a_slot[2];
  float sa2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = arow0 + 64 * i;
    // LDS row of tile row r: r itself, or (YM) 36 y + (2 | 26) + x (padded segments)
    arow_lds[i] = YM ? ((r % p.HW) >> 3) * T16Y_PITCH + (r / p.HW ? T16Y_S1 : T16Y_S0) + (r & 7) : r;
    a_ok2[i] = (r < TM) & (m0 + r < p.M);
    a_slot[i] = !YM || r < TM;  // (YM: a row past the tile has no slot of its own)
    unsigned am_i = am;
    if (p.per_image) {
      const int img = min((m0 + r) / p.HW, p.B - 1);
      am_i = p.a_amax0[img];
      if (p.a_amax1) am_i = max(am_i, p.a_amax1[img]);
    }
    sa2[i] = pow2f(scale_exp(am_i));
  }
  if constexpr (YM) {  // everything: the padding slots (and the rows of a short tile) are never written again
#pragma unroll
    for (int i = 0; i < 2 * ABUF / (256 * 16); ++i)  // (2 * ABUF = 18 * 4096)
      *reinterpret_cast<u32x4*>(lds_raw + (i * 256 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();  // (the staging below writes some of the same slots from other threads)
  } else {  // zero rows 128..143 of every chunk plane of both buffers: 2 * 2 * 4 * 16 = 256 vectors
    const int pl = tid >> 4, r = tid & 15;  // plane index (buffer, part, chunk), row
    *reinterpret_cast<u32x4*>(lds_raw + (pl >> 3) * T16_ABUF + ((pl >> 2) & 1) * T16_PP + (pl & 3) * T16_CP +
                              (128 + r) * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  unsigned amask[RB];  // per 16-row block: one bit per tap for the shifted pixel's validity
#pragma unroll
  for (int t = 0; t < RB; ++t) {
    const int r = (mbw + t) * 16 + lr;
    const int im = r / p.HW;
    const int q = r - im * p.HW;
    // (YM: block t is image row wm * RB + t, the lane's pixel is x = lr & 7 of it)
    const int y = YM ? mbw + t : q / p.W, x = YM ? (lr & 7) : q - (q / p.W) * p.W;
    unsigned mk = 0;
    for (int tp = 0; tp < p.taps; ++tp) {
      const int yy = y + tp / p.ks - p.pad, xx = x + tp % p.ks - p.pad;
      mk |= (((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W)) ? (1u << tp) : 0u;
    }
    amask[t] = mk;
  }
  // fragment base of the lane: block t adds t * 256 -- (YM) image row wm * RB + t: t * 16 T16Y_PITCH
  const int abase = YM ? lq * CP + (mbw * T16Y_PITCH + (lr >> 3 ? T16Y_S1 : T16Y_S0) + (lr & 7)) * 16
                       : lq * CP + (mbw * 16 + lr) * 16;
  const int zrow = lq * CP + 128 * 16;
  const rsrc_t w_rsrc = make_rsrc(p.w, (unsigned)(2 * p.w_ps * 2));
  const unsigned w_pstride = (unsigned)(p.w_ps * 2);
  unsigned b_off[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ntile = (n0 >> 5) + wn * NT + j;
    b_off[j] = (ntile * 32 < p.N ? (unsigned)ntile * (unsigned)p.w_nchunks * 2048u : 0u) + (unsigned)lane * 16u;
  }
  auto load_b = [&](u32x4(&rb)[4 * NT], int kc) {
    const int so = kc * 2048;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          rb[(j * 2 + part) * 2 + nb] = __builtin_bit_cast(
              u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off[j] + part * w_pstride + nb * 1024u), so, 0));
  };
  u32x4 ra[4];  // fp32 activations: [chunk i][half]
  auto issue_a = [&](int cc) {
    const int c0 = cc * SBK;
    const bool first = c0 < p.a_split;
    const int Cs = first ? p.a_split : p.Cin - p.a_split;
    const int cl = first ? c0 : c0 - p.a_split;
    const rsrc_t a_rsrc = make_rsrc(first ? (const void*)p.a0 : (const void*)p.a1, (unsigned)((long)p.P * Cs * 4));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned oa = (unsigned)((m0 + arow0 + 64 * i) * Cs + cl + agrp * 8) * 4u;
      ra[2 * i] = load16(a_rsrc, a_ok2[i] ? oa : OOB);
      ra[2 * i + 1] = load16(a_rsrc, a_ok2[i] ? oa + 16u : OOB);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 q[2];
      split8h(ra[2 * i], ra[2 * i + 1], sa2[i], q);
#pragma unroll
      for (int part = 0; part < 2; ++part)
        if (a_slot[i])
          *reinterpret_cast<u32x4*>(lds_raw + buf * ABUF + part * PP + agrp * CP + arow_lds[i] * 16) = q[part];
    }
  };

  f32x4 acc[RB][NB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kc_begin < kc_end) {
    int cc = kc_begin / p.taps;
    int tap = kc_begin - cc * p.taps;
    int ky = tap / p.ks, kx = tap - ky * p.ks;
    int cur = 0;
    // weights two steps ahead in three register sets.  (Measured alternatives, M = 1024 / 64 000 gate GEMM: A fragments
    // double-buffered in registers with two weight sets 428-440 / 473 TF, with three sets (spills) 390 / 447; skipping
    // the MFMAs of row blocks that are all padding for a tap (10 % of a 5x5 conv on 8x8 maps) through three step
    // variants: spills, 188 / 228 TF; this form 437 / 477.  Round 3: s_setprio 1 / 3 around the MFMA block: 491 / 490 TF
    // against 492 at M = 64 000, 435-446 either way at M = 1024 -- nothing.  Rotating the A fragments through the step in
    // halves (blocks 2, 3 and then the NEXT tap's blocks 0, 1 requested under the MFMAs of the other half; no registers
    // added, reads unconditional so that the compiler's lgkmcnt waits stay exact -- behind the YM branches or an EXEC
    // mask it waits for every read in flight): 479 against 490 TF at M = 64 000, 441 against 455 at M = 1024, 3x3 -4 %:
    // the other wave of the SIMD already covers a wave's LDS round trip; what is short is issue slots.)
    u32x4 b0[4 * NT], b1[4 * NT], b2[4 * NT];
    issue_a(cc);
    load_b(b0, kc_begin);
    load_b(b1, kc_begin + 1);
    store_a(0);
    __syncthreads();
    // The next chunk's activations are requested where the previous chunk's have just been converted -- here and behind the
    // conversion at a chunk's last step -- never at the START of a step: between that request and its conversion every path
    // then passes at least one step's 8 weight loads, and the compiler's wait-counter model can prove it (`s_waitcnt
    // vmcnt(11)` ... `(8)` in front of the conversion).  Rounds 1-3 requested "when a chunk's first step begins" and converted
    // "when its last step ends", two run-time conditions of the same step body: the model had to assume the conversion may
    // follow the request directly, `vmcnt(3)` ... `(0)`, and since loads return in order that also drained the 16 weight loads
    // in flight for the next two steps -- an exposed L2 round trip per chunk (9 steps of a 3x3 conv, 25 of a 5x5).
    auto request_a = [&](int c) {
      if (!(RAC_EXP_TILE & 1) && c * p.taps < kc_end) issue_a((RAC_EXP_TILE & 2) ? kc_begin / p.taps : c);
    };
    if (!(KU != 0 && RAC_TILE_A_BEHIND)) request_a(cc + 1);
    // (unrolled instances: the request sits behind the weight request of a chunk's FIRST step instead -- loads return in order,
    // and the first weight set that has to wait for the activations is then the one read three steps from the request, not two)

    // (Refilling the weight sets in place by 16-column quarters, as conv16_rows_kernel does -- all three sets in flight,
    // 2 3/4 steps of cover -- needs the MFMAs column-block-major with the requests pinned between the blocks: measured
    // 12.18 -> 12.93 ms on the 5x5 gate GEMM at M = 64 000, 4.79 -> 4.95 on the 3x3: the order costs more than the cover gains.)
    auto stepk = [&](const u32x4(&rb)[4 * NT], int tap, int ky, int kx) {
      const int drow = (ky - p.pad) * (YM ? T16Y_PITCH : p.W) + (kx - p.pad);
      const int shift = drow * 16 + cur * ABUF + abase;
      // pixels outside the image read one of the 16 zero rows: the one on the bank slot this lane's shifted row
      // would have used, so that the read group stays conflict-free  (YM: the segments' own padding)
      const int zr = zrow + cur * ABUF + ((lr + drow) & 15) * 16;
      const unsigned bit = 1u << tap;
      f16x8 fb[NB][2];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int part = 0; part < 2; ++part)
            fb[j * 2 + nb][part] = __builtin_bit_cast(f16x8, rb[(j * 2 + part) * 2 + nb]);
#pragma unroll
      for (int h = 0; h < RB / 4; ++h) {
        f16x8 fa[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int mb = 4 * h + t;
          if (!FULL && (mb >= RBL || mbw + mb >= nmb)) continue;  // wave-uniform
#if !RAC_TILE_READ_ALL  // (reading dead blocks too -- straight-line LDS traffic, exact waits -- measured +0 % on 5x5, +1 % on 3x3)
          // the tap leaves the image: no work.  (Dropping the test for the taps of the kernel's centre row, which never leave
          // -- a compile-time fact in the unrolled instances -- merges such a step's four row blocks into one basic block of 48
          // MFMAs, and the compiler's schedule of that block is SLOWER: 12.55 -> 13.2 ms on the 5x5 gate GEMM at M = 64 000.)
          if (YM && (unsigned)(mbw + mb + ky - p.pad) >= (unsigned)p.H) continue;
#endif
          const int ao = YM ? shift + mb * (16 * T16Y_PITCH) : ((amask[mb] & bit) ? shift + mb * 256 : zr);
#pragma unroll
          for (int part = 0; part < 2; ++part)
            fa[t][part] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lds_raw + ao + part * PP));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            if (!FULL && (4 * h + t >= RBL || mbw + 4 * h + t >= nmb)) continue;
            if (YM && (unsigned)(mbw + 4 * h + t + ky - p.pad) >= (unsigned)p.H) continue;
            acc[4 * h + t][nb] = mma3(fa[t], fb[nb], acc[4 * h + t][nb]);
          }
      }
    };
    if constexpr (KU == 3) {
      // (p.ks == 3, p.pad == 1, whole chunks: the launcher's condition)
      const int c_end = kc_end / 9;
      for (; cc < c_end; ++cc) {
        const bool more = cc + 1 < c_end;
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
          if (t9 == 0) load_b(b2, cc * 9 + 2);
          if (RAC_TILE_A_BEHIND && t9 == 0 && more) request_a(cc + 1);
          if (t9 % 3 == 1) load_b(b0, cc * 9 + t9 + 2);
          if (t9 % 3 == 2) load_b(b1, cc * 9 + t9 + 2);
          if (t9 % 3 == 0 && t9) load_b(b2, cc * 9 + t9 + 2);
          stepk(t9 % 3 == 0 ? b0 : (t9 % 3 == 1 ? b1 : b2), t9, t9 / 3, t9 % 3);
        }
        if (more) {
          store_a(cur ^ 1);
          __syncthreads();
          cur ^= 1;
          if (!RAC_TILE_A_BEHIND) request_a(cc + 2);
        }
      }
    } else if constexpr (KU == 5) {
      // 25 taps per chunk, ring of 3: a chunk's first step sits at ring phase (25 x chunks so far) mod 3 = chunks mod 3.  One
      // instance of the chunk per phase, three chunks per trip; inside an instance the phase is static: 8 groups of 3 steps
      // (a run-time loop whose iterations all leave the ring in the same registers) + the 25th step.
      const int c_end = kc_end / 25;
      auto chunk5 = [&](auto phase, int c) {
        constexpr int P = decltype(phase)::value;
        const bool more = c + 1 < c_end;
        const int kc0 = c * 25;
        u32x4(&s0)[4 * NT] = P == 0 ? b0 : (P == 1 ? b1 : b2);
        u32x4(&s1)[4 * NT] = P == 0 ? b1 : (P == 1 ? b2 : b0);
        u32x4(&s2)[4 * NT] = P == 0 ? b2 : (P == 1 ? b0 : b1);
        int t = 0, y = 0, x = 0;
        auto adv = [&] {
          ++t;
          x = x == 4 ? 0 : x + 1;
          y = x == 0 ? y + 1 : y;
        };
        for (int g8 = 0; g8 < 8; ++g8) {
          load_b(s2, kc0 + t + 2);
          if (RAC_TILE_A_BEHIND && g8 == 0 && more) request_a(c + 1);
          stepk(s0, t, y, x);
          adv();
          load_b(s0, kc0 + t + 2);
          stepk(s1, t, y, x);
          adv();
          load_b(s1, kc0 + t + 2);
          stepk(s2, t, y, x);
          adv();
        }
        load_b(s2, kc0 + 26);
        stepk(s0, 24, 4, 4);
        if (more) {
          store_a(cur ^ 1);
          __syncthreads();
          cur ^= 1;
          if (!RAC_TILE_A_BEHIND) request_a(c + 2);
        }
      };
      while (cc < c_end) {
        chunk5(std::integral_constant<int, 0>{}, cc);
        if (++cc >= c_end) break;
        chunk5(std::integral_constant<int, 1>{}, cc);
        if (++cc >= c_end) break;
        chunk5(std::integral_constant<int, 2>{}, cc);
        ++cc;
      }
    } else {
    auto step = [&](const u32x4(&rb)[4 * NT], int kc) {
      const bool last_tap = tap == p.taps - 1;
      const bool more = kc + 1 < kc_end;
      stepk(rb, tap, ky, kx);
      if (last_tap && more) {
        store_a(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        request_a(cc + 2);
      }
      cc = last_tap ? cc + 1 : cc;
      tap = last_tap ? 0 : tap + 1;
      kx = (kx + 1 == p.ks) ? 0 : kx + 1;
      ky = last_tap ? 0 : (kx == 0 ? ky + 1 : ky);
    };

    // Round 3, timing builds at M = 64 000 / 1024 (wrong results, what each stream costs): this loop 508 / 433 TF; the
    // weight operands loaded once 585 / 497 (+15 %); no fragment reads from LDS 617 / 513 (+21 %); 15 % of the MFMAs
    // skipped (YM) +3..5 %: no single stream bounds it -- issue slots and the latencies of both operand paths share it.
    for (int kc = kc_begin; kc < kc_end; kc += 3) {
      load_b(b2, kc + 2);
      step(b0, kc);
      if (kc + 1 < kc_end) {
        load_b(b0, kc + 3);
        step(b1, kc + 1);
      }
      if (kc + 2 < kc_end) {
        load_b(b1, kc + 4);
        step(b2, kc + 2);
      }
    }
    }
  }

  if (p.per_image) __syncthreads();  // ia_sh (a workgroup with an empty K range has not passed a barrier yet)
  if constexpr (WM == 2) {
    if (p.lstm_h) {
      conv16_lstm_epilogue(p, acc, m0, mbw, min(nmb, mbw + RBL), n0 + wn * NT * 32, pow2f(-ka), pow2f(-kw),
                           p.per_image ? ia_sh : nullptr, YM ? p.HW : 0);
      return;
    }
  }
  conv16_epilogue(p, acc, m0, mbw, min(nmb, mbw + RBL), n0 + wn * NT * 32, bz, pow2f(-ka), pow2f(-kw), nullptr,
                  p.per_image ? ia_sh : nullptr, p.out_amax, YM ? p.HW : 0);
}

// ---------------------------------------------------------------------------------------------------------
// The 64 -> 4 output head (ConvTranspose2d(64, 4, 3, 1, 1) + bias + Sigmoid, vgg_64.py:218-220) on the matrix pipe as a
// TAP-STACKED 1 x 1 conv followed by a shifted sum:
//     y[p][c] = sigmoid(bias[c] + sum_tap Z[tap][p + d_tap][c]),    Z[tap][q][c] = sum_ci w[tap][ci][c] x[q][ci].
// Z is one GEMM with M = 9 taps x 4 channels = 36 rows (3 blocks of 16), K = 64, N = the pixels of a tile's halo: the
// weights are the MFMA's A operand (48 VGPRs for both parts, for the life of the workgroup), 16 pixels the B operand,
// loaded STRAIGHT from HBM as the lane's 8 consecutive channels and split in registers -- every input element is read,
// converted and multiplied once (the conv form reads and multiplies it nine times: 36 KB of LDS fragment reads per 16
// pixels, here none).  A result block is [tap-channel][pixel]: the lane of k-group q holds the float4 Z[4 j + q][pixel]
// [0..3], stored to LDS with one 16-byte write; after a barrier every output pixel sums its nine float4s in a fixed order
// (fp32, the same for every batch).  Per 16 x 16 pixel tile (18 x 18 halo = 21 blocks of 16): 378 MFMAs, 1.3 k VALU
// instructions per SIMD for the split, 47 KB + 37 KB of LDS traffic -- all far below the tile's 83 KB from HBM / L2:
// memory-bound (1.05 GB per 4 M pixels plus the halos' L2 hits).  Measured at 4 M pixels: this form 285 us (3.7 TB/s of
// input; floor 0.17-0.19 ms); the 3 x 3 conv form on the matrix pipe (weights 144 VGPRs, halo tile split into LDS once,
// read by nine taps) with specialised producer / consumer waves 361, without 459; exact-fp32 FMAs (rac_head_fwd) 515;
// the rows kernel's 32-column tile 845.
constexpr int H16_T = 16, H16_P = H16_T + 2, H16_NPX = H16_P * H16_P, H16_NBLK = (H16_NPX + 15) / 16, H16_ZS = H16_NBLK * 16;
__global__ __launch_bounds__(256, 2) void head16_kernel(const float* __restrict__ x, const unsigned* __restrict__ a_amax,
                                                        int per_image, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                        int n_tiles) {
  __shared__ __attribute__((aligned(16))) float zs[9 * H16_ZS * 4];  // Z[tap][halo pixel][4]
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  float m = 0.f;
  for (int i = tid; i < 9 * 64 * 4; i += 256) m = fmaxf(m, fabsf(wt[i]));
#pragma unroll
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (lane == 0) red[wid] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int kw = scale_exp(__builtin_bit_cast(unsigned, m));
  const float sw = pow2f(kw);
  // A fragments: row lr of block j = (tap 4 j + lr / 4, channel lr % 4); k = 32 s + 8 lq + i
  f16x8 fw[3][2][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tap = 4 * j + (lr >> 2), co = lr & 3;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // (clamped address + select: a load behind a branch is waited for on the spot, 24 round trips in a row)
        const float t = wt[(min(tap, 8) * 64 + s * 32 + lq * 8 + i) * 4 + co];
        const float v = tap < 9 ? t * sw : 0.f;
        const _Float16 a = (_Float16)v;
        fw[j][s][0][i] = a;
        fw[j][s][1][i] = (_Float16)(v - (float)a);
      }
  }
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias);
  const int tiles_x = W >> 4, tiles_img = (H >> 4) * tiles_x;
  constexpr int NB = (H16_NBLK + 3) / 4;  // blocks per wave: wid, wid + 4, ...

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b = tile / tiles_img, ti = tile - b * tiles_img;
    const int ty0 = (ti / tiles_x) << 4, tx0 = (ti % tiles_x) << 4;
    const int ka = scale_exp(a_amax[per_image ? b : 0]);
    const float sa = pow2f(ka), inv = pow2f(-ka) * pow2f(-kw);
    const rsrc_t r = make_rsrc(x + (long)b * H * W * 64, (unsigned)((long)H * W * 64 * 4));
    // [buffer][step s, half]: the lane's 8 channels of both K steps, one block ahead (two ahead costs the third wave per
    // SIMD: 309 against 285 us; an XCD-contiguous tile order changes nothing: the halos' second readers hit the MALL)
    u32x4 v[2][4];
    auto issue = [&](u32x4(&d)[4], int blk) {
      const int hp = blk * 16 + lr;
      const int row = hp / H16_P, col = hp - row * H16_P;
      const int gy = ty0 + row - 1, gx = tx0 + col - 1;
      const bool ok = (hp < H16_NPX) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
      const unsigned off = ok ? (unsigned)(((gy * W + gx) * 64 + lq * 8) * 4) : OOB;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        d[2 * s] = load16(r, ok ? off + s * 128u : OOB);
        d[2 * s + 1] = load16(r, ok ? off + s * 128u + 16u : OOB);
      }
    };
    issue(v[0], wid);
    __syncthreads();  // the previous tile's sums have read zs
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int blk = wid + 4 * n;
      if (blk >= H16_NBLK) break;
      if (blk + 4 < H16_NBLK) issue(v[(n + 1) & 1], blk + 4);
      f16x8 fx[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 q[2];
        split8h(v[n & 1][2 * s], v[n & 1][2 * s + 1], sa, q);
        fx[s][0] = __builtin_bit_cast(f16x8, q[0]);
        fx[s][1] = __builtin_bit_cast(f16x8, q[1]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) acc = mma3(fw[j][s], fx[s], acc);
        const int tap = 4 * j + lq;  // acc[r] = Z[tap][pixel lr][channel r]
        if (tap < 9) *reinterpret_cast<f32x4*>(zs + ((tap * H16_ZS) + blk * 16 + lr) * 4) = acc * inv;
      }
    }
    __syncthreads();
    {
      const int py = tid >> 4, px = tid & 15;
      f32x4 o = b4;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        o += *reinterpret_cast<const f32x4*>(zs + (tap * H16_ZS + (py + 2 - ky) * H16_P + px + 2 - kx) * 4);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = sigmoid_acc(o[c]);
      *reinterpret_cast<f32x4*>(y + (((long)b * H + ty0 + py) * W + tx0 + px) * 4) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// First encoder layer of the frozen model on the matrix pipe, straight from the NCHW planes (vgg_64.py:8-18 on
// dynamics.py:578-582's input): out[b][y][x][co] = act(scale[co] conv3x3([img * (zmask == 0) | mask])[co] + shift[co]).
// K = 9 taps x CIN (27..72) is one to three MFMA steps; with the roles swapped -- the weights the A operand (M = 64 output
// channels, 4 blocks: 32 VGPRs per step for the life of the workgroup), 16 pixels of an image row the B operand, gathered
// from the tile's halo planes in LDS -- a result block is [channel][pixel]: a lane stores 4 consecutive channels of its
// pixel with one 16-byte store and the four lanes of a pixel write 64 contiguous bytes.
// Operand scales need no tensor maxima here: a COLUMN of B (a pixel's 9 x CIN inputs) and a ROW of A (a channel's
// weights) may each carry their own power of two (C[m][n] is scaled by both, exactly), so every pixel is scaled by the
// maximum of its own inputs and every channel by the maximum of its own weights, found with two lane exchanges: better
// conditioned than one scale per image, and an image's result cannot depend on anything but the image.
// Per 16 pixels: 12 KS MFMAs and ~150 VALU instructions against 4 KB of output: bound by the 1.05 GB write per 4 M pixels.
// (The exact-fp32 FMA form, rac_first_layer_fwd: 1728 CIN / 3 FMAs per pixel, VALU-bound, 0.53 ms there.)
template <int CIN>
__global__ __launch_bounds__(256, 2) void first16_kernel(const float* __restrict__ img, const float* __restrict__ zmask,
                                                         const float* __restrict__ mask, const float* __restrict__ w,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int act, float* __restrict__ out, unsigned* amax, int per_image,
                                                         int H, int W, int n_tiles) {
  constexpr int KK = 9 * CIN, KS = (KK + 31) / 32, TP = 18, TPW = 20, PL = TP * TPW;
  __shared__ float in_sh[CIN * PL];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // A fragments: row lr of block j = channel 16 j + lr, k = 32 s + 8 lq + i = (tap, ci) flattened as in w[co][tap][ci]
  f16x8 fw[4][KS][2];
  float osc[4][4], osh[4][4];  // the epilogue's per-channel scale (x 2^-k of the channel's weight scale) and shift
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* wr = w + (long)(16 * j + lr) * KK;
    float wv[KS][8], mx = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 32 * s + 8 * lq + i;
        const float t = wr[min(k, KK - 1)];  // (clamped address + select: no load behind a branch)
        wv[s][i] = k < KK ? t : 0.f;
        mx = fmaxf(mx, fabsf(wv[s][i]));
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const int kw = scale_exp(__builtin_bit_cast(unsigned, mx));
    const float sw = pow2f(kw);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = wv[s][i] * sw;
        const _Float16 a = (_Float16)v;
        fw[j][s][0][i] = a;
        fw[j][s][1][i] = (_Float16)(v - (float)a);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // this lane's result rows are channels 16 j + 4 lq + r: their scale sits in lane 4 lq + r
      const int kwr = __shfl(kw, 4 * lq + r);
      const int co = 16 * j + 4 * lq + r;
      osc[j][r] = (scale ? scale[co] : 1.f) * pow2f(-kwr);
      osh[j][r] = scale ? shift[co] : 0.f;
    }
  }
  // B gather: the lane's k values of step s as offsets into the halo planes, relative to its pixel
  int goff[KS][8];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = min(32 * s + 8 * lq + i, KK - 1), tap = k / CIN, ci = k - tap * CIN;
      goff[s][i] = ci * PL + (tap / 3) * TPW + tap % 3;
    }
  const float slope = act == RAC_ACT_LEAKY02 ? 0.2f : 1.f;
  const int tiles_x = W >> 4, tiles_img = (H >> 4) * tiles_x;
  const long HW = (long)H * W;

  // the halo planes of a tile: CIN x 18 x 18 values, NST per thread, requested one tile ahead (a tile's life is otherwise
  // mostly the round trips of these small loads)
  constexpr int NST = (CIN * TP * TP + 255) / 256;
  float pv[NST];
  auto fetch = [&](int tile) {
    const int b = tile / tiles_img, ti = tile - b * tiles_img;
    const int ty0 = (ti / tiles_x) << 4, tx0 = (ti % tiles_x) << 4;
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int i = tid + 256 * it;
      const int ci = i / (TP * TP), q = i - ci * TP * TP;
      const int qy = q / TP, qx = q - qy * TP;
      const int yy = ty0 + qy - 1, xx = tx0 + qx - 1;
      const bool ok = (i < CIN * TP * TP) & ((unsigned)yy < (unsigned)H) & ((unsigned)xx < (unsigned)W);
      const long pix = ok ? (long)yy * W + xx : 0;
      const float* src = ci < 3 ? img + ((long)b * 3 + (ok ? ci : 0)) * HW + pix
                                : mask + ((long)b * (CIN - 3) + (ok ? ci - 3 : 0)) * HW + pix;
      float v = ok ? *src : 0.f;
      if (zmask) {
        const float z = zmask[(long)b * HW + pix];
        if (ok & (ci < 3) & (z != 0.f)) v = v * 0.f;
      }
      pv[it] = v;
    }
  };
  if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b = tile / tiles_img, ti = tile - b * tiles_img;
    const int ty0 = (ti / tiles_x) << 4, tx0 = (ti % tiles_x) << 4;
    __syncthreads();  // the previous tile's gathers
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int i = tid + 256 * it;
      if (i >= CIN * TP * TP) break;
      const int ci = i / (TP * TP), q = i - ci * TP * TP;
      in_sh[ci * PL + (q / TP) * TPW + q % TP] = pv[it];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
    unsigned mxo = 0;
#pragma unroll 1
    for (int n = 0; n < 4; ++n) {
      const int py = 4 * wid + n;  // the block = image row py of the tile, pixel lr
      const float* base = in_sh + py * TPW + lr;
      float xv[KS][8], mx = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float t = base[goff[s][i]];
          xv[s][i] = (32 * s + 8 * lq + i < KK) ? t : 0.f;
          mx = fmaxf(mx, fabsf(xv[s][i]));
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const int ka = scale_exp(__builtin_bit_cast(unsigned, mx));
      const float sa = pow2f(ka), ia = pow2f(-ka);
      f16x8 fx[KS][2];
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = xv[s][i] * sa;
          const _Float16 a = (_Float16)v;
          fx[s][0][i] = a;
          fx[s][1][i] = (_Float16)(v - (float)a);
        }
      float* op = out + (((long)b * H + ty0 + py) * W + tx0 + lr) * 64 + 4 * lq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = mma3(fw[j][s], fx[s], acc);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[r] * ia, osc[j][r], osh[j][r]);
          v = v > 0.f ? v : slope * v;
          mxo = max(mxo, absbits(v));
          o[r] = v;
        }
        *reinterpret_cast<f32x4*>(op + 16 * j) = o;
      }
    }
    if (amax) amax_commit_block(mxo, per_image ? amax + b : amax);
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same for maps LARGER than a tile (the 16x16 / 32x32 / 64x64 vgg maps, the 16x16 ConvLSTM maps of a 128x128
// model, the 12x16 maps of 48x64 frames): a tile = R whole image rows (R | H, R * W <= 128 and a multiple of 16);
// its pixels plus `pad` image rows above and below (the halo; zeros outside the image) are staged per channel chunk
// as one CONTIGUOUS pixel range of the input.  The y shift of a tap lands in the halo, only the x shift can leave
// the row (one bit per kernel column and lane selects a zero row).  LDS image [part 2][8-channel group 4]
// [staged row + 16 zero rows][16 B], two buffers (the next chunk is stored while this one is read: one barrier
// per channel chunk); weights one chunk-tap ahead.
// NV = 16-byte staging vectors per thread and part.  Waves: WM along the rows x 4 / WM along the columns, NT
// 32-column weight tiles each: (2, 2) 128-column workgroups, waves 64 rows x 64 columns; (2, 1) 64 columns (the
// 64-channel layers); (4, 1) 32 columns (narrow heads: the 4-channel output head stores only N of them).
// ---------------------------------------------------------------------------------------------------------
// D = weight register sets: a step's B operands are requested D - 1 steps ahead (2 and 4 measure the same on every
// shape: the weights are L2 hits that arrive within one step).
// FAST: 3 x 3 taps, full 128-row tiles, whole channel chunks per K range: the 9 taps of a chunk are unrolled over a ring
// of 3 weight sets (no per-step bookkeeping or branches) and the epilogue's per-column parameters are fetched before the
// main loop instead of behind it (one exposed memory round trip less): 3-10 % on the 64- and 128-column shapes.
// What bounds the short-K shapes (64-channel layers on 64x64 maps, 18 steps per tile; SQ counters and in-kernel clocks,
// round 2): the matrix pipe is busy 27 % of the time; a wave spends 38 % of its life in s_waitcnt (the first chunk's
// HBM round trip alone is 4-6 us of a 22 us tile) and issues 3.4 other VALU instructions per MFMA (staging with its
// 2x halo, tap addressing, the epilogue), which share the SIMD's issue port with the MFMAs.  Dropping the fragment
// reads, the weight loads or the MFMAs themselves from the loop changes the kernel time by 5-10 % each.
#ifndef RAC_ROWS_PIN
#define RAC_ROWS_PIN 1
#endif
// KSF: kernel size of the FAST form (3, or 5: the 5x5 ConvLSTM gate convs on the 16x16 latents of a 128x128 model --
// round 5; whole-row tiles only, 128-column workgroups only).
template <int NV, int WM, int NT, int D, bool FAST = false, int KSF = 3>
__global__ __launch_bounds__(256, 2) void conv16_rows_kernel(Conv16P p) {
  constexpr int PF = KSF / 2, TAPSF = KSF * KSF;  // (FAST form: pad, taps per channel chunk)
  constexpr int WN = 4 / WM;
  constexpr int MB = 8 / WM;            // 16-row blocks per wave
  constexpr int NB = 2 * NT;            // 16-column blocks per wave
  constexpr int BNW = WN * NT * 32;     // columns per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_group == 1) {
    const int mt = gridDim.x, nt = gridDim.y;
    const int lin = bx + mt * (by + nt * bz);
    const int xcd = lin & 7, s = lin >> 3;
    const int grp = (s / mt) * 8 + xcd;
    bx = s % mt;
    by = grp % nt;
    bz = grp / nt;
  }
  if (p.xcd_group == 6) {
    // one column tile, K unsplit: consecutive tiles share halo rows; an XCD takes runs of 8 consecutive tiles (ids 64 q + 8 j
    // + x -> tile 64 q + 8 x + j), so a halo line's second reader finds it in its own L2
    const int mt = gridDim.x;
    if (bx < (mt & ~63)) bx = (bx & ~63) | ((bx & 7) << 3) | ((bx >> 3) & 7);
  }
  if (p.xcd_group == 2) {
    // two column tiles, K unsplit: the two workgroups of an M-tile read the same activations.  Hardware order sends
    // consecutive workgroup ids to consecutive XCDs (8 L2s) and, with the column tile as the slow grid index, runs the two half
    // a launch apart: every activation line comes from HBM twice.  Here ids 16 q + 8 n + x become (M-tile 8 q + x, column tile
    // n): the pair runs on ONE XCD, 8 ids apart -- one HBM fetch per line, the second reader finds it in (or on its way to) L2.
    // (... and, as in mode 6, an XCD takes RUNS of 8 consecutive M-tiles -- ids 128 q + 16 j + 8 n + x -> M-tile
    // 64 q + 8 x + j -- so that the halo rows two neighbouring tiles share come from one L2 as well)
    const int mt = gridDim.x, lin = bx + mt * by;
    const int big = (2 * mt) & ~127;
    if (lin < big) {
      const int s = lin >> 3;
      bx = ((s >> 4) << 6) | ((lin & 7) << 3) | ((s >> 1) & 7);
      by = s & 1;
    } else if (lin < ((2 * mt) & ~15)) {
      const int l2 = lin - big;
      bx = (big >> 1) + (((l2 >> 4) << 3) | (l2 & 7));
      by = (l2 >> 3) & 1;
    } else {  // (the last < 16 ids: the remaining M-tiles in the natural order)
      const int r = lin - ((2 * mt) & ~15), base = ((2 * mt) & ~15) >> 1;
      bx = base + (r >> 1);
      by = r & 1;
    }
  }
  const int wn = wid % WN, wm = wid / WN;
  const int TM = p.tile_m;
  const int nmb = TM >> 4;
  const bool seg = FAST && KSF == 3 && p.seg_w;  // 2-D tile (unrolled 3x3 form only)
  SegOrigin so{};
  if (seg) so = seg_origin(p, bx);
  const int m0 = seg ? so.m0 : bx * TM, n0 = by * BNW;
  const int kc_begin = bz * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);
  const int img = p.per_image ? m0 / p.HW : 0;  // a tile is R rows of ONE image: one scale, one output slot
  unsigned am = p.a_amax0[img];
  if (p.a_amax1) am = max(am, p.a_amax1[img]);
  const int ka = scale_exp(am), kw = scale_exp(*p.w_amax);
  const float sa = pow2f(ka);
  const int halo = p.pad * p.W;
  // staged pixel rows: the tile's image rows + the halo rows (a multiple of 16); 2-D tile: the (seg_h + 2) x (seg_w + 2) halo tile
  const int nrows = seg ? (p.seg_h + 2) * (p.seg_w + 2) : TM + 2 * halo;
  // FAST: every staged image row sits in W + 2 PF LDS rows, PF zero rows either side, so a horizontal tap that leaves
  // the image reads zeros by address: no per-lane validity mask / compare / select per (tap, block) -- as the tile
  // kernel's padded segments.  (2-D tile: the halo columns are staged like any other pixel, zeros outside the image.)
  // Otherwise: the staged rows + 16 zero rows that the select points at.
  const int WD = seg ? p.seg_w : p.W;  // pixels of a tile row
  const int WP = WD + 2 * PF;
  // one 8-channel group; (FAST) rounded to 256 B: the four k-groups of a read must start on the same 16-byte slot
  const int cplane = FAST ? (((seg ? nrows : (nrows / p.W) * WP) + 15) & ~15) * 16 : (nrows + 16) * 16;
  const int pplane = 4 * cplane;
  const int abuf = 2 * pplane;           // one buffer
  if constexpr (FAST) {
#if !RAC_EXP_ROWS_NOZERO  // (timing / counter builds: which LDS stream of this kernel conflicts?  tools/build_variant.sh)
    for (int o = tid * 16; o < 2 * abuf; o += 4096) *reinterpret_cast<u32x4*>(lds_raw + o) = u32x4{0u, 0u, 0u, 0u};
#endif
    __syncthreads();  // (the staging below writes some of the same rows from other threads)
  } else {  // zero rows of both buffers: 2 buffers x 2 parts x 4 groups x 16 rows = 256 vectors
    const int pl = tid >> 4, r = tid & 15;
    *reinterpret_cast<u32x4*>(lds_raw + (pl >> 3) * abuf + ((pl >> 2) & 1) * pplane + (pl & 3) * cplane +
                              (nrows + r) * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  const int y_tile = (m0 % p.HW) / p.W;
  int s_off[NV], s_row[NV], s_grp[NV], s_pix[NV];
  bool s_ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + 256 * i;
    const int g = v / nrows, row = v - g * nrows;
    s_grp[i] = g;
    s_row[i] = row;
    if (seg) {  // staged slot `row` = halo-tile position (row / WP, row % WP) = image pixel (y0 - 1 + .., x0 - 1 + ..)
      s_off[i] = g < 4 ? g * cplane + row * 16 : -1;
      const int y = so.y0 - 1 + row / WP, x = so.x0 - 1 + row % WP;
      s_ok[i] = (g < 4) & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
      s_row[i] = (so.img * p.H + y) * p.W + x;  // (2-D tile: s_row holds the source PIXEL; s_pix its half-resolution twin)
      s_pix[i] = (so.img * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1);
      continue;
    }
    s_off[i] = g < 4 ? g * cplane + (FAST ? (row / p.W) * WP + PF + row % p.W : row) * 16 : -1;
    const int y = y_tile - p.pad + row / p.W;
    s_ok[i] = (g < 4) & ((unsigned)y < (unsigned)p.H) & (m0 - halo + row < p.M);
    s_pix[i] = m0 - halo + row;
    if (p.a0_up && s_ok[i]) {  // the pixel of the half-resolution source under this one
      const int img = s_pix[i] / p.HW, x = row % p.W;
      s_pix[i] = (img * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1);
    }
  }
  unsigned amask[MB];
#pragma unroll
  for (int t = 0; t < MB; ++t) {
    const int r = (wm * MB + t) * 16 + lr;
    const int x = r % p.W;
    unsigned mk = 0;
    for (int kx = 0; kx < p.ks; ++kx) mk |= ((unsigned)(x + kx - p.pad) < (unsigned)p.W) ? (1u << kx) : 0u;
    amask[t] = mk;
  }
  const int abase = lq * cplane + (halo + wm * MB * 16 + lr) * 16;  // block t adds t * 256
  const int zrow = lq * cplane + nrows * 16;
  int ablk[MB];  // FAST: block t's fragment base in the padded rows (a 16-pixel block lies inside one image row)
#pragma unroll
  for (int t = 0; t < MB; ++t) {
    const int r0 = (wm * MB + t) * 16;
    ablk[t] = lq * cplane + ((PF + r0 / WD) * WP + PF + r0 % WD + lr) * 16;
  }
  const rsrc_t w_rsrc = make_rsrc(p.w, (unsigned)(2 * p.w_ps * 2));
  const unsigned w_pstride = (unsigned)(p.w_ps * 2);
  unsigned b_off[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ntile = (n0 >> 5) + wn * NT + j;
    b_off[j] = (ntile * 32 < p.N ? (unsigned)ntile * (unsigned)p.w_nchunks * 2048u : 0u) + (unsigned)lane * 16u;
  }
  auto load_b = [&](u32x4(&rb)[4 * NT], int kc) {
    const int wo = kc * 2048;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          rb[(j * 2 + part) * 2 + nb] = __builtin_bit_cast(
              u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off[j] + part * w_pstride + nb * 1024u), wo, 0));
  };
  u32x4 ra[2 * NV];
  auto issue_a = [&](int cc) {
    const int c0 = cc * SBK;
    const bool first = c0 < p.a_split;
    const int Cs = first ? p.a_split : p.Cin - p.a_split;
    const int cl = first ? c0 : c0 - p.a_split;
    const bool up = first && p.a0_up;
    const rsrc_t a_rsrc = make_rsrc(first ? (const void*)p.a0 : (const void*)p.a1,
                                    (unsigned)((long)(up ? p.P >> 2 : p.P) * Cs * 4));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int px = up ? s_pix[i] : (seg ? s_row[i] : m0 - halo + s_row[i]);
      const unsigned oa = (unsigned)(px * Cs + cl + s_grp[i] * 8) * 4u;
      ra[2 * i] = load16(a_rsrc, s_ok[i] ? oa : OOB);
      ra[2 * i + 1] = load16(a_rsrc, s_ok[i] ? oa + 16u : OOB);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (s_off[i] < 0) continue;
      u32x4 q[2];
      split8h(ra[2 * i], ra[2 * i + 1], sa, q);
#if !RAC_EXP_ROWS_NOSTORE
#pragma unroll
      for (int part = 0; part < 2; ++part)
        *reinterpret_cast<u32x4*>(lds_raw + buf * abuf + part * pplane + s_off[i]) = q[part];
#else
      if (q[0].x == 0x12345678u && q[1].y == 0x9abcdef0u) *reinterpret_cast<u32x4*>(lds_raw) = q[0];
#endif
    }
  };

  f32x4 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float pre[NB][3];
  if constexpr (FAST) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = n0 + wn * NT * 32 + nb * 16 + lr;
      const int nc = n < p.n_store ? n : 0;
      pre[nb][0] = p.bias ? p.bias[nc] : 0.f;
      pre[nb][1] = p.scale ? p.scale[nc] : 1.f;
      pre[nb][2] = p.scale ? p.shift[nc] : 0.f;
    }
    static_assert(D == 3, "the taps of a chunk walk a ring of 2 or 3 weight register sets");
    const int c_begin = kc_begin / TAPSF, c_end = kc_end / TAPSF;
    if constexpr (KSF == 5) {
      // 5 x 5: 25 taps per chunk.  Unrolled like the 3 x 3 form the chunk body is 2 x 25 taps x 48 MFMAs = 160 KB of code (the
      // instruction cache holds 64 KB); here the taps run in a loop whose body is ONE PAIR of taps (weight sets 0 and 1, each
      // refilled in place by quarters for the tap two ahead), the tap's offset in the padded rows is a scalar, and the chunk
      // switch (conversion + stores of the next chunk, one barrier) hangs off the pair whose tap is a chunk's last.  What the
      // padded rows buy is the same as in the 3 x 3 form: no per-lane mask, compare and select per (tap, row block).
      static_assert(NT == 2, "the 5x5 form exists for 128-column workgroups");
      if (c_begin < c_end) {
        u32x4 bs[2][4 * NT];
        issue_a(c_begin);
        load_b(bs[0], kc_begin);
        load_b(bs[1], min(kc_begin + 1, kc_end - 1));
        store_a(0);
        __syncthreads();
        if (c_begin + 1 < c_end) issue_a(c_begin + 1);  // (right behind a conversion: see conv16_tile_kernel)
        int cc = c_begin, ky = 0, kx = 0, cur = 0;
        auto load_b_quarter = [&](u32x4(&rb)[4 * NT], int nb, int kc) {
          const int j = nb >> 1, h = nb & 1;
#pragma unroll
          for (int part = 0; part < 2; ++part)
            rb[(j * 2 + part) * 2 + h] = __builtin_bit_cast(
                u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off[j] + part * w_pstride + h * 1024u), kc * 2048, 0));
        };
        auto one_tap = [&](u32x4(&rb)[4 * NT], int kc) {
          const int shift = ((ky - PF) * WP + (kx - PF)) * 16 + cur * abuf;  // wave-uniform: the tap in the padded rows
          f16x8 fb[NB][2];
#pragma unroll
          for (int j2 = 0; j2 < NT; ++j2)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
              for (int part = 0; part < 2; ++part) fb[j2 * 2 + nb][part] = __builtin_bit_cast(f16x8, rb[(j2 * 2 + part) * 2 + nb]);
          f16x8 fa[MB][2];
#pragma unroll
          for (int t = 0; t < MB; ++t) {
            const int ao = ablk[t] + shift;
#pragma unroll
            for (int part = 0; part < 2; ++part)
              fa[t][part] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lds_raw + ao + part * pplane));
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int t = 0; t < MB; ++t) acc[t][nb] = mma3(fa[t], fb[nb], acc[t][nb]);
            __builtin_amdgcn_sched_barrier(0);
            load_b_quarter(rb, nb, min(kc + 2, kc_end - 1));
            __builtin_amdgcn_sched_barrier(0);
          }
          const bool last_tap = (ky == KSF - 1) & (kx == KSF - 1);
          if (last_tap) {
            if (cc + 1 < c_end) {
              store_a(cur ^ 1);
              __syncthreads();
              cur ^= 1;
              if (cc + 2 < c_end) issue_a(cc + 2);
            }
            ++cc, ky = 0, kx = 0;
          } else {
            kx = kx + 1 == KSF ? 0 : kx + 1;
            ky += kx == 0;
          }
        };
        for (int kc = kc_begin; kc < kc_end; kc += 2) {
          one_tap(bs[0], kc);
          if (kc + 1 < kc_end) one_tap(bs[1], kc + 1);
        }
      }
    } else
    if (c_begin < c_end) {
      // weight register sets: 128-column workgroups (NT = 2: 32 VGPRs per set) keep TWO and request one tap ahead -- with
      // three the kernel needs more than 256 VGPRs, and the scheduler, to stay inside, sank every request to just before the
      // MFMAs that read it: each tap then waited out an L2 round trip (seen in the ISA: s_waitcnt vmcnt(7) straight after
      // eight loads).  Nine taps are odd, so with two sets the set of a tap alternates between chunks: the chunk body exists
      // for both parities.  The narrower forms keep three sets, two taps ahead.
      // RAC_ROWS_REFILL: no set is kept free for the requests -- all R sets are in flight, and a set is refilled IN PLACE, one
      // 16-column quarter (two 16-byte loads) at a time, as soon as the MFMAs that read that quarter have been issued
      // (the MFMAs of a tap run column-block-major for this): the quarter for tap t + R is requested 1/4 .. 4/4 of the way
      // through tap t and first read the same way through tap t + R -- R - 1/4 taps of cover for every quarter instead
      // of R - 1, from the same registers.  +2-3 % on the 16x16 256-channel layers, ~1 % on the 32x32 ones; a SIX-set ring on
      // the 64-column form (16 VGPRs per set) measured exactly the three-set time: L2 latency is covered, what the weight
      // stream costs these kernels (a timing build without it: +12 % here, +33 % before the refill) is not waiting time.
      constexpr int R = (NT == 2 && !RAC_ROWS_RING3) ? 2 : 3, AHEAD = RAC_ROWS_REFILL ? R : R - 1;
      u32x4 bs[R][4 * NT];
      issue_a(c_begin);
#pragma unroll
      for (int j = 0; j < AHEAD; ++j) load_b(bs[j], min(kc_begin + j, kc_end - 1));
      store_a(0);
      __syncthreads();
      int cur = 0;
      auto load_b_quarter = [&](u32x4(&rb)[4 * NT], int nb, int kc) {
        const int j = nb >> 1, h = nb & 1;
#pragma unroll
        for (int part = 0; part < 2; ++part)
          rb[(j * 2 + part) * 2 + h] = __builtin_bit_cast(
              u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off[j] + part * w_pstride + h * 1024u), kc * 2048, 0));
      };
      auto chunk = [&](auto par, int cc) {
        constexpr int P = decltype(par)::value;
        const bool more = cc + 1 < c_end;
        if (more) issue_a(cc + 1);
        const int kc0 = cc * TAPSF;
        const int bufo = cur * abuf;
#pragma unroll
        for (int tap = 0; tap < TAPSF; ++tap) {
          const int ky = tap / KSF, kx = tap % KSF;
#if !(RAC_EXP_ROWS_NOB) && !RAC_ROWS_REFILL  // (timing build: the weight fragments loaded once per workgroup)
          load_b(bs[(tap + AHEAD + P) % R], min(kc0 + tap + AHEAD, kc_end - 1));
#endif
#if RAC_ROWS_PIN
          // pin the requests here: in this one large basic block the scheduler otherwise moves them towards their use.
          // (Also pinning a tap's fragment reads before its MFMAs, or rotating them through the tap in halves as
          // conv16_rows_persist_kernel does: +4 % on the 64-column form, nothing on the 128-column one -- not done.)
          __builtin_amdgcn_sched_barrier(0);
#endif
          const int shift = ((ky - PF) * WP + (kx - PF)) * 16 + bufo;  // wave-uniform: the tap in the padded rows
          f16x8 fb[NB][2];
#pragma unroll
          for (int j2 = 0; j2 < NT; ++j2)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
              for (int part = 0; part < 2; ++part)
                fb[j2 * 2 + nb][part] = __builtin_bit_cast(f16x8, bs[(tap + P) % R][(j2 * 2 + part) * 2 + nb]);
          f16x8 fa[MB][2];
#pragma unroll
          for (int t = 0; t < MB; ++t) {
            const int ao = ablk[t] + shift;
#pragma unroll
            for (int part = 0; part < 2; ++part)
              fa[t][part] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lds_raw + ao + part * pplane));
          }
#if RAC_ROWS_REFILL
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int t = 0; t < MB; ++t) acc[t][nb] = mma3(fa[t], fb[nb], acc[t][nb]);
#if RAC_ROWS_REFILL_PIN & 1
            __builtin_amdgcn_sched_barrier(0);
#endif
#if !(RAC_EXP_ROWS_NOB)
            load_b_quarter(bs[(tap + P) % R], nb, min(kc0 + tap + R, kc_end - 1));
#endif
#if RAC_ROWS_REFILL_PIN & 2
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
#else
#pragma unroll
          for (int t = 0; t < MB; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mma3(fa[t], fb[nb], acc[t][nb]);
#endif
        }
        if (more) {
          store_a(cur ^ 1);
          __syncthreads();
          cur ^= 1;
        }
      };
      if constexpr (R == 2) {
        static_assert(TAPSF % 2 == 1, "an odd number of taps: the set of a chunk's first tap alternates");
        for (int cc = c_begin; cc < c_end; cc += 2) {
          chunk(std::integral_constant<int, 0>{}, cc);
          if (cc + 1 < c_end) chunk(std::integral_constant<int, 1>{}, cc + 1);
        }
      } else {
        static_assert(TAPSF % R == 0, "ring phase");
        for (int cc = c_begin; cc < c_end; ++cc) chunk(std::integral_constant<int, 0>{}, cc);
      }
    }
  } else
  if (kc_begin < kc_end) {
    int cc = kc_begin / p.taps;
    int tap = kc_begin - cc * p.taps;
    int ky = tap / p.ks, kx = tap - ky * p.ks;
    int cur = 0;
    u32x4 bs[D][4 * NT];
    issue_a(cc);
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load_b(bs[j], kc_begin + j);
    store_a(0);
    __syncthreads();
    // (the next chunk's activations are requested right behind a conversion, never at the start of a step: see
    // conv16_tile_kernel -- the wait in front of the conversion then leaves the weight ring in flight)
    if ((cc + 1) * p.taps < kc_end) issue_a(cc + 1);

    auto step = [&](const u32x4(&rb)[4 * NT], int kc) {
      const bool last_tap = tap == p.taps - 1;
      const bool more = kc + 1 < kc_end;
      const int drow = (ky - p.pad) * p.W + (kx - p.pad);
      const int shift = drow * 16 + cur * abuf + abase;
      const int zr = zrow + cur * abuf + ((lr + halo + drow) & 15) * 16;  // the zero row on this lane's own bank slot
      const unsigned bit = 1u << kx;
      f16x8 fb[NB][2];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int part = 0; part < 2; ++part)
            fb[j * 2 + nb][part] = __builtin_bit_cast(f16x8, rb[(j * 2 + part) * 2 + nb]);
      constexpr int HB = MB < 4 ? MB : 4;  // fragment reads in groups of up to 4 blocks
#pragma unroll
      for (int h = 0; h < MB / HB; ++h) {
        f16x8 fa[HB][2];
#pragma unroll
        for (int t = 0; t < HB; ++t) {
          const int mb = HB * h + t;
          if (wm * MB + mb >= nmb) continue;  // wave-uniform: past the tile's rows
          const int ao = (amask[mb] & bit) ? shift + mb * 256 : zr;
#pragma unroll
          for (int part = 0; part < 2; ++part)
            fa[t][part] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lds_raw + ao + part * pplane));
        }
#pragma unroll
        for (int t = 0; t < HB; ++t) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            if (wm * MB + HB * h + t >= nmb) continue;
            acc[HB * h + t][nb] = mma3(fa[t], fb[nb], acc[HB * h + t][nb]);
          }
#if RAC_ROWS_GENERIC_SEG
          __builtin_amdgcn_sched_barrier(0);  // one row block's MFMAs at a time (see the tile kernel's note on block branches)
#endif
        }
      }
      if (last_tap && more) {
        store_a(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        if ((cc + 2) * p.taps < kc_end) issue_a(cc + 2);
      }
      cc = last_tap ? cc + 1 : cc;
      tap = last_tap ? 0 : tap + 1;
      kx = (kx + 1 == p.ks) ? 0 : kx + 1;
      ky = last_tap ? 0 : (kx == 0 ? ky + 1 : ky);
    };

    for (int kc = kc_begin; kc < kc_end; kc += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        if (kc + j >= kc_end) break;
        load_b(bs[(j + D - 1) % D], min(kc + j + D - 1, kc_end - 1));  // (past the end: a harmless repeat)
        step(bs[j], kc + j);
      }
    }
  }
  conv16_epilogue(p, acc, m0, wm * MB, nmb, n0 + wn * NT * 32, bz, pow2f(-ka), pow2f(-kw), FAST ? pre : nullptr, nullptr,
                  p.out_amax ? p.out_amax + img : nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// PERSISTENT form of the unrolled 3x3 rows kernel for the narrow layers (64 / 32 output columns per workgroup: the
// 64-channel 64x64 layers and the 4-channel output head).  Their K is 18-36 steps per tile, so a tile's life is mostly
// fixed costs: the first chunk's HBM round trip with nothing to overlap it, the epilogue's stores, a fresh workgroup.
// Here a workgroup walks tiles bx, bx + gridDim.x, ... : while the LAST chunk of a tile is multiplied, the first chunk of
// the NEXT tile is already requested (and the weight ring wraps to the next tile's first two taps: same weights), so the
// epilogue's stores run under those loads and the next tile starts from registers.  Same arithmetic, same order of sums:
// bit-equal to conv16_rows_kernel<NV, WM, NT, 3, true>.
// ---------------------------------------------------------------------------------------------------------
template <int NV, int WM, int NT>
__global__ __launch_bounds__(256, 2) void conv16_rows_persist_kernel(Conv16P p) {
  constexpr int WN = 4 / WM;
  constexpr int MB = 8 / WM;
  constexpr int NB = 2 * NT;
  constexpr int BNW = WN * NT * 32;
  constexpr int TM = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15;
  const int by = blockIdx.y;
  const int wn = wid % WN, wm = wid / WN;
  const int nmb = TM >> 4;
  const int n0 = by * BNW;
  const int kc_begin = 0, kc_end = p.nchunks;  // (never K-split: the short-K layers)
  const int kw = scale_exp(*p.w_amax);
  const int halo = p.W;
  const bool seg = p.seg_w != 0;  // 2-D tiles (see Conv16P): seg_h rows x seg_w pixels + a one-pixel halo all round
  const int nrows = seg ? (p.seg_h + 2) * (p.seg_w + 2) : TM + 2 * halo;
  const int WD = seg ? p.seg_w : p.W;
  const int WP = WD + 2;  // padded rows, as conv16_rows_kernel's unrolled form: a zero LDS row either side of an image row
  const int cplane = (((seg ? nrows : (nrows / p.W) * WP) + 15) & ~15) * 16;  // (a multiple of 256 B, as there)
  const int pplane = 4 * cplane;
  const int abuf = 2 * pplane;
  for (int o = tid * 16; o < 2 * abuf; o += 4096) *reinterpret_cast<u32x4*>(lds_raw + o) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  // tile-independent staging roles
  int s_off[NV], s_row[NV], s_grp[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + 256 * i;
    const int g = v / nrows, row = v - g * nrows;
    s_grp[i] = g;
    s_row[i] = row;
    s_off[i] = g < 4 ? g * cplane + (seg ? row : (row / p.W) * WP + 1 + row % p.W) * 16 : -1;
  }
  const int lq = lane >> 4;
  int ablk[MB];
#pragma unroll
  for (int t = 0; t < MB; ++t) {
    const int r0 = (wm * MB + t) * 16;
    ablk[t] = lq * cplane + ((1 + r0 / WD) * WP + 1 + r0 % WD + lr) * 16;
  }
  const rsrc_t w_rsrc = make_rsrc(p.w, (unsigned)(2 * p.w_ps * 2));
  const unsigned w_pstride = (unsigned)(p.w_ps * 2);
  unsigned b_off[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ntile = (n0 >> 5) + wn * NT + j;
    b_off[j] = (ntile * 32 < p.N ? (unsigned)ntile * (unsigned)p.w_nchunks * 2048u : 0u) + (unsigned)lane * 16u;
  }
  auto load_b = [&](u32x4(&rb)[4 * NT], int kc) {
    const int wo = kc * 2048;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          rb[(j * 2 + part) * 2 + nb] = __builtin_bit_cast(
              u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off[j] + part * w_pstride + nb * 1024u), wo, 0));
  };
  // per-tile state: the tile being multiplied (cur) and the one whose first chunk is in flight (nxt)
  struct TileState {
    int m0, ka;
    bool ok[NV];
    int pix[NV];   // source pixel of the first source (half resolution under a0_up)
    int pix1[NV];  // 2-D tiles: the full-resolution pixel (the second source's, and the first's without a0_up)
  };
  auto setup = [&](int bx, TileState& t) {
    SegOrigin so{};
    if (seg) so = seg_origin(p, bx);
    t.m0 = seg ? so.m0 : bx * TM;
    const int img = p.per_image ? t.m0 / p.HW : 0;
    unsigned am = p.a_amax0[img];
    if (p.a_amax1) am = max(am, p.a_amax1[img]);
    t.ka = scale_exp(am);
    const int y_tile = (t.m0 % p.HW) / p.W;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int row = s_row[i];
      if (seg) {
        const int y = so.y0 - 1 + row / WP, x = so.x0 - 1 + row % WP;
        t.ok[i] = (s_grp[i] < 4) & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)p.W);
        t.pix1[i] = (so.img * p.H + y) * p.W + x;
        t.pix[i] = p.a0_up ? (so.img * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1) : t.pix1[i];
        continue;
      }
      const int y = y_tile - 1 + row / p.W;
      t.ok[i] = (s_grp[i] < 4) & ((unsigned)y < (unsigned)p.H) & (t.m0 - halo + row < p.M);
      t.pix[i] = t.m0 - halo + row;
      t.pix1[i] = t.pix[i];
      if (p.a0_up && t.ok[i]) {
        const int im = t.pix[i] / p.HW, x = row % p.W;
        t.pix[i] = (im * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1);
      }
    }
  };
  u32x4 ra[2 * NV];
  auto issue_a = [&](int cc, const TileState& t) {
    const int c0 = cc * SBK;
    const bool first = c0 < p.a_split;
    const int Cs = first ? p.a_split : p.Cin - p.a_split;
    const int cl = first ? c0 : c0 - p.a_split;
    const bool up = first && p.a0_up;
    const rsrc_t a_rsrc = make_rsrc(first ? (const void*)p.a0 : (const void*)p.a1,
                                    (unsigned)((long)(up ? p.P >> 2 : p.P) * Cs * 4));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const unsigned oa = (unsigned)((up ? t.pix[i] : t.pix1[i]) * Cs + cl + s_grp[i] * 8) * 4u;
      ra[2 * i] = load16(a_rsrc, t.ok[i] ? oa : OOB);
      ra[2 * i + 1] = load16(a_rsrc, t.ok[i] ? oa + 16u : OOB);
    }
  };
  auto store_a = [&](int buf, float sa) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (s_off[i] < 0) continue;
      u32x4 q[2];
      split8h(ra[2 * i], ra[2 * i + 1], sa, q);
#pragma unroll
      for (int part = 0; part < 2; ++part)
        *reinterpret_cast<u32x4*>(lds_raw + buf * abuf + part * pplane + s_off[i]) = q[part];
    }
  };
  float pre[NB][3];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + wn * NT * 32 + nb * 16 + lr;
    const int nc = n < p.n_store ? n : 0;
    pre[nb][0] = p.bias ? p.bias[nc] : 0.f;
    pre[nb][1] = p.scale ? p.scale[nc] : 1.f;
    pre[nb][2] = p.scale ? p.shift[nc] : 0.f;
  }
  const int n_tiles = (p.M + TM - 1) / TM;
  const int c_end = kc_end / 9;
  TileState cur_t, nxt_t;
  int bx = blockIdx.x;
  if (p.xcd_group == 5) {
    // neighbouring tiles share their halo pixels (180 staged per 128): workgroup ids go to the XCDs round robin, so with tile =
    // id the two readers of a halo pixel sit on different L2s.  Here an XCD's gridDim.x / 8 workgroups take CONSECUTIVE
    // tiles (whole images' worth): a halo line is fetched from HBM once, its second reader finds it in L2.
    const int per = gridDim.x >> 3;
    bx = (bx & 7) * per + (bx >> 3);
  }
  if (bx >= n_tiles) return;
  setup(bx, cur_t);
  u32x4 bs[3][4 * NT];
  issue_a(0, cur_t);
  load_b(bs[0], kc_begin);
  load_b(bs[1], min(kc_begin + 1, kc_end - 1));
  if (RAC_EXP_PERSIST & 16) load_b(bs[2], min(kc_begin + 2, kc_end - 1));
  int cur = 0;
  // A fragments rotate through the tap in halves: the second half's reads are issued before the first half's MFMAs, the NEXT
  // tap's first half before the second half's MFMAs, pinned with scheduling barriers.  (Left alone, the scheduler issued
  // every read right in front of the MFMAs that need it -- s_waitcnt lgkmcnt(0) after each: an LDS round trip per fragment
  // under 3-6 MFMAs.  -7 % here; in conv16_rows_kernel the same rotation measured +4 % on the 64-column form.)
  constexpr int HB = MB / 2;
  f16x8 fa[MB][2];
  auto read_frag = [&](int t, int tap, int bufo) {
    const int ky = tap / 3, kx = tap % 3;
    const int ao = ablk[t] + ((ky - 1) * WP + (kx - 1)) * 16 + bufo;
#pragma unroll
    for (int part = 0; part < 2; ++part)
      fa[t][part] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lds_raw + ao + part * pplane));
  };
  for (;;) {
    const int next_bx = bx + gridDim.x;
    const bool has_next = next_bx < n_tiles;
    store_a(cur, pow2f(cur_t.ka));  // the first chunk of this tile: requested while the previous tile was finishing
    __syncthreads();
#pragma unroll
    for (int t = 0; t < HB; ++t) read_frag(t, 0, cur * abuf);
    if (has_next) {
      if (RAC_EXP_PERSIST & 32)
        nxt_t = cur_t;
      else
        setup(next_bx, nxt_t);
      if (RAC_EXP_PERSIST & 64) {  // the arithmetic is done, the addresses are this tile's again (loads hit L2)
        const int keep = nxt_t.m0 ^ nxt_t.pix[0];
        nxt_t = cur_t;
        if (keep == 0x7fffffff) nxt_t.ka += 1;  // (keeps setup() alive)
      }
    }
    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cc = 0; cc < c_end; ++cc) {
      const bool more = cc + 1 < c_end;
      const int kc0 = cc * 9;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // the ring wraps to taps 0 and 1 of chunk 0 behind the last chunk: the next tile multiplies the same weights
        int knext = kc0 + tap + 2;
        knext = knext >= kc_end ? knext - kc_end : knext;
        if (!(RAC_EXP_PERSIST & 16)) load_b(bs[(tap + 2) % 3], knext);
        if (tap == RAC_A_REQUEST_TAP) {
          // the next chunk's activations are requested BEHIND this tap's weight request: loads return in order, so the first
          // weight set that has to wait for them is the one requested at the next tap and read three taps from here (requested
          // in front of this tap's, it was the one read two taps from here)
          if (more)
            issue_a(cc + 1, cur_t);
          else if (has_next)
            issue_a(0, nxt_t);
        }
        __builtin_amdgcn_sched_barrier(0);
        f16x8 fb[NB][2];
#pragma unroll
        for (int j2 = 0; j2 < NT; ++j2)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int part = 0; part < 2; ++part)
              fb[j2 * 2 + nb][part] = __builtin_bit_cast(f16x8, bs[tap % 3][(j2 * 2 + part) * 2 + nb]);
        if (!(RAC_EXP_PERSIST & 8) || tap == 0) {
#pragma unroll
          for (int t = HB; t < MB; ++t) read_frag(t, tap, cur * abuf);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < HB; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mma3(fa[t], fb[nb], acc[t][nb]);
        __builtin_amdgcn_sched_barrier(0);
        if (tap == 8 && more) {
          store_a(cur ^ 1, pow2f(cur_t.ka));
          __syncthreads();
          cur ^= 1;
        }
        if ((tap < 8 || more) && (!(RAC_EXP_PERSIST & 8) || tap == 8)) {
#pragma unroll
          for (int t = 0; t < HB; ++t) read_frag(t, (tap + 1) % 9, cur * abuf);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = HB; t < MB; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mma3(fa[t], fb[nb], acc[t][nb]);
      }
    }
    const int img = p.per_image ? cur_t.m0 / p.HW : 0;
    if (RAC_EXP_PERSIST & 2) {
      unsigned mx = 0;
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = max(mx, absbits(acc[i][j][r]));
      if (p.out_amax) amax_commit(mx, p.out_amax + img);
    } else {
      conv16_epilogue(p, acc, cur_t.m0, wm * MB, nmb, n0 + wn * NT * 32, 0, pow2f(-cur_t.ka), pow2f(-kw), pre, nullptr,
                      p.out_amax ? p.out_amax + img : nullptr);
    }
    if (!has_next) break;
    // the next tile's first chunk goes to the buffer nobody reads any more (the last chunk sits in `cur`)
    cur ^= 1;
    cur_t = nxt_t;
    bx = next_bx;
  }
}

// ---------------------------------------------------------------------------------------------------------
// max |x| of one or two fp32 arrays as a bit pattern (non-negative floats order like unsigned integers):
// *amax = max(*amax, bits(max |x|)).  The slot must hold 0 (or an earlier maximum) on entry.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float4* x0, long n0v, const float4* x1, long n1v,
                                                     unsigned* amax) {
  __shared__ unsigned sh[4];
  unsigned m = 0;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n0v + n1v; i += stride) {
    const float4 v = i < n0v ? x0[i] : x1[i - n0v];
    const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7FFFFFFFu, b = __builtin_bit_cast(unsigned, v.y) & 0x7FFFFFFFu,
                   c = __builtin_bit_cast(unsigned, v.z) & 0x7FFFFFFFu, d = __builtin_bit_cast(unsigned, v.w) & 0x7FFFFFFFu;
    m = max(m, max(max(a, b), max(c, d)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
    // one atomic per workgroup, and only if it can raise the slot (all adders hit ONE address)
    if (m > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, m);
  }
}

// Conv weight [Cout][taps][Cin] fp32 -> fp16 parts of (w * scale) in MFMA fragment order:
//   out[part][R/32][Kc/32][tap][nb 2][lane 64][8],  lane = 16 q + r mod 16, row r = 32 tile + 16 nb + lane mod 16,
//   k = 32 chunk + 8 q + j
// forward:    rows r = co, k = ci, value w[r][tap][k]
// transposed: rows r = ci, k = co, value w[k][taps-1-tap][r]   (the conv that IS the data gradient)
__device__ __forceinline__ void weight_frag16_block(const float* w, const unsigned* w_amax, unsigned short* out, int Cout,
                                                    int Cin, int taps, int transposed, long ps, long block) {
  const int R = transposed ? Cin : Cout, Kc = transposed ? Cout : Cin;
  const int cch = Kc >> 5;
  const long cell = block * 2 + (threadIdx.x >> 7);  // ((nt * cch + cc) * taps + tap)
  if (cell >= (long)(R >> 5) * cch * taps) return;
  const float s = pow2f(scale_exp(*w_amax));
  const int tap = (int)(cell % taps);
  const int cc = (int)((cell / taps) % cch);
  const int nt = (int)(cell / ((long)taps * cch));
  const int nb = (threadIdx.x >> 6) & 1, lane = threadIdx.x & 63;
  const int r = nt * 32 + nb * 16 + (lane & 15);
  const int k0 = cc * 32 + 8 * (lane >> 4);
  u32x4 lo, hi;
  if (!transposed) {
    const u32x4* src = reinterpret_cast<const u32x4*>(w + ((long)r * taps + tap) * Cin + k0);
    lo = src[0], hi = src[1];
  } else {
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = __builtin_bit_cast(unsigned, w[((long)(k0 + j) * taps + (taps - 1 - tap)) * Cin + r]);
    lo = u32x4{v[0], v[1], v[2], v[3]}, hi = u32x4{v[4], v[5], v[6], v[7]};
  }
  u32x4 q[2];
  split8h(lo, hi, s, q);
  const long o = (cell * 2 + nb) * 512 + lane * 8;
  *reinterpret_cast<u32x4*>(out + o) = q[0];
  *reinterpret_cast<u32x4*>(out + ps + o) = q[1];
}

__global__ void weight_frag16_kernel(const float* w, const unsigned* w_amax, unsigned short* out, int Cout, int Cin,
                                     int taps, int transposed, long ps) {
  weight_frag16_block(w, w_amax, out, Cout, Cin, taps, transposed, ps, blockIdx.x);
}

// Many tensors in one launch (every conv weight of a model after an optimiser step): job j owns the workgroups
// [block_begin_j, block_begin_{j+1}); the job table lives in device memory.
template <typename Job>
__device__ __forceinline__ int job_of_block(const Job* jobs, int n_jobs, long block) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_begin <= block) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ void weight_frag16_multi_kernel(const rac_frag_job* jobs, int n_jobs) {
  const int j = job_of_block(jobs, n_jobs, blockIdx.x);
  const rac_frag_job q = jobs[j];
  weight_frag16_block(q.w, q.w_amax, q.parts, q.Cout, q.Cin, q.ksize * q.ksize, q.transposed, q.part_stride,
                      blockIdx.x - q.block_begin);
}

__global__ __launch_bounds__(256) void absmax_multi_kernel(const rac_absmax_job* jobs, int n_jobs) {
  __shared__ unsigned sh[4];
  const int j = job_of_block(jobs, n_jobs, blockIdx.x);
  const rac_absmax_job q = jobs[j];
  const long nb = (j + 1 < n_jobs ? jobs[j + 1].block_begin : (long)gridDim.x) - q.block_begin;
  const float4* x = reinterpret_cast<const float4*>(q.x);
  const long nv = q.n >> 2;
  unsigned m = 0;
  for (long i = (blockIdx.x - q.block_begin) * 256 + threadIdx.x; i < nv; i += nb * 256) {
    const float4 v = x[i];
    m = max(m, max(max(absbits(v.x), absbits(v.y)), max(absbits(v.z), absbits(v.w))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
    if (m > __hip_atomic_load(q.amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(q.amax, m);
  }
}


// ---------------------------------------------------------------------------------------------------------
// Adam step AND the next step's operand parts in one pass over a conv weight (torch.optim.Adam.step(), trainer.py:461, +
// rac_absmax + rac_weight_frag_split x 2): the separate passes read the 954 MB of weights three times after Adam wrote
// them.  A workgroup owns two 32 x 32 (co, ci) cells of one tap, as weight_frag16_kernel does: its threads update 8
// consecutive input channels each (p, g, m, v in, p, m, v out), write the forward fragment, and pass the new values
// through LDS for the transposed, tap-flipped fragment of the data gradient's weight.  The parts need their scale
// BEFORE the new maximum exists: the caller provides an upper bound of max |p_new| (old exact maximum + the largest
// step Adam can take), whose exponent is the scale the convs then undo; the exact new maximum is folded into amax_out
// for the next bound.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_frag_multi_kernel(const rac_adam_frag_job* jobs, int n_jobs, long total_blocks,
                                                              float b1, float b2, float eps, float step_size,
                                                              float inv_sqrt_bc2) {
  __shared__ float tile[2][32][33];
  __shared__ unsigned mx_sh[4];
  // one block of work = two 32 x 32 cells; a workgroup takes blocks blockIdx.x, + gridDim.x, ... (the grid is the block
  // count, or a bounded number of workgroups: rac_adam_frag_multi_bounded -- a pass that shares the chip with other
  // streams' kernels leaves them registers and LDS)
  for (long blk = blockIdx.x; blk < total_blocks; blk += gridDim.x) {
  const int j = job_of_block(jobs, n_jobs, (int)blk);
  const rac_adam_frag_job q = jobs[j];
  const int taps = q.ksize * q.ksize, cch = q.Cin >> 5;
  const int half = threadIdx.x >> 7;
  const long cell = (blk - q.block_begin) * 2 + half;  // ((nt * cch + cc) * taps + tap)
  // (walking the cells along the input channels first -- consecutive workgroups on consecutive 256-byte pieces of the same
  // weight rows, the fragment writes then 50 KB apart -- measured 2.02 ms against 1.85 for this order)
  const bool active = cell < (long)(q.Cout >> 5) * cch * taps;
  const int tap = (int)(cell % taps);
  const int cc = (int)((cell / taps) % cch);
  const int nt = (int)(cell / ((long)taps * cch));
  const int nb = (threadIdx.x >> 6) & 1, lane = threadIdx.x & 63;
  const int rl = nb * 16 + (lane & 15), kl = 8 * (lane >> 4);  // row (co) and first column (ci) inside the cell
  const float s = pow2f(scale_exp(*q.scale_slot));
  unsigned mx = 0;
  if (active) {
    const long off = ((long)(nt * 32 + rl) * taps + tap) * q.Cin + cc * 32 + kl;
    f32x4 P[2], G[2], M[2], V[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      P[h] = *reinterpret_cast<const f32x4*>(q.p + off + 4 * h);
      G[h] = *reinterpret_cast<const f32x4*>(q.g + off + 4 * h);
      M[h] = *reinterpret_cast<const f32x4*>(q.m + off + 4 * h);
      V[h] = *reinterpret_cast<const f32x4*>(q.v + off + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pp = P[h][e], mm = M[h][e], vv = V[h][e];
        adam_update(pp, G[h][e], mm, vv, b1, b2, eps, step_size, inv_sqrt_bc2);  // the same bits as adam_kernel
        M[h][e] = mm, V[h][e] = vv, P[h][e] = pp;
        mx = max(mx, absbits(pp));
        tile[half][rl][kl + 4 * h + e] = pp;
      }
      *reinterpret_cast<f32x4*>(q.p + off + 4 * h) = P[h];
      *reinterpret_cast<f32x4*>(q.m + off + 4 * h) = M[h];
      *reinterpret_cast<f32x4*>(q.v + off + 4 * h) = V[h];
    }
    if (q.parts_fwd) {
      u32x4 parts[2];
      split8h(__builtin_bit_cast(u32x4, P[0]), __builtin_bit_cast(u32x4, P[1]), s, parts);
      const long o = (cell * 2 + nb) * 512 + lane * 8;
      *reinterpret_cast<u32x4*>(q.parts_fwd + o) = parts[0];
      *reinterpret_cast<u32x4*>(q.parts_fwd + q.part_stride + o) = parts[1];
    }
  }
  __syncthreads();
  if (active && q.parts_t) {
    // transposed cell: rows = this cell's input channels, k = its output channels, tap flipped
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[half][kl + e][rl];
    u32x4 parts[2];
    split8h(u32x4{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]),
                  __builtin_bit_cast(unsigned, v[3])},
            u32x4{__builtin_bit_cast(unsigned, v[4]), __builtin_bit_cast(unsigned, v[5]), __builtin_bit_cast(unsigned, v[6]),
                  __builtin_bit_cast(unsigned, v[7])}, s, parts);
    const long cell_t = ((long)cc * (q.Cout >> 5) + nt) * taps + (taps - 1 - tap);
    const long o = (cell_t * 2 + nb) * 512 + lane * 8;
    *reinterpret_cast<u32x4*>(q.parts_t + o) = parts[0];
    *reinterpret_cast<u32x4*>(q.parts_t + q.part_stride + o) = parts[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  if (lane == 0) mx_sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = max(max(mx_sh[0], mx_sh[1]), max(mx_sh[2], mx_sh[3]));
    if (mx > __hip_atomic_load(q.amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(q.amax_out, mx);
  }
  }
}

// bound[i] = bits(exact[i] as float + margin); exact[i] = 0, for the listed slots: the scale of the parts the fused
// Adam pass writes, and the zeroed accumulator of the maximum it measures
__global__ void amax_bound_kernel(unsigned* exact, unsigned* bound, const int* idx, int n, float margin) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int i = idx[j];
  bound[i] = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, exact[i]) + margin);
  exact[i] = 0;
}


// ---------------------------------------------------------------------------------------------------------
// Weight gradient   dw[co][ky][kx][ci] (+)= sum_p dy[p][co] * x[p + (ky - pad, kx - pad)][ci]
// as a GEMM over pixels (M = co, N = ci per tap, K = pixels), both operands read in their natural NHWC layout:
// no transposed or shifted copies exist anywhere.
//   * K walks pixel COLUMNS: one K block = the 32 pixels (image, y) x {one column c} of 32 consecutive image rows
//     (any order of the K index is allowed as long as both operands use the same one).  All pixels of a block share
//     their x coordinate, so a horizontal tap kx is either valid for the whole block or skipped -- no masks, and the
//     border taps cost no MFMAs.
//   * a workgroup = (128 co) x (64 ci) x (one kernel row ky, all KS kernel columns): it stages the dy tile of column c
//     and keeps a ring of input columns c - pad .. c + pad (+1 being loaded), so ONE staged input column serves every
//     horizontal tap; the vertical shift ky is folded into the staging address (rows outside the image are zeros).
//   * fp32 -> two fp16 parts on the way into LDS ([pixel][channel] rows of 256 B, 32-byte segments XOR-swizzled by
//     the row), and ds_read_b64_tr_b16 hands the MFMA its K-major fragments: the transposition is free.
//   * time steps (and images) are just more K: up to 16 (dy, x0, x1) triples per launch, one read-modify-write of dw.
// Deterministic: no atomics; a K split writes slabs that rac_slab_accumulate adds in a fixed order.
// ---------------------------------------------------------------------------------------------------------
struct Wgrad16P {
  int Bimg, H, W, Cin, Cout, C0, T;
  int R, G;      // image rows (Bimg * H) and 32-row groups per step
  int nsplit, accumulate;
  int nseg;      // wgrad16_allky_kernel: column segments per image row (K units = groups x segments)
  int x1_skip;   // leading steps whose x1 is all zeros: the workgroups of the x1 half start behind them
  int presplit;  // the operand pointers are fp16 part pairs [2][elements] (rac_split_steps), already scaled
  const float* dy[RAC_WGRAD_MAX_STEPS];
  const float* x0[RAC_WGRAD_MAX_STEPS];
  const float* x1[RAC_WGRAD_MAX_STEPS];
  const unsigned* dy_amax[RAC_WGRAD_MAX_STEPS];
  const unsigned* x0_amax[RAC_WGRAD_MAX_STEPS];
  const unsigned* x1_amax[RAC_WGRAD_MAX_STEPS];
  float* dw;
  float* slabs;
  long slab_stride;
};

typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define RAC_LDS_PTR(T, base, off) reinterpret_cast<__attribute__((address_space(3))) T*>( \
    (__attribute__((address_space(3))) unsigned char*)(base) + (off))

// Swizzle of the [pixel row][256 B] operand images of the weight-gradient kernels: the 32-byte segment index of row r is
// XORed with wg_swz(r).  A half-wave of a transposing read (ds_read_b64_tr_b16) touches rows {a .. a+3} and {a+8 .. a+11},
// 32 bytes each, and needs eight different values there: the low two bits separate four consecutive rows, bit 2 the
// two groups (measured conflict free, tools/micro/lds_conflicts.hip).
__device__ __forceinline__ int wg_swz(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// the 16x16x32 operand of lane (i = l & 15, g = l >> 4): column i of pixel rows 8g .. 8g+7 of a [row][256 B] image
__device__ __forceinline__ f16x8 tr_frag(const unsigned char* lds, int addr) {
  typedef __fp16 h16x8 __attribute__((__vector_size__(8 * sizeof(__fp16))));
  const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(RAC_LDS_PTR(h16x4, lds, addr));
  const h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(RAC_LDS_PTR(h16x4, lds, addr + 1024));
  // one register tuple: the two 64-bit reads land in the halves of the MFMA operand, no moves
  return __builtin_bit_cast(f16x8, (h16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// WMW = waves along the output channels: 2 -> workgroup 128 co x 64 ci, waves 64 co x 32 ci; 1 -> 64 co x 64 ci, waves
// 64 co x 16 ci (64-channel layers: no all-zero half tile; its dy tile is staged by the first two waves only).
// PRE: the operand pointers are fp16 part pairs [2][elements] (rac_split_steps), already scaled: staging is a copy.
//
// Round 5: the column step is straight-line code.  (Round 4's SQ counters: 0.9-1.3 other vector instructions per MFMA and
// the matrix pipe 43-48 % busy; its ISA re-derived every step's addresses from the cursors -- a scalar and a vector
// division, 64-bit multiplies, a descriptor from the pointer table with its s_load + wait -- branched on the run-time
// `presplit` flag around every load and store, and added a run-time slot offset to every fragment address.)  Now:
//   * the step loop is unrolled over the NR ring phases (NR = 2 PAD + 2 is even, so the dy buffer is a phase constant
//     too): every LDS address of a step is a lane constant plus an instruction immediate;
//   * a lane's global offset is constant for a whole 32-row group (its row and channels), the column walks in the
//     buffer instructions' SCALAR offset (+ one pixel per step), and the descriptor, the lane's validity (rows past the
//     tensor, vertical taps outside the image -> the OOB offset, which reads zeros) and the group's scalar base are set up
//     once per group (every W steps) under a uniform branch;
//   * the operand format is a template parameter.
template <int N>
struct StepPhase {
  static constexpr int v = N;
};
constexpr unsigned OOBV = 0xFFFFFF00u;  // past every range (tensors are < 0xFFFFFF00 bytes) and OOBV + immediate does not wrap

__device__ __forceinline__ u32x4 load16s(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

template <int KS, int WMW, bool PRE>
__global__ __launch_bounds__(256, 2) void wgrad16_kernel(Wgrad16P p) {
  constexpr int PAD = KS / 2, NR = 2 * PAD + 2;
  // AHEAD = column steps between an operand request and its LDS stores.  2: a second set of staging registers; the
  // stores of the set requested a step ago are spread over this step's MFMA blocks from the first one on.  1 (the 5x5
  // 128-co form sits at 256 registers): requested at the top of the step, stored between its later blocks.
  constexpr int AHEAD = (KS == 5 && WMW == 2) ? 1 : 2;
  constexpr int DYB = 16384, XSLOT = 8192, X_BASE = 2 * DYB;
  constexpr unsigned EB = PRE ? 2u : 4u;  // bytes per operand element in memory
  static_assert(NR % 2 == 0, "the dy buffer index is a function of the ring phase");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int WN = 4 / WMW, NU = 4 / WN;  // waves along ci, 16-ci blocks per wave
  const int wm = wid / WN, wn = wid % WN;
  const int co0 = blockIdx.x * (64 * WMW), ci0 = blockIdx.y * 64;
  const int ky = blockIdx.z % KS, split = blockIdx.z / KS;
  const bool first = ci0 < p.C0;
  const int Cs = first ? p.C0 : p.Cin - p.C0;
  const int cl0 = first ? ci0 : ci0 - p.C0;

  float sd = 1.f, sx = 1.f;
  int kd, kxs;
  {
    unsigned amd = 0, amx = 0;
    for (int t = 0; t < p.T; ++t) {
      amd = max(amd, *p.dy_amax[t]);
      amx = max(amx, *p.x0_amax[t]);
      if (p.x1_amax[t]) amx = max(amx, *p.x1_amax[t]);
    }
    kd = scale_exp(amd), kxs = scale_exp(amx);
    sd = pow2f(kd), sx = pow2f(kxs);
  }

  const int g_skip = first ? 0 : p.x1_skip * p.G;
  const int NG = p.T * p.G;
  const int gpb = (NG - g_skip + p.nsplit - 1) / p.nsplit;
  const int g_begin = g_skip + split * gpb;
  const int g_end = min(NG, g_begin + gpb);
  const int S = max(0, g_end - g_begin) * p.W;  // column steps of this workgroup

  // ---- staging roles: row kk = tid >> 3 of the 32-row block; dy: 16 channels (segment tid & 7), x: 8 channels ----
  const int skk = tid >> 3, ssub = tid & 7;
  const int ssw = wg_swz(skk);
  // dy roles of the 64-co form: threads 0..127 (waves 0 and 1), row tid >> 2, 16-channel segment tid & 3
  const bool dy_role = WMW == 2 || wid < 2;  // (wave-uniform)
  const int dkk = WMW == 2 ? skk : (tid >> 2) & 31, dsub = WMW == 2 ? ssub : tid & 3;
  const int dsw = wg_swz(dkk);
  const int dy_lds = dkk * 256 + ((dsub ^ dsw) * 32);                       // + buf * DYB + part * 8192
  // A dy thread stores its 16 channels as the two 16-byte halves of one 32-byte segment.  LDS stores retire 8 lanes x 16
  // bytes per clock into 32 banks (addresses mod 128 B: measured, tools/micro/lds_conflicts.hip), and the 8 lanes of one
  // store instruction sit either in the SAME half of the eight segments of a row (128-co form: segments s and s + 4
  // collide) or of four segments of two rows (64-co form: the rows collide) -- round 3's 2-way conflict on every dy
  // store (18-20 % of the kernels' LDS cycles, profiles/r03i_sq_summary.md).  So half of them take their halves in the
  // opposite order: by bit 2 of the segment, or by the row's parity.  The loads are issued in the matching order: no
  // data moves, only lane-constant address bits.
  const unsigned dflip = (unsigned)(WMW == 2 ? ((dsub ^ dsw) >> 2) & 1 : dkk & 1) * 16u;
  const int x_lds = X_BASE + skk * 256 + (((ssub >> 1) ^ ssw) * 32) + (ssub & 1) * 16;  // part 1: segment ^ 4 -> ^ 128
  const bool dy_ch_ok = co0 + dsub * 16 < p.Cout;
  const bool x_ch_ok = cl0 + ssub * 8 < Cs;

  // ---- loaders: (time step, 32-row group, column) cursors; the dy cursor runs one column ahead of the MFMAs, the x
  //      cursor PAD + 1 columns.  Memory offset of (row r = 32 G + kk, column c, channel ch) = ((r W + c) C + ch) EB is
  //      split three ways: the DESCRIPTOR's base carries the group (32 G W C EB, + the step's pointer), the lane offset
  //      carries (kk W C + ch) EB -- a constant of the lane -- and the instruction's SCALAR offset carries the column
  //      (c C EB).  A descriptor of (rows left in the tensor) W C EB bytes zero-fills the rows past the tensor's end
  //      (gfx950 range-checks lane + scalar offset against it: measured -- a part-1 load through the part-0 descriptor
  //      with the part stride as its scalar offset returns zeros; operand parts therefore get a descriptor each), and
  //      lanes that are invalid for the whole launch (channels past the tensor; vertical taps outside the image when 32
  //      rows are whole images) carry the out-of-range offset as their constant: entering a group is scalar work. ----
  const unsigned dy_col = (unsigned)p.Cout * EB, dy_row = (unsigned)p.W * dy_col, dy_grp = 32u * dy_row;
  const unsigned dy_part = PRE ? (unsigned)((long)p.R * dy_row) : 0u;
  const unsigned dy_lane = ((unsigned)(dkk * p.W) * (unsigned)p.Cout + (unsigned)(co0 + dsub * 16)) * EB;
  // fp32: four 16-byte pieces at {0, 16, 32, 48} ^ 2 dflip (channels 8..15 first for half of the lanes) = two lane
  // offsets + the immediate 16; parts: the 32 bytes of a part as two halves, in the lane's order, part 1 = scalar + part
  const unsigned d_v0 = dy_ch_ok ? dy_lane + (PRE ? dflip : 2u * dflip) : OOBV;
  const unsigned d_v1 = dy_ch_ok ? dy_lane + (PRE ? (dflip ^ 16u) : (32u ^ (2u * dflip))) : OOBV;
  const unsigned x_col = (unsigned)Cs * EB, x_row = (unsigned)p.W * x_col, x_grp = 32u * x_row;
  const unsigned x_part = PRE ? (unsigned)((long)p.R * x_row) : 0u;
  const unsigned x_lane = ((unsigned)(skk * p.W) * (unsigned)Cs + (unsigned)(cl0 + ssub * 8)) * EB;
  // the vertical tap moves the descriptor's base by ky - PAD image rows: the row a lane reads is r + ky - PAD, valid iff
  // its image row y = r mod H + ky - PAD stays inside the image (which also keeps it inside the tensor)
  const long x_shift = (long)(ky - PAD) * (long)x_row;
  const bool h32 = __builtin_amdgcn_readfirstlane((32 % p.H) == 0 ? 1 : 0) != 0;  // a lane's image row is then the same in every group
  const bool y_ok0 = (unsigned)(skk % p.H + ky - PAD) < (unsigned)p.H;
  const float* const* xsrc = first ? p.x0 : p.x1;

  int dT = g_begin / p.G, dG = g_begin - dT * p.G, dC = 0;
  int xT = dT, xG = dG, xC = 0;
  rsrc_t d_rs = make_rsrc(p.dy[0], 0u), x_rs = d_rs, d_rs1 = d_rs, x_rs1 = d_rs;  // (rs1: the second parts)
  unsigned d_so = 0, x_so = 0;
  unsigned x_v = (x_ch_ok & (y_ok0 | !h32)) ? x_lane : OOBV;
  u32x4 rdS[AHEAD][4], rxS[AHEAD][2];
  auto issue_dy = [&](u32x4 (&rd)[4]) {
    if (dC == 0) {  // (uniform) entering a group
      const char* base = reinterpret_cast<const char*>(p.dy[dT]) + (long)dG * dy_grp;
      d_rs = make_rsrc(base, (unsigned)(p.R - dG * 32) * dy_row);
      if constexpr (PRE) d_rs1 = make_rsrc(base + dy_part, (unsigned)(p.R - dG * 32) * dy_row);
      d_so = 0;
    }
    if (dy_role) {
      if constexpr (PRE) {
        rd[0] = load16s(d_rs, d_v0, d_so);
        rd[1] = load16s(d_rs, d_v1, d_so);
        rd[2] = load16s(d_rs1, d_v0, d_so);
        rd[3] = load16s(d_rs1, d_v1, d_so);
      } else {
        rd[0] = load16s(d_rs, d_v0, d_so);
        rd[1] = load16s(d_rs, d_v0 + 16u, d_so);
        rd[2] = load16s(d_rs, d_v1, d_so);
        rd[3] = load16s(d_rs, d_v1 + 16u, d_so);
      }
    }
    d_so += dy_col;
    if (++dC == p.W) {
      dC = 0;
      if (++dG == p.G) dG = 0, ++dT;
    }
  };
  auto issue_x = [&](u32x4 (&rx)[2]) {
    if (xC == 0) {
      const char* base = reinterpret_cast<const char*>(xsrc[xT]) + ((long)xG * x_grp + x_shift);
      x_rs = make_rsrc(base, (unsigned)(p.R - xG * 32) * x_row);
      if constexpr (PRE) x_rs1 = make_rsrc(base + x_part, (unsigned)(p.R - xG * 32) * x_row);
      x_so = 0;
      if (!h32) {  // (uniform; 6- / 12-row maps) the lane's image row changes with the group
        const int y = (xG * 32 + skk) % p.H + ky - PAD;
        x_v = (x_ch_ok & ((unsigned)y < (unsigned)p.H)) ? x_lane : OOBV;
      }
    }
    if constexpr (PRE) {
      rx[0] = load16s(x_rs, x_v, x_so);
      rx[1] = load16s(x_rs1, x_v, x_so);
    } else {
      rx[0] = load16s(x_rs, x_v, x_so);
      rx[1] = load16s(x_rs, x_v + 16u, x_so);
    }
    x_so += x_col;
    if (++xC == p.W) {
      xC = 0;
      if (++xG == p.G) xG = 0, ++xT;
    }
  };
  // The six 16-byte LDS stores of a step's staged operands, as separately placeable pieces (a ds_write_b128 moves its
  // data to the LDS for ~13 cycles, at half rate when one wave stores alone: 24 KB per workgroup and step cost 12-18 % of
  // the kernel as a block behind the MFMAs -- timing builds, profiles/r05_wgrad_decomposition.md -- so each piece sits behind
  // one block of 12 MFMAs).  Pieces 0..3: the dy tile (fp32 operands: pieces 0 / 2 convert a pair of vectors, 1 / 3 store
  // the second part), 4..5: the input column.
  u32x4 qd[2], qx[2];
  auto stage_piece = [&](int j, u32x4 (&rd)[4], u32x4 (&rx)[2], int buf, int slot, bool do_dy, bool do_x) {
    if (j < 4) {
      if (!(dy_role && do_dy)) return;
      unsigned char* d = lds_raw + buf * DYB + dy_lds;
      if constexpr (PRE) {  // rd[0..1]: the halves of part 0 in the lane's order, rd[2..3]: of part 1
        *reinterpret_cast<u32x4*>(d + (j >> 1) * 8192 + ((j & 1) ? (dflip ^ 16u) : dflip)) = rd[j];
      } else {
        if (j == 0) split8h(rd[0], rd[1], sd, qd);
        if (j == 2) split8h(rd[2], rd[3], sd, qd);
        *reinterpret_cast<u32x4*>(d + (j & 1) * 8192 + ((j & 2) ? (dflip ^ 16u) : dflip)) = qd[j & 1];
      }
    } else {
      if (!do_x) return;
      if constexpr (PRE) {
        *reinterpret_cast<u32x4*>(lds_raw + slot * XSLOT + ((j & 1) ? (x_lds ^ 128) : x_lds)) = rx[j & 1];
      } else {
        if (j == 4) split8h(rx[0], rx[1], sx, qx);
        *reinterpret_cast<u32x4*>(lds_raw + slot * XSLOT + ((j & 1) ? (x_lds ^ 128) : x_lds)) = qx[j & 1];
      }
    }
  };

  // ---- fragment addresses (lane constants) ----
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  const int fsw = wg_swz(8 * fg + fq);  // (= that of row + 4, the second half of a fragment)
  const int lane_base = (8 * fg + fq) * 256 + fp * 8;
  int offA[4], offB[2][NU];
#pragma unroll
  for (int t = 0; t < 4; ++t) offA[t] = lane_base + (((wm * 4 + t) ^ fsw) * 32);
#pragma unroll
  for (int part = 0; part < 2; ++part)
#pragma unroll
    for (int u = 0; u < NU; ++u) offB[part][u] = X_BASE + lane_base + (((part * 4 + wn * NU + u) ^ fsw) * 32);

  f32x4 acc[KS][4][NU];
#pragma unroll
  for (int k = 0; k < KS; ++k)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < NU; ++u) acc[k][t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (WMW == 1) {  // the channel segments of the dy tiles that nobody stages (co 64..127 of the 256-byte rows)
#pragma unroll
    for (int v = 0; v < 8; ++v)
      *reinterpret_cast<u32x4*>(lds_raw + (v * 256 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
  }
  if (S > 0) {
    // prologue: input columns 0 .. PAD and the dy tile of column 0 (AHEAD = 2: and the requests of column 1's operands)
    for (int e = 0; e <= PAD && e < S; ++e) {
      issue_x(rxS[0]);
#pragma unroll
      for (int j = 4; j < 6; ++j) stage_piece(j, rdS[0], rxS[0], 0, e, false, true);
    }
    issue_dy(rdS[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_piece(j, rdS[0], rxS[0], 0, 0, true, false);
    if constexpr (AHEAD == 2) {  // what step 0 stores: requested "in step -1", set 1
      if (1 < S) issue_dy(rdS[1]);
      if (PAD + 1 < S) issue_x(rxS[1]);
    }
    __syncthreads();
    int c = 0, s = 0;  // column of the current step inside its image row, step index
    // one column step at ring phase PH (= s mod NR): the current input column sits in slot PH, the dy tile in buffer PH & 1
    auto step = [&](auto phase) {
      constexpr int PH = decltype(phase)::v;
      constexpr int dbuf = (PH & 1) * DYB;
      constexpr int LSET = PH % AHEAD, SSET = (PH + 1) % AHEAD;  // register sets: requested into / stored from in this step
      constexpr int NB = KS * NU;                                 // MFMA blocks of a step (12 MFMAs each)
      // first block with a store piece behind it, blocks that take pieces.  AHEAD = 1 is the 5x5 128-co form, whose 160
      // accumulator + 48 fragment + 24 staging registers leave no room for store addresses while fragments are live
      // (the allocator spills ACCUMULATORS to scratch if pieces sit between its blocks): its pieces follow the last
      // blocks, where the fragment registers are dead
      // (fp32 operands add the conversion's temporaries: all pieces behind the last block there)
      constexpr int LATE = PRE ? RAC_WGRAD_LATE_BLOCKS : 0;
      constexpr int FIRST = AHEAD == 2 ? 0 : NB - 1 - LATE;
      constexpr int NSLOT = AHEAD == 2 ? NB - 1 : LATE + 1;
      const bool st_dy = s + 1 < S, st_x = s + PAD + 1 < S;
      if (!(RAC_EXP_WGRAD & 2)) {
        if (s + AHEAD < S) issue_dy(rdS[LSET]);
        if (s + PAD + AHEAD < S) issue_x(rxS[LSET]);
      }
      auto pieces_behind = [&](int blk) {  // (blk is a compile-time constant at every call site)
        if (RAC_EXP_WGRAD & 4) return;
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if (FIRST + (j * NSLOT) / 6 == blk)
            stage_piece(j, rdS[SSET], rxS[SSET], (PH + 1) & 1, (PH + PAD + 1) % NR, st_dy, st_x);
      };
      f16x8 fa[4][2];
      if constexpr (NU == 2 && RAC_WGRAD_ROLL) {
        // The input fragments roll through the taps in halves: the 16-ci block u of tap k + 1 is read into the registers of
        // block u of tap k as soon as that block's 12 MFMAs are issued, under the 12 MFMAs of the other block -- every tap's
        // fragments are in flight for half a tap before their first use instead of being waited for in front of it.  Reads
        // are unconditional (a tap that leaves the image row reads a stale slot), only its MFMAs are skipped.
        f16x8 fb[2][2];
        auto read_fb = [&](int u, int k) {
          const int sl = (PH + k - PAD + NR) % NR;
#pragma unroll
          for (int part = 0; part < 2; ++part) fb[u][part] = tr_frag(lds_raw, sl * XSLOT + offB[part][u]);
        };
        // the first block's operands first: its MFMAs start while the rest of the step's fragments are still in flight
        read_fb(0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int part = 1; part >= 0; --part) fa[t][part] = tr_frag(lds_raw, dbuf + part * 8192 + offA[t]);
        read_fb(1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const bool valid = (unsigned)(c + k - PAD) < (unsigned)p.W;  // uniform: the tap stays inside the image row
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (valid) {
#pragma unroll
              for (int t = 0; t < 4; ++t) acc[k][t][u] = mma3(fa[t], fb[u], acc[k][t][u]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < KS) read_fb(u, k + 1);
            pieces_behind(k * 2 + u);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int part = 0; part < 2; ++part) fa[t][part] = tr_frag(lds_raw, dbuf + part * 8192 + offA[t]);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          if ((unsigned)(c + k - PAD) < (unsigned)p.W) {  // uniform: the tap stays inside the image row
            const int sl = (PH + k - PAD + NR) % NR;
            f16x8 fb[NU][2];
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
              for (int part = 0; part < 2; ++part) fb[u][part] = tr_frag(lds_raw, sl * XSLOT + offB[part][u]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int u = 0; u < NU; ++u) acc[k][t][u] = mma3(fa[t], fb[u], acc[k][t][u]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < NU; ++u) pieces_behind(k * NU + u);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!(RAC_EXP_WGRAD & 1)) __syncthreads();
      c = (c + 1 == p.W) ? 0 : c + 1;
      ++s;
    };
    for (;;) {
      step(StepPhase<0>{});
      if (s == S) break;
      step(StepPhase<1>{});
      if (s == S) break;
      step(StepPhase<2>{});
      if (s == S) break;
      step(StepPhase<3>{});
      if (s == S) break;
      if constexpr (NR == 6) {
        step(StepPhase<4>{});
        if (s == S) break;
        step(StepPhase<5>{});
        if (s == S) break;
      }
    }
  }

  // ---- epilogue: rows (co) = 4 (lane >> 4) + reg of each 16-row block, column (ci) = lane & 15 ----
  const float id = pow2f(-kd), ix = pow2f(-kxs);
  const int lr = lane & 15, lq = lane >> 4;
  float* dst = split == 0 ? p.dw : p.slabs + (long)(split - 1) * p.slab_stride;
  const bool add = split == 0 && p.accumulate;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int cil = cl0 + wn * (16 * NU) + u * 16 + lr;  // channel inside its source
    if (cil >= Cs) continue;
    const int ci = (first ? 0 : p.C0) + cil;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wm * 64 + t * 16 + 4 * lq + r;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const long idx = (((long)co * KS + ky) * KS + k) * p.Cin + ci;
          const float v = acc[k][t][u][r] * id * ix;
          dst[idx] = add ? dst[idx] + v : v;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The 3x3 weight gradient of the THIN layers (Cout <= 128 on 32x32 / 64x64 maps: K = 80 000 .. 330 000 pixels against a
// 64 x 576 .. 128 x 2304 output): wgrad16_kernel gives each kernel ROW its own workgroup, so every dy tile is staged once
// per (input-channel tile, ky) and every input column three times -- 645 MB fetched for a 0.3 MB result, 0.07-0.13 of the
// pipe.  Here one workgroup (64 co x 64 ci) keeps ALL NINE taps: a K block is still 32 consecutive image rows x one
// pixel column, but the input column is staged with one halo row above and below (34 rows) and the three vertical taps
// read it at row offsets 0 / 1 / 2 -- dy and x leave HBM once per (co tile, ci tile).  Needs H % 32 == 0 (a 32-row group
// never straddles two images: only its halo rows can fall outside, and they are zeroed as a whole).
// LDS rows are [pixel row][256 B] with the 32-byte segments XOR-swizzled by the row index, as in wgrad16_kernel; a
// fragment of a shifted window starts at an arbitrary row, so the two 4-row halves of a transposing read get their own
// swizzled addresses.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad16_allky_kernel(Wgrad16P p) {
  constexpr int KS = 3, NR = 4;
  constexpr int DYB = 16384, XROWS = 34, XSLOT = 40 * 256, X_BASE = 2 * DYB;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // waves along ci: 64 co x 16 ci each
  const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64;
  const int split = blockIdx.z;
  const bool first = ci0 < p.C0;
  const int Cs = first ? p.C0 : p.Cin - p.C0;
  const int cl0 = first ? ci0 : ci0 - p.C0;

  unsigned amd = 0, amx = 0;
  for (int t = 0; t < p.T; ++t) {
    amd = max(amd, *p.dy_amax[t]);
    amx = max(amx, *p.x0_amax[t]);
    if (p.x1_amax[t]) amx = max(amx, *p.x1_amax[t]);
  }
  const int kd = scale_exp(amd), kxs = scale_exp(amx);
  const float sd = pow2f(kd), sx = pow2f(kxs);

  // K units: (32-row group, column segment) pairs -- p.nseg segments of cseg columns per image row; a workgroup walks
  // its units one after the other (each with its own 1-column halo left and right)
  const int cseg = p.W / p.nseg;
  const int NU = p.T * p.G * p.nseg;
  const int upw = (NU + p.nsplit - 1) / p.nsplit;
  const int u_begin = split * upw;
  const int u_end = min(NU, u_begin + upw);

  // ---- staging roles ----
  // dy: threads 0..127, row tid >> 2 of the 32-row block, 16-channel segment tid & 3 (co 64..127 of the rows stay zero)
  const bool dy_role = tid < 128;
  const int dkk = (tid >> 2) & 31, dsub = tid & 3;
  const int dsw = wg_swz(dkk);
  const int dy_lds = dkk * 256 + ((dsub ^ dsw) * 32);
  const unsigned dflip = (unsigned)(dkk & 1) * 16u;  // (see wgrad16_kernel: conflict-free staging stores)
  const bool dy_ch_ok = co0 + dsub * 16 < p.Cout;
  // x: staged row i = tid >> 3 (and 32 + (tid >> 3) for tid < 16), 8 channels tid & 7; row i holds image row
  // 32 * group + i - 1
  const int ssub = tid & 7;
  const bool x_ch_ok = cl0 + ssub * 8 < Cs;
  auto x_lds_of = [&](int i) {
    const int sw = wg_swz(i);
    return X_BASE + i * 256 + (((ssub >> 1) ^ sw) * 32) + (ssub & 1) * 16;
  };
  const int xi0 = tid >> 3, xi1 = 32 + (tid >> 3);
  const bool x_two = tid < 16;
  const int x_lds0 = x_lds_of(xi0), x_lds1 = x_lds_of(xi1);
  u32x4 rd[4], rx[4];
  auto issue_dy = [&](int gidx, int col) {  // dy tile of column `col` of group `gidx` (= step * G + group)
    if (dy_role) {
      const int t = gidx / p.G, gr = gidx - t * p.G;
      const int r = gr * 32 + dkk;
      const bool ok = (r < p.R) & dy_ch_ok;
      const rsrc_t rs = make_rsrc(p.dy[t], (unsigned)((long)p.R * p.W * p.Cout * 4));
      const unsigned off = (unsigned)(((long)r * p.W + col) * p.Cout + co0 + dsub * 16) * 4u;
#pragma unroll
      for (int v = 0; v < 4; ++v) rd[v] = load16(rs, ok ? off + ((16u * v) ^ (2u * dflip)) : OOB);
    }
  };
  auto issue_x = [&](int gidx, int col) {
    const int t = gidx / p.G, gr = gidx - t * p.G;
    const rsrc_t rs = make_rsrc(first ? p.x0[t] : p.x1[t], (unsigned)((long)p.R * p.W * Cs * 4));
    const int r0 = gr * 32 - 1;
    // the halo rows belong to the group's image unless the group starts / ends it
    const bool top_ok = (gr * 32) % p.H != 0, bot_ok = (gr * 32 + 32) % p.H != 0;
    {
      const int r = r0 + xi0;
      const bool ok = (r >= 0) & (r < p.R) & x_ch_ok & (xi0 != 0 || top_ok);
      const unsigned off = (unsigned)(((long)r * p.W + col) * Cs + cl0 + ssub * 8) * 4u;
      rx[0] = load16(rs, ok ? off : OOB);
      rx[1] = load16(rs, ok ? off + 16u : OOB);
    }
    if (x_two) {  // rows 32 and 33
      const int r = r0 + xi1;
      const bool ok = (r < p.R) & x_ch_ok & (xi1 != 33 || bot_ok);
      const unsigned off = (unsigned)(((long)r * p.W + col) * Cs + cl0 + ssub * 8) * 4u;
      rx[2] = load16(rs, ok ? off : OOB);
      rx[3] = load16(rs, ok ? off + 16u : OOB);
    }
  };
  auto store_dy = [&](int buf) {
    if (!dy_role) return;
    u32x4 q0[2], q1[2];
    split8h(rd[0], rd[1], sd, q0);
    split8h(rd[2], rd[3], sd, q1);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      unsigned char* d = lds_raw + buf * DYB + part * 8192 + dy_lds;
      *reinterpret_cast<u32x4*>(d + dflip) = q0[part];
      *reinterpret_cast<u32x4*>(d + (dflip ^ 16u)) = q1[part];
    }
  };
  auto store_x = [&](int slot) {
    u32x4 q[2];
    split8h(rx[0], rx[1], sx, q);
    *reinterpret_cast<u32x4*>(lds_raw + slot * XSLOT + x_lds0) = q[0];
    *reinterpret_cast<u32x4*>(lds_raw + slot * XSLOT + (x_lds0 ^ 128)) = q[1];
    if (x_two) {
      split8h(rx[2], rx[3], sx, q);
      *reinterpret_cast<u32x4*>(lds_raw + slot * XSLOT + x_lds1) = q[0];
      *reinterpret_cast<u32x4*>(lds_raw + slot * XSLOT + (x_lds1 ^ 128)) = q[1];
    }
  };

  // ---- fragment addresses (lane constants) ----
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
  const int fsw = wg_swz(8 * fg + fq);
  const int lane_base = (8 * fg + fq) * 256 + fp * 8;
  int offA[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) offA[t] = lane_base + ((t ^ fsw) * 32);
  // input window of vertical tap ky: staged rows ky + (8 fg + fq) and + 4, each with the swizzle of ITS row
  int offB[KS][2][2];  // [ky][part][half]
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = ky + 8 * fg + fq + 4 * h;
      const int sw = wg_swz(i);
#pragma unroll
      for (int part = 0; part < 2; ++part) offB[ky][part][h] = X_BASE + i * 256 + fp * 8 + (((part * 4 + wn) ^ sw) * 32);
    }
  typedef __fp16 h16x8 __attribute__((__vector_size__(8 * sizeof(__fp16))));
  auto frag2 = [&](int lo_addr, int hi_addr) {
    const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(RAC_LDS_PTR(h16x4, lds_raw, lo_addr));
    const h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(RAC_LDS_PTR(h16x4, lds_raw, hi_addr));
    return __builtin_bit_cast(f16x8, (h16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  f32x4 acc[KS][KS][4];
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[ky][k][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // zero what nobody stages: co 64..127 of the dy rows, rows 34..39 of the input slots
#pragma unroll
  for (int v = 0; v < 8; ++v) *reinterpret_cast<u32x4*>(lds_raw + (v * 256 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
  for (int v = tid; v < NR * 6 * 16; v += 256)
    *reinterpret_cast<u32x4*>(lds_raw + X_BASE + (v / 96) * XSLOT + XROWS * 256 + (v % 96) * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  for (int u = u_begin; u < u_end; ++u) {
    const int gidx = u / p.nseg, c0 = (u - gidx * p.nseg) * cseg, c1 = c0 + cseg;
    // ring slot of column c: (c - c0 + 1) & 3 -- the left halo column c0 - 1 in slot 0
    if (c0 > 0) {
      issue_x(gidx, c0 - 1);
      store_x(0);
    }
    issue_x(gidx, c0);
    store_x(1);
    if (c0 + 1 < p.W) {
      issue_x(gidx, c0 + 1);
      store_x(2);
    }
    issue_dy(gidx, c0);
    store_dy(0);
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
      const bool more_dy = c + 1 < c1;
      const bool more_x = more_dy && c + 2 < p.W;  // column c + 2 serves step c + 1 (its right neighbour)
      if (more_dy) issue_dy(gidx, c + 1);
      if (more_x) issue_x(gidx, c + 2);
      const int dbuf = ((c - c0) & 1) * DYB;
      f16x8 fa[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int part = 0; part < 2; ++part) fa[t][part] = tr_frag(lds_raw, dbuf + part * 8192 + offA[t]);
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        if ((unsigned)(c + k - 1) >= (unsigned)p.W) continue;  // uniform: the tap leaves the image row
        const int so = ((c - c0 + k) & (NR - 1)) * XSLOT;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          f16x8 fb[2];
#pragma unroll
          for (int part = 0; part < 2; ++part) fb[part] = frag2(so + offB[ky][part][0], so + offB[ky][part][1]);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[ky][k][t] = mma3(fa[t], fb, acc[ky][k][t]);
        }
      }
      if (more_dy) store_dy((c - c0 + 1) & 1);
      if (more_x) store_x((c - c0 + 3) & (NR - 1));
      __syncthreads();
    }
  }

  const float id = pow2f(-kd), ix = pow2f(-kxs);
  const int lr = lane & 15, lq = lane >> 4;
  float* dst = split == 0 ? p.dw : p.slabs + (long)(split - 1) * p.slab_stride;
  const bool add = split == 0 && p.accumulate;
  const int cil = cl0 + wn * 16 + lr;
  if (cil < Cs) {
    const int ci = (first ? 0 : p.C0) + cil;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + t * 16 + 4 * lq + r;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            const long idx = (((long)co * KS + ky) * KS + k) * p.Cin + ci;
            const float v = acc[ky][k][t][r] * id * ix;
            dst[idx] = add ? dst[idx] + v : v;
          }
      }
  }
}

// out[i] += sum_s slabs[s * stride + i]
__global__ void slab_accumulate_kernel(const float4* slabs, int n_slabs, long stride4, float4* out, long n4) {
  const long st = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += st) {
    float4 a = out[i];
    for (int s = 0; s < n_slabs; ++s) {
      const float4 b = slabs[s * stride4 + i];
      a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
    out[i] = a;
  }
}

// fp32 tensors of up to 16 time steps -> their two fp16 parts under ONE power-of-two scale (from the maximum over the
// given amax slots): parts[t] = [2][n] halves, h1 = fp16(x s), h2 = fp16(x s - h1).  The weight-gradient kernel reads
// every operand tile in dozens of workgroups (one per input-channel tile and kernel row); splitting once here takes
// the conversion out of all of them.
struct SplitSteps {
  const float* x[RAC_WGRAD_MAX_STEPS];
  unsigned short* parts[RAC_WGRAD_MAX_STEPS];
  const unsigned* amax[2 * RAC_WGRAD_MAX_STEPS];
  int n_amax;
  long n8;  // elements / 8 per step
};
__global__ __launch_bounds__(256) void split_steps_kernel(SplitSteps p) {
  unsigned am = 0;
  for (int i = 0; i < p.n_amax; ++i) am = max(am, *p.amax[i]);
  const float s = pow2f(scale_exp(am));
  const int t = blockIdx.y;
  const u32x4* src = reinterpret_cast<const u32x4*>(p.x[t]);
  u32x4* d0 = reinterpret_cast<u32x4*>(p.parts[t]);
  u32x4* d1 = d0 + p.n8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.n8; i += (long)gridDim.x * 256) {
    u32x4 q[2];
    split8h(src[2 * i], src[2 * i + 1], s, q);
    d0[i] = q[0];
    d1[i] = q[1];
  }
}

// the same for MANY small slabs (rac_thin_wgrad's per-workgroup partial sums): 16 lanes share the slabs of a 16-byte
// column and combine through LDS in a fixed order
__global__ __launch_bounds__(256) void slab_accumulate_many_kernel(const float4* slabs, int n_slabs, long stride4,
                                                                   float4* out, long n4) {
  __shared__ float4 sh[256];
  const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + col;
  float4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < n4)
    for (int s = lane; s < n_slabs; s += 16) {
      const float4 b = slabs[s * stride4 + i];
      a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
  sh[threadIdx.x] = a;
  __syncthreads();
  if (lane == 0 && i < n4) {
    float4 r = out[i];
    for (int k = 0; k < 16; ++k) {
      const float4 b = sh[k * 16 + col];
      r.x += b.x, r.y += b.y, r.z += b.z, r.w += b.w;
    }
    out[i] = r;
  }
}

}  // namespace rac

using namespace rac;

extern "C" int rac_absmax(const float* x0, int64_t n0, const float* x1, int64_t n1, uint32_t* amax, void* stream) {
  RAC_REQUIRE(x0 && n0 > 0 && n1 >= 0 && (n1 == 0 || x1) && amax, "rac_absmax: bad args");
  RAC_REQUIRE(n0 % 4 == 0 && n1 % 4 == 0 && aligned16(x0) && (!x1 || aligned16(x1)),
              "rac_absmax: element counts must be multiples of 4, pointers 16-byte aligned");
  long nb = ((n0 + n1) / 4 + 1023) / 1024;  // >= 4 vectors per thread
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(absmax_kernel, dim3((int)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(x0), (long)(n0 / 4), reinterpret_cast<const float4*>(x1),
                     (long)(n1 / 4), amax);
  return check_launch("rac_absmax");
}

extern "C" int rac_weight_frag_split(const float* w, const uint32_t* w_amax, uint16_t* parts, int32_t Cout, int32_t Cin,
                                     int32_t ksize, int32_t transposed, int64_t part_stride, void* stream) {
  RAC_REQUIRE(w && w_amax && parts && Cout > 0 && Cin > 0 && ksize >= 1 && (ksize & 1), "rac_weight_frag_split: bad args");
  RAC_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0, "rac_weight_frag_split: channel counts must be multiples of 32");
  const long n = (long)Cout * Cin * ksize * ksize;
  RAC_REQUIRE(part_stride >= n && part_stride % 8 == 0 && aligned16(w) && aligned16(parts),
              "rac_weight_frag_split: part stride / alignment");
  const long cells = n / 1024;  // (row tile, k chunk, tap) cells of 32 x 32 weights
  hipLaunchKernelGGL(weight_frag16_kernel, dim3((unsigned)((cells + 1) / 2)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w, w_amax, parts, Cout, Cin, ksize * ksize, transposed,
                     (long)part_stride);
  return check_launch("rac_weight_frag_split");
}

extern "C" int64_t rac_absmax_blocks(int64_t n) {
  long nb = (n / 4 + 1023) / 1024;
  return nb < 1 ? 1 : (nb > 512 ? 512 : nb);
}

extern "C" int64_t rac_weight_frag_blocks(int32_t Cout, int32_t Cin, int32_t ksize) {
  return ((long)Cout * Cin * ksize * ksize / 1024 + 1) / 2;
}

extern "C" int rac_absmax_multi(const rac_absmax_job* jobs, int32_t n_jobs, int64_t total_blocks, void* stream) {
  RAC_REQUIRE(jobs && n_jobs > 0 && total_blocks >= n_jobs && total_blocks < 0x7FFFFFFFL, "rac_absmax_multi: bad args");
  hipLaunchKernelGGL(absmax_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), jobs, n_jobs);
  return check_launch("rac_absmax_multi");
}

extern "C" int rac_weight_frag_split_multi(const rac_frag_job* jobs, int32_t n_jobs, int64_t total_blocks, void* stream) {
  RAC_REQUIRE(jobs && n_jobs > 0 && total_blocks >= n_jobs && total_blocks < 0x7FFFFFFFL,
              "rac_weight_frag_split_multi: bad args");
  hipLaunchKernelGGL(weight_frag16_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), jobs, n_jobs);
  return check_launch("rac_weight_frag_split_multi");
}

extern "C" int rac_adam_frag_multi(const rac_adam_frag_job* jobs, int32_t n_jobs, int64_t total_blocks, float lr, float beta1,
                                   float beta2, float eps, int32_t step, void* stream) {
  RAC_REQUIRE(jobs && n_jobs > 0 && total_blocks >= n_jobs && total_blocks < 0x7FFFFFFFL && step >= 1,
              "rac_adam_frag_multi: bad args");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_frag_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     jobs, n_jobs, (long)total_blocks, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)));
  return check_launch("rac_adam_frag_multi");
}

extern "C" int rac_adam_frag_multi_bounded(const rac_adam_frag_job* jobs, int32_t n_jobs, int64_t total_blocks,
                                           int32_t max_workgroups, float lr, float beta1, float beta2, float eps, int32_t step,
                                           void* stream) {
  RAC_REQUIRE(jobs && n_jobs > 0 && total_blocks > 0 && total_blocks < (1L << 31) && step >= 1 && max_workgroups >= 1,
              "rac_adam_frag_multi_bounded: bad args");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const long grid = total_blocks < max_workgroups ? total_blocks : max_workgroups;
  hipLaunchKernelGGL(adam_frag_multi_kernel, dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs,
                     n_jobs, (long)total_blocks, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)));
  return check_launch("rac_adam_frag_multi_bounded");
}

extern "C" int rac_amax_bound(uint32_t* exact, uint32_t* bound, const int32_t* idx, int32_t n, float margin, void* stream) {
  RAC_REQUIRE(exact && bound && idx && n > 0 && margin >= 0.f, "rac_amax_bound: bad args");
  hipLaunchKernelGGL(amax_bound_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), exact, bound,
                     idx, n, margin);
  return check_launch("rac_amax_bound");
}

// image rows per tile of conv16_rows_kernel: R | H, R * W <= 128 and a multiple of 16 (0: none)
static int rows_tile_m(int H, int W) {
  for (int r = 128 / W; r >= 1; --r)
    if (H % r == 0 && (r * W) % 16 == 0) return r * W;
  return 0;
}

extern "C" int rac_conv2d_split_supported(int32_t H, int32_t W, int32_t ksize, int32_t Cin, int32_t Cout, int32_t a_split) {
  if (H <= 0 || W <= 0 || ksize < 3 || ksize > 5 || !(ksize & 1) || Cin % 32 || Cout % 32) return 0;
  if (a_split > 0 && a_split < Cin && a_split % 32) return 0;
  const int HW = H * W;
  if (HW <= 128) return ((128 / HW) * HW) % 16 == 0;
  const int tm = W <= 128 ? rows_tile_m(H, W) : 0;
  if (tm && (tm + 2 * (ksize / 2) * W) * 4 <= 1024) return 1;
  return ksize == 3 && W >= 64 && W % 16 == 0 && H % 8 == 0;  // 2-D tiles of the unrolled 3x3 form
}

static int conv16_launch(const rac_conv_args* a, const uint32_t* a_amax0, const uint32_t* a_amax1, int64_t w_part_stride,
                         int32_t w_cin, const uint32_t* w_amax, uint32_t* out_amax, const float* lstm_c_prev, float* lstm_h,
                         float* lstm_c, void* stream, bool pool_query = false) {
  // pool_query: nothing is launched; the return value says whether this conv can write its own 2 x 2 max-pooled output
  // (1) or not (0) -- rac_conv2d_fwd_split_pool_ok
  RAC_REQUIRE(a && a->mode == RAC_CONV_FWD, "rac_conv2d_fwd_split: forward mode only");
  RAC_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0 && a->a0 && a->w && (a->out0 || lstm_h) && a_amax0 &&
                  w_amax,
              "rac_conv2d_fwd_split: bad args");
  const int a_split = (a->a1 && a->a_split > 0 && a->a_split < a->Cin) ? a->a_split : a->Cin;
  const int n_rows = (a->Cout + 31) / 32 * 32;  // weight rows: Cout zero-padded to whole 32-column tiles
  RAC_REQUIRE(rac_conv2d_split_supported(a->H, a->W, a->ksize, a->Cin, n_rows, a_split),
              "rac_conv2d_fwd_split: shape not supported (k 3 or 5, channel counts %% 32 == 0, whole images or whole "
              "image rows per 128-pixel tile): use rac_conv2d");
  Conv16P p{};
  p.B = a->B, p.H = a->H, p.W = a->W, p.ks = a->ksize, p.pad = a->ksize / 2;
  p.Cin = a->Cin, p.Cout = a->Cout, p.act = a->act;
  p.split_k = a->split_k > 1 ? a->split_k : 1;
  p.slab_stride = a->slab_stride;
  p.a0 = a->a0, p.a1 = a->a1;
  p.w = reinterpret_cast<const unsigned short*>(a->w);
  p.w_ps = w_part_stride;
  p.a_amax0 = a_amax0, p.a_amax1 = a_amax1, p.w_amax = w_amax, p.out_amax = out_amax;
  p.out0 = a->out0;
  p.bias = a->bias, p.scale = a->scale, p.shift = a->shift, p.stats = a->stats;
  p.stats_rows = a->stats ? a->stats_rows : 0;
  p.HW = a->H * a->W;
  p.P = a->B * p.HW;
  p.M = p.P, p.N = n_rows, p.n_store = a->Cout;
  p.taps = a->ksize * a->ksize;
  p.a_split = a_split;
  p.a0_up = a->a0_up ? 1 : 0;
  p.per_image = a->amax_per_image ? 1 : 0;
  RAC_REQUIRE(!p.per_image || (p.split_k == 1 && (p.HW > 128 || p.HW % 16 == 0)),
              "rac_conv2d_fwd_split: amax_per_image needs split_k 1 and H*W a multiple of 16");
  p.lstm_c_prev = lstm_c_prev, p.lstm_h = lstm_h, p.lstm_c = lstm_c, p.lstm_g = a->Cout / 4;
  RAC_REQUIRE(!lstm_h || (lstm_c_prev && lstm_c && a->bias && p.split_k == 1 && p.HW <= 128 && a->Cout % 64 == 0),
              "rac_convlstm_cell_fwd_split: needs c_prev / h / c / bias, split_k 1, a map that fits a tile, 4g % 64 == 0");
  RAC_REQUIRE(!p.a0_up || (p.HW > 128 && a->H % 2 == 0 && a->W % 2 == 0),
              "rac_conv2d_fwd_split: a0_up needs even H, W and a map larger than a 128-pixel tile");
  RAC_REQUIRE(aligned16(a->a0) && aligned16(a->w) && (!a->a1 || aligned16(a->a1)), "rac_conv2d_fwd_split: alignment");
  RAC_REQUIRE((long)p.P * (p.a_split > a->Cin - p.a_split ? p.a_split : a->Cin - p.a_split) * 4 < 0xFFFFFF00L,
              "rac_conv2d_fwd_split: operand larger than 4 GiB");
  if (w_cin <= 0) w_cin = a->Cin;
  RAC_REQUIRE(w_cin >= a->Cin && w_cin % SBK == 0, "rac_conv2d_fwd_split: w_cin must be a multiple of 32, >= Cin");
  RAC_REQUIRE(w_part_stride >= (long)n_rows * p.taps * w_cin && 2 * w_part_stride * 2 < 0xFFFFFF00L,
              "rac_conv2d_fwd_split: weight part stride");
  RAC_REQUIRE(p.split_k == 1 || a->slab_stride >= (long)p.M * p.n_store, "rac_conv2d_fwd_split: slab_stride too small");
  RAC_REQUIRE(p.stats_rows >= 0 && (p.stats_rows == 0 || ((long)a->B * a->H * a->W) % p.stats_rows == 0),
              "rac_conv2d_fwd_split: stats_rows must divide B*H*W");
  p.cchunks = a->Cin / SBK;
  p.nchunks = p.taps * p.cchunks;
  p.w_nchunks = p.taps * (w_cin / SBK);
  p.cps = cdiv(p.nchunks, p.split_k);
  RAC_REQUIRE((long)(n_rows / 32) * p.w_nchunks * 2048L < 0xFFFFFF00L, "rac_conv2d_fwd_split: weight part too large");
  static const char* xg = getenv("RAC_XCD_GROUP");
  const bool want_xcd = xg ? atoi(xg) != 0 : true;
  if (p.HW > 128) {
    // 2-D tiles (8 image rows x 16 pixels + a one-pixel halo: 180 staged pixels per 128) for the unrolled 3x3 form on
    // maps 64 or more pixels wide: a whole-row tile stages (rows + 2) W pixels -- 264 per 128 on a 64-wide map -- and has
    // no room at all for a 128-wide one (its halo alone is 256 pixels).  RAC_ROWS_TILE2D=0: whole-row tiles only.
    const char* no2d = getenv("RAC_ROWS_TILE2D");
    const bool rows_fit = a->W <= 128 && rows_tile_m(a->H, a->W) && (rows_tile_m(a->H, a->W) + 2 * p.pad * a->W) * 4 <= 1024;
    static const int min_w2d = [] { const char* e = getenv("RAC_ROWS_TILE2D_MINW"); return e ? atoi(e) : 64; }();
    bool tile2d = a->ksize == 3 && a->W >= min_w2d && a->W % 16 == 0 && a->H % 8 == 0 && !(no2d && atoi(no2d) == 0 && rows_fit);
    if (tile2d && p.cps % 9 != 0) {
      if (rows_fit)
        tile2d = false;  // a K split that cuts a channel chunk: the generic loop on whole-row tiles
      else
        p.cps = cdiv(p.cps, 9) * 9;  // (only whole chunks per split; the last splits may be short or empty)
    }
    RAC_REQUIRE(tile2d || rows_fit, "rac_conv2d_fwd_split: this map needs 2-D tiles (3x3, W %% 16 == 0, H %% 8 == 0)");
    if (tile2d) {
      p.seg_w = 16, p.seg_h = 8;
      p.seg_tx = a->W / 16, p.seg_tpi = (a->H / 8) * p.seg_tx;
    }
    p.tile_m = tile2d ? 128 : rows_tile_m(a->H, a->W);
    RAC_REQUIRE(p.stats_rows % p.tile_m == 0, "rac_conv2d_fwd_split: stats_rows must be a multiple of the tile rows");
    const int nrows = tile2d ? 10 * 18 : p.tile_m + 2 * p.pad * a->W;
    const int nv = cdiv(nrows * 4, 256) < 2 ? 2 : cdiv(nrows * 4, 256);
    RAC_REQUIRE(nv <= 4, "rac_conv2d_fwd_split: halo too large for the LDS image");
    typedef void (*rows_fn)(Conv16P);
    static const rows_fn fns[3][3] = {
        {conv16_rows_kernel<2, 2, 2, 2>, conv16_rows_kernel<3, 2, 2, 2>, conv16_rows_kernel<4, 2, 2, 2>},   // 128 columns
        {conv16_rows_kernel<2, 2, 1, 2>, conv16_rows_kernel<3, 2, 1, 2>, conv16_rows_kernel<4, 2, 1, 2>},   // 64 columns
        {conv16_rows_kernel<2, 4, 1, 2>, conv16_rows_kernel<3, 4, 1, 2>, conv16_rows_kernel<4, 4, 1, 2>}};  // 32 columns
    // 3 x 3, full tiles, chunk-aligned K ranges: the unrolled-tap form
    static const rows_fn fast_fns[3][3] = {
        {conv16_rows_kernel<2, 2, 2, 3, true>, conv16_rows_kernel<3, 2, 2, 3, true>, conv16_rows_kernel<4, 2, 2, 3, true>},
        {conv16_rows_kernel<2, 2, 1, 3, true>, conv16_rows_kernel<3, 2, 1, 3, true>, conv16_rows_kernel<4, 2, 1, 3, true>},
        {conv16_rows_kernel<2, 4, 1, 3, true>, conv16_rows_kernel<3, 4, 1, 3, true>, conv16_rows_kernel<4, 4, 1, 3, true>}};
    // 5 x 5 (the ConvLSTM gate convs on 16x16 latents: BASELINE configs[4]), 128-column workgroups: the same unrolled form
    static const rows_fn fast5_fns[3] = {conv16_rows_kernel<2, 2, 2, 3, true, 5>, conv16_rows_kernel<3, 2, 2, 3, true, 5>,
                                         conv16_rows_kernel<4, 2, 2, 3, true, 5>};
    size_t lds_rows = (size_t)2 * 2 * 4 * (nrows + 16) * 16;
    static bool rows_attr = false;
    if (!rows_attr) {
      for (int i = 0; i < 21; ++i) {
        const rows_fn f = i < 9 ? fns[i / 3][i % 3] : (i < 18 ? fast_fns[(i - 9) / 3][i % 3] : fast5_fns[i - 18]);
        if (!f) continue;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(f),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * 2 * 4 * (256 + 16) * 16);
        if (e != hipSuccess) {
          set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
          return RAC_ELAUNCH;
        }
      }
      rows_attr = true;
    }
    const int width = p.N <= 32 ? 2 : (p.N <= 64 ? 1 : 0);
    dim3 grid(cdiv(p.M, p.tile_m), cdiv(p.N, width == 2 ? 32 : (width == 1 ? 64 : 128)), p.split_k);
    p.xcd_group = want_xcd && grid.x > 1 && (grid.y * grid.z) % 8 == 0;
    static const char* nopair = getenv("RAC_ROWS_PAIR_XCD");  // (=0: hardware order for the two-column-tile layers)
    if (want_xcd && !p.xcd_group && grid.y == 2 && grid.z == 1 && grid.x >= 64 && !(nopair && atoi(nopair) == 0)) p.xcd_group = 2;
    static const char* nohalo1 = getenv("RAC_ROWS_HALO_XCD");  // (=0: hardware order for the one-column-tile layers)
    if (want_xcd && !p.xcd_group && grid.y == 1 && grid.z == 1 && grid.x >= 128 && !(nohalo1 && atoi(nohalo1) == 0))
      p.xcd_group = 6;
    static const char* nofast = getenv("RAC_ROWS_GENERIC");  // A/B switch: always the generic loop
    rows_fn fn = fns[width][nv - 2];
    const bool fast = a->ksize == 3 && p.tile_m == 128 && p.cps % 9 == 0 && fast_fns[width][nv - 2] &&
                      (tile2d || !(nofast && atoi(nofast)));
    RAC_REQUIRE(fast || !tile2d, "rac_conv2d_fwd_split: 2-D tiles need the unrolled 3x3 form");
    const char* nofast5 = getenv("RAC_ROWS_FAST5");  // (=0: the generic loop for 5x5 convs on maps larger than a tile)
    const size_t lds5 = (size_t)2 * 2 * 4 * ((((nrows / a->W) * (a->W + 4)) + 15) & ~15) * 16;
    if (a->ksize == 5 && width == 0 && p.tile_m == 128 && p.cps % 25 == 0 && !(nofast && atoi(nofast)) &&
        !(nofast5 && atoi(nofast5) == 0) && lds5 <= (size_t)2 * 2 * 4 * (256 + 16) * 16) {
      fn = fast5_fns[nv - 2];
      lds_rows = lds5;
    }
    if (fast) {
      fn = fast_fns[width][nv - 2];
      // padded rows: W + 2 LDS rows per image row (2-D tile: the halo tile itself), the plane rounded to 256 B
      lds_rows = (size_t)2 * 2 * 4 * (((tile2d ? nrows : (nrows / a->W) * (a->W + 2)) + 15) & ~15) * 16;
    }
    {
      // The 2 x 2 max pool in the epilogue (out1): the epilogue's predicate-free path must be the one every wave takes
      // (full tiles, full column blocks, no statistics, offsets inside 4 GiB), a 16-row block must lie in one image row
      // with the block under it in the same wave (blocks per tile row 1 or 2, an even number of tile rows per wave), and the
      // tile's first image row must be even.
      const int bpr = (tile2d ? 16 : a->W) / 16, mb_wave = width == 2 ? 2 : 4, bnw = width == 2 ? 32 : (width == 1 ? 64 : 128);
      const bool pool_ok = fast && p.split_k == 1 && !a->stats && !lstm_h && a->H % 2 == 0 && a->W % 16 == 0 &&
                           (bpr == 1 || bpr == 2) && mb_wave % (2 * bpr) == 0 && (p.tile_m / (16 * bpr)) % 2 == 0 &&
                           p.M % p.tile_m == 0 && a->Cout % bnw == 0 && (long)p.M * a->Cout * 4 < (1l << 32) && RAC_EPILOGUE_FAST;
      if (pool_query) return pool_ok ? 1 : 0;
      if (a->out1) {
        RAC_REQUIRE(pool_ok, "rac_conv2d_fwd_split: this conv cannot pool in its epilogue (rac_conv2d_fwd_split_pool_ok)");
        p.pool_out = a->out1, p.pool_bpr = bpr;
      }
    }
    // narrow layers (64 / 32 columns), unsplit K, many more tiles than the chip holds: persistent workgroups that walk
    // the tiles and request the next tile's first chunk under the current tile's last one
    static const rows_fn persist_fns[2][3] = {
        {conv16_rows_persist_kernel<2, 2, 1>, conv16_rows_persist_kernel<3, 2, 1>, conv16_rows_persist_kernel<4, 2, 1>},
        {conv16_rows_persist_kernel<2, 4, 1>, conv16_rows_persist_kernel<3, 4, 1>, conv16_rows_persist_kernel<4, 4, 1>}};
    static const char* nopersist = getenv("RAC_ROWS_PERSIST");
    static const int persist_wgs = [] { const char* e = getenv("RAC_ROWS_PERSIST_WGS"); return e ? atoi(e) : 512; }();
    // (a static split of few tiles per workgroup loses to the hardware's dynamic one: measured +11 % on the 8 000-tile 32x32
    // layers, -10 % on the 32 000-tile 64x64 ones)
    if (fast && width >= 1 && p.split_k == 1 && !(nopersist && atoi(nopersist) == 0) &&
        ((int)grid.x >= 32 * persist_wgs || ((int)grid.x >= 4 * persist_wgs && (int)grid.x % persist_wgs == 0))) {
      static bool persist_attr = false;
      if (!persist_attr) {
        for (int i = 0; i < 6; ++i) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(persist_fns[i / 3][i % 3]),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 4 * (256 + 16) * 16);
          if (e != hipSuccess) {
            set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
            return RAC_ELAUNCH;
          }
        }
        persist_attr = true;
      }
      static const char* nohalo = getenv("RAC_PERSIST_HALO_XCD");  // (=0: tile = workgroup id)
      p.xcd_group = (want_xcd && persist_wgs % 8 == 0 && !(nohalo && atoi(nohalo) == 0)) ? 5 : 0;
      hipLaunchKernelGGL(persist_fns[width - 1][nv - 2], dim3(persist_wgs, grid.y, 1), dim3(256), lds_rows,
                         reinterpret_cast<hipStream_t>(stream), p);
      return check_launch("rac_conv2d_fwd_split(image rows, persistent)");
    }
    hipLaunchKernelGGL(fn, grid, dim3(256), lds_rows, reinterpret_cast<hipStream_t>(stream), p);
    return check_launch("rac_conv2d_fwd_split(image rows)");
  }
  if (pool_query) return 0;
  RAC_REQUIRE(!a->out1, "rac_conv2d_fwd_split: maps that fit a tile do not pool in the epilogue");
  p.tile_m = (128 / p.HW) * p.HW;  // whole images per workgroup
  RAC_REQUIRE(p.stats_rows % p.tile_m == 0, "rac_conv2d_fwd_split: stats_rows must be a multiple of the tile rows");
  dim3 grid(cdiv(p.M, p.tile_m), cdiv(p.N, SBN), p.split_k);
  p.xcd_group = want_xcd && grid.x > 1 && (grid.y * grid.z) % 8 == 0;
  // many M-tiles, K unsplit: the workgroups that share an activation tile run back to back on one XCD (RAC_TILE_PAIR_XCD=0:
  // the weight-slab grouping above / hardware order)
  static const char* pairn = getenv("RAC_TILE_PAIR_XCD");
  const bool pair_ok = want_xcd && grid.z == 1 && grid.x >= 64 && !(pairn && atoi(pairn) == 0);
  if (pair_ok && p.xcd_group && grid.y % 8 == 0 && grid.y > 8) p.xcd_group = 3;
  // (Layers with 2 or 4 column tiles keep the hardware order: running an M-tile's column tiles back to back on one XCD
  // puts all their weight slabs -- 4 x 2.4 MB for the planner's 8x8 vgg layers -- through one 4 MB L2 at once: measured
  // 0.402 -> 0.433 ms on 256 -> 512.)
  size_t lds_tile = 2 * T16_ABUF;  // 36,864 B
  // waves 2 x 2 (template argument 2); the 1 x 4 arrangement of the same kernel measured 10 % slower
  static const char* noym = getenv("RAC_TILE_YMAJOR");
  const bool ym = a->W == 8 && p.tile_m == 2 * p.HW && !(noym && atoi(noym) == 0);  // 8x8 / 6x8 maps: skip vertical padding
  typedef void (*tile_fn)(Conv16P);
  // 3x3 convs whose K ranges are whole chunks: the unrolled-tap instance (RAC_TILE_UNROLL3=0: the generic loop)
  static const char* nou3 = getenv("RAC_TILE_UNROLL3");
  const bool u3 = a->ksize == 3 && p.cps % 9 == 0 && !(nou3 && atoi(nou3) == 0);
  static const char* nou5 = getenv("RAC_TILE_UNROLL5");
  const bool u5 = a->ksize == 5 && p.cps % 25 == 0 && !(nou5 && atoi(nou5) == 0);
  const tile_fn fn =
      u5 ? (p.tile_m == 128 ? (ym ? (tile_fn)conv16_tile_kernel<2, true, true, 5> : (tile_fn)conv16_tile_kernel<2, true, false, 5>)
                            : (ym ? (tile_fn)conv16_tile_kernel<2, false, true, 5> : (tile_fn)conv16_tile_kernel<2, false, false, 5>)) :
      u3 ? (p.tile_m == 128 ? (ym ? (tile_fn)conv16_tile_kernel<2, true, true, 3> : (tile_fn)conv16_tile_kernel<2, true, false, 3>)
                            : (ym ? (tile_fn)conv16_tile_kernel<2, false, true, 3> : (tile_fn)conv16_tile_kernel<2, false, false, 3>))
         : (p.tile_m == 128 ? (ym ? (tile_fn)conv16_tile_kernel<2, true, true> : (tile_fn)conv16_tile_kernel<2, true>)
                            : (ym ? (tile_fn)conv16_tile_kernel<2, false, true> : (tile_fn)conv16_tile_kernel<2, false>));
  if (ym) {
    lds_tile = 2 * T16Y_ABUF;  // 73,728 B (+ 512 B static): two workgroups per CU
    static bool ym_attr = false;
    if (!ym_attr) {
      for (tile_fn f : {(tile_fn)conv16_tile_kernel<2, true, true>, (tile_fn)conv16_tile_kernel<2, false, true>,
                        (tile_fn)conv16_tile_kernel<2, true, true, 3>, (tile_fn)conv16_tile_kernel<2, false, true, 3>,
                        (tile_fn)conv16_tile_kernel<2, true, true, 5>, (tile_fn)conv16_tile_kernel<2, false, true, 5>}) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * T16Y_ABUF);
        if (e != hipSuccess) {
          set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
          return RAC_ELAUNCH;
        }
      }
      ym_attr = true;
    }
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), lds_tile, reinterpret_cast<hipStream_t>(stream), p);
  return check_launch("rac_conv2d_fwd_split(whole images)");
}

extern "C" int rac_conv2d_fwd_split(const rac_conv_args* a, const uint32_t* a_amax0, const uint32_t* a_amax1,
                                    int64_t w_part_stride, int32_t w_cin, const uint32_t* w_amax, uint32_t* out_amax,
                                    void* stream) {
  return conv16_launch(a, a_amax0, a_amax1, w_part_stride, w_cin, w_amax, out_amax, nullptr, nullptr, nullptr, stream);
}

extern "C" int rac_conv2d_fwd_split_pool_ok(const rac_conv_args* a, int32_t w_cin) {
  static const uint32_t one = 0x3F800000u;  // (argument checks only: nothing is read or launched)
  if (!a || a->mode != RAC_CONV_FWD || !a->a0 || !a->w || !a->out0) return 0;
  const int n_rows = (a->Cout + 31) / 32 * 32;
  const long taps = (long)a->ksize * a->ksize;
  const int r = conv16_launch(a, &one, nullptr, (int64_t)n_rows * taps * (w_cin > 0 ? w_cin : a->Cin), w_cin, &one, nullptr,
                              nullptr, nullptr, nullptr, nullptr, true);
  return r == 1 ? 1 : 0;
}

extern "C" int rac_convlstm_cell_fwd_split(const rac_conv_args* a, const uint32_t* a_amax0, const uint32_t* a_amax1,
                                           int64_t w_part_stride, int32_t w_cin, const uint32_t* w_amax, const float* c_prev,
                                           float* h_out, float* c_out, void* stream) {
  RAC_REQUIRE(c_prev && h_out && c_out, "rac_convlstm_cell_fwd_split: null state pointer");
  return conv16_launch(a, a_amax0, a_amax1, w_part_stride, w_cin, w_amax, nullptr, c_prev, h_out, c_out, stream);
}

extern "C" int rac_first_layer_fwd_split(const float* img, const float* zmask, const float* mask, int32_t Cm, const float* w,
                                         const float* scale, const float* shift, int32_t act, float* out,
                                         uint32_t* out_amax, int32_t amax_per_image, int32_t B, int32_t H, int32_t W,
                                         int32_t Cout, void* stream) {
  RAC_REQUIRE(img && w && out && B > 0 && H > 0 && W > 0 && Cm >= 0 && Cm <= 5 && (Cm == 0 || mask),
              "rac_first_layer_fwd_split: bad args (3 image planes + at most 5 mask / heatmap planes)");
  RAC_REQUIRE(Cout == 64 && H % 16 == 0 && W % 16 == 0 && (scale == nullptr) == (shift == nullptr) && aligned16(out) &&
                  (act == RAC_ACT_NONE || act == RAC_ACT_LEAKY02),
              "rac_first_layer_fwd_split: Cout 64, H and W multiples of 16, act none / leaky");
  const long n_tiles = (long)B * (H / 16) * (W / 16);
  RAC_REQUIRE(n_tiles < (1L << 31), "rac_first_layer_fwd_split: too many tiles");
  const int grid = (int)(n_tiles < 1024 ? n_tiles : 1024);  // persistent: the weight fragments are built once per workgroup
  typedef void (*fn_t)(const float*, const float*, const float*, const float*, const float*, const float*, int, float*,
                       unsigned*, int, int, int, int);
  static const fn_t fns[6] = {first16_kernel<3>, first16_kernel<4>, first16_kernel<5>,
                              first16_kernel<6>, first16_kernel<7>, first16_kernel<8>};
  hipLaunchKernelGGL(fns[Cm], dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img, zmask, mask, w, scale,
                     shift, act, out, out_amax, amax_per_image ? 1 : 0, H, W, (int)n_tiles);
  return check_launch("rac_first_layer_fwd_split");
}

extern "C" int rac_head_fwd_split(const float* x, const uint32_t* x_amax, int32_t amax_per_image, const float* w_taps,
                                  const float* bias, float* y, int32_t B, int32_t H, int32_t W, void* stream) {
  RAC_REQUIRE(x && x_amax && w_taps && bias && y && B > 0 && H > 0 && W > 0, "rac_head_fwd_split: bad args");
  RAC_REQUIRE(H % 16 == 0 && W % 16 == 0 && aligned16(x) && aligned16(y) && aligned16(bias),
              "rac_head_fwd_split: H % 16 == 0, W % 16 == 0, 16-byte aligned buffers");
  RAC_REQUIRE((long)H * W * 64 * 4 < (1L << 32), "rac_head_fwd_split: image too large for one buffer descriptor");
  const long n_tiles = (long)B * (H / 16) * (W / 16);
  RAC_REQUIRE(n_tiles < (1L << 31), "rac_head_fwd_split: too many tiles");
  const int grid = (int)(n_tiles < 768 ? n_tiles : 768);  // persistent: the weight fragments are built once per workgroup
  hipLaunchKernelGGL(head16_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, x_amax,
                     amax_per_image, w_taps, bias, y, H, W, (int)n_tiles);
  return check_launch("rac_head_fwd_split");
}

extern "C" int rac_split_steps(const float* const* xs, uint16_t* const* parts, int32_t T, int64_t n,
                               const uint32_t* const* amax, int32_t n_amax, void* stream) {
  RAC_REQUIRE(xs && parts && amax && T >= 1 && T <= RAC_WGRAD_MAX_STEPS && n > 0 && n % 8 == 0 && n_amax >= 1 &&
                  n_amax <= 2 * RAC_WGRAD_MAX_STEPS,
              "rac_split_steps: bad args (n %% 8 == 0, 1 <= T <= 16, 1 <= n_amax <= 32)");
  SplitSteps p{};
  for (int t = 0; t < T; ++t) {
    RAC_REQUIRE(xs[t] && parts[t] && aligned16(xs[t]) && aligned16(parts[t]), "rac_split_steps: null / unaligned step");
    p.x[t] = xs[t], p.parts[t] = parts[t];
  }
  for (int i = 0; i < n_amax; ++i) {
    RAC_REQUIRE(amax[i], "rac_split_steps: null amax slot");
    p.amax[i] = amax[i];
  }
  p.n_amax = n_amax, p.n8 = n / 8;
  long nb = (p.n8 + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(split_steps_kernel, dim3((unsigned)nb, T), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
  return check_launch("rac_split_steps");
}

extern "C" int rac_slab_accumulate(const float* slabs, int32_t n_slabs, int64_t slab_stride, float* out, int64_t n,
                                   void* stream) {
  RAC_REQUIRE(slabs && out && n_slabs >= 1 && n > 0 && slab_stride >= n, "rac_slab_accumulate: bad args");
  RAC_REQUIRE(n % 4 == 0 && slab_stride % 4 == 0 && aligned16(slabs) && aligned16(out), "rac_slab_accumulate: alignment");
  long nb = (n / 4 + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (n_slabs >= 64 && n <= (1 << 20)) {
    hipLaunchKernelGGL(slab_accumulate_many_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(slabs), n_slabs,
                       (long)(slab_stride / 4), reinterpret_cast<float4*>(out), (long)(n / 4));
    return check_launch("rac_slab_accumulate");
  }
  hipLaunchKernelGGL(slab_accumulate_kernel, dim3((int)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(slabs), n_slabs, (long)(slab_stride / 4),
                     reinterpret_cast<float4*>(out), (long)(n / 4));
  return check_launch("rac_slab_accumulate");
}

extern "C" int rac_conv2d_wgrad_split(const rac_wgrad_args* a, void* stream) {
  RAC_REQUIRE(a && a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0 && a->dw, "rac_conv2d_wgrad_split: bad args");
  RAC_REQUIRE(a->ksize == 3 || a->ksize == 5, "rac_conv2d_wgrad_split: ksize must be 3 or 5");
  RAC_REQUIRE(a->T >= 1 && a->T <= RAC_WGRAD_MAX_STEPS, "rac_conv2d_wgrad_split: 1 <= T <= RAC_WGRAD_MAX_STEPS");
  Wgrad16P p{};
  p.Bimg = a->B, p.H = a->H, p.W = a->W, p.Cin = a->Cin, p.Cout = a->Cout, p.T = a->T;
  bool two = false;
  for (int t = 0; t < a->T; ++t) {
    RAC_REQUIRE(a->dy[t] && a->x0[t] && a->dy_amax[t] && a->x0_amax[t], "rac_conv2d_wgrad_split: null operand");
    RAC_REQUIRE(aligned16(a->dy[t]) && aligned16(a->x0[t]) && (!a->x1[t] || aligned16(a->x1[t])),
                "rac_conv2d_wgrad_split: alignment");
    RAC_REQUIRE(!a->x1[t] || a->x1_amax[t], "rac_conv2d_wgrad_split: x1 needs its amax slot");
    RAC_REQUIRE(t == 0 || (a->x1[t] != nullptr) == two, "rac_conv2d_wgrad_split: every step needs the same sources");
    two = a->x1[t] != nullptr;
    p.dy[t] = a->dy[t], p.x0[t] = a->x0[t], p.x1[t] = a->x1[t];
    p.dy_amax[t] = a->dy_amax[t], p.x0_amax[t] = a->x0_amax[t], p.x1_amax[t] = a->x1[t] ? a->x1_amax[t] : nullptr;
  }
  p.C0 = (two && a->a_split > 0 && a->a_split < a->Cin) ? a->a_split : a->Cin;
  RAC_REQUIRE(two == (p.C0 < a->Cin), "rac_conv2d_wgrad_split: a_split must split Cin exactly when x1 is given");
  RAC_REQUIRE(a->Cout % 16 == 0 && p.C0 % 8 == 0 && (a->Cin - p.C0) % 8 == 0,
              "rac_conv2d_wgrad_split: Cout % 16 == 0 and input channel counts % 8 == 0");
  RAC_REQUIRE(!two || p.C0 % 64 == 0, "rac_conv2d_wgrad_split: a_split must be a multiple of 64");
  p.R = a->B * a->H;
  p.G = cdiv(p.R, 32);
  const long rowbytes = (long)p.R * a->W * 4;
  RAC_REQUIRE(rowbytes * a->Cout < 0xFFFFFF00L && rowbytes * (p.C0 > a->Cin - p.C0 ? p.C0 : a->Cin - p.C0) < 0xFFFFFF00L,
              "rac_conv2d_wgrad_split: operand larger than 4 GiB");
  p.nsplit = a->nsplit >= 1 ? a->nsplit : 1;
  RAC_REQUIRE(p.nsplit <= a->T * p.G * (a->all_ky && a->col_segments > 1 ? a->col_segments : 1) && p.nsplit <= 1024,
              "rac_conv2d_wgrad_split: more K splits than 32-row groups (x column segments)");
  const long n = (long)a->Cout * a->ksize * a->ksize * a->Cin;
  RAC_REQUIRE(p.nsplit == 1 || (a->slabs && a->slab_stride >= n), "rac_conv2d_wgrad_split: slabs for the K split");
  p.accumulate = a->accumulate;
  p.dw = a->dw, p.slabs = a->slabs, p.slab_stride = a->slab_stride;
  RAC_REQUIRE(a->x1_zero_steps >= 0 && a->x1_zero_steps <= a->T && (two || a->x1_zero_steps == 0),
              "rac_conv2d_wgrad_split: x1_zero_steps");
  p.x1_skip = a->x1_zero_steps;
  p.presplit = a->presplit ? 1 : 0;
  RAC_REQUIRE(!p.presplit || (a->Cout % 16 == 0 && p.C0 % 8 == 0), "rac_conv2d_wgrad_split: presplit operand alignment");
  // thin 3x3 layers on large maps: all nine taps per workgroup, dy and x fetched once (wgrad16_allky_kernel)
  static const char* noallky = getenv("RAC_WGRAD_ALLKY");
  if (a->ksize == 3 && a->Cout <= 128 && a->H % 32 == 0 && !p.presplit && p.x1_skip == 0 && a->all_ky &&
      !(noallky && atoi(noallky) == 0)) {
    p.nseg = a->col_segments > 1 ? a->col_segments : 1;
    RAC_REQUIRE(a->W % p.nseg == 0 && p.nsplit <= a->T * p.G * p.nseg,
                "rac_conv2d_wgrad_split: col_segments must divide W; nsplit <= T * groups * col_segments");
    dim3 grid(cdiv(a->Cout, 64), cdiv(a->Cin - p.C0, 64) + cdiv(p.C0, 64), p.nsplit);
    const int lds = 2 * 16384 + 4 * 40 * 256;
    static bool attr_allky = false;
    if (!attr_allky) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16_allky_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
        return RAC_ELAUNCH;
      }
      attr_allky = true;
    }
    hipLaunchKernelGGL(wgrad16_allky_kernel, grid, dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
    return check_launch("rac_conv2d_wgrad_split(all taps)");
  }
  const bool co64 = a->Cout <= 64;  // 64 co x 64 ci workgroups: no all-zero half of the 128-co tile
  const int ct = cdiv(a->Cout, co64 ? 64 : 128), nt = cdiv(a->Cin - p.C0, 64) + cdiv(p.C0, 64);
  dim3 grid(ct, nt, a->ksize * p.nsplit);
  typedef void (*fn_t)(Wgrad16P);
  static const fn_t fns[8] = {(fn_t)wgrad16_kernel<3, 2, false>, (fn_t)wgrad16_kernel<3, 1, false>,
                              (fn_t)wgrad16_kernel<5, 2, false>, (fn_t)wgrad16_kernel<5, 1, false>,
                              (fn_t)wgrad16_kernel<3, 2, true>,  (fn_t)wgrad16_kernel<3, 1, true>,
                              (fn_t)wgrad16_kernel<5, 2, true>,  (fn_t)wgrad16_kernel<5, 1, true>};
  const int fi = p.presplit * 4 + (a->ksize == 5) * 2 + co64;
  const fn_t fn = fns[fi];
  // (experiment, RAC_WGRAD_LDS_MIN=<bytes>: a dynamic-LDS request of at least that size -- above 80 KB only ONE workgroup
  // fits a CU, which leaves half of every SIMD's slots to the kernels of another stream)
  static const int lds_min = [] { const char* e = getenv("RAC_WGRAD_LDS_MIN"); return e ? atoi(e) : 0; }();
  const int lds_need = 2 * 16384 + (a->ksize + 1) * 8192;
  const int lds = lds_need > lds_min ? lds_need : lds_min;
  static bool attr_done[8] = {false, false, false, false, false, false, false, false};
  if (!attr_done[fi]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return RAC_ELAUNCH;
    }
    attr_done[fi] = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
  return check_launch("rac_conv2d_wgrad_split");
}

RAC_DEVICE_CODE_END
