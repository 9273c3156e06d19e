// Split-precision implicit-GEMM convolution (forward only) for the frozen-model gate GEMMs of the CEM
// rollouts: operands arrive as three bf16 parts (x = p1 + p2 + p3 exactly), six part-products run on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Same tiling as rac_igemm.hip's fast path (128x128 tile,
// 4 waves x 2x2 accumulators, K chunks of 32, branch-free buffer loads with hardware zero fill), but one
// LDS buffer per workgroup (48 KB: three parts of both operands, XOR-swizzled 64-byte rows) so that two to
// three workgroups share a CU and cover each other's barrier phases; the next chunk's 12 loads per lane are
// in flight while the current chunk's 48 MFMAs run.
#include <stdlib.h>

#include "rac_common.h"

namespace rac {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct SplitP {
  int B, H, W, ks, pad, Cin, Cout, act, split_k, a_split;
  long slab_stride;
  const unsigned short *a0, *a1, *w;
  long a0_ps, a1_ps, w_ps;  // part strides in elements
  float* out0;
  const float *bias, *scale, *shift;
  double* stats;
  long stats_rows;  // rows per statistics group (0: one group)
  int M, N, HW, P, taps, cchunks, nchunks, cps;
  int w_chunk_major;  // weights stored [Cout][Cin/32][taps][32] (tap-inner streaming order) instead of [Cout][taps][Cin]
  int xcd_group;      // remap workgroup ids so that the M-tiles sharing one weight slab run on one XCD (one L2)
  int tile_m;         // igemm_split_bdirect16_kernel: output rows per workgroup (whole images, multiple of 16, <= 128)
};

constexpr unsigned OOBS = 0xFFFFFFF0u;
constexpr int SBM = 128, SBN = 128, SBK = 32;
// igemm_split_bdirect_kernel's LDS image: 80-byte rows, 129 rows per part plane (row 128 = zeros), 3 planes per buffer
constexpr int BD_ROW = 80, BD_PLANE = 129 * BD_ROW, BD_ABUF = 3 * BD_PLANE;

__device__ __forceinline__ const void* uniform_vptr(const void* p) {
  unsigned long long v = reinterpret_cast<unsigned long long>(p);
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ rsrc_t mk_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_vptr(p)), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 ld16(rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}

__device__ __forceinline__ void split3(float a, unsigned short& q1, unsigned short& q2, unsigned short& q3) {
  const __bf16 p1 = (__bf16)a;
  const float r1 = a - (float)p1;
  const __bf16 p2 = (__bf16)r1;
  const __bf16 p3 = (__bf16)(r1 - (float)p2);
  q1 = __builtin_bit_cast(unsigned short, p1);
  q2 = __builtin_bit_cast(unsigned short, p2);
  q3 = __builtin_bit_cast(unsigned short, p3);
}

// eight fp32 values (two 16-byte vectors) -> their three bf16 parts, packed as the 16-byte operand vectors
__device__ __forceinline__ void split8(u32x4 lo, u32x4 hi, u32x4 (&q)[3]) {
  const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  unsigned short a[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(__builtin_bit_cast(float, w[j]), a[0][j], a[1][j], a[2][j]);
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    u32x4 o;
    o.x = a[part][0] | ((unsigned)a[part][1] << 16);
    o.y = a[part][2] | ((unsigned)a[part][3] << 16);
    o.z = a[part][4] | ((unsigned)a[part][5] << 16);
    o.w = a[part][6] | ((unsigned)a[part][7] << 16);
    q[part] = o;
  }
}

// LDS image of one operand: [part 3][row 128][4 chunks of 16 B], chunk index XOR-swizzled by (row >> 2) & 3 so that
// the 16 lanes of a ds_read_b128 group (16 consecutive rows, same logical chunk) hit 16 distinct 4-bank slots.
__device__ __forceinline__ int lds_off(int part, int row, int chunk) {  // in 16-byte units
  return (part * 128 + row) * 4 + (chunk ^ ((row >> 2) & 3));
}


// The 48 MFMAs of one 32-deep K chunk for a wave's 2x2 accumulators: both operands from the swizzled LDS image.
__device__ __forceinline__ void split_mma_chunk(const u32x4* As, const u32x4* Bs, f32x16 (&acc)[2][2], int wm, int wn,
                                                int li, int lh) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 fa[2][3], fb[2][3];
    const int chunk = 2 * s + lh;  // MFMA k = 16 s + 8 h + j
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        fa[t][part] = __builtin_bit_cast(bf16x8, As[lds_off(part, (wm * 2 + t) * 32 + li, chunk)]);
        fb[t][part] = __builtin_bit_cast(bf16x8, Bs[lds_off(part, (wn * 2 + t) * 32 + li, chunk)]);
      }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x16 c = acc[mt][nt];
        // smallest terms first: (3,1) (2,2) (1,3) | (2,1) (1,2) | (1,1)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][2], fb[nt][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[nt][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[nt][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[nt][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[nt][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[nt][0], c, 0, 0, 0);
        acc[mt][nt] = c;
      }
  }
}

__global__ __launch_bounds__(256, 2) void igemm_split_kernel(SplitP p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];  // [2 operands][3][128][4] x 16 B = 48 KB
  u32x4* As = lds;
  u32x4* Bs = lds + 3 * 128 * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.x * SBM, n0 = blockIdx.y * SBN;
  const int kc_begin = blockIdx.z * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);

  // staging map: lane handles rows {tid/4, tid/4 + 64} x chunk (tid & 3) x 3 parts, for A and for B
  const int srow = tid >> 2, schunk = tid & 3;
  int a_pix[2], a_y[2], a_x[2], b_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + srow + 64 * i;
    if (m < p.M) {
      int b = m / p.HW;
      int r = m - b * p.HW;
      int y = r / p.W;
      a_pix[i] = m, a_y[i] = y, a_x[i] = r - y * p.W;
    } else {
      a_pix[i] = 0, a_y[i] = -100000, a_x[i] = 0;
    }
    int n = n0 + srow + 64 * i;
    b_row[i] = (n < p.N) ? n * p.taps * p.Cin : -1;
  }
  const rsrc_t w_rsrc = mk_rsrc(p.w, (unsigned)(3 * p.w_ps * 2));

  u32x4 ra[6], rb[6];  // [part][row]
  // byte offsets of one chunk's loads (part 0; parts 1, 2 add the part stride) and the A-side descriptor
  auto offsets = [&](int kc, unsigned (&oa)[2], unsigned (&ob)[2], rsrc_t& a_rsrc, unsigned& a_pstride) {
    const bool live = kc < kc_end;
    const int tap = kc / p.cchunks;
    const int cc = kc - tap * p.cchunks;
    const int ky = tap / p.ks, kx = tap - ky * p.ks;
    const int dy = ky - p.pad, dx = kx - p.pad;
    const int c0 = cc * SBK;
    const bool first = c0 < p.a_split;
    const int Cs = first ? p.a_split : p.Cin - p.a_split;
    const int cl = (first ? c0 : c0 - p.a_split) + schunk * 8;
    const long aps = first ? p.a0_ps : p.a1_ps;
    a_rsrc = mk_rsrc(first ? p.a0 : p.a1, (unsigned)(3 * aps * 2));
    a_pstride = (unsigned)(aps * 2);
    const int shift = dy * p.W + dx;
    const int s0 = tap * p.Cin + c0 + schunk * 8;
    const bool cok = live & (c0 + schunk * 8 < p.Cin);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int yy = a_y[i] + dy, xx = a_x[i] + dx;
      bool ok = live & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W) & (cl < Cs);
      oa[i] = ok ? (unsigned)((a_pix[i] + shift) * Cs + cl) * 2u : OOBS;
      ob[i] = (cok & (b_row[i] >= 0)) ? (unsigned)(b_row[i] + s0) * 2u : OOBS;
    }
  };
  const unsigned w_pstride = (unsigned)(p.w_ps * 2);
  auto issue = [&](const unsigned (&oa)[2], const unsigned (&ob)[2], rsrc_t a_rsrc, unsigned a_pstride) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        // OOBS + part stride stays out of range: the descriptor covers 3 parts < 4 GiB, OOBS is 2^32 - 16
        ra[part * 2 + i] = ld16(a_rsrc, oa[i] == OOBS ? OOBS : oa[i] + part * a_pstride);
        rb[part * 2 + i] = ld16(w_rsrc, ob[i] == OOBS ? OOBS : ob[i] + part * w_pstride);
      }
  };
  auto store = [&]() {
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        As[lds_off(part, srow + 64 * i, schunk)] = ra[part * 2 + i];
        Bs[lds_off(part, srow + 64 * i, schunk)] = rb[part * 2 + i];
      }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kc_begin < kc_end) {
    unsigned oa[2], ob[2], na[2], nb[2], aps, naps;
    rsrc_t ars, nrs;
    offsets(kc_begin, oa, ob, ars, aps);
    issue(oa, ob, ars, aps);
    offsets(kc_begin + 1, oa, ob, ars, aps);
    for (int kc = kc_begin; kc < kc_end; ++kc) {
      store();  // waits for the loads of chunk kc
      __syncthreads();
      issue(oa, ob, ars, aps);  // chunk kc+1 (offsets from one chunk ago): in flight under the MFMAs
      __builtin_amdgcn_sched_barrier(0);
      offsets(kc + 2, na, nb, nrs, naps);  // address arithmetic interleaves with the MFMAs below
      split_mma_chunk(As, Bs, acc, wm, wn, li, lh);
      __syncthreads();  // every wave is done reading before the next chunk overwrites the buffer
#pragma unroll
      for (int i = 0; i < 2; ++i) oa[i] = na[i], ob[i] = nb[i];
      ars = nrs, aps = naps;
    }
  }

  // ---- epilogue (same semantics as rac_conv2d FWD) ----
  const bool slab = p.split_k > 1;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + (wn * 2 + nt) * 32 + li;
    const bool nok = n < p.N;
    float bias = 0.f, sc = 1.f, sh = 0.f;
    if (!slab && nok) {
      if (p.bias) bias = p.bias[n];
      if (p.scale) {
        sc = p.scale[n];
        sh = p.shift[n];
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * 2 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.M || !nok) continue;
        float v = acc[mt][nt][r];
        if (slab) {
          p.out0[(long)blockIdx.z * p.slab_stride + (long)m * p.N + n] = v;
          continue;
        }
        v += bias;
        s1 += v;
        s2 += v * v;
        v = v * sc + sh;
        if (p.act == RAC_ACT_LEAKY02)
          v = v > 0.f ? v : 0.2f * v;
        else if (p.act == RAC_ACT_SIGMOID)
          v = sigmoid_acc(v);
        p.out0[(long)m * p.N + n] = v;
      }
    }
    if (p.stats && !slab) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lh == 0 && nok) {
        double* sg = p.stats + (p.stats_rows ? (long)(m0 / p.stats_rows) * 2 * p.N : 0L);
        atomicAdd(sg + n, (double)s1);
        atomicAdd(sg + p.N + n, (double)s2);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// Tap-inner forward kernel for latent-resolution convs (H*W divides 128: the 8x8 ConvLSTM maps).
// A 128-pixel M-tile then holds WHOLE images, so every tap of the k x k window reads pixels of the same tile:
// the activation chunk (128 pixels x 32 channels x 3 parts) is staged in LDS ONCE per channel chunk and all
// k*k taps reuse it through shifted fragment reads (rows that leave the image read a zero line), instead of
// being re-fetched from L2 / Infinity Cache per tap.  Only the weight chunk streams per tap (double-buffered):
// half the CU <- L2 bytes of the tap-outer kernel and 1/25 of its activation traffic at k = 5; PMC had
// shown that kernel pulling ~10 TB/s through the fabric (80x the unique bytes).
// K order is (channel chunk, tap) instead of (tap, channel chunk): only the summation order changes.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void igemm_split_tapinner_kernel(SplitP p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  u32x4* As = lds;                      // [3][128][4]
  u32x4* Bs = lds + 3 * 128 * 4;        // [2 buffers][3][128][4]
  u32x4* Zs = lds + 3 * 3 * 128 * 4;    // 4 x 16 B of zeros
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_group) {
    // Workgroups are dealt to the 8 XCDs round-robin in linear-id order, so the few M-tiles that read the same
    // (N-tile, K-slab) weight bytes would land on 8 different L2s and each pull the slab from HBM.  Remap:
    // XCD x takes groups x, x+8, ...; the M-tiles of a group are consecutive in that XCD's queue.
    const int mt = gridDim.x, nt = gridDim.y;
    const int lin = bx + mt * (by + nt * bz);
    const int xcd = lin & 7, s = lin >> 3;
    const int grp = (s / mt) * 8 + xcd;
    bx = s % mt;
    by = grp % nt;
    bz = grp / nt;
  }
  const int m0 = bx * SBM, n0 = by * SBN;
  const int kc_begin = bz * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);
  if (tid < 4) Zs[tid] = u32x4{0u, 0u, 0u, 0u};

  const int srow = tid >> 2, schunk = tid & 3;
  int b_row[2];
  bool a_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_ok[i] = m0 + srow + 64 * i < p.M;
    int n = n0 + srow + 64 * i;
    b_row[i] = (n < p.N) ? n * p.taps * p.Cin : -1;
  }
  // fragment rows of this lane: position inside its image, for the tap shift
  int f_img[2], f_y[2], f_x[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = (wm * 2 + t) * 32 + li;
    const int im = r / p.HW;
    const int q = r - im * p.HW;
    f_img[t] = im * p.HW;
    f_y[t] = q / p.W;
    f_x[t] = q - f_y[t] * p.W;
  }
  const rsrc_t w_rsrc = mk_rsrc(p.w, (unsigned)(3 * p.w_ps * 2));
  const unsigned w_pstride = (unsigned)(p.w_ps * 2);

  u32x4 ra[6], rb[6];
  auto issue_a = [&](int cc, bool live) {  // activation chunk `cc`, unshifted pixels of this tile
    const int c0 = cc * SBK;
    const bool first = c0 < p.a_split;
    const int Cs = first ? p.a_split : p.Cin - p.a_split;
    const int cl = (first ? c0 : c0 - p.a_split) + schunk * 8;
    const long aps = first ? p.a0_ps : p.a1_ps;
    const rsrc_t a_rsrc = mk_rsrc(first ? p.a0 : p.a1, (unsigned)(3 * aps * 2));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = live & a_ok[i] & (cl < Cs);
      const unsigned oa = (unsigned)((m0 + srow + 64 * i) * Cs + cl) * 2u;
#pragma unroll
      for (int part = 0; part < 3; ++part)
        ra[part * 2 + i] = ld16(a_rsrc, ok ? oa + (unsigned)(part * aps * 2) : OOBS);
    }
  };
  auto issue_b = [&](int kc, int cc, int tap) {  // weight chunk kc = (cc, tap)
    const bool live = kc < kc_end;
    const int c0 = cc * SBK;
    const int s0 = p.w_chunk_major ? (cc * p.taps + tap) * SBK + schunk * 8 : tap * p.Cin + c0 + schunk * 8;
    const bool cok = live & (c0 + schunk * 8 < p.Cin);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = cok & (b_row[i] >= 0);
      const unsigned ob = (unsigned)(b_row[i] + s0) * 2u;
#pragma unroll
      for (int part = 0; part < 3; ++part) rb[part * 2 + i] = ld16(w_rsrc, ok ? ob + part * w_pstride : OOBS);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kc_begin < kc_end) {
    // (cc, tap, ky, kx) of the current chunk are carried and stepped: no division in the loop
    int cc = kc_begin / p.taps;
    int tap = kc_begin - cc * p.taps;
    int ky = tap / p.ks, kx = tap - ky * p.ks;
    issue_a(cc, true);
    issue_b(kc_begin, cc, tap);
    for (int kc = kc_begin; kc < kc_end; ++kc) {
      const int buf = (kc - kc_begin) & 1;
      u32x4* Bb = Bs + buf * (3 * 128 * 4);
      const bool new_a = (kc == kc_begin) | (tap == 0);
      if (new_a) {
        __syncthreads();  // every wave has finished the previous channel chunk's fragment reads
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
          for (int i = 0; i < 2; ++i) As[lds_off(part, srow + 64 * i, schunk)] = ra[part * 2 + i];
      }
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int i = 0; i < 2; ++i) Bb[lds_off(part, srow + 64 * i, schunk)] = rb[part * 2 + i];
      __syncthreads();
      const bool last_tap = tap == p.taps - 1;
      const int ncc = last_tap ? cc + 1 : cc, ntap = last_tap ? 0 : tap + 1;
      issue_b(kc + 1, ncc, ntap);  // weights of the next tap: in flight under the MFMAs
      if (last_tap) issue_a(cc + 1, kc + 1 < kc_end);  // next channel chunk's activations
      __builtin_amdgcn_sched_barrier(0);

      // tap shift on the A fragment rows: rows that leave their image read the zero line
      const int dy = ky - p.pad, dx = kx - p.pad;
      int arow[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int yy = f_y[t] + dy, xx = f_x[t] + dx;
        const bool ok = ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
        arow[t] = ok ? f_img[t] + yy * p.W + xx : -1;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 fa[2][3], fb[2][3];
        const int chunk = 2 * s + lh;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int part = 0; part < 3; ++part) {
            const u32x4* src = arow[t] >= 0 ? As + lds_off(part, arow[t], chunk) : Zs + chunk;
            fa[t][part] = __builtin_bit_cast(bf16x8, *src);
            fb[t][part] = __builtin_bit_cast(bf16x8, Bb[lds_off(part, (wn * 2 + t) * 32 + li, chunk)]);
          }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            f32x16 c = acc[mt][nt];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][2], fb[nt][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[nt][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[nt][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[nt][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[nt][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[nt][0], c, 0, 0, 0);
            acc[mt][nt] = c;
          }
      }
      cc = ncc;
      tap = ntap;
      kx = (kx + 1 == p.ks) ? 0 : kx + 1;
      ky = last_tap ? 0 : (kx == 0 ? ky + 1 : ky);
    }
  }

  // ---- epilogue (same semantics as rac_conv2d FWD) ----
  const bool slab = p.split_k > 1;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + (wn * 2 + nt) * 32 + li;
    const bool nok = n < p.N;
    float bias = 0.f, sc = 1.f, sh = 0.f;
    if (!slab && nok) {
      if (p.bias) bias = p.bias[n];
      if (p.scale) {
        sc = p.scale[n];
        sh = p.shift[n];
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * 2 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.M || !nok) continue;
        float v = acc[mt][nt][r];
        if (slab) {
          p.out0[(long)bz * p.slab_stride + (long)m * p.N + n] = v;
          continue;
        }
        v += bias;
        s1 += v;
        s2 += v * v;
        v = v * sc + sh;
        if (p.act == RAC_ACT_LEAKY02)
          v = v > 0.f ? v : 0.2f * v;
        else if (p.act == RAC_ACT_SIGMOID)
          v = sigmoid_acc(v);
        p.out0[(long)m * p.N + n] = v;
      }
    }
    if (p.stats && !slab) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lh == 0 && nok) {
        double* sg = p.stats + (p.stats_rows ? (long)(m0 / p.stats_rows) * 2 * p.N : 0L);
        atomicAdd(sg + n, (double)s1);
        atomicAdd(sg + p.N + n, (double)s2);
      }
    }
  }
}



// ---------------------------------------------------------------------------------------------------------
// Tap-inner forward kernel with the WEIGHT operand streamed straight into MFMA registers (no LDS, no barrier).
// Timing knock-outs on the kernel above showed where its time went at M = 32000: MFMAs alone 10.6 ms (the pipe's
// floor at the 1.9 GHz it holds), + weight loads 1.7 ms, + their ds_write_b128 staging 1.4 ms (a store moves its
// VGPRs to the LDS at 13 cycles per wave-instruction and nothing hides it), + the per-chunk barrier 0.7 ms.
// Here the weights are stored once (cached copy) in FRAGMENT ORDER,
//     w[part][Cout/32][Cin/32][tap][s 0..1][lane 0..63][8 bf16],   lane = 32 h + (n mod 32), k = 16 s + 8 h + j,
// so the B operand of every MFMA is ONE fully coalesced 1 KB buffer_load_dwordx4 per wave, held two chunks
// ahead in registers.  The four waves of a workgroup split the 128-column tile by columns (32 each) and all read
// the whole 128-pixel activation image from LDS (staged once per channel chunk, double-buffered: one barrier
// per k*k taps instead of one per tap).  Same arithmetic and summation order as the kernel above.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void igemm_split_bdirect_kernel(SplitP p) {
  // LDS: two activation buffers [part 3][row 0..128][80 B]; a row is 64 B of data (32 channels) + 16 B of pad, so
  // that any 16 consecutive rows fall on distinct 4-bank slots (80 r / 4 mod 64 has period 16) and a tap shift is
  // ONE wave-uniform byte offset; row 128 of every part plane is a zero line for the pixels that leave the image.
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_group) {  // see igemm_split_tapinner_kernel
    const int mt = gridDim.x, nt = gridDim.y;
    const int lin = bx + mt * (by + nt * bz);
    const int xcd = lin & 7, s = lin >> 3;
    const int grp = (s / mt) * 8 + xcd;
    bx = s % mt;
    by = grp % nt;
    bz = grp / nt;
  }
  const int m0 = bx * SBM, n0 = by * SBN;
  const int kc_begin = bz * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);
  if (tid < 2 * 3 * 5) {  // zero lines (80 B = 5 x 16) of both buffers
    const int b = tid / 15, r = tid - b * 15;
    *reinterpret_cast<u32x4*>(lds_raw + b * BD_ABUF + (r / 5) * BD_PLANE + 128 * BD_ROW + (r % 5) * 16) =
        u32x4{0u, 0u, 0u, 0u};
  }

  const int srow = tid >> 2, schunk = tid & 3;
  bool a_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_ok[i] = m0 + srow + 64 * i < p.M;
  // fragment rows of this lane (the four 32-row blocks of the tile): LDS byte offset of the unshifted row, and
  // one bit per tap telling whether the shifted pixel stays inside its image
  int abase[4];
  unsigned amask[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = t * 32 + li;
    const int im = r / p.HW;
    const int q = r - im * p.HW;
    const int y = q / p.W, x = q - y * p.W;
    abase[t] = r * BD_ROW + lh * 16;
    unsigned mk = 0;
    for (int tp = 0; tp < p.taps; ++tp) {
      const int yy = y + tp / p.ks - p.pad, xx = x + tp % p.ks - p.pad;
      mk |= (((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W)) ? (1u << tp) : 0u;
    }
    amask[t] = mk;
  }
  const int zrow = 128 * BD_ROW + lh * 16;
  // this wave's 32 weight columns: a contiguous stream of 2 KB per (channel chunk, tap) and part.  Reads past the
  // K range or of a dead column tile stay inside the buffer's range check and are never used.
  const int ntile = (n0 >> 5) + wid;
  const rsrc_t w_rsrc = mk_rsrc(p.w, (unsigned)(3 * p.w_ps * 2));
  const unsigned w_pstride = (unsigned)(p.w_ps * 2);
  unsigned b_off[3];
#pragma unroll
  for (int part = 0; part < 3; ++part)
    b_off[part] = (ntile * 32 < p.N ? (unsigned)ntile * (unsigned)p.nchunks * 2048u : 0u) + (unsigned)lane * 16u +
                  part * w_pstride;

  auto load_b = [&](u32x4(&rb)[6], int kc) {
    const int so = kc * 2048;  // wave-uniform: goes to the instruction's scalar offset
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        rb[part * 2 + s] = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off[part] + s * 1024u), so, 0));
  };
  // activations are read as fp32 [pixel][channel] and split into their bf16 parts on the way into LDS (once per
  // channel chunk = once per k*k taps: ~130 VALU instructions against 1200 MFMAs), so no split pass precedes the conv
  u32x4 ra[4];
  auto issue_a = [&](int cc) {  // activation chunk `cc`, unshifted pixels of this tile
    const int c0 = cc * SBK;
    const bool first = c0 < p.a_split;
    const int Cs = first ? p.a_split : p.Cin - p.a_split;
    const int cl = (first ? c0 : c0 - p.a_split) + schunk * 8;
    const rsrc_t a_rsrc = mk_rsrc(first ? (const void*)p.a0 : (const void*)p.a1, (unsigned)((long)p.P * Cs * 4));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned oa = (unsigned)((m0 + srow + 64 * i) * Cs + cl) * 4u;
      ra[2 * i] = ld16(a_rsrc, a_ok[i] ? oa : OOBS);
      ra[2 * i + 1] = ld16(a_rsrc, a_ok[i] ? oa + 16u : OOBS);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 q[3];
      split8(ra[2 * i], ra[2 * i + 1], q);
#pragma unroll
      for (int part = 0; part < 3; ++part)
        *reinterpret_cast<u32x4*>(lds_raw + buf * BD_ABUF + part * BD_PLANE + (srow + 64 * i) * BD_ROW + schunk * 16) =
            q[part];
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (kc_begin < kc_end) {
    int cc = kc_begin / p.taps;
    int tap = kc_begin - cc * p.taps;
    int ky = tap / p.ks, kx = tap - ky * p.ks;
    int cur = 0;
    bool fresh = true;  // first chunk of a channel chunk inside this K range
    u32x4 b0[6], b1[6], b2[6];
    issue_a(cc);
    load_b(b0, kc_begin);
    load_b(b1, kc_begin + 1);
    store_a(0);
    __syncthreads();

    auto step = [&](const u32x4(&rb)[6], int kc) {
      const bool last_tap = tap == p.taps - 1;
      const bool more = kc + 1 < kc_end;
      if (fresh && (cc + 1) * p.taps < kc_end) issue_a(cc + 1);  // next channel chunk: in flight for k*k taps
      fresh = false;
      // tap shift: one uniform byte offset on the fragment rows; rows that leave their image read the zero line
      const int shift = ((ky - p.pad) * p.W + (kx - p.pad)) * BD_ROW + cur * BD_ABUF;
      const int zr = zrow + cur * BD_ABUF;
      const unsigned bit = 1u << tap;
      int aoff[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) aoff[t] = (amask[t] & bit) ? abase[t] + shift : zr;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 fa[4][3], fb[3];
#pragma unroll
        for (int part = 0; part < 3; ++part) fb[part] = __builtin_bit_cast(bf16x8, rb[part * 2 + s]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int part = 0; part < 3; ++part) {
            fa[t][part] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const u32x4*>(lds_raw + aoff[t] + part * BD_PLANE + s * 32));
          }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          f32x16 c = acc[mt];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][2], fb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[0], c, 0, 0, 0);
          acc[mt] = c;
        }
      }
      if (last_tap && more) {  // the other buffer was last read before the previous such barrier
        store_a(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        fresh = true;
      }
      cc = last_tap ? cc + 1 : cc;
      tap = last_tap ? 0 : tap + 1;
      kx = (kx + 1 == p.ks) ? 0 : kx + 1;
      ky = last_tap ? 0 : (kx == 0 ? ky + 1 : ky);
    };

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

  // ---- epilogue (same semantics as rac_conv2d FWD): this wave owns columns n0 + 32 wid .. + 31 ----
  const bool slab = p.split_k > 1;
  const int n = n0 + wid * 32 + li;
  const bool nok = n < p.N;
  float bias = 0.f, sc = 1.f, sh = 0.f;
  if (!slab && nok) {
    if (p.bias) bias = p.bias[n];
    if (p.scale) {
      sc = p.scale[n];
      sh = p.shift[n];
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= p.M || !nok) continue;
      float v = acc[mt][r];
      if (slab) {
        p.out0[(long)bz * p.slab_stride + (long)m * p.N + n] = v;
        continue;
      }
      v += bias;
      s1 += v;
      s2 += v * v;
      v = v * sc + sh;
      if (p.act == RAC_ACT_LEAKY02)
        v = v > 0.f ? v : 0.2f * v;
      else if (p.act == RAC_ACT_SIGMOID)
        v = sigmoid_acc(v);
      p.out0[(long)m * p.N + n] = v;
    }
  }
  if (p.stats && !slab) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lh == 0 && nok) {
      double* sg = p.stats + (p.stats_rows ? (long)(m0 / p.stats_rows) * 2 * p.N : 0L);
      atomicAdd(sg + n, (double)s1);
      atomicAdd(sg + p.N + n, (double)s2);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// igemm_split_bdirect_kernel on v_mfma_f32_16x16x32_bf16 (weight layout 3).  Same algorithm, same sums; the matrix
// instruction is the 16x16x32 form because the chip holds a higher clock on it under this MFMA-bound load
// (MI355X_MICROARCH.md DVFS item 7: ~1.12-1.15x the FLOP/s of the 32x32x16 form at equal cycles per FLOP).
//   A operand: lane l holds row l & 15, k = 8 (l >> 4) .. +7  -> the LDS image is CHUNK-major,
//     [part][16-byte chunk 0..3][row 0..143][16 B] (rows 128.. are zeros), so that the 16 lanes of a ds_read_b128
//     group always touch 16 distinct rows mod 16 = 16 distinct bank slots, and a tap shift is still one uniform offset.
//   B operand: lane l holds column l & 15, k = 8 (l >> 4) .. +7 -> weights in the matching fragment order
//     [part][Cout/32][Cin/32][tap][nb 0..1][lane][8]  (nb = 16-column half of the wave's 32 columns).
//   C: 8 x 2 accumulators of 16x16 per wave (128 rows x 32 columns), col = l & 15, row = 4 (l >> 4) + reg.
//   Rows per workgroup = the largest whole number of images within 128 pixels that is a multiple of 16 (128 for 8x8
//   maps, 96 = two 6x8 maps of 48x64 frames): blocks past it are skipped.
// ---------------------------------------------------------------------------------------------------------
constexpr int B16_CP = 144 * 16, B16_PP = 4 * B16_CP, B16_ABUF = 3 * B16_PP;  // chunk plane, part plane, buffer (bytes)

__global__ __launch_bounds__(256, 2) void igemm_split_bdirect16_kernel(SplitP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_group) {  // see igemm_split_tapinner_kernel
    const int mt = gridDim.x, nt = gridDim.y;
    const int lin = bx + mt * (by + nt * bz);
    const int xcd = lin & 7, s = lin >> 3;
    const int grp = (s / mt) * 8 + xcd;
    bx = s % mt;
    by = grp % nt;
    bz = grp / nt;
  }
  // rows per workgroup: the largest whole number of images that fits 128 pixels (128 for the 8x8 maps, 96 = two
  // images for the reference's default 48x64 frames, whose latent maps are 6x8), a multiple of 16
  const int TM = p.tile_m;
  const int nmb = TM >> 4;  // live 16-row blocks of the wave's 8
  const int m0 = bx * TM, n0 = by * SBN;
  const int kc_begin = bz * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);
  // zero rows 128..143 of every chunk plane of both buffers: 2 * 3 * 4 * 16 = 384 vectors
  for (int v = tid; v < 384; v += 256) {
    const int pl = v >> 4, r = v & 15;  // plane index (buffer, part, chunk), row
    *reinterpret_cast<u32x4*>(lds_raw + (pl / 12) * B16_ABUF + ((pl / 4) % 3) * B16_PP + (pl & 3) * B16_CP +
                              (128 + r) * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  // staging: thread -> row tid & 127, chunks (tid >> 7) and (tid >> 7) + 2 (8 consecutive lanes = 8 rows of one
  // chunk plane: distinct bank slots on the ds_write side too)
  const int srow = tid & 127, sch = tid >> 7;
  const bool a_ok = (srow < TM) & (m0 + srow < p.M);
  // fragment rows of this lane (eight 16-row blocks): one bit per tap and block for the shifted pixel's validity
  unsigned amask[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int r = t * 16 + lr;  // blocks t >= nmb are never used
    const int im = r / p.HW;
    const int q = r - im * p.HW;
    const int y = q / p.W, x = q - y * p.W;
    unsigned mk = 0;
    for (int tp = 0; tp < p.taps; ++tp) {
      const int yy = y + tp / p.ks - p.pad, xx = x + tp % p.ks - p.pad;
      mk |= (((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W)) ? (1u << tp) : 0u;
    }
    amask[t] = mk;
  }
  const int abase = lq * B16_CP + lr * 16;       // block t adds t * 256
  const int zrow = lq * B16_CP + 128 * 16;
  const int ntile = (n0 >> 5) + wid;
  const rsrc_t w_rsrc = mk_rsrc(p.w, (unsigned)(3 * p.w_ps * 2));
  const unsigned w_pstride = (unsigned)(p.w_ps * 2);
  unsigned b_off[3];
#pragma unroll
  for (int part = 0; part < 3; ++part)
    b_off[part] = (ntile * 32 < p.N ? (unsigned)ntile * (unsigned)p.nchunks * 2048u : 0u) + (unsigned)lane * 16u +
                  part * w_pstride;
  auto load_b = [&](u32x4(&rb)[6], int kc) {
    const int so = kc * 2048;
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        rb[part * 2 + nb] = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off[part] + nb * 1024u), so, 0));
  };
  u32x4 ra[4];  // fp32 activations: [chunk i][half]
  auto issue_a = [&](int cc) {
    const int c0 = cc * SBK;
    const bool first = c0 < p.a_split;
    const int Cs = first ? p.a_split : p.Cin - p.a_split;
    const int cl = first ? c0 : c0 - p.a_split;
    const rsrc_t a_rsrc = mk_rsrc(first ? (const void*)p.a0 : (const void*)p.a1, (unsigned)((long)p.P * Cs * 4));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned oa = (unsigned)((m0 + srow) * Cs + cl + (sch + 2 * i) * 8) * 4u;
      ra[2 * i] = ld16(a_rsrc, a_ok ? oa : OOBS);
      ra[2 * i + 1] = ld16(a_rsrc, a_ok ? oa + 16u : OOBS);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 q[3];
      split8(ra[2 * i], ra[2 * i + 1], q);
#pragma unroll
      for (int part = 0; part < 3; ++part)
        *reinterpret_cast<u32x4*>(lds_raw + buf * B16_ABUF + part * B16_PP + (sch + 2 * i) * B16_CP + srow * 16) = q[part];
    }
  };

  f32x4 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kc_begin < kc_end) {
    int cc = kc_begin / p.taps;
    int tap = kc_begin - cc * p.taps;
    int ky = tap / p.ks, kx = tap - ky * p.ks;
    int cur = 0;
    bool fresh = true;
    u32x4 b0[6], b1[6], b2[6];
    issue_a(cc);
    load_b(b0, kc_begin);
    load_b(b1, kc_begin + 1);
    store_a(0);
    __syncthreads();

    auto step = [&](const u32x4(&rb)[6], int kc) {
      const bool last_tap = tap == p.taps - 1;
      const bool more = kc + 1 < kc_end;
      if (fresh && (cc + 1) * p.taps < kc_end) issue_a(cc + 1);
      fresh = false;
      const int drow = (ky - p.pad) * p.W + (kx - p.pad);
      const int shift = drow * 16 + cur * B16_ABUF + abase;
      // pixels outside the image read one of the 16 zero rows: the one on the bank slot this lane's shifted row
      // would have used, so that the read group stays conflict-free
      const int zr = zrow + cur * B16_ABUF + ((lr + drow) & 15) * 16;
      const unsigned bit = 1u << tap;
      bf16x8 fb[2][3];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int part = 0; part < 3; ++part) fb[nb][part] = __builtin_bit_cast(bf16x8, rb[part * 2 + nb]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8 fa[4][3];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int mb = 4 * h + t;
          if (mb >= nmb) continue;  // wave-uniform
          const int ao = (amask[mb] & bit) ? shift + mb * 256 : zr;
#pragma unroll
          for (int part = 0; part < 3; ++part)
            fa[t][part] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds_raw + ao + part * B16_PP));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            if (4 * h + t >= nmb) continue;
            f32x4 c = acc[4 * h + t][nb];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][2], fb[nb][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][1], fb[nb][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][0], fb[nb][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][1], fb[nb][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][0], fb[nb][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][0], fb[nb][0], c, 0, 0, 0);
            acc[4 * h + t][nb] = c;
          }
      }
      if (last_tap && more) {
        store_a(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        fresh = true;
      }
      cc = last_tap ? cc + 1 : cc;
      tap = last_tap ? 0 : tap + 1;
      kx = (kx + 1 == p.ks) ? 0 : kx + 1;
      ky = last_tap ? 0 : (kx == 0 ? ky + 1 : ky);
    };

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

  // ---- epilogue (same semantics as rac_conv2d FWD): col = l & 15 (+16 nb), rows 4 (l >> 4) + reg of each block ----
  const bool slab = p.split_k > 1;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = n0 + wid * 32 + nb * 16 + lr;
    const bool nok = n < p.N;
    float bias = 0.f, sc = 1.f, sh = 0.f;
    if (!slab && nok) {
      if (p.bias) bias = p.bias[n];
      if (p.scale) {
        sc = p.scale[n];
        sh = p.shift[n];
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mb * 16 + 4 * lq + r;
        if (mb >= nmb || m >= p.M || !nok) continue;
        float v = acc[mb][nb][r];
        if (slab) {
          p.out0[(long)bz * p.slab_stride + (long)m * p.N + n] = v;
          continue;
        }
        v += bias;
        s1 += v;
        s2 += v * v;
        v = v * sc + sh;
        if (p.act == RAC_ACT_LEAKY02)
          v = v > 0.f ? v : 0.2f * v;
        else if (p.act == RAC_ACT_SIGMOID)
          v = sigmoid_acc(v);
        p.out0[(long)m * p.N + n] = v;
      }
    }
    if (p.stats && !slab) {
      s1 += __shfl_xor(s1, 16);
      s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lq == 0 && nok) {
        double* sg = p.stats + (p.stats_rows ? (long)(m0 / p.stats_rows) * 2 * p.N : 0L);
        atomicAdd(sg + n, (double)s1);
        atomicAdd(sg + p.N + n, (double)s2);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same kernel for maps LARGER than a tile (W = 16, 32 or 64 and 128 / W rows of one image per tile: the 16x16
// and 32x32 vgg maps, the 16x16 ConvLSTM maps of a 128x128 model).  The tile's pixels plus `pad` image rows above
// and below (the halo; zeros outside the image) are staged per channel chunk as (R + 2 pad) * W rows of the padded
// LDS image, which is one CONTIGUOUS pixel range of the input.  A tap shift is again one wave-uniform byte offset;
// only the x shift can leave the row (one bit per kernel column and lane selects the zero row), the y shift lands
// in the halo.  One LDS buffer (up to 257 rows x 3 parts = 61.7 KB, two workgroups per CU), two barriers per
// channel chunk; weights in fragment order straight into registers, one chunk ahead.
// ---------------------------------------------------------------------------------------------------------
// NV = 16-byte staging vectors per thread and part: ceil((R + 2 pad) * W * 4 / 256).
// TN = 32-column groups per workgroup: 4 (128 columns, waves 1 x 4, each all 128 rows) or 2 (64 columns for the
// 64-channel layers: waves 2 x 2, each 64 rows x 32 columns).
template <int NV, int TN>
__global__ __launch_bounds__(256, 2) void igemm_split_bdirect_rows_kernel(SplitP p) {
  constexpr int MT = TN;           // 32-row blocks per wave
  constexpr int BNW = TN * 32;     // columns per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_group) {  // see igemm_split_tapinner_kernel
    const int mt = gridDim.x, nt = gridDim.y;
    const int lin = bx + mt * (by + nt * bz);
    const int xcd = lin & 7, s = lin >> 3;
    const int grp = (s / mt) * 8 + xcd;
    bx = s % mt;
    by = grp % nt;
    bz = grp / nt;
  }
  const int wn = wid % TN, wm = wid / TN;
  const int m0 = bx * SBM, n0 = by * BNW;
  const int kc_begin = bz * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);
  const int halo = p.pad * p.W;             // pixel rows of halo above (and below) the tile
  const int nrows = SBM + 2 * halo;         // staged pixel rows; row `nrows` of every plane is the zero row
  const int plane = (nrows + 1) * BD_ROW;
  if (tid < 3 * 5) *reinterpret_cast<u32x4*>(lds_raw + (tid / 5) * plane + nrows * BD_ROW + (tid % 5) * 16) =
      u32x4{0u, 0u, 0u, 0u};

  // staging: vector v = tid + 256 i covers LDS row v / 4 (input pixel m0 - halo + row), 16-byte chunk v % 4
  const int y_tile = (m0 % p.HW) / p.W;     // first image row of the tile
  int s_off[NV];                            // LDS byte offset, or -1: nothing to stage
  bool s_ok[NV];                            // inside the image (else zeros)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + 256 * i;
    const int row = v >> 2;
    s_off[i] = row < nrows ? row * BD_ROW + (v & 3) * 16 : -1;
    const int y = y_tile - p.pad + row / p.W;
    s_ok[i] = (row < nrows) & ((unsigned)y < (unsigned)p.H) & (m0 - halo + row < p.M);
  }
  // fragment rows of this lane: unshifted LDS byte offset, and one bit per kernel column for the x shift
  int abase[MT];
  unsigned amask[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int r = (wm * MT + t) * 32 + li;
    const int x = r % p.W;
    abase[t] = (halo + r) * BD_ROW + lh * 16;
    unsigned mk = 0;
    for (int kx = 0; kx < p.ks; ++kx) mk |= ((unsigned)(x + kx - p.pad) < (unsigned)p.W) ? (1u << kx) : 0u;
    amask[t] = mk;
  }
  const int zrow = nrows * BD_ROW + lh * 16;
  const int ntile = (n0 >> 5) + wn;
  const rsrc_t w_rsrc = mk_rsrc(p.w, (unsigned)(3 * p.w_ps * 2));
  const unsigned w_pstride = (unsigned)(p.w_ps * 2);
  unsigned b_off[3];
#pragma unroll
  for (int part = 0; part < 3; ++part)
    b_off[part] = (ntile * 32 < p.N ? (unsigned)ntile * (unsigned)p.nchunks * 2048u : 0u) + (unsigned)lane * 16u +
                  part * w_pstride;
  auto load_b = [&](u32x4(&rb)[6], int kc) {
    const int so = kc * 2048;
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        rb[part * 2 + s] = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off[part] + s * 1024u), so, 0));
  };
  u32x4 ra[2 * NV];  // fp32 activations, split into their bf16 parts on the way into LDS (see the kernel above)
  auto issue_a = [&](int cc) {
    const int c0 = cc * SBK;
    const bool first = c0 < p.a_split;
    const int Cs = first ? p.a_split : p.Cin - p.a_split;
    const int cl = (first ? c0 : c0 - p.a_split) + (tid & 3) * 8;
    const rsrc_t a_rsrc = mk_rsrc(first ? (const void*)p.a0 : (const void*)p.a1, (unsigned)((long)p.P * Cs * 4));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int pix = m0 - halo + ((tid + 256 * i) >> 2);
      const unsigned oa = (unsigned)(pix * Cs + cl) * 4u;
      ra[2 * i] = ld16(a_rsrc, s_ok[i] ? oa : OOBS);
      ra[2 * i + 1] = ld16(a_rsrc, s_ok[i] ? oa + 16u : OOBS);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (s_off[i] < 0) continue;
      u32x4 q[3];
      split8(ra[2 * i], ra[2 * i + 1], q);
#pragma unroll
      for (int part = 0; part < 3; ++part) *reinterpret_cast<u32x4*>(lds_raw + part * plane + s_off[i]) = q[part];
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (kc_begin < kc_end) {
    int cc = kc_begin / p.taps;
    int tap = kc_begin - cc * p.taps;
    int ky = tap / p.ks, kx = tap - ky * p.ks;
    bool fresh = true;
    u32x4 b0[6], b1[6];
    issue_a(cc);
    load_b(b0, kc_begin);
    store_a();
    __syncthreads();

    auto step = [&](const u32x4(&rb)[6], int kc) {
      const bool last_tap = tap == p.taps - 1;
      const bool more = kc + 1 < kc_end;
      if (fresh && (cc + 1) * p.taps < kc_end) issue_a(cc + 1);
      fresh = false;
      const int shift = ((ky - p.pad) * p.W + (kx - p.pad)) * BD_ROW;
      const unsigned bit = 1u << kx;
      int aoff[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) aoff[t] = (amask[t] & bit) ? abase[t] + shift : zrow;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 fa[MT][3], fb[3];
#pragma unroll
        for (int part = 0; part < 3; ++part) fb[part] = __builtin_bit_cast(bf16x8, rb[part * 2 + s]);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int part = 0; part < 3; ++part)
            fa[t][part] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const u32x4*>(lds_raw + aoff[t] + part * plane + s * 32));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x16 c = acc[mt];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][2], fb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[0], c, 0, 0, 0);
          acc[mt] = c;
        }
      }
      if (last_tap && more) {
        __syncthreads();  // every wave has finished this channel chunk's fragment reads
        store_a();
        __syncthreads();
        fresh = true;
      }
      cc = last_tap ? cc + 1 : cc;
      tap = last_tap ? 0 : tap + 1;
      kx = (kx + 1 == p.ks) ? 0 : kx + 1;
      ky = last_tap ? 0 : (kx == 0 ? ky + 1 : ky);
    };

    for (int kc = kc_begin; kc < kc_end; kc += 2) {
      load_b(b1, kc + 1);
      step(b0, kc);
      if (kc + 1 < kc_end) {
        load_b(b0, kc + 2);
        step(b1, kc + 1);
      }
    }
  }

  // ---- epilogue (same semantics as rac_conv2d FWD): this wave owns columns n0 + 32 wn .. + 31 ----
  const bool slab = p.split_k > 1;
  const int n = n0 + wn * 32 + li;
  const bool nok = n < p.N;
  float bias = 0.f, sc = 1.f, sh = 0.f;
  if (!slab && nok) {
    if (p.bias) bias = p.bias[n];
    if (p.scale) {
      sc = p.scale[n];
      sh = p.shift[n];
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= p.M || !nok) continue;
      float v = acc[mt][r];
      if (slab) {
        p.out0[(long)bz * p.slab_stride + (long)m * p.N + n] = v;
        continue;
      }
      v += bias;
      s1 += v;
      s2 += v * v;
      v = v * sc + sh;
      if (p.act == RAC_ACT_LEAKY02)
        v = v > 0.f ? v : 0.2f * v;
      else if (p.act == RAC_ACT_SIGMOID)
        v = sigmoid_acc(v);
      p.out0[(long)m * p.N + n] = v;
    }
  }
  if (p.stats && !slab) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lh == 0 && nok) {
      double* sg = p.stats + (p.stats_rows ? (long)(m0 / p.stats_rows) * 2 * p.N : 0L);
      atomicAdd(sg + n, (double)s1);
      atomicAdd(sg + p.N + n, (double)s2);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// igemm_split_bdirect_rows_kernel on v_mfma_f32_16x16x32_bf16 (weight layout 3, maps larger than a tile).
// LDS image chunk-major: [part][8-channel group 0..3][staged pixel row 0..nrows+15][16 B], the 16 rows after the
// staged range are zeros (x shifts that leave the image row).  TN = 4: waves 1 x 4 (128 rows x 32 columns each,
// 8 x 2 accumulators of 16x16); TN = 2: waves 2 x 2 (64 rows x 32 columns, 4 x 2 accumulators).
// ---------------------------------------------------------------------------------------------------------
template <int NV, int TN>
__global__ __launch_bounds__(256, 2) void igemm_split_bdirect_rows16_kernel(SplitP p) {
  constexpr int MB = 2 * TN;       // 16-row blocks per wave
  constexpr int BNW = TN * 32;     // columns per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_group) {  // see igemm_split_tapinner_kernel
    const int mt = gridDim.x, nt = gridDim.y;
    const int lin = bx + mt * (by + nt * bz);
    const int xcd = lin & 7, s = lin >> 3;
    const int grp = (s / mt) * 8 + xcd;
    bx = s % mt;
    by = grp % nt;
    bz = grp / nt;
  }
  const int wn = wid % TN, wm = wid / TN;
  // rows per workgroup: R whole image rows with R | H, R * W <= 128 and a multiple of 16 (128 for 16x16 / 32x32 /
  // 64x64 maps; 96 = six rows of the 12x16 maps of 48x64 frames)
  const int TM = p.tile_m;
  const int nmb = TM >> 4;  // live 16-row blocks of the workgroup
  const int m0 = bx * TM, n0 = by * BNW;
  const int kc_begin = bz * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);
  const int halo = p.pad * p.W;
  const int nrows = TM + 2 * halo;           // staged pixel rows (a multiple of 16)
  const int cplane = (nrows + 16) * 16;      // one 8-channel group: staged rows + 16 zero rows
  const int pplane = 4 * cplane;
  if (tid < 3 * 4 * 16)
    *reinterpret_cast<u32x4*>(lds_raw + (tid >> 6) * pplane + ((tid >> 4) & 3) * cplane + (nrows + (tid & 15)) * 16) =
        u32x4{0u, 0u, 0u, 0u};

  // staging: vector v = tid + 256 i -> 8-channel group v / nrows, staged row v % nrows (consecutive lanes =
  // consecutive rows of one group: distinct bank slots)
  const int y_tile = (m0 % p.HW) / p.W;
  int s_off[NV], s_row[NV], s_grp[NV];
  bool s_ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + 256 * i;
    const int g = v / nrows, row = v - g * nrows;
    s_grp[i] = g;
    s_row[i] = row;
    s_off[i] = g < 4 ? g * cplane + row * 16 : -1;
    const int y = y_tile - p.pad + row / p.W;
    s_ok[i] = (g < 4) & ((unsigned)y < (unsigned)p.H) & (m0 - halo + row < p.M);
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
  const int abase = lq * cplane + (halo + wm * MB * 16 + lr) * 16;   // block t adds t * 256
  const int zrow = lq * cplane + nrows * 16;
  const int ntile = (n0 >> 5) + wn;
  const rsrc_t w_rsrc = mk_rsrc(p.w, (unsigned)(3 * p.w_ps * 2));
  const unsigned w_pstride = (unsigned)(p.w_ps * 2);
  unsigned b_off[3];
#pragma unroll
  for (int part = 0; part < 3; ++part)
    b_off[part] = (ntile * 32 < p.N ? (unsigned)ntile * (unsigned)p.nchunks * 2048u : 0u) + (unsigned)lane * 16u +
                  part * w_pstride;
  auto load_b = [&](u32x4(&rb)[6], int kc) {
    const int so = kc * 2048;
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        rb[part * 2 + nb] = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off[part] + nb * 1024u), so, 0));
  };
  u32x4 ra[2 * NV];
  auto issue_a = [&](int cc) {
    const int c0 = cc * SBK;
    const bool first = c0 < p.a_split;
    const int Cs = first ? p.a_split : p.Cin - p.a_split;
    const int cl = first ? c0 : c0 - p.a_split;
    const rsrc_t a_rsrc = mk_rsrc(first ? (const void*)p.a0 : (const void*)p.a1, (unsigned)((long)p.P * Cs * 4));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const unsigned oa = (unsigned)((m0 - halo + s_row[i]) * Cs + cl + s_grp[i] * 8) * 4u;
      ra[2 * i] = ld16(a_rsrc, s_ok[i] ? oa : OOBS);
      ra[2 * i + 1] = ld16(a_rsrc, s_ok[i] ? oa + 16u : OOBS);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (s_off[i] < 0) continue;
      u32x4 q[3];
      split8(ra[2 * i], ra[2 * i + 1], q);
#pragma unroll
      for (int part = 0; part < 3; ++part) *reinterpret_cast<u32x4*>(lds_raw + part * pplane + s_off[i]) = q[part];
    }
  };

  f32x4 acc[MB][2];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kc_begin < kc_end) {
    int cc = kc_begin / p.taps;
    int tap = kc_begin - cc * p.taps;
    int ky = tap / p.ks, kx = tap - ky * p.ks;
    bool fresh = true;
    u32x4 b0[6], b1[6];
    issue_a(cc);
    load_b(b0, kc_begin);
    store_a();
    __syncthreads();

    auto step = [&](const u32x4(&rb)[6], int kc) {
      const bool last_tap = tap == p.taps - 1;
      const bool more = kc + 1 < kc_end;
      if (fresh && (cc + 1) * p.taps < kc_end) issue_a(cc + 1);
      fresh = false;
      const int drow = (ky - p.pad) * p.W + (kx - p.pad);
      const int shift = drow * 16 + abase;
      const int zr = zrow + ((lr + halo + drow) & 15) * 16;  // the zero row on this lane's own bank slot
      const unsigned bit = 1u << kx;
      bf16x8 fb[2][3];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int part = 0; part < 3; ++part) fb[nb][part] = __builtin_bit_cast(bf16x8, rb[part * 2 + nb]);
#pragma unroll
      for (int h = 0; h < MB / 4; ++h) {
        bf16x8 fa[4][3];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int mb = 4 * h + t;
          if (wm * MB + mb >= nmb) continue;  // wave-uniform: past the tile's rows
          const int ao = (amask[mb] & bit) ? shift + mb * 256 : zr;
#pragma unroll
          for (int part = 0; part < 3; ++part)
            fa[t][part] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds_raw + ao + part * pplane));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            if (wm * MB + 4 * h + t >= nmb) continue;
            f32x4 c = acc[4 * h + t][nb];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][2], fb[nb][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][1], fb[nb][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][0], fb[nb][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][1], fb[nb][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][0], fb[nb][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][0], fb[nb][0], c, 0, 0, 0);
            acc[4 * h + t][nb] = c;
          }
      }
      if (last_tap && more) {
        __syncthreads();
        store_a();
        __syncthreads();
        fresh = true;
      }
      cc = last_tap ? cc + 1 : cc;
      tap = last_tap ? 0 : tap + 1;
      kx = (kx + 1 == p.ks) ? 0 : kx + 1;
      ky = last_tap ? 0 : (kx == 0 ? ky + 1 : ky);
    };

    for (int kc = kc_begin; kc < kc_end; kc += 2) {
      load_b(b1, kc + 1);
      step(b0, kc);
      if (kc + 1 < kc_end) {
        load_b(b0, kc + 2);
        step(b1, kc + 1);
      }
    }
  }

  const bool slab = p.split_k > 1;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = n0 + wn * 32 + nb * 16 + lr;
    const bool nok = n < p.N;
    float bias = 0.f, sc = 1.f, sh = 0.f;
    if (!slab && nok) {
      if (p.bias) bias = p.bias[n];
      if (p.scale) {
        sc = p.scale[n];
        sh = p.shift[n];
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (wm * MB + mb) * 16 + 4 * lq + r;
        if (wm * MB + mb >= nmb || m >= p.M || !nok) continue;
        float v = acc[mb][nb][r];
        if (slab) {
          p.out0[(long)bz * p.slab_stride + (long)m * p.N + n] = v;
          continue;
        }
        v += bias;
        s1 += v;
        s2 += v * v;
        v = v * sc + sh;
        if (p.act == RAC_ACT_LEAKY02)
          v = v > 0.f ? v : 0.2f * v;
        else if (p.act == RAC_ACT_SIGMOID)
          v = sigmoid_acc(v);
        p.out0[(long)m * p.N + n] = v;
      }
    }
    if (p.stats && !slab) {
      s1 += __shfl_xor(s1, 16);
      s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lq == 0 && nok) {
        double* sg = p.stats + (p.stats_rows ? (long)(m0 / p.stats_rows) * 2 * p.N : 0L);
        atomicAdd(sg + n, (double)s1);
        atomicAdd(sg + p.N + n, (double)s2);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient on the split-precision pipe.
//   dw[co][tap][ci] += sum_p dy[p][co] * x[p + tap][ci]
// The bf16 MFMA wants 8 consecutive k (= pixels) per lane for both operands, so both are consumed TRANSPOSED:
//   A = dyT[co][p] parts, B = xT_dx[ci][p] parts, where xT_dx is x shifted by dx along the image row with the
//   out-of-row pixels zeroed (one copy per kernel column; rac_transpose_split builds them).  The row shift dy is a
//   multiple of W >= 8 pixels, i.e. a whole number of 16-byte vectors, applied in the load offset with the
//   rows that leave the image returned as zeros by the buffer range check.
// One workgroup = (128 output channels) x (128 input channels of one tap); K = pixels in chunks of 32.
// ---------------------------------------------------------------------------------------------------------
struct WgradSplitP {
  int H, W, ks, pad, Cin, Cout, a_split, split_k, P, HW, taps, nchunks, cps, ntile_per_tap;
  const unsigned short *x0t, *x1t, *dyt;  // [ks][3][C0][P], [ks][3][C1][P], [3][Cout][P]
  float* dw;
  unsigned long long magic_hw, magic_w;
  int wr, wr_shift;  // 8-pixel vectors per image row (W / 8, a power of two) and its log2: fragment-order kernel
};

__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(WgradSplitP p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  u32x4* As = lds;
  u32x4* Bs = lds + 3 * 128 * 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.x * SBM;
  const int tap = blockIdx.y / p.ntile_per_tap;
  const int n0 = (blockIdx.y - tap * p.ntile_per_tap) * SBN;
  const int ky = tap / p.ks, kx = tap - ky * p.ks;
  const int dy = ky - p.pad;
  const int kc_begin = blockIdx.z * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);
  const int srow = tid >> 2, schunk = tid & 3;

  // block-uniform source of this channel range (virtual concat [x0 | x1])
  const bool first = n0 < p.a_split;
  const int Cs = first ? p.a_split : p.Cin - p.a_split;
  const int nl0 = first ? n0 : n0 - p.a_split;
  const long xps = (long)Cs * p.P;  // part stride of the chosen source
  const unsigned short* xbase = (first ? p.x0t : p.x1t) + (long)kx * 3 * xps;
  const rsrc_t x_rsrc = mk_rsrc(xbase, (unsigned)(3 * xps * 2));
  const long dps = (long)p.Cout * p.P;
  const rsrc_t d_rsrc = mk_rsrc(p.dyt, (unsigned)(3 * dps * 2));

  int a_base[2], b_base[2];  // row * P, or -1
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int co = m0 + srow + 64 * i;
    a_base[i] = co < p.Cout ? co * p.P : -1;
    int cl = nl0 + srow + 64 * i;
    b_base[i] = (n0 + srow + 64 * i < p.Cin && cl < Cs) ? cl * p.P : -1;
  }

  u32x4 ra[6], rb[6];
  auto issue = [&](int kc) {
    const bool live = kc < kc_end;
    const int px = kc * SBK + schunk * 8;  // first pixel of this lane's 8-pixel vector (never straddles an image row)
    const bool pok = live & (px < p.P);
    const int b = (int)(((unsigned long long)(unsigned)px * p.magic_hw) >> 40);
    const int r = px - b * p.HW;
    const int y = (int)(((unsigned long long)(unsigned)r * p.magic_w) >> 40);
    const bool yok = pok & ((unsigned)(y + dy) < (unsigned)p.H);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool oka = pok & (a_base[i] >= 0);
      const bool okb = yok & (b_base[i] >= 0);
      const unsigned oa = (unsigned)(a_base[i] + px) * 2u;
      const unsigned ob = (unsigned)(b_base[i] + px + dy * p.W) * 2u;
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        ra[part * 2 + i] = ld16(d_rsrc, oka ? oa + (unsigned)(part * dps * 2) : OOBS);
        rb[part * 2 + i] = ld16(x_rsrc, okb ? ob + (unsigned)(part * xps * 2) : OOBS);
      }
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        As[lds_off(part, srow + 64 * i, schunk)] = ra[part * 2 + i];
        Bs[lds_off(part, srow + 64 * i, schunk)] = rb[part * 2 + i];
      }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kc_begin < kc_end) {
    issue(kc_begin);
    for (int kc = kc_begin; kc < kc_end; ++kc) {
      store();
      __syncthreads();
      issue(kc + 1);
      __builtin_amdgcn_sched_barrier(0);
      split_mma_chunk(As, Bs, acc, wm, wn, li, lh);
      __syncthreads();
    }
  }

  // dw[co][tap][ci] += acc   (C/D layout: col = lane & 31 -> ci, row -> co)
  const long wrow = (long)p.taps * p.Cin;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + (wn * 2 + nt) * 32 + li;
    if (n >= p.Cin || (first ? n >= p.a_split : false)) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * 2 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.Cout) continue;
        float* dst = p.dw + m * wrow + (long)tap * p.Cin + n;
        if (p.split_k > 1)
          atomicAdd(dst, acc[mt][nt][r]);
        else
          *dst += acc[mt][nt][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient with the INPUT operand streamed straight into MFMA registers.
// x^T is stored in fragment order (rac_transpose_split layout 1): [dx][part][ci/32][r = p/8][ci mod 32][8 pixels],
// so the B operand of an MFMA (lane = 32 h + ci mod 32 holds pixels 16 s + 8 h .. +7 of the K chunk) is ONE
// coalesced 1 KB load, and the row shift dy of a tap is a constant offset of dy*W/8 vectors with the rows that
// leave the image selected to zero.  The four waves split the 128 input channels of the tile (32 each); dy^T
// (A operand, [part][Cout][P]) is staged through a double-buffered padded LDS image: one barrier per K chunk,
// no LDS traffic for the inputs.  Same arithmetic and summation order as wgrad_split_kernel.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad_split_bdirect_kernel(WgradSplitP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];  // 2 x [part 3][128 rows][80 B]
  constexpr int PLANE = 128 * BD_ROW, ABUF = 3 * PLANE;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * SBM;
  const int tap = blockIdx.y / p.ntile_per_tap;
  const int n0 = (blockIdx.y - tap * p.ntile_per_tap) * SBN;
  const int ky = tap / p.ks, kx = tap - ky * p.ks;
  const int dy = ky - p.pad;
  const int kc_begin = blockIdx.z * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);
  const int srow = tid >> 2, schunk = tid & 3;

  // block-uniform source of this channel range (virtual concat [x0 | x1])
  const bool first = n0 < p.a_split;
  const int Cs = first ? p.a_split : p.Cin - p.a_split;
  const int nl0 = first ? n0 : n0 - p.a_split;
  const long xps = (long)Cs * p.P;  // part stride of the chosen source
  const unsigned short* xbase = (first ? p.x0t : p.x1t) + (long)kx * 3 * xps;
  const rsrc_t x_rsrc = mk_rsrc(xbase, (unsigned)(3 * xps * 2));
  const long dps = (long)p.Cout * p.P;
  const rsrc_t d_rsrc = mk_rsrc(p.dyt, (unsigned)(3 * dps * 2));

  // B stream of this wave: channel tile ct of the source, vectors r = p / 8 (512 B each), shifted by dy rows
  const int ct = (nl0 >> 5) + wid;
  const bool n_live = (n0 + wid * 32 < p.Cin) & (ct * 32 < Cs);
  const int R = p.P >> 3;
  const int rshift = dy * p.wr;
  unsigned b_off[3];
#pragma unroll
  for (int part = 0; part < 3; ++part)
    b_off[part] = (unsigned)(((long)ct * R * 32 + li) * 16 + part * xps * 2);
  auto load_b = [&](u32x4(&rb)[6], int kc) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int r = kc * 4 + 2 * s + lh;  // this lane's 8-pixel vector of the chunk
      const int irow = r >> p.wr_shift;                                                    // image row, all images
      const int img = (int)(((unsigned long long)(unsigned)irow * p.magic_hw) >> 40);      // irow / H
      const int y = irow - img * p.H;
      const bool ok = n_live & (kc < kc_end) & (r < R) & ((unsigned)(y + dy) < (unsigned)p.H);
      const unsigned o = (unsigned)(r + rshift) * 512u;
#pragma unroll
      for (int part = 0; part < 3; ++part) rb[part * 2 + s] = ld16(x_rsrc, ok ? b_off[part] + o : OOBS);
    }
  };

  int a_base[2];  // row * P, or -1
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int co = m0 + srow + 64 * i;
    a_base[i] = co < p.Cout ? co * p.P : -1;
  }
  u32x4 ra[6];
  auto issue_a = [&](int kc) {
    const int px = kc * SBK + schunk * 8;
    const bool pok = (kc < kc_end) & (px < p.P);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool oka = pok & (a_base[i] >= 0);
      const unsigned oa = (unsigned)(a_base[i] + px) * 2u;
#pragma unroll
      for (int part = 0; part < 3; ++part) ra[part * 2 + i] = ld16(d_rsrc, oka ? oa + (unsigned)(part * dps * 2) : OOBS);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<u32x4*>(lds_raw + buf * ABUF + part * PLANE + (srow + 64 * i) * BD_ROW + schunk * 16) =
            ra[part * 2 + i];
  };
  int abase[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) abase[t] = (t * 32 + li) * BD_ROW + lh * 16;

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (kc_begin < kc_end) {
    u32x4 b0[6], b1[6], b2[6];
    issue_a(kc_begin);
    load_b(b0, kc_begin);
    load_b(b1, kc_begin + 1);
    store_a(0);
    issue_a(kc_begin + 1);
    __syncthreads();

    // chunk kc: dy^T from LDS buffer (kc - kc_begin) & 1, inputs from `rb`; `ra` holds chunk kc + 1 on entry
    auto step = [&](const u32x4(&rb)[6], int kc) {
      const int cur = (kc - kc_begin) & 1;
      store_a(cur ^ 1);   // chunk kc + 1 (its buffer was last read in chunk kc - 1, before the previous barrier)
      issue_a(kc + 2);
      const int aoff = cur * ABUF;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 fa[4][3], fb[3];
#pragma unroll
        for (int part = 0; part < 3; ++part) fb[part] = __builtin_bit_cast(bf16x8, rb[part * 2 + s]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int part = 0; part < 3; ++part)
            fa[t][part] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const u32x4*>(lds_raw + aoff + abase[t] + part * PLANE + s * 32));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          f32x16 c = acc[mt];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][2], fb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][1], fb[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][0], fb[0], c, 0, 0, 0);
          acc[mt] = c;
        }
      }
      __syncthreads();
    };

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

  // dw[co][tap][ci] += acc   (C/D layout: col = lane & 31 -> ci, row -> co): this wave owns ci n0 + 32 wid .. + 31
  const long wrow = (long)p.taps * p.Cin;
  const int n = n0 + wid * 32 + li;
  if (n < p.Cin && !(first && n >= p.a_split)) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.Cout) continue;
        float* dst = p.dw + m * wrow + (long)tap * p.Cin + n;
        if (p.split_k > 1)
          atomicAdd(dst, acc[mt][r]);
        else
          *dst += acc[mt][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// wgrad_split_bdirect_kernel on v_mfma_f32_16x16x32_bf16 (x_layout 2).  dy^T comes in TILE ORDER
// (rac_transpose_split layout 2): [part][Cout/128][P/32][8-pixel group 0..3][co mod 128][8 pixels], i.e. every
// (Cout tile, K chunk, part) is one contiguous 8 KB block that IS the chunk-major LDS image of the A operand
// (conflict-free for the 16-lane read groups of this instruction): staging is a linear, fully coalesced copy.
// x^T stays in the fragment order of the 32x32x16 kernel; a lane loads column ci mod 16 (+16 nb) of pixel vector
// 4 kc + (l >> 4).  8 x 2 accumulators of 16x16 per wave.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad_split_bdirect16_kernel(WgradSplitP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];  // 2 x [part 3][group 4][128 rows][16 B]
  constexpr int PPL = 4 * 128 * 16, ABUF = 3 * PPL;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * SBM;
  const int tap = blockIdx.y / p.ntile_per_tap;
  const int n0 = (blockIdx.y - tap * p.ntile_per_tap) * SBN;
  const int ky = tap / p.ks, kx = tap - ky * p.ks;
  const int dy = ky - p.pad;
  const int kc_begin = blockIdx.z * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);

  // block-uniform source of this channel range (virtual concat [x0 | x1])
  const bool first = n0 < p.a_split;
  const int Cs = first ? p.a_split : p.Cin - p.a_split;
  const int nl0 = first ? n0 : n0 - p.a_split;
  const long xps = (long)Cs * p.P;
  const unsigned short* xbase = (first ? p.x0t : p.x1t) + (long)kx * 3 * xps;
  const rsrc_t x_rsrc = mk_rsrc(xbase, (unsigned)(3 * xps * 2));
  const long dps = (long)p.Cout * p.P;
  const rsrc_t d_rsrc = mk_rsrc(p.dyt, (unsigned)(3 * dps * 2));

  const int ct = (nl0 >> 5) + wid;
  const bool n_live = (n0 + wid * 32 < p.Cin) & (ct * 32 < Cs);
  const int R = p.P >> 3;
  const int rshift = dy * p.wr;
  unsigned b_off[3];
#pragma unroll
  for (int part = 0; part < 3; ++part) b_off[part] = (unsigned)(((long)ct * R * 32 + lr) * 16 + part * xps * 2);
  auto load_b = [&](u32x4(&rb)[6], int kc) {
    const int r = kc * 4 + lq;  // this lane's 8-pixel vector of the chunk
    const int irow = r >> p.wr_shift;
    const int img = (int)(((unsigned long long)(unsigned)irow * p.magic_hw) >> 40);
    const int y = irow - img * p.H;
    const bool ok = n_live & (kc < kc_end) & (r < R) & ((unsigned)(y + dy) < (unsigned)p.H);
    const unsigned o = (unsigned)(r + rshift) * 512u;
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) rb[part * 2 + nb] = ld16(x_rsrc, ok ? b_off[part] + o + nb * 256u : OOBS);
  };

  // dy^T: the (Cout tile, chunk, part) block of 512 vectors, two per thread
  const unsigned a_tile = (unsigned)(((long)(m0 >> 7) * p.nchunks) * 8192);
  u32x4 ra[6];
  auto issue_a = [&](int kc) {
    const bool ok = kc < kc_end;
    const unsigned o = a_tile + (unsigned)kc * 8192u + (unsigned)tid * 16u;
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        ra[part * 2 + i] = ld16(d_rsrc, ok ? o + i * 4096u + (unsigned)(part * dps * 2) : OOBS);
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<u32x4*>(lds_raw + buf * ABUF + part * PPL + i * 4096 + tid * 16) = ra[part * 2 + i];
  };
  const int abase = lq * 2048 + lr * 16;

  f32x4 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kc_begin < kc_end) {
    u32x4 b0[6], b1[6], b2[6];
    issue_a(kc_begin);
    load_b(b0, kc_begin);
    load_b(b1, kc_begin + 1);
    store_a(0);
    issue_a(kc_begin + 1);
    __syncthreads();

    auto step = [&](const u32x4(&rb)[6], int kc) {
      const int cur = (kc - kc_begin) & 1;
      store_a(cur ^ 1);  // chunk kc + 1
      issue_a(kc + 2);
      const int aoff = cur * ABUF + abase;
      bf16x8 fb[2][3];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int part = 0; part < 3; ++part) fb[nb][part] = __builtin_bit_cast(bf16x8, rb[part * 2 + nb]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8 fa[4][3];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int part = 0; part < 3; ++part)
            fa[t][part] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const u32x4*>(lds_raw + aoff + (4 * h + t) * 256 + part * PPL));
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x4 c = acc[4 * h + t][nb];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][2], fb[nb][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][1], fb[nb][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][0], fb[nb][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][1], fb[nb][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][0], fb[nb][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][0], fb[nb][0], c, 0, 0, 0);
            acc[4 * h + t][nb] = c;
          }
      }
      __syncthreads();
    };

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

  // dw[co][tap][ci] += acc: col = l & 15 (+16 nb) -> ci, rows 4 (l >> 4) + reg of each 16-row block -> co
  const long wrow = (long)p.taps * p.Cin;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = n0 + wid * 32 + nb * 16 + lr;
    if (n >= p.Cin || (first && n >= p.a_split)) continue;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mb * 16 + 4 * lq + r;
        if (m >= p.Cout) continue;
        float* dst = p.dw + m * wrow + (long)tap * p.Cin + n;
        if (p.split_k > 1)
          atomicAdd(dst, acc[mb][nb][r]);
        else
          *dst += acc[mb][nb][r];
      }
  }
}

// out[dxi][part][c][p] = part-th bf16 part of (x in-row ? in[p + dx][c] : 0), dx = dxi - pad, dxi < ndx
// (ndx = 1: plain transpose + split).  32x32 tiles through LDS: coalesced on both sides.
__global__ void transpose_split_kernel(const float* in, unsigned short* out, int P, int C, int W, int ndx, int pad,
                                       long ld) {
  __shared__ float tile[32][33];
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int dxi = blockIdx.z;
  const int dx = dxi - pad;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pr = p0 + ty + 8 * i;  // output pixel
    const int c = c0 + tx;
    float v = 0.f;
    if (pr < P && c < C) {
      const int x = pr % W;
      if ((unsigned)(x + dx) < (unsigned)W) v = in[(long)(pr + dx) * C + c];
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
  const long ps = (long)C * ld;  // ld >= P: row stride of the (possibly wider, time-batched) output
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;
    const int pr = p0 + tx;
    if (c < C && pr < P) {
      const float a = tile[tx][ty + 8 * i];
      const __bf16 q1 = (__bf16)a;
      const float r1 = a - (float)q1;
      const __bf16 q2 = (__bf16)r1;
      const __bf16 q3 = (__bf16)(r1 - (float)q2);
      unsigned short* o = out + (long)dxi * 3 * ps + (long)c * ld + pr;
      o[0] = __builtin_bit_cast(unsigned short, q1);
      o[ps] = __builtin_bit_cast(unsigned short, q2);
      o[2 * ps] = __builtin_bit_cast(unsigned short, q3);
    }
  }
}

// Fragment-order variant: out[dxi][part][c / 32][r = (p_off + p) / 8][c mod 32][8 pixels], ld = pixels per row of the
// whole (time-batched) operand.  One 32 x 32 tile = four 512-byte runs [c mod 32][8 pixels], written 2 B per lane
// with the 256 threads covering one run contiguously.
__global__ void transpose_split_frag_kernel(const float* in, unsigned short* out, int P, int C, int W, int ndx, int pad,
                                            long ld, long p_off) {
  __shared__ float tile[32][33];
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int dxi = blockIdx.z;
  const int dx = dxi - pad;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pr = p0 + ty + 8 * i;
    const int c = c0 + tx;
    float v = 0.f;
    if (pr < P && c < C) {
      const int x = pr % W;
      if ((unsigned)(x + dx) < (unsigned)W) v = in[(long)(pr + dx) * C + c];
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
  const long ps = (long)C * ld;
  const int j = threadIdx.x & 7, cl = threadIdx.x >> 3;  // pixel within the vector, channel within the tile
  if (c0 + cl >= C) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pr = p0 + 8 * i + j;
    if (pr >= P) continue;
    const float a = tile[8 * i + j][cl];
    const __bf16 q1 = (__bf16)a;
    const float r1 = a - (float)q1;
    const __bf16 q2 = (__bf16)r1;
    const __bf16 q3 = (__bf16)(r1 - (float)q2);
    const long r = (p_off + pr) >> 3;
    unsigned short* o = out + (long)dxi * 3 * ps + (long)(c0 >> 5) * (ld * 32) + r * 256 + cl * 8 + j;
    o[0] = __builtin_bit_cast(unsigned short, q1);
    o[ps] = __builtin_bit_cast(unsigned short, q2);
    o[2 * ps] = __builtin_bit_cast(unsigned short, q3);
  }
}

// Tile-order variant for dy^T (layout 2, ndx = 1): out[part][c / 128][(p_off + p) / 32][(p mod 32) / 8][c mod 128][8],
// the chunk-major A image of wgrad_split_bdirect16_kernel.  One 32 x 32 tile = four 512-byte runs.
__global__ void transpose_split_tile_kernel(const float* in, unsigned short* out, int P, int C, long ld, long p_off) {
  __shared__ float tile[32][33];
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pr = p0 + ty + 8 * i;
    const int c = c0 + tx;
    tile[ty + 8 * i][tx] = (pr < P && c < C) ? in[(long)pr * C + c] : 0.f;
  }
  __syncthreads();
  const long ps = (long)C * ld;
  const int j = threadIdx.x & 7, cl = threadIdx.x >> 3;
  if (c0 + cl >= C) return;
  const long blk = ((long)(c0 >> 7) * (ld >> 5) + ((p_off + p0) >> 5)) * 4096;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float a = tile[8 * g + j][cl];
    unsigned short q1, q2, q3;
    split3(a, q1, q2, q3);
    unsigned short* o = out + blk + g * 1024 + ((c0 & 127) + cl) * 8 + j;
    o[0] = q1;
    o[ps] = q2;
    o[2 * ps] = q3;
  }
}

__global__ void split_bf16x3_kernel(const float* x, unsigned short* parts, long n, long ps) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned short q1, q2, q3;
    split3(x[i], q1, q2, q3);
    parts[i] = q1;
    parts[ps + i] = q2;
    parts[2 * ps + i] = q3;
  }
}

typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
// n % 4 == 0, ps % 4 == 0, 16-byte aligned x / 8-byte aligned parts: four values per lane
__global__ void split_bf16x3_vec4_kernel(const float4* x, u16x4* parts, long n4, long ps4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 a = x[i];
    u16x4 o1, o2, o3;
    unsigned short q1, q2, q3;
    split3(a.x, q1, q2, q3); o1.x = q1; o2.x = q2; o3.x = q3;
    split3(a.y, q1, q2, q3); o1.y = q1; o2.y = q2; o3.y = q3;
    split3(a.z, q1, q2, q3); o1.z = q1; o2.z = q2; o3.z = q3;
    split3(a.w, q1, q2, q3); o1.w = q1; o2.w = q2; o3.w = q3;
    parts[i] = o1;
    parts[ps4 + i] = o2;
    parts[2 * ps4 + i] = o3;
  }
}

// Conv weight [Cout][taps][Cin] fp32 -> bf16 parts in MFMA fragment order (weight layout 2 of rac_conv2d_fwd_split):
//   out[part][R/32][Kc/32][tap][s][lane = 32 h + r mod 32][j],  k = 32 chunk + 16 s + 8 h + j
// forward:    rows r = co, k = ci, value w[r][tap][k]
// transposed: rows r = ci, k = co, value w[k][taps-1-tap][r]   (the conv that IS the data gradient)
// One workgroup = two (row tile, k chunk, tap) cells: every wave writes 1 KB contiguous per part.
__global__ void weight_frag_split_kernel(const float* w, unsigned short* out, int Cout, int Cin, int taps, int transposed,
                                         long ps, int mfma16) {
  const int R = transposed ? Cin : Cout, Kc = transposed ? Cout : Cin;
  const int cch = Kc >> 5;
  const long cell = (long)blockIdx.x * 2 + (threadIdx.x >> 7);  // ((nt * cch + cc) * taps + tap)
  if (cell >= (long)(R >> 5) * cch * taps) return;
  const int tap = (int)(cell % taps);
  const int cc = (int)((cell / taps) % cch);
  const int nt = (int)(cell / ((long)taps * cch));
  const int s = (threadIdx.x >> 6) & 1, lane = threadIdx.x & 63;
  // 32x32x16 operand: lane = 32 h + r mod 32, k = 16 s + 8 h + j;  16x16x32 operand (layout 3): s = 16-column half,
  // lane = 16 q + r mod 16, k = 8 q + j
  const int r = mfma16 ? nt * 32 + s * 16 + (lane & 15) : nt * 32 + (lane & 31);
  const int k0 = mfma16 ? cc * 32 + 8 * (lane >> 4) : cc * 32 + 16 * s + 8 * (lane >> 5);
  float v[8];
  if (!transposed) {
    const float4* src = reinterpret_cast<const float4*>(w + ((long)r * taps + tap) * Cin + k0);
    const float4 a = src[0], b = src[1];
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w[((long)(k0 + j) * taps + (taps - 1 - tap)) * Cin + r];
  }
  unsigned short q[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(v[j], q[0][j], q[1][j], q[2][j]);
  const long o = (cell * 2 + s) * 512 + lane * 8;
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    u32x4 pk;
    pk.x = q[part][0] | ((unsigned)q[part][1] << 16);
    pk.y = q[part][2] | ((unsigned)q[part][3] << 16);
    pk.z = q[part][4] | ((unsigned)q[part][5] << 16);
    pk.w = q[part][6] | ((unsigned)q[part][7] << 16);
    *reinterpret_cast<u32x4*>(out + part * ps + o) = pk;
  }
}

}  // namespace rac

using namespace rac;

extern "C" int rac_split_bf16x3(const float* x, uint16_t* parts, int64_t n, int64_t part_stride, void* stream) {
  RAC_REQUIRE(x && parts && n > 0 && part_stride >= n, "rac_split_bf16x3: bad args");
  if (n % 4 == 0 && part_stride % 4 == 0 && aligned16(x) && (reinterpret_cast<uintptr_t>(parts) & 7) == 0) {
    long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(split_bf16x3_vec4_kernel, dim3((int)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<u16x4*>(parts), (long)(n / 4),
                       (long)(part_stride / 4));
    return check_launch("rac_split_bf16x3");
  }
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3((int)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, parts,
                     (long)n, (long)part_stride);
  return check_launch("rac_split_bf16x3");
}

extern "C" int rac_weight_frag_split(const float* w, uint16_t* parts, int32_t Cout, int32_t Cin, int32_t ksize,
                                     int32_t transposed, int64_t part_stride, int32_t w_layout, void* stream) {
  RAC_REQUIRE(w_layout == 2 || w_layout == 3, "rac_weight_frag_split: w_layout must be 2 or 3");
  RAC_REQUIRE(w && parts && Cout > 0 && Cin > 0 && ksize >= 1 && (ksize & 1), "rac_weight_frag_split: bad args");
  RAC_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0, "rac_weight_frag_split: channel counts must be multiples of 32");
  const long n = (long)Cout * Cin * ksize * ksize;
  RAC_REQUIRE(part_stride >= n && part_stride % 8 == 0 && aligned16(w) && aligned16(parts),
              "rac_weight_frag_split: part stride / alignment");
  const long cells = n / 1024;  // (row tile, k chunk, tap) cells of 32 x 32 weights
  hipLaunchKernelGGL(weight_frag_split_kernel, dim3((unsigned)((cells + 1) / 2)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), w, parts, Cout, Cin, ksize * ksize, transposed,
                     (long)part_stride, w_layout == 3 ? 1 : 0);
  return check_launch("rac_weight_frag_split");
}

extern "C" int rac_conv2d_fwd_split(const rac_conv_args* a, int64_t a0_ps, int64_t a1_ps, int64_t w_ps,
                                    int32_t w_layout, void* stream) {
  RAC_REQUIRE(a && a->mode == RAC_CONV_FWD, "rac_conv2d_fwd_split: forward mode only");
  RAC_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0 && a->a0 && a->w && a->out0,
              "rac_conv2d_fwd_split: bad args");
  RAC_REQUIRE(a->ksize >= 1 && (a->ksize & 1) && a->ksize <= 7, "rac_conv2d_fwd_split: ksize must be odd");
  SplitP p{};
  p.B = a->B, p.H = a->H, p.W = a->W, p.ks = a->ksize, p.pad = a->ksize / 2;
  p.Cin = a->Cin, p.Cout = a->Cout, p.act = a->act;
  p.split_k = a->split_k > 1 ? a->split_k : 1;
  p.slab_stride = a->slab_stride;
  p.a0 = reinterpret_cast<const unsigned short*>(a->a0);
  p.a1 = reinterpret_cast<const unsigned short*>(a->a1);
  p.w = reinterpret_cast<const unsigned short*>(a->w);
  p.a0_ps = a0_ps, p.a1_ps = a1_ps, p.w_ps = w_ps;
  p.out0 = a->out0;
  p.bias = a->bias, p.scale = a->scale, p.shift = a->shift, p.stats = a->stats;
  p.stats_rows = a->stats ? a->stats_rows : 0;
  RAC_REQUIRE(p.stats_rows >= 0 && (p.stats_rows % 128 == 0 || w_layout == 3) &&
                  (p.stats_rows == 0 || ((long)a->B * a->H * a->W) % p.stats_rows == 0),
              "rac_conv2d_fwd_split: stats_rows must be a multiple of 128 that divides B*H*W");
  p.HW = a->H * a->W;
  p.P = a->B * p.HW;
  p.M = p.P, p.N = a->Cout;
  p.taps = a->ksize * a->ksize;
  p.a_split = (a->a1 && a->a_split > 0 && a->a_split < a->Cin) ? a->a_split : a->Cin;
  RAC_REQUIRE(a->Cin % 8 == 0 && p.a_split % 8 == 0, "rac_conv2d_fwd_split: channel counts must be multiples of 8");
  RAC_REQUIRE(p.a_split == a->Cin || p.a_split % SBK == 0, "rac_conv2d_fwd_split: a_split must be a multiple of 32");
  RAC_REQUIRE(aligned16(a->a0) && aligned16(a->w) && (!a->a1 || aligned16(a->a1)), "rac_conv2d_fwd_split: alignment");
  if (w_layout >= 2) {  // fragment-order weights: the activations are fp32 [pixel][channel], split inside the kernel
    RAC_REQUIRE(a0_ps == 0 && a1_ps == 0, "rac_conv2d_fwd_split: w_layout 2 / 3 take fp32 activations (part strides 0)");
    RAC_REQUIRE((long)p.P * (p.a_split > a->Cin - p.a_split ? p.a_split : a->Cin - p.a_split) * 4 < 0xFFFFFF00L,
                "rac_conv2d_fwd_split: operand larger than 4 GiB");
  } else {
    RAC_REQUIRE(a0_ps >= (long)p.P * p.a_split && (p.a_split == a->Cin || a1_ps >= (long)p.P * (a->Cin - p.a_split)),
                "rac_conv2d_fwd_split: part strides too small");
    RAC_REQUIRE(3 * a0_ps * 2 < 0xFFFFFF00L && 3 * a1_ps * 2 < 0xFFFFFF00L,
                "rac_conv2d_fwd_split: operand larger than 4 GiB");
  }
  RAC_REQUIRE(w_ps >= (long)p.Cout * p.taps * p.Cin, "rac_conv2d_fwd_split: weight part stride too small");
  RAC_REQUIRE(3 * w_ps * 2 < 0xFFFFFF00L, "rac_conv2d_fwd_split: operand larger than 4 GiB");
  RAC_REQUIRE(p.split_k == 1 || a->slab_stride >= (long)p.M * p.N, "rac_conv2d_fwd_split: slab_stride too small");
  p.cchunks = cdiv(a->Cin, SBK);
  p.nchunks = p.taps * p.cchunks;
  p.cps = cdiv(p.nchunks, p.split_k);
  constexpr size_t lds = 2 * 3 * 128 * 4 * 16;  // 49,152 B
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_split_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return RAC_ELAUNCH;
    }
    attr_done = true;
  }
  dim3 grid(cdiv(p.M, SBM), cdiv(p.N, SBN), p.split_k);
  static const bool no_tapinner = getenv("RAC_SPLIT_TAPOUTER") != nullptr;  // A/B switch for benchmarks
  p.w_chunk_major = 0;
  RAC_REQUIRE(w_layout >= 0 && w_layout <= 3, "rac_conv2d_fwd_split: w_layout must be 0 .. 3");
  const bool w_chunk_major = w_layout == 1;
  RAC_REQUIRE(w_layout == 0 || (w_layout >= 2 && p.HW > SBM) ||
                  (a->Cin % SBK == 0 && p.HW <= SBM && (SBM % p.HW == 0 || w_layout == 3) && p.taps > 1),
              "rac_conv2d_fwd_split: chunk-major / fragment-order weights need Cin % 32 == 0, k > 1 and whole images "
              "per 128-pixel tile");
  // XCD x runs the (N-tile, K-slab) columns x, x+8, ...: with few M-tiles (training batch) the launch is otherwise
  // bound by weight re-reads across the 8 L2s (2.79 -> 0.42 GB per launch); at M = 64 000 it still trims the fabric
  // traffic (12.2 -> 10.1 GB) at equal or slightly better speed
  static const char* xg = getenv("RAC_XCD_GROUP");
  p.xcd_group = (xg ? atoi(xg) != 0 : 1) && grid.x > 1 && (grid.y * grid.z) % 8 == 0;
  if (w_layout >= 2 && p.HW > SBM) {
    // maps larger than a tile: whole image rows per tile plus a halo (igemm_split_bdirect_rows*_kernel)
    RAC_REQUIRE(a->Cout % 32 == 0 && a->ksize <= 5 && a->Cin % SBK == 0 && p.taps > 1,
                "rac_conv2d_fwd_split: fragment-order weights on large maps need channel counts % 32 == 0, 1 < k <= 5");
    int rows = 0;  // image rows per tile: R | H, R * W <= 128; the 32x32x16 form needs exactly 128 pixels
    for (int r = SBM / a->W; r >= 1; --r)
      if (a->H % r == 0 && (r * a->W) % 16 == 0) {
        rows = r;
        break;
      }
    p.tile_m = rows * a->W;
    RAC_REQUIRE(rows > 0 && a->W <= SBM && (w_layout == 3 || p.tile_m == SBM),
                "rac_conv2d_fwd_split: no whole-row tile for this map (W must be <= 128; layout 2 needs W | 128 and "
                "H a multiple of 128 / W)");
    RAC_REQUIRE(p.stats_rows % p.tile_m == 0, "rac_conv2d_fwd_split: stats_rows must be a multiple of the tile rows");
    RAC_REQUIRE((long)(a->Cout / 32) * p.nchunks * 2048L < 0xFFFFFF00L, "rac_conv2d_fwd_split: weight part too large");
    grid.x = cdiv(p.M, p.tile_m);
    const int nrows = p.tile_m + 2 * p.pad * a->W;
    const int nv = cdiv(nrows * 4, 256) < 2 ? 2 : cdiv(nrows * 4, 256);
    RAC_REQUIRE(nv <= 4, "rac_conv2d_fwd_split: halo too large for the LDS image");
    const bool m16 = w_layout == 3;
    const size_t lds_rows = m16 ? (size_t)3 * 4 * (nrows + 16) * 16 : (size_t)3 * (nrows + 1) * BD_ROW;
    static bool rows_attr = false;
    typedef void (*rows_fn)(SplitP);
    static const rows_fn fns[4][3] = {
        {igemm_split_bdirect_rows_kernel<2, 4>, igemm_split_bdirect_rows_kernel<3, 4>, igemm_split_bdirect_rows_kernel<4, 4>},
        {igemm_split_bdirect_rows_kernel<2, 2>, igemm_split_bdirect_rows_kernel<3, 2>, igemm_split_bdirect_rows_kernel<4, 2>},
        {igemm_split_bdirect_rows16_kernel<2, 4>, igemm_split_bdirect_rows16_kernel<3, 4>,
         igemm_split_bdirect_rows16_kernel<4, 4>},
        {igemm_split_bdirect_rows16_kernel<2, 2>, igemm_split_bdirect_rows16_kernel<3, 2>,
         igemm_split_bdirect_rows16_kernel<4, 2>}};
    if (!rows_attr) {
      const int max_lds = 3 * (256 + 1) * BD_ROW;
      for (int i = 0; i < 12; ++i) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[i / 3][i % 3]),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        if (e != hipSuccess) {
          set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
          return RAC_ELAUNCH;
        }
      }
      rows_attr = true;
    }
    // 64-channel layers: 64-column workgroups (waves 2 x 2) instead of half-empty 128-column ones
    const int narrow = p.N <= 64 ? 1 : 0;
    if (narrow) grid.y = cdiv(p.N, 64);
    p.xcd_group = p.xcd_group && (grid.y * grid.z) % 8 == 0;
    hipLaunchKernelGGL(fns[narrow + (m16 ? 2 : 0)][nv - 2], grid, dim3(256), lds_rows,
                       reinterpret_cast<hipStream_t>(stream), p);
    return check_launch("rac_conv2d_fwd_split(weights direct, image rows)");
  }
  if (w_layout == 3) {
    RAC_REQUIRE(a->Cout % 32 == 0 && a->ksize <= 5,
                "rac_conv2d_fwd_split: fragment-order weights need Cout % 32 == 0 and k <= 5");
    p.tile_m = (SBM / p.HW) * p.HW;  // whole images per workgroup
    RAC_REQUIRE(p.tile_m % 16 == 0, "rac_conv2d_fwd_split: w_layout 3 needs (128 / (H*W)) * H*W to be a multiple of 16");
    RAC_REQUIRE(p.stats_rows % p.tile_m == 0, "rac_conv2d_fwd_split: stats_rows must be a multiple of the tile rows");
    grid.x = cdiv(p.M, p.tile_m);
    p.xcd_group = p.xcd_group && grid.x > 1;
    RAC_REQUIRE((long)(a->Cout / 32) * p.nchunks * 2048L < 0xFFFFFF00L, "rac_conv2d_fwd_split: weight part too large");
    constexpr size_t lds_b16 = 2 * B16_ABUF;  // 55,296 B
    static bool b16_attr = false;
    if (!b16_attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_split_bdirect16_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b16);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
        return RAC_ELAUNCH;
      }
      b16_attr = true;
    }
    hipLaunchKernelGGL(igemm_split_bdirect16_kernel, grid, dim3(256), lds_b16, reinterpret_cast<hipStream_t>(stream), p);
    return check_launch("rac_conv2d_fwd_split(weights direct, 16x16x32)");
  }
  if (w_layout == 2) {
    RAC_REQUIRE(a->Cout % 32 == 0 && a->ksize <= 5,
                "rac_conv2d_fwd_split: fragment-order weights need Cout % 32 == 0 and k <= 5");
    RAC_REQUIRE((long)(a->Cout / 32) * p.nchunks * 2048L < 0xFFFFFF00L, "rac_conv2d_fwd_split: weight part too large");
    constexpr size_t lds_bd = 2 * BD_ABUF;  // two activation buffers incl. their zero lines = 61,920 B
    static bool bd_attr = false;
    if (!bd_attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_split_bdirect_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bd);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
        return RAC_ELAUNCH;
      }
      bd_attr = true;
    }
    hipLaunchKernelGGL(igemm_split_bdirect_kernel, grid, dim3(256), lds_bd, reinterpret_cast<hipStream_t>(stream), p);
    return check_launch("rac_conv2d_fwd_split(weights direct)");
  }
  if ((w_chunk_major || !no_tapinner) && p.HW <= SBM && SBM % p.HW == 0 && p.taps > 1) {
    p.w_chunk_major = w_chunk_major;
    // whole images per M-tile: activations staged once per channel chunk, taps inner
    constexpr size_t lds_ti = (3 * 3 * 128 * 4 + 4) * 16;  // A + 2 x B + zero line = 73,792 B
    static bool ti_attr = false;
    if (!ti_attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_split_tapinner_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ti);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
        return RAC_ELAUNCH;
      }
      ti_attr = true;
    }
    hipLaunchKernelGGL(igemm_split_tapinner_kernel, grid, dim3(256), lds_ti, reinterpret_cast<hipStream_t>(stream), p);
    return check_launch("rac_conv2d_fwd_split(tap-inner)");
  }
  hipLaunchKernelGGL(igemm_split_kernel, grid, dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
  return check_launch("rac_conv2d_fwd_split");
}

extern "C" int rac_transpose_split(const float* x, uint16_t* out, int32_t P, int32_t C, int32_t W, int32_t ndx,
                                   int64_t ld, int32_t layout, int64_t p_off, void* stream) {
  RAC_REQUIRE(x && out && P > 0 && C > 0 && W > 0 && ndx >= 1 && (ndx & 1) && (ld == 0 || ld >= P) && p_off >= 0,
              "rac_transpose_split: bad args");
  const long ldl = ld ? (long)ld : (long)P;
  if (layout == 1) {
    RAC_REQUIRE(C % 32 == 0 && W % 8 == 0 && P % 8 == 0 && ldl % 8 == 0 && p_off % 8 == 0 && p_off + P <= ldl,
                "rac_transpose_split: fragment order needs C % 32 == 0 and 8-pixel aligned rows");
    hipLaunchKernelGGL(transpose_split_frag_kernel, dim3(cdiv(P, 32), cdiv(C, 32), ndx), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, out, P, C, W, ndx, ndx / 2, ldl, (long)p_off);
    return check_launch("rac_transpose_split(fragment order)");
  }
  if (layout == 2) {
    RAC_REQUIRE(ndx == 1 && C % 128 == 0 && P % 32 == 0 && ldl % 32 == 0 && p_off % 32 == 0 && p_off + P <= ldl,
                "rac_transpose_split: tile order is for dy (ndx 1), C % 128 == 0 and 32-pixel aligned ranges");
    hipLaunchKernelGGL(transpose_split_tile_kernel, dim3(cdiv(P, 32), cdiv(C, 32)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, out, P, C, ldl, (long)p_off);
    return check_launch("rac_transpose_split(tile order)");
  }
  RAC_REQUIRE(layout == 0, "rac_transpose_split: layout must be 0, 1 or 2");
  hipLaunchKernelGGL(transpose_split_kernel, dim3(cdiv(P, 32), cdiv(C, 32), ndx), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, out + p_off, P, C, W, ndx, ndx / 2, ldl);
  return check_launch("rac_transpose_split");
}

extern "C" int rac_conv2d_wgrad_split(const rac_conv_args* a, int32_t x_layout, void* stream) {
  RAC_REQUIRE(a && a->mode == RAC_CONV_WGRAD, "rac_conv2d_wgrad_split: weight-gradient mode only");
  RAC_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0 && a->a0 && a->w && a->out0,
              "rac_conv2d_wgrad_split: bad args");
  RAC_REQUIRE(a->ksize >= 1 && (a->ksize & 1) && a->ksize <= 7, "rac_conv2d_wgrad_split: ksize must be odd");
  RAC_REQUIRE(a->W % 8 == 0, "rac_conv2d_wgrad_split: image width must be a multiple of 8 (16-byte pixel vectors)");
  WgradSplitP p{};
  p.H = a->H, p.W = a->W, p.ks = a->ksize, p.pad = a->ksize / 2;
  p.Cin = a->Cin, p.Cout = a->Cout;
  p.HW = a->H * a->W;
  p.P = a->B * p.HW;
  p.taps = a->ksize * a->ksize;
  p.a_split = (a->a1 && a->a_split > 0 && a->a_split < a->Cin) ? a->a_split : a->Cin;
  RAC_REQUIRE(p.a_split == a->Cin || p.a_split % SBN == 0,
              "rac_conv2d_wgrad_split: a_split must be a multiple of 128 (block-uniform source)");
  p.x0t = reinterpret_cast<const unsigned short*>(a->a0);
  p.x1t = reinterpret_cast<const unsigned short*>(a->a1);
  p.dyt = reinterpret_cast<const unsigned short*>(a->w);
  p.dw = a->out0;
  RAC_REQUIRE(aligned16(a->a0) && aligned16(a->w) && (!a->a1 || aligned16(a->a1)), "rac_conv2d_wgrad_split: alignment");
  const long big = (long)3 * p.P * (a->Cout > a->Cin ? a->Cout : a->Cin) * 2;
  RAC_REQUIRE(big < 0xFFFFFF00L, "rac_conv2d_wgrad_split: operand larger than 4 GiB");
  p.magic_hw = ((1ULL << 40) + p.HW - 1) / p.HW;
  p.magic_w = ((1ULL << 40) + a->W - 1) / a->W;
  p.nchunks = cdiv(p.P, SBK);
  p.ntile_per_tap = cdiv(a->Cin, SBN);
  const long tiles = (long)cdiv(a->Cout, SBM) * p.ntile_per_tap * p.taps;
  int split = a->split_k;
  if (split <= 0) {  // auto: atomics when the tile count cannot fill the chip, or to even out the last round
    split = 1;
    if (tiles < 512) {
      split = (int)((768 + tiles - 1) / tiles);
    } else {
      // 512 workgroups run at a time (2 per CU): pick the K split whose last round is fullest, while every
      // workgroup keeps >= 32 chunks to amortise its read-modify-write of the 64 KB output tile
      double best = 0.;
      for (int sk = 1; sk <= 4 && p.nchunks / sk >= 32; ++sk) {
        const double rounds = (double)tiles * sk / 512.;
        const double eff = rounds / (double)(long)(rounds + 0.999999) - 0.04 * (sk - 1);
        if (eff > best) best = eff, split = sk;
      }
    }
    if (split > p.nchunks / 8) split = p.nchunks / 8 > 0 ? p.nchunks / 8 : 1;
    if (split > 64) split = 64;
    static const char* fk = getenv("RAC_WGRAD_SPLITK");  // experiments
    if (fk && atoi(fk) > 0) split = atoi(fk);
  }
  p.split_k = split;
  p.cps = cdiv(p.nchunks, split);
  constexpr size_t lds = 2 * 3 * 128 * 4 * 16;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return RAC_ELAUNCH;
    }
    attr_done = true;
  }
  dim3 grid(cdiv(a->Cout, SBM), p.ntile_per_tap * p.taps, split);
  if (x_layout == 2) {
    const int c1 = a->Cin - p.a_split;
    RAC_REQUIRE(p.a_split % 32 == 0 && c1 % 32 == 0 && p.P % 32 == 0 && a->Cout % 128 == 0,
                "rac_conv2d_wgrad_split: x_layout 2 needs channel counts % 32 == 0, Cout % 128 == 0, pixels % 32 == 0");
    p.wr = a->W / 8;
    RAC_REQUIRE((p.wr & (p.wr - 1)) == 0, "rac_conv2d_wgrad_split: fragment-order inputs need W / 8 a power of two");
    p.wr_shift = __builtin_ctz((unsigned)p.wr);
    p.magic_hw = ((1ULL << 40) + a->H - 1) / a->H;
    constexpr size_t lds_b16 = 2 * 3 * 4 * 128 * 16;  // 49,152 B
    static bool b16_attr = false;
    if (!b16_attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_bdirect16_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b16);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
        return RAC_ELAUNCH;
      }
      b16_attr = true;
    }
    hipLaunchKernelGGL(wgrad_split_bdirect16_kernel, grid, dim3(256), lds_b16, reinterpret_cast<hipStream_t>(stream), p);
    return check_launch("rac_conv2d_wgrad_split(inputs direct, 16x16x32)");
  }
  if (x_layout == 1) {
    const int c1 = a->Cin - p.a_split;
    RAC_REQUIRE(p.a_split % 32 == 0 && c1 % 32 == 0 && p.P % 8 == 0,
                "rac_conv2d_wgrad_split: fragment-order inputs need channel counts % 32 == 0");
    p.wr = a->W / 8;
    RAC_REQUIRE((p.wr & (p.wr - 1)) == 0, "rac_conv2d_wgrad_split: fragment-order inputs need W / 8 a power of two");
    p.wr_shift = __builtin_ctz((unsigned)p.wr);
    p.magic_hw = ((1ULL << 40) + a->H - 1) / a->H;  // here: division by H of an image-row index
    constexpr size_t lds_bd = 2 * 3 * 128 * BD_ROW;  // 61,440 B
    static bool bd_attr = false;
    if (!bd_attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_split_bdirect_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bd);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
        return RAC_ELAUNCH;
      }
      bd_attr = true;
    }
    hipLaunchKernelGGL(wgrad_split_bdirect_kernel, grid, dim3(256), lds_bd, reinterpret_cast<hipStream_t>(stream), p);
    return check_launch("rac_conv2d_wgrad_split(inputs direct)");
  }
  RAC_REQUIRE(x_layout == 0, "rac_conv2d_wgrad_split: x_layout must be 0, 1 or 2");
  hipLaunchKernelGGL(wgrad_split_kernel, grid, dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
  return check_launch("rac_conv2d_wgrad_split");
}
