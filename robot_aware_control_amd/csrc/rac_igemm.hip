// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32): forward, data-gradient and weight-gradient of the
// stride-1 "same" convolutions of the SVG model, NHWC activations.
//
//   C[M][N] (+)= sum_k A(m,k) * B(k,n)
//   FWD   : m = pixel, n = co,       k = (tap, ci)   A = im2col(x)        B = w[n][tap][ci]
//   DGRAD : m = pixel, n = ci,       k = (tap, co)   A = im2col(dy, -tap) B = w[co][tap][n]
//   WGRAD : m = co,    n = (tap,ci), k = pixel       A = dy[k][m]         B = x[k+tap][ci]
//
// One workgroup = 4 waves (256 threads) computing a BM x BN tile; each wave owns
// (BM/WM) x (BN/WN) as MT x NT accumulators of 32x32.  K is walked in chunks of 32:
// the next chunk's global loads (16 B per lane, zero-filled at the borders) are
// issued before the MFMAs of the current chunk and written to the other LDS
// buffer after them, one barrier per chunk.
//
// LDS tile forms (chosen by what is contiguous in HBM):
//   KC: T[row][k]  row stride 36 floats; a lane reads 4 consecutive k (ds_read_b128)
//   MC: T[k][row]  row stride BM/BN floats; a lane reads one float per MFMA (ds_read_b32)
// Both use the same k assignment: MFMA (j,t) of lane half h consumes k = 8j + 4h + t.
#include <stdlib.h>

#include "rac_common.h"

namespace rac {

constexpr int BK = 32;
constexpr int LDK = 36;

struct ConvP {
  int mode, B, H, W, ks, pad, Cin, Cout, act, split_k, accumulate, a_split, o_split;
  long slab_stride;
  const float *a0, *a1, *w;
  float *out0, *out1;
  const float *bias, *scale, *shift;
  double* stats;
  long stats_rows;  // rows per statistics group (0: one group); tiles never straddle groups
  int M, N, HW, P, taps, cchunks, nchunks, cps, ntile_per_tap;
  unsigned long long magic_hw, magic_w;  // ceil(2^40 / d): n / d == (n * magic) >> 40 for n * d < 2^40
};

template <bool VEC>
__device__ __forceinline__ f32x4 ld4(const float* p, int nvalid) {
  // nvalid: how many of the 4 elements are in range (VEC: 0 or 4)
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (VEC) {
    if (nvalid > 0) v = *reinterpret_cast<const f32x4*>(p);
  } else {
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
  }
  return v;
}

// element (pixel q, channel c) of the virtual concat [a0 (Csplit ch) | a1 (C - Csplit ch)]
template <bool VEC>
__device__ __forceinline__ f32x4 ld_cat(const float* a0, const float* a1, int C, int Csplit, long q, int c,
                                        bool pix_ok) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!pix_ok) return v;
  if (VEC) {
    if (c < C) {
      const float* p = (c < Csplit) ? a0 + q * Csplit + c : a1 + q * (C - Csplit) + (c - Csplit);
      v = *reinterpret_cast<const f32x4*>(p);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int ce = c + e;
      if (ce < C) v[e] = (ce < Csplit) ? a0[q * Csplit + ce] : a1[q * (C - Csplit) + (ce - Csplit)];
    }
  }
  return v;
}

// Epilogue shared by both kernels.
// C/D layout of 32x32 f32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
template <int MODE, int MT, int NT>
__device__ __forceinline__ void epilogue(const ConvP& p, f32x16 (&acc)[MT][NT], int m0, int n0, int wg_tap, int wm,
                                         int wn, int li, int lh) {
  if (MODE == RAC_CONV_WGRAD) {
    const long wrow = (long)p.taps * p.Cin;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + (wn * NT + nt) * 32 + li;
      if (n >= p.Cin) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m >= p.M) continue;
          float* dst = p.out0 + m * wrow + (long)wg_tap * p.Cin + n;
          const float v = acc[mt][nt][r];
          if (p.split_k > 1)
            atomicAdd(dst, v);
          else if (p.accumulate)
            *dst += v;
          else
            *dst = v;
        }
      }
    }
    return;
  }

  const bool slab = p.split_k > 1;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + (wn * NT + nt) * 32 + li;
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
        if (p.o_split > 0) {
          if (n < p.o_split)
            p.out0[(long)m * p.o_split + n] = v;
          else
            p.out1[(long)m * (p.N - p.o_split) + (n - p.o_split)] = v;
        } else {
          p.out0[(long)m * p.N + n] = v;
        }
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

template <int MODE, int BM, int BN, int WM, int WN, bool AV, bool BV>
__global__ __launch_bounds__(256) void igemm_kernel(ConvP p) {
  constexpr int MT = BM / (32 * WM);
  constexpr int NT = BN / (32 * WN);
  constexpr bool A_KC = (MODE != RAC_CONV_WGRAD);
  constexpr bool B_KC = (MODE == RAC_CONV_FWD);
  constexpr int A_SZ = A_KC ? BM * LDK : BK * BM;
  constexpr int B_SZ = B_KC ? BN * LDK : BK * BN;
  // per-thread staging passes
  constexpr int A_PASS = A_KC ? BM / 32 : (BK * BM / 4) / 256;
  constexpr int B_PASS = B_KC ? BN / 32 : (BK * BN / 4) / 256;
  constexpr int A_TPR = BM / 4;  // MC form: threads per k-row
  constexpr int B_TPR = BN / 4;
  static_assert(A_PASS >= 1 && B_PASS >= 1, "tile too small for 256 threads");

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int wm = wid / WN;
  const int wn = wid % WN;

  const int m0 = blockIdx.x * BM;
  int n0, wg_tap = 0;
  if (MODE == RAC_CONV_WGRAD) {
    wg_tap = blockIdx.y / p.ntile_per_tap;
    n0 = (blockIdx.y - wg_tap * p.ntile_per_tap) * BN;
  } else {
    n0 = blockIdx.y * BN;
  }
  const int kc_begin = blockIdx.z * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);

  // ---- per-thread row bookkeeping for the pixel-row (KC) A loader ----
  int a_pix[A_KC ? A_PASS : 1], a_y[A_KC ? A_PASS : 1], a_x[A_KC ? A_PASS : 1];
  if (A_KC) {
    const int arow = tid >> 3;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      int m = m0 + arow + 32 * i;
      if (m < p.M) {
        int b = m / p.HW;
        int r = m - b * p.HW;
        int y = r / p.W;
        a_pix[i] = m;
        a_y[i] = y;
        a_x[i] = r - y * p.W;
      } else {
        a_pix[i] = 0;
        a_y[i] = -100000;
        a_x[i] = 0;
      }
    }
  }
  int wg_dy = 0, wg_dx = 0;
  if (MODE == RAC_CONV_WGRAD) {
    wg_dy = wg_tap / p.ks - p.pad;
    wg_dx = wg_tap % p.ks - p.pad;
  }

  f32x4 ra[A_PASS], rb[B_PASS];

  auto load_chunk = [&](int kc) {
    if (MODE != RAC_CONV_WGRAD) {
      const int tap = kc / p.cchunks;
      const int cc = kc - tap * p.cchunks;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      const int dy = (MODE == RAC_CONV_FWD) ? ky - p.pad : p.pad - ky;
      const int dx = (MODE == RAC_CONV_FWD) ? kx - p.pad : p.pad - kx;
      const int CA = (MODE == RAC_CONV_FWD) ? p.Cin : p.Cout;  // channels of the A-side tensor
      const int csplit = (MODE == RAC_CONV_FWD) ? p.a_split : CA;
      {
        const int c = cc * BK + (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
          int yy = a_y[i] + dy, xx = a_x[i] + dx;
          bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
          long q = (long)a_pix[i] + dy * p.W + dx;
          ra[i] = ld_cat<AV>(p.a0, p.a1, CA, csplit, q, c, ok);
        }
      }
      if (MODE == RAC_CONV_FWD) {  // B rows = output channel n, contiguous over ci
        const int c = cc * BK + (tid & 7) * 4;
        const int brow = tid >> 3;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
          int n = n0 + brow + 32 * i;
          int nv = (n < p.N) ? min(4, p.Cin - c) : 0;
          rb[i] = ld4<BV>(p.w + ((long)n * p.taps + tap) * p.Cin + c, nv);
        }
      } else {  // DGRAD: B rows = co within the chunk, contiguous over n = ci
        const int col = tid % B_TPR, r0 = tid / B_TPR;
        const int n = n0 + col * 4;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
          int co = cc * BK + r0 + (256 / B_TPR) * i;
          int nv = (co < p.Cout) ? min(4, p.N - n) : 0;
          rb[i] = ld4<BV>(p.w + ((long)co * p.taps + tap) * p.Cin + n, nv);
        }
      }
    } else {
      {  // A = dy[pixel][co], contiguous over m = co
        const int col = tid % A_TPR, r0 = tid / A_TPR;
        const int m = m0 + col * 4;
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
          long px = (long)kc * BK + r0 + (256 / A_TPR) * i;
          int nv = (px < p.P) ? min(4, p.M - m) : 0;
          ra[i] = ld4<AV>(p.w + px * p.Cout + m, nv);
        }
      }
      {  // B = x[pixel + tap][ci]
        const int col = tid % B_TPR, r0 = tid / B_TPR;
        const int c = n0 + col * 4;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
          int px = kc * BK + r0 + (256 / B_TPR) * i;
          bool ok = false;
          long q = 0;
          if (px < p.P) {
            int b = px / p.HW;
            int r = px - b * p.HW;
            int y = r / p.W;
            int x = r - y * p.W;
            int yy = y + wg_dy, xx = x + wg_dx;
            ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            q = (long)px + wg_dy * p.W + wg_dx;
          }
          rb[i] = ld_cat<BV>(p.a0, p.a1, p.Cin, p.a_split, q, c, ok);
        }
      }
    }
  };

  auto store_chunk = [&](int buf) {
    float* As = smem + buf * (A_SZ + B_SZ);
    float* Bs = As + A_SZ;
    if (A_KC) {
#pragma unroll
      for (int i = 0; i < A_PASS; ++i)
        *reinterpret_cast<f32x4*>(As + ((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4) = ra[i];
    } else {
#pragma unroll
      for (int i = 0; i < A_PASS; ++i)
        *reinterpret_cast<f32x4*>(As + (tid / A_TPR + (256 / A_TPR) * i) * BM + (tid % A_TPR) * 4) = ra[i];
    }
    if (B_KC) {
#pragma unroll
      for (int i = 0; i < B_PASS; ++i)
        *reinterpret_cast<f32x4*>(Bs + ((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4) = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < B_PASS; ++i)
        *reinterpret_cast<f32x4*>(Bs + (tid / B_TPR + (256 / B_TPR) * i) * BN + (tid % B_TPR) * 4) = rb[i];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute_chunk = [&](int buf) {
    const float* As = smem + buf * (A_SZ + B_SZ);
    const float* Bs = As + A_SZ;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 fa[MT], fb[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = (wm * MT + mt) * 32 + li;
        if (A_KC) {
          fa[mt] = *reinterpret_cast<const f32x4*>(As + row * LDK + 8 * j + 4 * lh);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) fa[mt][t] = As[(8 * j + 4 * lh + t) * BM + row];
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int row = (wn * NT + nt) * 32 + li;
        if (B_KC) {
          fb[nt] = *reinterpret_cast<const f32x4*>(Bs + row * LDK + 8 * j + 4 * lh);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) fb[nt][t] = Bs[(8 * j + 4 * lh + t) * BN + row];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][t], fb[nt][t], acc[mt][nt], 0, 0, 0);
    }
  };

  // ---- main loop ----
  if (kc_begin < kc_end) {
    load_chunk(kc_begin);
    store_chunk(0);
    __syncthreads();
    for (int kc = kc_begin; kc < kc_end; ++kc) {
      const int buf = (kc - kc_begin) & 1;
      const bool more = kc + 1 < kc_end;
      if (more) load_chunk(kc + 1);
      compute_chunk(buf);
      if (more) store_chunk(buf ^ 1);
      __syncthreads();
    }
  }

  epilogue<MODE, MT, NT>(p, acc, m0, n0, wg_tap, wm, wn, li, lh);
}


// --------------------------------------------------------------------------------------------
// Fast path: same tiling / LDS forms / k assignment as igemm_kernel, but
//   * every global load is a branch-free `buffer_load_dwordx4` through a wave-uniform buffer
//     descriptor: out-of-image taps, partial channel chunks, tile tails and the chunk past the
//     end are out-of-range offsets that the hardware returns as zeros -- no exec-mask branches,
//     32-bit offset arithmetic only, so the whole K-chunk body is ONE basic block;
//   * the next chunk's loads are issued in four slices between the four j-steps of MFMAs and the
//     MFMA fragments are double-buffered, so address VALU, LDS reads and HBM latency sit under
//     the 64-cycle fp32 MFMAs instead of in front of them.
// Requires 16-B aligned operands, channel counts % 4 == 0, a chunk-uniform concat source
// (FWD: a_split % 32 == 0; WGRAD: a_split % BN == 0) and operands < 4 GiB.
// --------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ int udiv40(int n, unsigned long long magic) {
  return (int)(((unsigned long long)(unsigned)n * magic) >> 40);
}
constexpr unsigned OOB = 0xFFFFFFF0u;

__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  unsigned long long v = reinterpret_cast<unsigned long long>(p);
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ rsrc_t make_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(p)), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_ld4(rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}

template <int MODE, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm_fast_kernel(ConvP p) {
  constexpr int MT = BM / (32 * WM);
  constexpr int NT = BN / (32 * WN);
  constexpr bool A_KC = (MODE != RAC_CONV_WGRAD);
  constexpr bool B_KC = (MODE == RAC_CONV_FWD);
  constexpr int A_SZ = A_KC ? BM * LDK : BK * BM;
  constexpr int B_SZ = B_KC ? BN * LDK : BK * BN;
  constexpr int A_PASS = A_KC ? BM / 32 : (BK * BM / 4) / 256;
  constexpr int B_PASS = B_KC ? BN / 32 : (BK * BN / 4) / 256;
  constexpr int A_TPR = BM / 4;
  constexpr int B_TPR = BN / 4;
  constexpr int A_RPP = 256 / A_TPR;  // MC form: k-rows covered per pass
  constexpr int B_RPP = 256 / B_TPR;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int wm = wid / WN;
  const int wn = wid % WN;

  const int m0 = blockIdx.x * BM;
  int n0, wg_tap = 0;
  if (MODE == RAC_CONV_WGRAD) {
    wg_tap = blockIdx.y / p.ntile_per_tap;
    n0 = (blockIdx.y - wg_tap * p.ntile_per_tap) * BN;
  } else {
    n0 = blockIdx.y * BN;
  }
  const int kc_begin = blockIdx.z * p.cps;
  const int kc_end = min(kc_begin + p.cps, p.nchunks);

  // ---- loop-invariant per-thread row data ----
  int a_pix[A_PASS], a_y[A_PASS], a_x[A_PASS];  // pixel-row A loader (FWD / DGRAD)
  int b_row[B_PASS];                            // FWD: n*taps*Cin (or -1)
  const int kcol = (tid & 7) * 4;               // KC form: channel offset inside the chunk
  if (A_KC) {
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      int m = m0 + (tid >> 3) + 32 * i;
      if (m < p.M) {
        int b = m / p.HW;
        int r = m - b * p.HW;
        int y = r / p.W;
        a_pix[i] = m, a_y[i] = y, a_x[i] = r - y * p.W;
      } else {
        a_pix[i] = 0, a_y[i] = -100000, a_x[i] = 0;
      }
    }
  }
  if (MODE == RAC_CONV_FWD) {
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      int n = n0 + (tid >> 3) + 32 * i;
      b_row[i] = (n < p.N) ? n * p.taps * p.Cin : -1;
    }
  }
  const int wg_dy = (MODE == RAC_CONV_WGRAD) ? wg_tap / p.ks - p.pad : 0;
  const int wg_dx = (MODE == RAC_CONV_WGRAD) ? wg_tap % p.ks - p.pad : 0;

  // descriptors that do not change over the K loop
  const rsrc_t w_rsrc = make_rsrc(p.w, MODE == RAC_CONV_WGRAD ? (unsigned)p.P * p.Cout * 4u
                                                                : (unsigned)p.Cout * p.taps * p.Cin * 4u);
  // WGRAD: the x source of this block's channel range (block-uniform)
  const bool wg_first = n0 < p.a_split;
  const int wg_Cs = wg_first ? p.a_split : p.Cin - p.a_split;
  const int wg_cl = (wg_first ? n0 : n0 - p.a_split) + (tid % B_TPR) * 4;
  const rsrc_t x_rsrc = make_rsrc(wg_first ? p.a0 : (p.a1 ? p.a1 : p.a0), (unsigned)p.P * wg_Cs * 4u);

  // Byte offsets of one chunk's loads (OOB where the hardware must return zeros) and the A-side descriptor.
  auto chunk_offsets = [&](int kc, unsigned (&oa)[A_PASS], unsigned (&ob)[B_PASS], rsrc_t& a_rsrc) {
    const bool live = kc < kc_end;
    if (MODE != RAC_CONV_WGRAD) {
      const int tap = kc / p.cchunks;
      const int cc = kc - tap * p.cchunks;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      const int dy = (MODE == RAC_CONV_FWD) ? ky - p.pad : p.pad - ky;
      const int dx = (MODE == RAC_CONV_FWD) ? kx - p.pad : p.pad - kx;
      const int c0 = cc * BK;
      const int CA = (MODE == RAC_CONV_FWD) ? p.Cin : p.Cout;
      const int csplit = (MODE == RAC_CONV_FWD) ? p.a_split : CA;
      const bool first = c0 < csplit;
      const int Cs = first ? csplit : CA - csplit;
      const int cl = (first ? c0 : c0 - csplit) + kcol;
      a_rsrc = make_rsrc(first ? p.a0 : p.a1, (unsigned)p.P * Cs * 4u);
      const int shift = dy * p.W + dx;
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        int yy = a_y[i] + dy, xx = a_x[i] + dx;
        bool ok = live & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W) & (cl < Cs);
        oa[i] = ok ? (unsigned)((a_pix[i] + shift) * Cs + cl) * 4u : OOB;
      }
      if (MODE == RAC_CONV_FWD) {
        const int s0 = tap * p.Cin + c0 + kcol;
        const bool cok = live & (c0 + kcol < p.Cin);
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) ob[i] = (cok & (b_row[i] >= 0)) ? (unsigned)(b_row[i] + s0) * 4u : OOB;
      } else {
        const int n = n0 + (tid % B_TPR) * 4;
#pragma unroll
        for (int i = 0; i < B_PASS; ++i) {
          int co = c0 + tid / B_TPR + B_RPP * i;
          bool ok = live & (co < p.Cout) & (n < p.N);
          ob[i] = ok ? (unsigned)((co * p.taps + tap) * p.Cin + n) * 4u : OOB;
        }
      }
    } else {
      a_rsrc = w_rsrc;
      const int p0 = kc * BK;
      const int m = m0 + (tid % A_TPR) * 4;
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        int px = p0 + tid / A_TPR + A_RPP * i;
        bool ok = live & (px < p.P) & (m < p.M);
        oa[i] = ok ? (unsigned)(px * p.Cout + m) * 4u : OOB;
      }
      const int shift = wg_dy * p.W + wg_dx;
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) {
        int px = p0 + tid / B_TPR + B_RPP * i;
        int b = udiv40(px, p.magic_hw);
        int r = px - b * p.HW;
        int y = udiv40(r, p.magic_w);
        int x = r - y * p.W;
        int yy = y + wg_dy, xx = x + wg_dx;
        bool ok = live & (px < p.P) & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W) & (wg_cl < wg_Cs);
        ob[i] = ok ? (unsigned)((px + shift) * wg_Cs + wg_cl) * 4u : OOB;
      }
    }
  };
  auto issue_loads = [&](const unsigned (&oa)[A_PASS], const unsigned (&ob)[B_PASS], rsrc_t a_rsrc,
                         f32x4 (&ra)[A_PASS], f32x4 (&rb)[B_PASS]) {
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) ra[i] = buf_ld4(a_rsrc, oa[i]);
    const rsrc_t b_rsrc = (MODE == RAC_CONV_WGRAD) ? x_rsrc : w_rsrc;
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) rb[i] = buf_ld4(b_rsrc, ob[i]);
  };

  // LDS store of staged load number q (A loads first, then B) into buffer `buf`
  auto store_one = [&](int buf, int q, const f32x4 (&ra)[A_PASS], const f32x4 (&rb)[B_PASS]) {
    float* As = smem + buf * (A_SZ + B_SZ);
    float* Bs = As + A_SZ;
    if (q < A_PASS) {
      const int i = q;
      if (A_KC)
        *reinterpret_cast<f32x4*>(As + ((tid >> 3) + 32 * i) * LDK + kcol) = ra[i];
      else
        *reinterpret_cast<f32x4*>(As + (tid / A_TPR + A_RPP * i) * BM + (tid % A_TPR) * 4) = ra[i];
    } else {
      const int i = q - A_PASS;
      if (B_KC)
        *reinterpret_cast<f32x4*>(Bs + ((tid >> 3) + 32 * i) * LDK + kcol) = rb[i];
      else
        *reinterpret_cast<f32x4*>(Bs + (tid / B_TPR + B_RPP * i) * BN + (tid % B_TPR) * 4) = rb[i];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto read_frags = [&](const float* As, const float* Bs, int j, f32x4 (&fa)[MT], f32x4 (&fb)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = (wm * MT + mt) * 32 + li;
      if (A_KC) {
        fa[mt] = *reinterpret_cast<const f32x4*>(As + row * LDK + 8 * j + 4 * lh);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[mt][t] = As[(8 * j + 4 * lh + t) * BM + row];
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int row = (wn * NT + nt) * 32 + li;
      if (B_KC) {
        fb[nt] = *reinterpret_cast<const f32x4*>(Bs + row * LDK + 8 * j + 4 * lh);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[nt][t] = Bs[(8 * j + 4 * lh + t) * BN + row];
      }
    }
  };

  if (kc_begin < kc_end) {
    constexpr int NLOADS = A_PASS + B_PASS;
    unsigned oa[A_PASS], ob[B_PASS];
    rsrc_t ars;
    f32x4 ra0[A_PASS], rb0[B_PASS], ra1[A_PASS], rb1[B_PASS];  // two staging sets: loads run two chunks ahead

    // One K-chunk.  On entry: LDS[buf] holds chunk kc, (cra, crb) hold chunk kc+1 (issued one chunk ago),
    // (oa, ob, ars) address chunk kc+2.  The body issues kc+2 into (nra, nrb), derives the offsets of kc+3,
    // and writes kc+1 into LDS[buf^1] between the MFMAs of the third j-step, so that only the barrier is
    // left at the end of the chunk.
    auto body = [&](int kc, f32x4 (&cra)[A_PASS], f32x4 (&crb)[B_PASS], f32x4 (&nra)[A_PASS], f32x4 (&nrb)[B_PASS]) {
      const int buf = (kc - kc_begin) & 1;
      const float* As = smem + buf * (A_SZ + B_SZ);
      const float* Bs = As + A_SZ;
      issue_loads(oa, ob, ars, nra, nrb);
      __builtin_amdgcn_sched_barrier(0);
      unsigned na[A_PASS], nb[B_PASS];
      rsrc_t nrs;
      chunk_offsets(kc + 3, na, nb, nrs);
      f32x4 fa[2][MT], fb[2][NT];
      read_frags(As, Bs, 0, fa[0], fb[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < 3) read_frags(As, Bs, j + 1, fa[(j + 1) & 1], fb[(j + 1) & 1]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j & 1][mt][t], fb[j & 1][nt][t], acc[mt][nt], 0,
                                                                 0, 0);
              const int q = (t * MT + mt) * NT + nt;  // compile-time after unrolling
              if (j == 2 && q < NLOADS) {  // one LDS store per MFMA gap, pinned there
                __builtin_amdgcn_sched_barrier(0);
                store_one(buf ^ 1, q, cra, crb);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
        if (j == 2 && NLOADS > 4 * MT * NT) {
#pragma unroll
          for (int q = 4 * MT * NT; q < NLOADS; ++q) store_one(buf ^ 1, q, cra, crb);
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) oa[i] = na[i];
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) ob[i] = nb[i];
      ars = nrs;
    };

    chunk_offsets(kc_begin, oa, ob, ars);
    issue_loads(oa, ob, ars, ra0, rb0);
    chunk_offsets(kc_begin + 1, oa, ob, ars);
    issue_loads(oa, ob, ars, ra1, rb1);
    chunk_offsets(kc_begin + 2, oa, ob, ars);
#pragma unroll
    for (int q = 0; q < NLOADS; ++q) store_one(0, q, ra0, rb0);
    __syncthreads();
    for (int kc = kc_begin; kc < kc_end; kc += 2) {
      body(kc, ra1, rb1, ra0, rb0);
      if (kc + 1 < kc_end) body(kc + 1, ra0, rb0, ra1, rb1);
    }
  }
  epilogue<MODE, MT, NT>(p, acc, m0, n0, wg_tap, wm, wn, li, lh);
}

template <int MODE, int BM, int BN, int WM, int WN>
static int launch_fast(const ConvP& p, dim3 grid, hipStream_t st) {
  constexpr bool A_KC = (MODE != RAC_CONV_WGRAD);
  constexpr bool B_KC = (MODE == RAC_CONV_FWD);
  constexpr int A_SZ = A_KC ? BM * LDK : BK * BM;
  constexpr int B_SZ = B_KC ? BN * LDK : BK * BN;
  constexpr size_t lds = 2 * (A_SZ + B_SZ) * sizeof(float);
  auto k = igemm_fast_kernel<MODE, BM, BN, WM, WN>;
  static bool attr_done = false;
  if (!attr_done) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
        return RAC_ELAUNCH;
      }
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  return check_launch("rac_conv2d(fast)");
}

template <int MODE, int BM, int BN, int WM, int WN, bool AV, bool BV>
static int launch(const ConvP& p, dim3 grid, hipStream_t st) {
  constexpr bool A_KC = (MODE != RAC_CONV_WGRAD);
  constexpr bool B_KC = (MODE == RAC_CONV_FWD);
  constexpr int A_SZ = A_KC ? BM * LDK : BK * BM;
  constexpr int B_SZ = B_KC ? BN * LDK : BK * BN;
  constexpr size_t lds = 2 * (A_SZ + B_SZ) * sizeof(float);
  auto k = igemm_kernel<MODE, BM, BN, WM, WN, AV, BV>;
  static bool attr_done = false;  // one flag per instantiation
  if (!attr_done) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
      if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(lds=%zu): %s", lds, hipGetErrorString(e));
        return RAC_ELAUNCH;
      }
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  return check_launch("rac_conv2d");
}

template <int MODE, int BM, int BN, int WM, int WN>
static int launch_vec(const ConvP& p, dim3 grid, hipStream_t st, bool av, bool bv) {
  if (av && bv) return launch<MODE, BM, BN, WM, WN, true, true>(p, grid, st);
  if (av) return launch<MODE, BM, BN, WM, WN, true, false>(p, grid, st);
  if (bv) return launch<MODE, BM, BN, WM, WN, false, true>(p, grid, st);
  return launch<MODE, BM, BN, WM, WN, false, false>(p, grid, st);
}

template <int MODE>
static int launch_mode(ConvP& p, hipStream_t st, bool av, bool bv) {
  // tile choice: narrow-N tile for heads, 128x128 when it still fills the chip, else 64x64
  const long gm = p.M, gn = (MODE == RAC_CONV_WGRAD) ? p.Cin : p.N;
  const long tapmul = (MODE == RAC_CONV_WGRAD) ? p.taps : 1;
  int bm, bn;
  static const int forced = [] {  // RAC_IGEMM_TILE=128|64|32 pins the tile (benchmark A/B only)
    const char* e = getenv("RAC_IGEMM_TILE");
    return e ? atoi(e) : 0;
  }();
  if (forced == 128 && gn > 32) {
    bm = bn = 128;
  } else if (forced == 64 && gn > 32) {
    bm = bn = 64;
  } else if (gn <= 32) {
    bm = 128, bn = 32;
  } else if (gn <= 64 && gm >= 128 * 256) {  // 64-channel layers at full resolution: tall tile, 2 accumulators / wave
    bm = 128, bn = 64;
  } else {
    long t128 = (long)cdiv(gm, 128) * cdiv(gn, 128) * tapmul;
    if (gm >= 128 && gn >= 128 && t128 * (p.split_k > 0 ? p.split_k : 1) >= 192)
      bm = bn = 128;
    else
      bm = bn = 64;
  }
  const long tiles = (long)cdiv(gm, bm) * cdiv(gn, bn) * tapmul;
  if (MODE == RAC_CONV_WGRAD && p.split_k <= 0) {  // auto split for the weight gradient (atomics, no workspace)
    int s = 1;
    if (tiles < 512) s = (int)min((long)cdiv(768, tiles), (long)max(1, p.nchunks / 8));
    p.split_k = max(1, min(s, 128));
  }
  if (p.split_k <= 0) p.split_k = 1;
  p.cps = cdiv(p.nchunks, p.split_k);
  p.ntile_per_tap = cdiv(gn, bn);
  dim3 grid(cdiv(gm, bm), (unsigned)(cdiv(gn, bn) * tapmul), p.split_k);
  static const bool no_fast = getenv("RAC_IGEMM_GENERIC") != nullptr;  // A/B switch for benchmarks
  bool fast = av && bv && !no_fast;
  {
    const long a_ch = (MODE == RAC_CONV_DGRAD) ? p.Cout : p.Cin;
    const long w_bytes = (MODE == RAC_CONV_WGRAD) ? (long)p.P * p.Cout * 4 : (long)p.Cout * p.taps * p.Cin * 4;
    fast = fast && (long)p.P * a_ch * 4 < 0xFFFFFF00L && w_bytes < 0xFFFFFF00L;
    const bool concat = p.a_split < ((MODE == RAC_CONV_DGRAD) ? p.Cout : p.Cin);
    if (MODE == RAC_CONV_FWD && concat) fast = fast && (p.a_split % BK == 0);
    if (MODE == RAC_CONV_WGRAD && concat) fast = fast && (p.a_split % bn == 0);
  }
  if (fast) {
    if (bn == 32) return launch_fast<MODE, 128, 32, 4, 1>(p, grid, st);
    if (bm == 128 && bn == 64) return launch_fast<MODE, 128, 64, 2, 2>(p, grid, st);
    if (bm == 128) return launch_fast<MODE, 128, 128, 2, 2>(p, grid, st);
    return launch_fast<MODE, 64, 64, 2, 2>(p, grid, st);
  }
  if (bn == 32) return launch_vec<MODE, 128, 32, 4, 1>(p, grid, st, av, bv);
  if (bm == 128 && bn == 64) return launch_vec<MODE, 128, 64, 2, 2>(p, grid, st, av, bv);
  if (bm == 128) return launch_vec<MODE, 128, 128, 2, 2>(p, grid, st, av, bv);
  return launch_vec<MODE, 64, 64, 2, 2>(p, grid, st, av, bv);
}

}  // namespace rac

extern "C" int rac_conv2d(const rac_conv_args* a, void* stream) {
  using namespace rac;
  RAC_REQUIRE(a != nullptr, "rac_conv2d: null args");
  RAC_REQUIRE(a->mode >= 0 && a->mode <= 2, "rac_conv2d: bad mode %d", a->mode);
  RAC_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0, "rac_conv2d: bad shape");
  RAC_REQUIRE(a->ksize >= 1 && (a->ksize & 1) && a->ksize <= 7, "rac_conv2d: ksize must be odd, got %d", a->ksize);
  RAC_REQUIRE(a->a0 && a->w && a->out0, "rac_conv2d: null tensor pointer");
  RAC_REQUIRE((long)a->B * a->H * a->W < (1L << 31) / 4, "rac_conv2d: too many pixels");
  ConvP p{};
  p.mode = a->mode;
  p.B = a->B, p.H = a->H, p.W = a->W, p.ks = a->ksize, p.pad = a->ksize / 2;
  p.Cin = a->Cin, p.Cout = a->Cout, p.act = a->act, p.split_k = a->split_k, p.accumulate = a->accumulate;
  p.slab_stride = a->slab_stride;
  p.a0 = a->a0, p.a1 = a->a1, p.w = a->w, p.out0 = a->out0, p.out1 = a->out1;
  p.bias = a->bias, p.scale = a->scale, p.shift = a->shift, p.stats = a->stats;
  p.stats_rows = a->stats ? a->stats_rows : 0;
  RAC_REQUIRE(p.stats_rows >= 0 && p.stats_rows % 128 == 0 &&
                  (p.stats_rows == 0 || ((long)a->B * a->H * a->W) % p.stats_rows == 0),
              "rac_conv2d: stats_rows must be a multiple of 128 that divides B*H*W");
  p.HW = a->H * a->W;
  p.P = a->B * p.HW;
  p.taps = a->ksize * a->ksize;
  p.o_split = 0;
  p.magic_hw = ((1ULL << 40) + p.HW - 1) / p.HW;
  p.magic_w = ((1ULL << 40) + a->W - 1) / a->W;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  RAC_REQUIRE(!(a->scale && !a->shift), "rac_conv2d: scale without shift");

  if (a->mode == RAC_CONV_FWD) {
    p.M = p.P, p.N = a->Cout;
    p.a_split = (a->a1 && a->a_split > 0 && a->a_split < a->Cin) ? a->a_split : a->Cin;
    p.cchunks = cdiv(a->Cin, BK);
    p.nchunks = p.taps * p.cchunks;
    RAC_REQUIRE(a->split_k <= 1 || a->slab_stride >= (long)p.M * p.N, "rac_conv2d: slab_stride too small");
    bool av = (a->Cin % 4 == 0) && (p.a_split % 4 == 0) && aligned16(a->a0) && (!a->a1 || aligned16(a->a1));
    bool bv = (a->Cin % 4 == 0) && aligned16(a->w);
    return launch_mode<RAC_CONV_FWD>(p, st, av, bv);
  }
  if (a->mode == RAC_CONV_DGRAD) {
    p.M = p.P, p.N = a->Cin;
    p.a_split = a->Cout;
    p.cchunks = cdiv(a->Cout, BK);
    p.nchunks = p.taps * p.cchunks;
    if (a->out1 && a->o_split > 0 && a->o_split < a->Cin) p.o_split = a->o_split;
    RAC_REQUIRE(a->split_k <= 1 || (a->slab_stride >= (long)p.M * p.N && p.o_split == 0),
                "rac_conv2d: dgrad split-K needs a slab and a single destination");
    bool av = (a->Cout % 4 == 0) && aligned16(a->a0);
    bool bv = (a->Cin % 4 == 0) && aligned16(a->w);
    return launch_mode<RAC_CONV_DGRAD>(p, st, av, bv);
  }
  // WGRAD: a0/a1 = saved input x (virtual concat), w = dy, out0 = dw
  p.M = a->Cout, p.N = p.taps * a->Cin;
  p.a_split = (a->a1 && a->a_split > 0 && a->a_split < a->Cin) ? a->a_split : a->Cin;
  p.cchunks = 0;
  p.nchunks = cdiv(p.P, BK);
  bool av = (a->Cout % 4 == 0) && aligned16(a->w);
  bool bv = (a->Cin % 4 == 0) && (p.a_split % 4 == 0) && aligned16(a->a0) && (!a->a1 || aligned16(a->a1));
  return launch_mode<RAC_CONV_WGRAD>(p, st, av, bv);
}

RAC_DEVICE_CODE_END
