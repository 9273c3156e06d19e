// HBM-bound kernels around the convolutions: BatchNorm statistics / apply /
// backward, pooling, upsampling, tiling, ConvLSTM gate math, reparameterisation,
// bias gradients and the fused Adam step.  fp32 NHWC maps ([M = B*H*W][C]).
// Elementwise kernels move 16 B per lane when C % 4 == 0 and grid-stride over
// at most 2048 workgroups (256 CUs x 8).
#include <stdlib.h>

#include "rac_common.h"

namespace rac {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline int grid_for(long work_items) {
  long b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
// grid of a kernel that also commits a max |v| slot: fewer, longer workgroups (one atomic each on ONE address)
static inline int grid_for_amax(long work_items, const void* amax) {
  const int g = grid_for(work_items);
  return amax && g > 512 ? 512 : g;  // measured on the cfg2 train step: 512 -> 34.5 ms, 2048 -> 35.5, 128 -> 38-40
}

// ------------------------------------------------------------------ BatchNorm
// `G` groups of `count` rows each (time steps batched along the row axis: every group is one BatchNorm call of
// the reference).  stats [G][2][C]; scale/shift/mean/invstd [G][C]; the running statistics take the groups' momentum
// updates in order, `n_updates` times each.
__global__ void bn_finalize_kernel(const double* stats, long count, const float* gamma, const float* beta,
                                   float* rmean, float* rvar, float momentum, float eps, int n_updates, float* scale,
                                   float* shift, float* mean_o, float* invstd_o, int C, int G) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
  const float ga = gamma[c], be = beta[c];
  for (int g = 0; g < G; ++g) {
    const double* st = stats + (long)g * 2 * C;
    double mean = st[c] / (double)count;
    double var = st[C + c] / (double)count - mean * mean;
    if (var < 0) var = 0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float meanf = (float)mean;
    float sc = ga * invstd;
    scale[g * C + c] = sc;
    shift[g * C + c] = be - meanf * sc;
    mean_o[g * C + c] = meanf;
    invstd_o[g * C + c] = invstd;
    float unbiased = (float)(count > 1 ? var * (double)count / (double)(count - 1) : var);
    for (int i = 0; i < n_updates; ++i) {
      rm = (1.f - momentum) * rm + momentum * meanf;
      rv = (1.f - momentum) * rv + momentum * unbiased;
    }
  }
  if (rmean) {
    rmean[c] = rm;
    rvar[c] = rv;
  }
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == RAC_ACT_LEAKY02) return v > 0.f ? v : 0.2f * v;
  if (act == RAC_ACT_SIGMOID) return sigmoid_acc(v);
  return v;
}

// scale / shift are [G][C]; `ge` = elements (vectors) per group
__global__ void affine_act_kernel4(const f32x4* x, const f32x4* scale, const f32x4* shift, int act, f32x4* y, long n4,
                                   int C4, long ge, unsigned* amax) {
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C4) + (int)(i / ge) * C4;
    f32x4 v = x[i], s = scale[c], t = shift[c], o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = act_apply(v[e] * s[e] + t[e], act);
      mx = max(mx, absbits(o[e]));
    }
    y[i] = o;
  }
  if (amax) amax_commit_block(mx, amax);
}
__global__ void affine_act_kernel1(const float* x, const float* scale, const float* shift, int act, float* y, long n,
                                   int C, long ge, unsigned* amax) {
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C) + (int)(i / ge) * C;
    const float o = act_apply(x[i] * scale[c] + shift[c], act);
    mx = max(mx, absbits(o));
    y[i] = o;
  }
  if (amax) amax_commit_block(mx, amax);
}

// rac_bn_finalize + rac_affine_act in ONE launch (row-walking form, C = 4 * 2^k <= 1024): every workgroup derives the
// (scale, shift) of its thread's four channels from the fp64 statistics itself -- bn_finalize_kernel's arithmetic, term for
// term -- and streams its rows; the first workgroup of a group also stores the group's scale / shift / mean / invstd (what
// the backward pass reads), workgroup 0 applies the running statistics' momentum updates of all groups in order.
__global__ __launch_bounds__(256) void bn_apply_act_rows_kernel(const double* stats, long count, const float* gamma,
                                                                const float* beta, float* rmean, float* rvar, float momentum,
                                                                float eps, int n_updates, const f32x4* x, f32x4* y,
                                                                float* scale_o, float* shift_o, float* mean_o, float* invstd_o,
                                                                long Mg, int C4, int G, int act, int rows_per_block, int bpg,
                                                                unsigned* amax) {
  const int tid = threadIdx.x, C = 4 * C4;
  const int c4 = tid & (C4 - 1), rl = tid / C4, rpi = 256 / C4;
  const int g = blockIdx.x / bpg;
  const long r_begin = (long)g * Mg + (long)(blockIdx.x - g * bpg) * rows_per_block;
  const long r_end = min(r_begin + rows_per_block, (long)(g + 1) * Mg);
  f32x4 sc, sh;
  const bool writer = (blockIdx.x - g * bpg) == 0 && rl == 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * c4 + e;
    const double* st = stats + (long)g * 2 * C;
    const double mean = st[c] / (double)count;
    double var = st[C + c] / (double)count - mean * mean;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    sc[e] = gamma[c] * invstd;
    sh[e] = beta[c] - meanf * sc[e];
    if (writer) {
      scale_o[g * C + c] = sc[e];
      shift_o[g * C + c] = sh[e];
      mean_o[g * C + c] = meanf;
      invstd_o[g * C + c] = invstd;
    }
  }
  if (blockIdx.x == 0 && rl == 0 && rmean) {  // running statistics: every group's update, in order (bn_finalize_kernel)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * c4 + e;
      float rm = rmean[c], rv = rvar[c];
      for (int gg = 0; gg < G; ++gg) {
        const double* st = stats + (long)gg * 2 * C;
        const double mean = st[c] / (double)count;
        double var = st[C + c] / (double)count - mean * mean;
        if (var < 0) var = 0;
        const float meanf = (float)mean;
        const float unbiased = (float)(count > 1 ? var * (double)count / (double)(count - 1) : var);
        for (int i = 0; i < n_updates; ++i) {
          rm = (1.f - momentum) * rm + momentum * meanf;
          rv = (1.f - momentum) * rv + momentum * unbiased;
        }
      }
      rmean[c] = rm;
      rvar[c] = rv;
    }
  }
  unsigned mx = 0;
#pragma unroll 4
  for (long r = r_begin + rl; r < r_end; r += rpi) {
    const f32x4 v = x[r * C4 + c4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = act_apply(v[e] * sc[e] + sh[e], act);
      mx = max(mx, absbits(o[e]));
    }
    y[r * C4 + c4] = o;
  }
  if (amax) amax_commit_block(mx, amax);
}

// ---- small train-mode BatchNorm layers (one statistics group, a few MB: the 8x8 / 16x16 vgg layers of ONE time step) ----
// The two-launch forms (combine + statistics, then finalize + apply; reduce, then apply) cost 10-22 us per launch on these
// tensors -- ramp, a cold first touch, the atomics' tail -- whatever the bytes: a window that feeds its frames back runs them
// 95 times each (profiles/r06_sched_shapes.md).  Here a workgroup owns a SLICE of channels and ALL rows: the per-channel
// sums never leave it (no atomics, no second launch, a fixed summation order), and the second pass re-reads from the caches
// what the first one streamed.  grid = C / (4 QS) slices; block = QS channel quads x (256 / QS) row lanes.
template <int QS>
__device__ __forceinline__ void slice_totals(f32x4 a1, f32x4 a2, double (&t1)[4], double (&t2)[4], f32x4* sh) {
  // sum of the 256 / QS row lanes' partials of this thread's quad, in fp64, in lane order; every thread gets the totals
  const int tid = threadIdx.x, cl = tid & (QS - 1), nrl = 256 / QS;
  sh[tid] = a1;
  sh[256 + tid] = a2;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) t1[e] = 0.0, t2[e] = 0.0;
  for (int r = 0; r < nrl; ++r) {
    const f32x4 p1 = sh[r * QS + cl], p2 = sh[256 + r * QS + cl];
#pragma unroll
    for (int e = 0; e < 4; ++e) t1[e] += (double)p1[e], t2[e] += (double)p2[e];
  }
  __syncthreads();
}

template <int QS>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const f32x4* slabs, int n_slabs, long slab_stride4, f32x4* raw,
                                                           f32x4* y, const float* gamma, const float* beta, float* rmean,
                                                           float* rvar, float momentum, float eps, int n_updates,
                                                           float* scale_o, float* shift_o, float* mean_o, float* invstd_o,
                                                           long M, int C4, int act, unsigned* amax) {
  __shared__ f32x4 sh[512];
  const int tid = threadIdx.x, cl = tid & (QS - 1), rl = tid / QS, nrl = 256 / QS;
  const int c4 = blockIdx.x * QS + cl;
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
  for (long r = rl; r < M; r += nrl) {
    const long i = r * C4 + c4;
    f32x4 v = slabs[i];
    for (int sidx = 1; sidx < n_slabs; ++sidx) v += slabs[sidx * slab_stride4 + i];
    raw[i] = v;
    a1 += v;
    a2 += v * v;
  }
  double t1[4], t2[4];
  slice_totals<QS>(a1, a2, t1, t2, sh);
  f32x4 sc, shf;
#pragma unroll
  for (int e = 0; e < 4; ++e) {  // (bn_finalize_kernel's arithmetic)
    const int c = 4 * c4 + e;
    const double mean = t1[e] / (double)M;
    double var = t2[e] / (double)M - mean * mean;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    sc[e] = gamma[c] * invstd;
    shf[e] = beta[c] - meanf * sc[e];
    if (rl == 0) {
      scale_o[c] = sc[e], shift_o[c] = shf[e], mean_o[c] = meanf, invstd_o[c] = invstd;
      if (rmean) {
        const float unbiased = (float)(M > 1 ? var * (double)M / (double)(M - 1) : var);
        float rm = rmean[c], rv = rvar[c];
        for (int k = 0; k < n_updates; ++k) {
          rm = (1.f - momentum) * rm + momentum * meanf;
          rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        rmean[c] = rm, rvar[c] = rv;
      }
    }
  }
  unsigned mx = 0;
  for (long r = rl; r < M; r += nrl) {
    const long i = r * C4 + c4;
    const f32x4 v = raw[i];  // (this thread wrote it)
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = act_apply(v[e] * sc[e] + shf[e], act);
      mx = max(mx, absbits(o[e]));
    }
    y[i] = o;
  }
  if (amax) amax_commit_block(mx, amax);
}

template <int QS>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const f32x4* dy, const f32x4* x, const f32x4* scale, const f32x4* shift,
                                                           const f32x4* mean, const f32x4* invstd, f32x4* dx, float* dgamma,
                                                           float* dbeta, long M, int C4, unsigned* amax) {
  __shared__ f32x4 sh[512];
  const int tid = threadIdx.x, cl = tid & (QS - 1), rl = tid / QS, nrl = 256 / QS;
  const int c4 = blockIdx.x * QS + cl;
  const f32x4 sc = scale[c4], shf = shift[c4], mu = mean[c4], is = invstd[c4];
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
  for (long r = rl; r < M; r += nrl) {
    const f32x4 xv = x[r * C4 + c4], dv = dy[r * C4 + c4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = xv[e] * sc[e] + shf[e];
      const float dz = dv[e] * (z > 0.f ? 1.f : 0.2f);
      a1[e] += dz;
      a2[e] += dz * ((xv[e] - mu[e]) * is[e]);
    }
  }
  double t1[4], t2[4];
  slice_totals<QS>(a1, a2, t1, t2, sh);
  f32x4 m1, m2;
  const double invM = 1.0 / (double)M;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    m1[e] = (float)(t1[e] * invM), m2[e] = (float)(t2[e] * invM);
    if (rl == 0 && dgamma) {  // (one workgroup per channel: plain adds)
      dgamma[4 * c4 + e] += (float)t2[e];
      dbeta[4 * c4 + e] += (float)t1[e];
    }
  }
  unsigned mx = 0;
  for (long r = rl; r < M; r += nrl) {
    const f32x4 xv = x[r * C4 + c4], dv = dy[r * C4 + c4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // (bn_bwd_apply_kernel's arithmetic, element by element)
      const float z = xv[e] * sc[e] + shf[e];
      const float dz = dv[e] * (z > 0.f ? 1.f : 0.2f);
      const float xh = (xv[e] - mu[e]) * is[e];
      o[e] = sc[e] * (dz - m1[e] - xh * m2[e]);
      mx = max(mx, absbits(o[e]));
    }
    dx[r * C4 + c4] = o;
  }
  if (amax) amax_commit_block(mx, amax);
}

// Per-channel reductions over M rows.  grid = (row blocks, 64-channel groups), block = 4 row lanes x 64 channel
// lanes; partials combined through LDS, then one fp64 atomic per (block, channel).
// grid.x = G groups x `bpg` row blocks; `Mg` rows per group; sums [G][2][C], scale.. [G][C]
__global__ void bn_bwd_reduce_kernel(const float* dy, const float* x, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, double* sums, long Mg, int C,
                                     int rows_per_block, int bpg) {
  __shared__ float s1[256], s2[256];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int g = blockIdx.x / bpg;
  const long r_begin = (long)g * Mg + (long)(blockIdx.x - g * bpg) * rows_per_block;
  const long r_end = min(r_begin + rows_per_block, (long)(g + 1) * Mg);
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    const int gc = g * C + c;
    const float sc = scale[gc], sh = shift[gc], mu = mean[gc], is = invstd[gc];
    for (long r = r_begin + rl; r < r_end; r += 4) {
      float xv = x[r * C + c];
      float z = xv * sc + sh;
      float dz = dy[r * C + c] * (z > 0.f ? 1.f : 0.2f);
      a1 += dz;
      a2 += dz * ((xv - mu) * is);
    }
  }
  s1[threadIdx.x] = a1;
  s2[threadIdx.x] = a2;
  __syncthreads();
  if (threadIdx.x < 64 && c < C) {
    const int t = threadIdx.x;
    double* sg = sums + (long)g * 2 * C;
    atomicAdd(sg + c, (double)((s1[t] + s1[t + 64]) + (s1[t + 128] + s1[t + 192])));
    atomicAdd(sg + C + c, (double)((s2[t] + s2[t + 64]) + (s2[t + 128] + s2[t + 192])));
  }
}

__global__ void bn_bwd_apply_kernel(const float* dy, const float* x, const float* scale, const float* shift,
                                    const float* mean, const float* invstd, const double* sums, float* dx,
                                    float* dgamma, float* dbeta, long Mg, int C, int G, unsigned* amax) {
  const long ge = Mg * C, n = ge * G;
  const double invM = 1.0 / (double)Mg;
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C), g = (int)(i / ge);
    const int gc = g * C + c;
    float xv = x[i], sc = scale[gc];
    float z = xv * sc + shift[gc];
    float dz = dy[i] * (z > 0.f ? 1.f : 0.2f);
    float xh = (xv - mean[gc]) * invstd[gc];
    const double* sg = sums + (long)g * 2 * C;
    float m1 = (float)(sg[c] * invM), m2 = (float)(sg[C + c] * invM);
    const float o = sc * (dz - m1 - xh * m2);
    mx = max(mx, absbits(o));
    dx[i] = o;
  }
  if (amax) amax_commit_block(mx, amax);
  if (blockIdx.x == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      double a = 0., b = 0.;
      for (int g = 0; g < G; ++g) {
        a += sums[(long)g * 2 * C + C + c];
        b += sums[(long)g * 2 * C + c];
      }
      dgamma[c] += (float)a;
      dbeta[c] += (float)b;
    }
  }
}

// The same two passes for channel counts C = 4 * 2^k <= 1024 (every layer of the model), walking ROWS: a thread owns four
// channels -- its per-channel constants (scale, shift, mean, 1 / std, the two means of the reduce pass) are set up once,
// not re-derived per element through 64-bit index divisions and fp64 multiplies as in the element-indexed kernels above
// (which ran at 2.4-2.8 TB/s: instruction bound) -- and streams 16-byte vectors of rows r0 + tid / C4, + 256 / C4, ...
__global__ __launch_bounds__(256) void bn_bwd_reduce_rows_kernel(const f32x4* dy, const f32x4* x, const f32x4* scale,
                                                                 const f32x4* shift, const f32x4* mean, const f32x4* invstd,
                                                                 double* sums, long Mg, int C4, int C4s, int rows_per_block,
                                                                 int bpg) {
  // blockIdx.y = channel slice of C4s quads (C4s = C4: one slice, round 4's form): a workgroup ends in 8 fp64 atomics per
  // quad of ITS SLICE only, so a launch of W workgroups costs W / slices x 2 C atomics instead of W x 2 C (reduce_plan)
  __shared__ f32x4 s1[256], s2[256];
  const int tid = threadIdx.x;
  const int cl = tid & (C4s - 1), c4 = blockIdx.y * C4s + cl, rl = tid / C4s, rpi = 256 / C4s;
  const int g = blockIdx.x / bpg;
  const long r_begin = (long)g * Mg + (long)(blockIdx.x - g * bpg) * rows_per_block;
  const long r_end = min(r_begin + rows_per_block, (long)(g + 1) * Mg);
  const int gc = g * C4 + c4;
  const f32x4 sc = scale[gc], sh = shift[gc], mu = mean[gc], is = invstd[gc];
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (long r = r_begin + rl; r < r_end; r += rpi) {
    const f32x4 xv = x[r * C4 + c4], dv = dy[r * C4 + c4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = xv[e] * sc[e] + sh[e];
      const float dz = dv[e] * (z > 0.f ? 1.f : 0.2f);
      a1[e] += dz;
      a2[e] += dz * ((xv[e] - mu[e]) * is[e]);
    }
  }
  s1[tid] = a1;
  s2[tid] = a2;
  __syncthreads();
  if (tid < C4s) {
    f32x4 t1 = s1[tid], t2 = s2[tid];
    for (int k = 1; k < rpi; ++k) t1 += s1[k * C4s + tid], t2 += s2[k * C4s + tid];
    double* sg = sums + (long)g * 8 * C4 + 4 * c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(sg + e, (double)t1[e]);
      atomicAdd(sg + 4 * C4 + e, (double)t2[e]);
    }
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(const f32x4* dy, const f32x4* x, const float* scale,
                                                                const float* shift, const float* mean, const float* invstd,
                                                                const double* sums, f32x4* dx, float* dgamma, float* dbeta,
                                                                long Mg, int C4, int G, int rows_per_block, int bpg,
                                                                unsigned* amax) {
  const int tid = threadIdx.x, C = 4 * C4;
  const int c4 = tid & (C4 - 1), rl = tid / C4, rpi = 256 / C4;
  const int g = blockIdx.x / bpg;
  const long r_begin = (long)g * Mg + (long)(blockIdx.x - g * bpg) * rows_per_block;
  const long r_end = min(r_begin + rows_per_block, (long)(g + 1) * Mg);
  const double invM = 1.0 / (double)Mg;
  f32x4 sc, sh, mu, is, m1, m2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int gc = g * C + 4 * c4 + e;
    sc[e] = scale[gc], sh[e] = shift[gc], mu[e] = mean[gc], is[e] = invstd[gc];
    const double* sg = sums + (long)g * 2 * C;
    m1[e] = (float)(sg[4 * c4 + e] * invM), m2[e] = (float)(sg[C + 4 * c4 + e] * invM);
  }
  unsigned mx = 0;
#pragma unroll 4
  for (long r = r_begin + rl; r < r_end; r += rpi) {
    const f32x4 xv = x[r * C4 + c4], dv = dy[r * C4 + c4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // (the arithmetic of bn_bwd_apply_kernel, element by element)
      const float z = xv[e] * sc[e] + sh[e];
      const float dz = dv[e] * (z > 0.f ? 1.f : 0.2f);
      const float xh = (xv[e] - mu[e]) * is[e];
      o[e] = sc[e] * (dz - m1[e] - xh * m2[e]);
      mx = max(mx, absbits(o[e]));
    }
    dx[r * C4 + c4] = o;
  }
  if (amax) amax_commit_block(mx, amax);
  if (blockIdx.x == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      double a = 0., b = 0.;
      for (int gg = 0; gg < G; ++gg) {
        a += sums[(long)gg * 2 * C + C + c];
        b += sums[(long)gg * 2 * C + c];
      }
      dgamma[c] += (float)a;
      dbeta[c] += (float)b;
    }
  }
}

// ------------------------------------------------------------ pool / upsample
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* x, T* y, int B, int H, int W, int Cv) {
  const int Ho = H / 2, Wo = W / 2;
  const long n = (long)B * Ho * Wo * Cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % Cv);
    long q = i / Cv;
    int xo = (int)(q % Wo);
    long q2 = q / Wo;
    int yo = (int)(q2 % Ho);
    int b = (int)(q2 / Ho);
    const T* p = x + (((long)b * H + 2 * yo) * W + 2 * xo) * Cv + c;
    T v00 = p[0], v01 = p[Cv], v10 = p[(long)W * Cv], v11 = p[(long)W * Cv + Cv], o;
    if constexpr (sizeof(T) == 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
    } else {
      o = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
    }
    y[i] = o;
  }
}

// gradient goes to the first maximum in scan order (ATen max_pool2d_with_indices semantics)
__global__ void maxpool2_bwd_kernel(const float* x, const float* dy, float* dx, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const long n = (long)B * Ho * Wo * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long q = i / C;
    int xo = (int)(q % Wo);
    long q2 = q / Wo;
    int yo = (int)(q2 % Ho);
    int b = (int)(q2 / Ho);
    long base = (((long)b * H + 2 * yo) * W + 2 * xo) * C + c;
    long off[4] = {0, (long)C, (long)W * C, (long)W * C + C};
    float best = x[base];
    int arg = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      float v = x[base + off[k]];
      if (v > best || (v != v && best == best)) {
        best = v;
        arg = k;
      }
    }
    float g = dy[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) dx[base + off[k]] = (k == arg) ? g : 0.f;
  }
}

template <typename T>
__global__ void upsample2_fwd_kernel(const T* x, T* y, int B, int h, int w, int Cv) {
  const long n = (long)B * (2 * h) * (2 * w) * Cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % Cv);
    long q = i / Cv;
    int xo = (int)(q % (2 * w));
    long q2 = q / (2 * w);
    int yo = (int)(q2 % (2 * h));
    int b = (int)(q2 / (2 * h));
    y[i] = x[(((long)b * h + yo / 2) * w + xo / 2) * Cv + c];
  }
}
template <typename T>
__global__ void upsample2_bwd_kernel(const T* dy, T* dx, int B, int h, int w, int Cv) {
  const long n = (long)B * h * w * Cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % Cv);
    long q = i / Cv;
    int xi = (int)(q % w);
    long q2 = q / w;
    int yi = (int)(q2 % h);
    int b = (int)(q2 / h);
    const T* p = dy + (((long)b * 2 * h + 2 * yi) * (2 * w) + 2 * xi) * Cv + c;
    dx[i] = (p[0] + p[Cv]) + (p[(long)2 * w * Cv] + p[(long)2 * w * Cv + Cv]);
  }
}

// ------------------------------------------------------------ tiling / slices
__global__ void tilecat_kernel(const float* v0, int n0, const float* v1, int n1, const float* v2, int n2,
                               const float* m0, int c0, const float* m1, int c1, int pad, float* out, int B, int HW,
                               unsigned* amax) {
  const int Cv = n0 + n1 + n2 + c0 + c1;
  const int Ct = Cv + pad;
  const long n = (long)B * HW * Ct;
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % Ct);
    long q = i / Ct;  // b*HW + p
    int b = (int)(q / HW);
    float v;
    if (c < n0)
      v = v0[b * n0 + c];
    else if (c < n0 + n1)
      v = v1[b * n1 + (c - n0)];
    else if (c < n0 + n1 + n2)
      v = v2[b * n2 + (c - n0 - n1)];
    else if (c < n0 + n1 + n2 + c0)
      v = m0[q * c0 + (c - n0 - n1 - n2)];
    else if (c < Cv)
      v = m1[q * c1 + (c - n0 - n1 - n2 - c0)];
    else
      v = 0.f;
    mx = max(mx, absbits(v));
    out[i] = v;
  }
  if (amax) amax_commit_block(mx, amax);
}

// per-image maxima: workgroup b writes image b (the same values as tilecat_kernel) and leaves max |v| in amax[b]
__global__ __launch_bounds__(256) void tilecat_image_kernel(const float* v0, int n0, const float* v1, int n1, const float* v2,
                                                            int n2, const float* m0, int c0, const float* m1, int c1,
                                                            int pad, float* out, int HW, unsigned* amax) {
  const int Cv = n0 + n1 + n2 + c0 + c1;
  const int Ct = Cv + pad;
  const int b = blockIdx.x;
  const int n = HW * Ct;
  unsigned mx = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i % Ct;
    const long q = (long)b * HW + i / Ct;
    float v;
    if (c < n0)
      v = v0[b * n0 + c];
    else if (c < n0 + n1)
      v = v1[b * n1 + (c - n0)];
    else if (c < n0 + n1 + n2)
      v = v2[b * n2 + (c - n0 - n1)];
    else if (c < n0 + n1 + n2 + c0)
      v = m0[q * c0 + (c - n0 - n1 - n2)];
    else if (c < Cv)
      v = m1[q * c1 + (c - n0 - n1 - n2 - c0)];
    else
      v = 0.f;
    mx = max(mx, absbits(v));
    out[(long)b * n + i] = v;
  }
  __shared__ unsigned sh[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) amax[b] = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
}

// amax[r] = bits(max |x[r][:]|): one workgroup per row (image)
__global__ __launch_bounds__(256) void absmax_rows_kernel(const f32x4* x, long len4, unsigned* amax) {
  const f32x4* row = x + (long)blockIdx.x * len4;
  unsigned mx = 0;
  for (long i = threadIdx.x; i < len4; i += 256) {
    const f32x4 v = row[i];
    mx = max(mx, max(max(absbits(v.x), absbits(v.y)), max(absbits(v.z), absbits(v.w))));
  }
  __shared__ unsigned sh[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) amax[blockIdx.x] = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
}

__global__ void pad_rows_kernel(const float* src, int C, float* dst, int Cpad, long R) {
  const long n = R * Cpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long r = i / Cpad;
    int c = (int)(i - r * Cpad);
    dst[i] = c < C ? src[r * C + c] : 0.f;
  }
}
__global__ void unpad_add_kernel(const float* src, int Cpad, float* dst, int C, long R) {
  const long n = R * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long r = i / C;
    int c = (int)(i - r * C);
    dst[i] += src[r * Cpad + c];
  }
}

__global__ void slice_channels_kernel(const float* src, int Csrc, int off, int nc, float* dst, long M) {
  const long n = M * nc;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m = i / nc;
    int c = (int)(i - m * nc);
    dst[i] = src[m * Csrc + off + c];
  }
}

// dst[m] = [a[m][0:Ca] | b[m][0:Cb]]; a null source contributes zeros
__global__ void cat2_channels_kernel(const float* a, int Ca, const float* b, int Cb, float* dst, long M,
                                     unsigned* amax) {
  const int C = Ca + Cb;
  const long n = M * C;
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m = i / C;
    int c = (int)(i - m * C);
    const float v = c < Ca ? (a ? a[m * Ca + c] : 0.f) : (b ? b[m * Cb + (c - Ca)] : 0.f);
    mx = max(mx, absbits(v));
    dst[i] = v;
  }
  if (amax) amax_commit_block(mx, amax);
}

// Bias gradient of a conv applied at T time steps in ONE launch: colsum_acc_kernel below with every workgroup walking
// its row range in all T tensors (T times fewer atomics than T launches).
struct ColsumSteps {
  const float* x[RAC_WGRAD_MAX_STEPS];
  int T;
};
// `parts` given: the workgroup's column sums are STORED at parts[blockIdx.x][c] (colsum_parts_add_kernel then adds the
// row blocks in a fixed order: a bit-reproducible bias gradient); NULL: one fp32 atomic per (workgroup, column).
__global__ void colsum_steps_kernel(ColsumSteps p, float* out, float* parts, long M, int C, int rows_per_block) {
  __shared__ float s1[256];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;  // 4 row lanes
  const long r_begin = (long)blockIdx.x * rows_per_block;
  const long r_end = min(r_begin + rows_per_block, M);
  float a = 0.f;
  if (c < C)
    for (int t = 0; t < p.T; ++t) {
      const float* x = p.x[t];
      for (long r = r_begin + rl; r < r_end; r += 4) a += x[r * C + c];
    }
  s1[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < 64 && c < C) {
    const float v = (s1[threadIdx.x] + s1[threadIdx.x + 64]) + (s1[threadIdx.x + 128] + s1[threadIdx.x + 192]);
    if (parts)
      parts[(long)blockIdx.x * C + c] = v;
    else
      atomicAdd(out + c, v);
  }
}

// out[c] += sum over the row blocks of parts[b][c], in a FIXED association (the second stage of the deterministic column
// sums): a workgroup takes 64 columns, its four row lanes each add every fourth block in order, then lane sums 0 + 1 + 2 + 3.
__global__ __launch_bounds__(256) void colsum_parts_add_kernel(const float* parts, int nb, float* out, int C) {
  __shared__ float s1[256];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float a = 0.f;
  if (c < C)
    for (int b = rl; b < nb; b += 4) a += parts[(long)b * C + c];
  s1[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < 64 && c < C)
    out[c] += (s1[threadIdx.x] + s1[threadIdx.x + 64]) + (s1[threadIdx.x + 128] + s1[threadIdx.x + 192]);
}

// grid = (row blocks, 64-channel groups): bias gradients of the 4g-wide gate tensors have few rows (B*64) and many
// channels, so both dimensions are needed to fill the chip.
__global__ void colsum_acc_kernel(const float* x, float* out, float* parts, long M, int C, int rows_per_block) {
  __shared__ float s1[256];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;  // 4 row lanes
  const long r_begin = (long)blockIdx.x * rows_per_block;
  const long r_end = min(r_begin + rows_per_block, M);
  float a = 0.f;
  if (c < C)
    for (long r = r_begin + rl; r < r_end; r += 4) a += x[r * C + c];
  s1[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < 64 && c < C) {
    const float v = (s1[threadIdx.x] + s1[threadIdx.x + 64]) + (s1[threadIdx.x + 128] + s1[threadIdx.x + 192]);
    if (parts)
      parts[(long)blockIdx.x * C + c] = v;
    else
      atomicAdd(out + c, v);
  }
}

__global__ void slab_reduce_kernel(const float* slabs, int n_slabs, long slab_stride, const float* bias, float* out,
                                   long n, int N, unsigned* amax) {
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float a = bias ? bias[i % N] : 0.f;
    for (int s = 0; s < n_slabs; ++s) a += slabs[s * slab_stride + i];
    mx = max(mx, absbits(a));
    out[i] = a;
  }
  if (amax) amax_commit_block(mx, amax);
}

// 16-byte forms of the two combines (N, o_split, slab_stride multiples of 4; 16-byte aligned pointers)
__global__ void slab_reduce_kernel4(const f32x4* slabs, int n_slabs, long slab_stride4, const f32x4* bias, f32x4* out,
                                    long n4, int N4, unsigned* amax) {
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 a = bias ? bias[i % N4] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < n_slabs; ++s) a += slabs[s * slab_stride4 + i];
    mx = max(max(mx, max(absbits(a.x), absbits(a.y))), max(absbits(a.z), absbits(a.w)));
    out[i] = a;
  }
  if (amax) amax_commit_block(mx, amax);
}

__global__ void slab_reduce2_kernel4(const f32x4* slabs, int n_slabs, long slab_stride4, const f32x4* bias, f32x4* out0,
                                     f32x4* out1, long M, int N4, int o_split4, unsigned* amax0, unsigned* amax1) {
  const long n4 = M * N4;
  unsigned mx0 = 0, mx1 = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    long m = i / N4;
    int c = (int)(i - m * N4);
    f32x4 a = bias ? bias[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < n_slabs; ++s) a += slabs[s * slab_stride4 + i];
    const unsigned mv = max(max(absbits(a.x), absbits(a.y)), max(absbits(a.z), absbits(a.w)));
    if (c < o_split4) {
      out0[m * o_split4 + c] = a;
      mx0 = max(mx0, mv);
    } else {
      out1[m * (N4 - o_split4) + (c - o_split4)] = a;
      mx1 = max(mx1, mv);
    }
  }
  if (amax0) amax_commit_block(mx0, amax0);
  if (amax1) {
    __syncthreads();
    amax_commit_block(mx1, amax1);
  }
}

// split-K combine of a train-mode vgg layer's conv + its BatchNorm batch statistics in one pass (row-walking form, C =
// 4 * 2^k): out = sum of slabs; stats[g][0][c] += sum over the group's rows, stats[g][1][c] += sum of squares (fp64 atomics)
__global__ __launch_bounds__(256) void slab_reduce_stats_rows_kernel(const f32x4* slabs, int n_slabs, long slab_stride4,
                                                                     f32x4* out, double* stats, long Mg, int C4, int C4s,
                                                                     int rows_per_block, int bpg, unsigned* amax) {
  // (blockIdx.y = channel slice of C4s quads, as bn_bwd_reduce_rows_kernel)
  __shared__ f32x4 s1[256], s2[256];
  const int tid = threadIdx.x;
  const int cl = tid & (C4s - 1), c4 = blockIdx.y * C4s + cl, rl = tid / C4s, rpi = 256 / C4s;
  const int g = blockIdx.x / bpg;
  const long r_begin = (long)g * Mg + (long)(blockIdx.x - g * bpg) * rows_per_block;
  const long r_end = min(r_begin + rows_per_block, (long)(g + 1) * Mg);
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
  unsigned mx = 0;
#pragma unroll 2
  for (long r = r_begin + rl; r < r_end; r += rpi) {
    const long i = r * C4 + c4;
    f32x4 v = slabs[i];
    for (int sidx = 1; sidx < n_slabs; ++sidx) v += slabs[sidx * slab_stride4 + i];
    out[i] = v;
    a1 += v;
    a2 += v * v;
    mx = max(max(mx, max(absbits(v.x), absbits(v.y))), max(absbits(v.z), absbits(v.w)));
  }
  if (amax) amax_commit_block(mx, amax);
  s1[tid] = a1;
  s2[tid] = a2;
  __syncthreads();
  if (tid < C4s) {
    f32x4 t1 = s1[tid], t2 = s2[tid];
    for (int k = 1; k < rpi; ++k) t1 += s1[k * C4s + tid], t2 += s2[k * C4s + tid];
    double* sg = stats + (long)g * 8 * C4 + 4 * c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(sg + e, (double)t1[e]);
      atomicAdd(sg + 4 * C4 + e, (double)t2[e]);
    }
  }
}

__global__ void col_stats_kernel(const float* x, double* stats, long Mg, int C, int rows_per_block, int bpg) {
  __shared__ float s1[256], s2[256];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int g = blockIdx.x / bpg;  // stats [G][2][C], `Mg` rows per group
  const long r_begin = (long)g * Mg + (long)(blockIdx.x - g * bpg) * rows_per_block;
  const long r_end = min(r_begin + rows_per_block, (long)(g + 1) * Mg);
  float a1 = 0.f, a2 = 0.f;
  if (c < C)
    for (long r = r_begin + rl; r < r_end; r += 4) {
      float v = x[r * C + c];
      a1 += v;
      a2 += v * v;
    }
  s1[threadIdx.x] = a1;
  s2[threadIdx.x] = a2;
  __syncthreads();
  if (threadIdx.x < 64 && c < C) {
    const int t = threadIdx.x;
    double* sg = stats + (long)g * 2 * C;
    atomicAdd(sg + c, (double)((s1[t] + s1[t + 64]) + (s1[t + 128] + s1[t + 192])));
    atomicAdd(sg + C + c, (double)((s2[t] + s2[t + 64]) + (s2[t + 128] + s2[t + 192])));
  }
}

__global__ void act_bwd_kernel(const float* dy, const float* y, int act, float* dx, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float yv = y[i], g = dy[i];
    if (act == RAC_ACT_SIGMOID)
      g *= yv * (1.f - yv);
    else if (act == RAC_ACT_LEAKY02)
      g *= (yv > 0.f ? 1.f : 0.2f);
    dx[i] = g;
  }
}

__global__ void slab_reduce2_kernel(const float* slabs, int n_slabs, long slab_stride, const float* bias, float* out0,
                                    float* out1, long M, int N, int o_split, unsigned* amax0, unsigned* amax1) {
  const long n = M * N;
  unsigned mx0 = 0, mx1 = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m = i / N;
    int c = (int)(i - m * N);
    float a = bias ? bias[c] : 0.f;
    for (int s = 0; s < n_slabs; ++s) a += slabs[s * slab_stride + i];
    if (c < o_split) {
      out0[m * o_split + c] = a;
      mx0 = max(mx0, absbits(a));
    } else {
      out1[m * (N - o_split) + (c - o_split)] = a;
      mx1 = max(mx1, absbits(a));
    }
  }
  if (amax0) amax_commit_block(mx0, amax0);
  if (amax1) {
    __syncthreads();
    amax_commit_block(mx1, amax1);
  }
}

// ------------------------------------------------------------------- ConvLSTM
__global__ void lstm_cell_fwd_kernel(const float* slabs, int n_slabs, long slab_stride, const float* bias,
                                     const float* c_prev, float* h_out, float* c_out, float* act_out, long M, int g) {
  const long n = M * g;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m = i / g;
    int c = (int)(i - m * g);
    const float* row = slabs + m * 4 * g + c;
    float pi = bias[c], pf = bias[g + c], po = bias[2 * g + c], pg = bias[3 * g + c];
    for (int s = 0; s < n_slabs; ++s) {
      const float* r = row + s * slab_stride;
      pi += r[0];
      pf += r[g];
      po += r[2 * g];
      pg += r[3 * g];
    }
    float gi = sigmoid_acc(pi), gf = sigmoid_acc(pf), go = sigmoid_acc(po), gg = tanhf(pg);
    float cn = gf * c_prev[i] + gi * gg;
    c_out[i] = cn;
    h_out[i] = go * tanhf(cn);
    if (act_out) {
      float* a = act_out + m * 4 * g + c;
      a[0] = gi;
      a[g] = gf;
      a[2 * g] = go;
      a[3 * g] = gg;
    }
  }
}

__global__ void lstm_cell_bwd_kernel(const float* dh, const float* dc_next, const float* act, const float* c_prev,
                                     const float* c_new, float* dgates, float* dc_prev, long M, int g, unsigned* amax) {
  const long n = M * g;
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m = i / g;
    int c = (int)(i - m * g);
    const float* a = act + m * 4 * g + c;
    float gi = a[0], gf = a[g], go = a[2 * g], gg = a[3 * g];
    float tc = tanhf(c_new[i]);
    float dhv = dh ? dh[i] : 0.f;
    float dc = dhv * go * (1.f - tc * tc) + (dc_next ? dc_next[i] : 0.f);
    float* d = dgates + m * 4 * g + c;
    const float d0 = dc * gg * gi * (1.f - gi), d1 = dc * c_prev[i] * gf * (1.f - gf),
                d2 = dhv * tc * go * (1.f - go), d3 = dc * gi * (1.f - gg * gg);
    d[0] = d0, d[g] = d1, d[2 * g] = d2, d[3 * g] = d3;
    mx = max(max(mx, absbits(d0)), max(max(absbits(d1), absbits(d2)), absbits(d3)));
    dc_prev[i] = dc * gf;
  }
  if (amax) amax_commit_block(mx, amax);
}

__global__ void reparam_fwd_kernel(const float* mu, const float* lv, const float* eps, float* z, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    z[i] = eps[i] * expf(0.5f * lv[i]) + mu[i];
}
__global__ void reparam_bwd_kernel(const float* dz, const float* lv, const float* eps, float* dlv, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dlv[i] = dz[i] * eps[i] * (0.5f * expf(0.5f * lv[i]));
}

// ------------------------------------------------- gradient maps given as sums of sources (hand-scheduled BPTT)
// A gradient map [M][C] as the sum of up to three sources, each a stack of K-split slabs of a wider tensor read through a
// column window: src[s][m][col_off + c].  The recurrent core's backward hands the data-gradient convs' raw slabs straight
// to their consumers (the cell kernel below, rac_grad_sum, rac_reparam_head_bwd): no combine pass, no accumulation add.
struct GradSrcs {
  const float* p[3];
  long slab_stride[3];
  int n_slabs[3], row_stride[3], col_off[3];
  int n;
};

// (a window may start at any column -- e.g. behind the 10 tiled action / state channels of an input conv's gradient --
// so source loads promise 4-byte alignment only; gfx950 global loads take dword-aligned dwordx4 accesses)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ f32x4 grad_src_load4(const GradSrcs& S, long m, int c) {
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k < S.n) {
      const float* q = S.p[k] + m * S.row_stride[k] + S.col_off[k] + c;
      for (int s = 0; s < S.n_slabs[k]; ++s) {
        const f32x4u v = *reinterpret_cast<const f32x4u*>(q + s * S.slab_stride[k]);
        a += f32x4{v.x, v.y, v.z, v.w};
      }
    }
  return a;
}

__global__ void grad_sum_kernel(GradSrcs S, float* out, long M, int C, unsigned* amax) {
  const int C4 = C / 4;
  const long n4 = M * C4;
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C4;
    const int c = (int)(i - m * C4) * 4;
    const f32x4 a = grad_src_load4(S, m, c);
    mx = max(max(mx, max(absbits(a.x), absbits(a.y))), max(absbits(a.z), absbits(a.w)));
    *reinterpret_cast<f32x4*>(out + m * C + c) = a;
  }
  if (amax) amax_commit_block(mx, amax);
}

// lstm_cell_bwd_kernel with dh = sum of sources, four channels per thread
__global__ void lstm_cell_bwd_srcs_kernel(GradSrcs S, const float* dc_next, const float* act, const float* c_prev,
                                          const float* c_new, float* dgates, float* dc_prev, long M, int g,
                                          unsigned* amax) {
  const int g4 = g / 4;
  const long n4 = M * g4;
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long m = i / g4;
    const int c = (int)(i - m * g4) * 4;
    const f32x4 dh = grad_src_load4(S, m, c);
    const float* a = act + m * 4 * g + c;
    const f32x4 gi = *reinterpret_cast<const f32x4*>(a), gf = *reinterpret_cast<const f32x4*>(a + g),
                go = *reinterpret_cast<const f32x4*>(a + 2 * g), gg = *reinterpret_cast<const f32x4*>(a + 3 * g);
    const f32x4 cn = *reinterpret_cast<const f32x4*>(c_new + m * g + c), cp = *reinterpret_cast<const f32x4*>(c_prev + m * g + c);
    f32x4 dcn = {0.f, 0.f, 0.f, 0.f};
    if (dc_next) dcn = *reinterpret_cast<const f32x4*>(dc_next + m * g + c);
    f32x4 d0, d1, d2, d3, dcp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // the arithmetic of lstm_cell_bwd_kernel, element by element
      const float tc = tanhf(cn[e]);
      const float dhv = dh[e];
      const float dc = dhv * go[e] * (1.f - tc * tc) + dcn[e];
      d0[e] = dc * gg[e] * gi[e] * (1.f - gi[e]);
      d1[e] = dc * cp[e] * gf[e] * (1.f - gf[e]);
      d2[e] = dhv * tc * go[e] * (1.f - go[e]);
      d3[e] = dc * gi[e] * (1.f - gg[e] * gg[e]);
      dcp[e] = dc * gf[e];
      mx = max(max(mx, absbits(d0[e])), max(max(absbits(d1[e]), absbits(d2[e])), absbits(d3[e])));
    }
    float* d = dgates + m * 4 * g + c;
    *reinterpret_cast<f32x4*>(d) = d0;
    *reinterpret_cast<f32x4*>(d + g) = d1;
    *reinterpret_cast<f32x4*>(d + 2 * g) = d2;
    *reinterpret_cast<f32x4*>(d + 3 * g) = d3;
    *reinterpret_cast<f32x4*>(dc_prev + m * g + c) = dcp;
  }
  if (amax) amax_commit_block(mx, amax);
}

// backward of z = eps * exp(0.5 logvar) + mu feeding the merged mu | logvar head: dy[m] = [dz + dmu_add | dz * eps * 0.5 *
// exp(0.5 logvar) + dlv_add]  (dz a sum of sources; the addends are the KL term's gradients, nullable)
__global__ void reparam_head_bwd_kernel(GradSrcs S, const float* lv, const float* eps, const float* dmu_add,
                                        const float* dlv_add, float* dy, long M, int z, unsigned* amax) {
  const int z4 = z / 4;
  const long n4 = M * z4;
  unsigned mx = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long m = i / z4;
    const int c = (int)(i - m * z4) * 4;
    const f32x4 dz = grad_src_load4(S, m, c);
    const f32x4 l = *reinterpret_cast<const f32x4*>(lv + m * z + c), e = *reinterpret_cast<const f32x4*>(eps + m * z + c);
    f32x4 a = dz, b;
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = dz[k] * e[k] * (0.5f * expf(0.5f * l[k]));
    if (dmu_add) a += *reinterpret_cast<const f32x4*>(dmu_add + m * z + c);
    if (dlv_add) b += *reinterpret_cast<const f32x4*>(dlv_add + m * z + c);
    *reinterpret_cast<f32x4*>(dy + m * 2 * z + c) = a;
    *reinterpret_cast<f32x4*>(dy + m * 2 * z + z + c) = b;
#pragma unroll
    for (int k = 0; k < 4; ++k) mx = max(mx, max(absbits(a[k]), absbits(b[k])));
  }
  if (amax) amax_commit_block(mx, amax);
}

// ----------------------------------------------------------------------- Adam
__global__ void adam_kernel(f32x4* p, const f32x4* g, f32x4* m, f32x4* v, long n4, float* pt, const float* gt,
                            float* mt, float* vt, int tail, float b1, float b2, float eps, float step_size,
                            float inv_sqrt_bc2) {
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    adam_update(pp, gg, mm, vv, b1, b2, eps, step_size, inv_sqrt_bc2);
  };
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 P = p[i], G = g[i], Mv = m[i], V = v[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = P[e], me = Mv[e], ve = V[e];
      upd(pe, G[e], me, ve);
      P[e] = pe, Mv[e] = me, V[e] = ve;
    }
    p[i] = P;
    m[i] = Mv;
    v[i] = V;
  }
  if (blockIdx.x == 0 && threadIdx.x < tail) upd(pt[threadIdx.x], gt[threadIdx.x], mt[threadIdx.x], vt[threadIdx.x]);
}

// Adam over a list of [begin, begin + n) element ranges of the flat buffers (what rac_adam_frag_multi does not cover:
// biases, BatchNorm / GroupNorm affines, the few conv weights off the split-precision pipe); job table in device memory,
// a workgroup per 1024 float4.
__global__ void adam_ranges_kernel(f32x4* p, const f32x4* g, f32x4* m, f32x4* v, const rac_adam_range* ranges, int n_ranges,
                                   float b1, float b2, float eps, float step_size, float inv_sqrt_bc2) {
  int lo = 0, hi = n_ranges - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (ranges[mid].block_begin <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const rac_adam_range r = ranges[lo];
  const long first = r.begin4 + (blockIdx.x - r.block_begin) * 1024L, last = min(r.begin4 + r.n4, first + 1024L);
  for (long i = first + threadIdx.x; i < last; i += blockDim.x) {
    f32x4 P = p[i], G = g[i], Mv = m[i], V = v[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = P[e], me = Mv[e], ve = V[e];
      adam_update(pe, G[e], me, ve, b1, b2, eps, step_size, inv_sqrt_bc2);
      P[e] = pe, Mv[e] = me, V[e] = ve;
    }
    p[i] = P, m[i] = Mv, v[i] = V;
  }
}

}  // namespace rac

using namespace rac;
#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int rac_version(void) { return RAC_ABI_VERSION; }
const char* rac_device_arch(void) { return "gfx950"; }
const char* rac_last_error(void) { return g_err; }

int rac_bn_finalize(const double* stats, int64_t count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, int32_t n_updates, float* scale, float* shift,
                    float* mean, float* invstd, int32_t C, int32_t groups, void* stream) {
  RAC_REQUIRE(stats && gamma && beta && scale && shift && mean && invstd && C > 0 && count > 0 && groups >= 1,
              "rac_bn_finalize: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, ST(stream), stats, (long)count, gamma, beta,
                     running_mean, running_var, momentum, eps, n_updates, scale, shift, mean, invstd, C, groups);
  return check_launch("rac_bn_finalize");
}

int rac_affine_act(const float* x, const float* scale, const float* shift, int32_t act, float* y, int64_t M,
                   int32_t C, int32_t groups, uint32_t* y_amax, void* stream) {
  RAC_REQUIRE(x && scale && shift && y && M > 0 && C > 0 && groups >= 1 && M % groups == 0, "rac_affine_act: bad args");
  long n = (long)M * C;
  if (C % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(scale) && aligned16(shift)) {
    hipLaunchKernelGGL(affine_act_kernel4, dim3(grid_for_amax(n / 4, y_amax)), dim3(256), 0, ST(stream), (const f32x4*)x,
                       (const f32x4*)scale, (const f32x4*)shift, act, (f32x4*)y, n / 4, C / 4, n / 4 / groups, y_amax);
  } else {
    hipLaunchKernelGGL(affine_act_kernel1, dim3(grid_for_amax(n, y_amax)), dim3(256), 0, ST(stream), x, scale, shift, act, y, n, C,
                       n / groups, y_amax);
  }
  return check_launch("rac_affine_act");
}

// (row blocks, 64-channel groups) with ~1024 workgroups in total and >= 16 rows per block
static dim3 reduce_grid(long M, int C, int* rows_per_block) {
  const int cgroups = cdiv(C, 64);
  long nb = 1024 / cgroups;
  if (nb < 1) nb = 1;
  if (nb > (M + 15) / 16) nb = (M + 15) / 16;
  *rows_per_block = (int)((M + nb - 1) / nb);
  return dim3(cdiv(M, *rows_per_block), cgroups);
}

static int rows_per_block_for(long M, int* nblocks) {
  long nb = M / 64;
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  int rpb = (int)((M + nb - 1) / nb);
  *nblocks = (int)((M + rpb - 1) / rpb);
  return rpb;
}

// Grid of a per-channel reduction (rows form, C = 4 * 2^k <= 1024) that ends in fp64 atomics: channel slices of 64 channels
// (blockIdx.y) x `bpg` row blocks per statistics group.  A workgroup's atomics cover its slice only, so W workgroups per
// group cost W / slices x 2 C atomics (~0.25 ns each, measured: they, not the bytes, bounded these passes on everything but
// the 64x64 layers) against bytes / (W x ~30 GB/s) of streaming: least at W = 0.365 sqrt(bytes x slices / 2 C).
static bool reduce_plan(int C, long Mg, int groups, long bytes, int* bpg, int* rows_per_block, int* C4s) {
  const int C4 = C / 4;
  if (C % 4 || C4 < 1 || C4 > 256 || (C4 & (C4 - 1))) return false;
  static const bool off = getenv("RAC_BN_ROWS") && getenv("RAC_BN_ROWS")[0] == '0';
  if (off) return false;
  static const bool old = getenv("RAC_BN_REDUCE_OLD") && getenv("RAC_BN_REDUCE_OLD")[0] == '1';  // A/B: one slice
  static const double coef = [] { const char* e = getenv("RAC_BN_REDUCE_COEF"); return e ? atof(e) : 0.365; }();
  const int slice = (old || C4 <= 16) ? C4 : 16;
  const int n_slices = C4 / slice;
  const int rpi = 256 / slice;
  const double per_group = (double)bytes / groups;
  long W = old ? (bytes / 2 <= (24L << 20) ? 256 : 1024) / groups : (long)(coef * sqrt(per_group * n_slices / (2.0 * C)) + 0.5);
  long nb = W / n_slices;
  if (nb < 1) nb = 1;
  if (nb * n_slices * groups > 2048) nb = 2048 / (n_slices * groups) > 0 ? 2048 / (n_slices * groups) : 1;
  const long most = (Mg + 2L * rpi - 1) / (2L * rpi);  // at least two passes per workgroup
  if (nb > most) nb = most;
  long rpb = (Mg + nb - 1) / nb;
  rpb = (rpb + rpi - 1) / rpi * rpi;
  *rows_per_block = (int)rpb;
  *bpg = (int)((Mg + rpb - 1) / rpb);
  *C4s = slice;
  return true;
}

// the row-walking BatchNorm backward kernels: C = 4 * 2^k <= 1024; `bpg` workgroups per statistics group (about
// `max_blocks` in total), each a whole number of 256 / (C / 4)-row passes
static bool bn_rows_form(int C, long Mg, int groups, int max_blocks, int* bpg, int* rows_per_block) {
  const int C4 = C / 4;
  if (C % 4 || C4 < 1 || C4 > 256 || (C4 & (C4 - 1))) return false;
  static const bool off = getenv("RAC_BN_ROWS") && getenv("RAC_BN_ROWS")[0] == '0';  // A/B switch: the element-indexed kernels
  if (off) return false;
  const int rpi = 256 / C4;
  long nb = max_blocks / groups;
  if (nb < 1) nb = 1;
  const long most = (Mg + 4L * rpi - 1) / (4L * rpi);  // at least four passes per workgroup
  if (nb > most) nb = most;
  long rpb = (Mg + nb - 1) / nb;
  rpb = (rpb + rpi - 1) / rpi * rpi;
  *rows_per_block = (int)rpb;
  *bpg = (int)((Mg + rpb - 1) / rpb);
  return true;
}

int rac_bn_apply_act(const double* stats, int64_t count, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, int32_t n_updates, const float* x, int32_t act, float* y,
                     float* scale, float* shift, float* mean, float* invstd, int64_t M, int32_t C, int32_t groups,
                     uint32_t* y_amax, void* stream) {
  RAC_REQUIRE(stats && gamma && beta && x && y && scale && shift && mean && invstd && C > 0 && count > 0 && groups >= 1 &&
                  M > 0 && M % groups == 0 && (running_mean == nullptr) == (running_var == nullptr),
              "rac_bn_apply_act: bad args");
  int bpg, rpb;
  static const int max_blocks_env = [] { const char* e = getenv("RAC_BN_APPLY_BLOCKS"); return e ? atoi(e) : 0; }();
  const int max_blocks = max_blocks_env > 0 ? max_blocks_env : (y_amax ? 512 : 2048);
  RAC_REQUIRE(bn_rows_form(C, M / groups, groups, max_blocks, &bpg, &rpb) && aligned16(x) && aligned16(y),
              "rac_bn_apply_act: C must be 4 * 2^k <= 1024 and the maps 16-byte aligned (use rac_bn_finalize + rac_affine_act)");
  hipLaunchKernelGGL(bn_apply_act_rows_kernel, dim3(bpg * groups), dim3(256), 0, ST(stream), stats, (long)count, gamma, beta,
                     running_mean, running_var, momentum, eps, n_updates, (const f32x4*)x, (f32x4*)y, scale, shift, mean,
                     invstd, (long)(M / groups), C / 4, groups, act, rpb, bpg, y_amax);
  return check_launch("rac_bn_apply_act");
}

// channel quads per slice of the small-layer kernels: as wide as leaves >= 32 workgroups (1, 2 or 4 quads)
static int small_bn_qs(int C) {
  const int C4 = C / 4;
  if (C4 % 4 == 0 && C4 / 4 >= 32) return 4;
  if (C4 % 2 == 0 && C4 / 2 >= 32) return 2;
  return 1;
}

int rac_bn_small_ok(int64_t M, int32_t C) {
  static const bool off = getenv("RAC_BN_SMALL") && getenv("RAC_BN_SMALL")[0] == '0';
  static const long max_bytes = [] { const char* e = getenv("RAC_BN_SMALL_BYTES"); return e ? atol(e) : (4L << 20); }();
  return !off && M > 0 && C >= 32 && C % 4 == 0 && C <= 4096 && (long)M * C * 4 <= max_bytes;
}

int rac_bn_small_fwd(const float* slabs, int32_t n_slabs, int64_t slab_stride, float* raw, float* y, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum, float eps, int32_t n_updates,
                     float* scale, float* shift, float* mean, float* invstd, int64_t M, int32_t C, int32_t act, uint32_t* y_amax,
                     void* stream) {
  RAC_REQUIRE(slabs && raw && y && gamma && beta && scale && shift && mean && invstd && n_slabs >= 1 && M > 0 && C > 0 &&
                  C % 4 == 0 && slab_stride % 4 == 0 && (running_mean == nullptr) == (running_var == nullptr),
              "rac_bn_small_fwd: bad args");
  RAC_REQUIRE(aligned16(slabs) && aligned16(raw) && aligned16(y), "rac_bn_small_fwd: 16-byte aligned maps");
  const int qs = small_bn_qs(C), C4 = C / 4;
  const dim3 grid(C4 / qs);
#define RAC_SMALL_FWD(Q)                                                                                                  \
  hipLaunchKernelGGL(bn_small_fwd_kernel<Q>, grid, dim3(256), 0, ST(stream), (const f32x4*)slabs, n_slabs,                  \
                     (long)(slab_stride / 4), (f32x4*)raw, (f32x4*)y, gamma, beta, running_mean, running_var, momentum, eps, \
                     n_updates, scale, shift, mean, invstd, (long)M, C4, act, y_amax)
  if (qs == 4) RAC_SMALL_FWD(4);
  else if (qs == 2) RAC_SMALL_FWD(2);
  else RAC_SMALL_FWD(1);
#undef RAC_SMALL_FWD
  return check_launch("rac_bn_small_fwd");
}

int rac_bn_small_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                     const float* invstd, float* dx, float* dgamma, float* dbeta, int64_t M, int32_t C, uint32_t* dx_amax,
                     void* stream) {
  RAC_REQUIRE(dy && x && scale && shift && mean && invstd && dx && M > 0 && C > 0 && C % 4 == 0 &&
                  (dgamma == nullptr) == (dbeta == nullptr),
              "rac_bn_small_bwd: bad args");
  RAC_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(scale) && aligned16(shift) && aligned16(mean) &&
                  aligned16(invstd),
              "rac_bn_small_bwd: 16-byte aligned operands");
  const int qs = small_bn_qs(C), C4 = C / 4;
  const dim3 grid(C4 / qs);
#define RAC_SMALL_BWD(Q)                                                                                              \
  hipLaunchKernelGGL(bn_small_bwd_kernel<Q>, grid, dim3(256), 0, ST(stream), (const f32x4*)dy, (const f32x4*)x,        \
                     (const f32x4*)scale, (const f32x4*)shift, (const f32x4*)mean, (const f32x4*)invstd, (f32x4*)dx, dgamma, \
                     dbeta, (long)M, C4, dx_amax)
  if (qs == 4) RAC_SMALL_BWD(4);
  else if (qs == 2) RAC_SMALL_BWD(2);
  else RAC_SMALL_BWD(1);
#undef RAC_SMALL_BWD
  return check_launch("rac_bn_small_bwd");
}

int rac_bn_bwd_reduce(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                      const float* invstd, double* sums, int64_t M, int32_t C, int32_t groups, void* stream) {
  RAC_REQUIRE(dy && x && scale && shift && mean && invstd && sums && M > 0 && C > 0 && groups >= 1 && M % groups == 0,
              "rac_bn_bwd_reduce: bad args");
  int rpb;
  const long Mg = M / groups;
  int bpg_rows, rpb_rows;
  // Every workgroup ends in 8 fp64 atomics per channel quad onto the same 2 C addresses of its group: what bounds the pass
  // on all but the largest tensors is the NUMBER of those atomics (workgroups x 2 C; measured ~0.25 ns each: the 8x8 x 512
  // layer of one time step, 4 MB of operands, took 25 us on 128 workgroups = 131 k atomics), not the bytes -- see
  // reduce_plan: channel slices, and the workgroup count that balances streaming time against the atomics behind it.
  // (Copies of the accumulators, a workgroup adding into copy index mod R, were built and measured in round 4: the reduce
  // pass gains what the apply pass then loses summing the copies.)
  int c4s;
  if (reduce_plan(C, Mg, groups, 2L * M * C * 4, &bpg_rows, &rpb_rows, &c4s) && aligned16(dy) && aligned16(x) && aligned16(scale) &&
      aligned16(shift) && aligned16(mean) && aligned16(invstd)) {
    hipLaunchKernelGGL(bn_bwd_reduce_rows_kernel, dim3(bpg_rows * groups, C / 4 / c4s), dim3(256), 0, ST(stream),
                       (const f32x4*)dy, (const f32x4*)x, (const f32x4*)scale, (const f32x4*)shift, (const f32x4*)mean,
                       (const f32x4*)invstd, sums, Mg, C / 4, c4s, rpb_rows, bpg_rows);
    return check_launch("rac_bn_bwd_reduce");
  }
  dim3 grid = reduce_grid(Mg, C, &rpb);
  const int bpg = (int)grid.x;
  grid.x *= groups;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, ST(stream), dy, x, scale, shift, mean, invstd, sums, Mg,
                     C, rpb, bpg);
  return check_launch("rac_bn_bwd_reduce");
}

int rac_bn_bwd_apply(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                     const float* invstd, const double* sums, float* dx, float* dgamma, float* dbeta, int64_t M,
                     int32_t C, int32_t groups, uint32_t* dx_amax, void* stream) {
  RAC_REQUIRE(dy && x && scale && shift && mean && invstd && sums && dx && M > 0 && C > 0 && groups >= 1 &&
                  M % groups == 0,
              "rac_bn_bwd_apply: bad args");
  RAC_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "rac_bn_bwd_apply: dgamma/dbeta must come together");
  int bpg_rows, rpb_rows;
  if (bn_rows_form(C, M / groups, groups, dx_amax ? 512 : 2048, &bpg_rows, &rpb_rows) && aligned16(dy) && aligned16(x) &&
      aligned16(dx)) {
    hipLaunchKernelGGL(bn_bwd_apply_rows_kernel, dim3(bpg_rows * groups), dim3(256), 0, ST(stream), (const f32x4*)dy,
                       (const f32x4*)x, scale, shift, mean, invstd, sums, (f32x4*)dx, dgamma, dbeta, (long)(M / groups), C / 4,
                       groups, rpb_rows, bpg_rows, dx_amax);
    return check_launch("rac_bn_bwd_apply");
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for_amax((long)M * C, dx_amax)), dim3(256), 0, ST(stream), dy, x, scale, shift,
                     mean, invstd, sums, dx, dgamma, dbeta, (long)(M / groups), C, groups, dx_amax);
  return check_launch("rac_bn_bwd_apply");
}

int rac_maxpool2_fwd(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  RAC_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, "rac_maxpool2_fwd: bad args");
  long n = (long)B * (H / 2) * (W / 2) * C;
  if (C % 4 == 0 && aligned16(x) && aligned16(y))
    hipLaunchKernelGGL(maxpool2_fwd_kernel<f32x4>, dim3(grid_for(n / 4)), dim3(256), 0, ST(stream), (const f32x4*)x,
                       (f32x4*)y, B, H, W, C / 4);
  else
    hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, y, B, H, W, C);
  return check_launch("rac_maxpool2_fwd");
}

int rac_maxpool2_bwd(const float* x, const float* dy, float* dx, int32_t B, int32_t H, int32_t W, int32_t C,
                     void* stream) {
  RAC_REQUIRE(x && dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0,
              "rac_maxpool2_bwd: bad args");
  long n = (long)B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, dy, dx, B, H, W, C);
  return check_launch("rac_maxpool2_bwd");
}

int rac_upsample2_fwd(const float* x, float* y, int32_t B, int32_t h, int32_t w, int32_t C, void* stream) {
  RAC_REQUIRE(x && y && B > 0 && h > 0 && w > 0 && C > 0, "rac_upsample2_fwd: bad args");
  long n = (long)B * 4 * h * w * C;
  if (C % 4 == 0 && aligned16(x) && aligned16(y))
    hipLaunchKernelGGL(upsample2_fwd_kernel<f32x4>, dim3(grid_for(n / 4)), dim3(256), 0, ST(stream), (const f32x4*)x,
                       (f32x4*)y, B, h, w, C / 4);
  else
    hipLaunchKernelGGL(upsample2_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, y, B, h, w, C);
  return check_launch("rac_upsample2_fwd");
}

int rac_upsample2_bwd(const float* dy, float* dx, int32_t B, int32_t h, int32_t w, int32_t C, void* stream) {
  RAC_REQUIRE(dy && dx && B > 0 && h > 0 && w > 0 && C > 0, "rac_upsample2_bwd: bad args");
  long n = (long)B * h * w * C;
  if (C % 4 == 0 && aligned16(dy) && aligned16(dx))
    hipLaunchKernelGGL(upsample2_bwd_kernel<f32x4>, dim3(grid_for(n / 4)), dim3(256), 0, ST(stream), (const f32x4*)dy,
                       (f32x4*)dx, B, h, w, C / 4);
  else
    hipLaunchKernelGGL(upsample2_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), dy, dx, B, h, w, C);
  return check_launch("rac_upsample2_bwd");
}

int rac_tilecat_fwd(const float* v0, int32_t n0, const float* v1, int32_t n1, const float* v2, int32_t n2,
                    const float* m0, int32_t c0, const float* m1, int32_t c1, int32_t pad, float* out, int32_t B,
                    int32_t HW, uint32_t* out_amax, int32_t amax_per_image, void* stream) {
  RAC_REQUIRE(out && B > 0 && HW > 0 && pad >= 0, "rac_tilecat_fwd: bad args");
  RAC_REQUIRE((n0 == 0 || v0) && (n1 == 0 || v1) && (n2 == 0 || v2) && (c0 == 0 || m0) && (c1 == 0 || m1),
              "rac_tilecat_fwd: null source with non-zero width");
  long n = (long)B * HW * (n0 + n1 + n2 + c0 + c1 + pad);
  RAC_REQUIRE(n > 0, "rac_tilecat_fwd: empty");
  if (amax_per_image && out_amax) {
    RAC_REQUIRE(n / B < 0x7FFFFFFFL, "rac_tilecat_fwd: image too large");
    hipLaunchKernelGGL(tilecat_image_kernel, dim3(B), dim3(256), 0, ST(stream), v0, n0, v1, n1, v2, n2, m0, c0, m1, c1, pad,
                       out, HW, out_amax);
    return check_launch("rac_tilecat_fwd");
  }
  hipLaunchKernelGGL(tilecat_kernel, dim3(grid_for_amax(n, out_amax)), dim3(256), 0, ST(stream), v0, n0, v1, n1, v2, n2, m0, c0, m1,
                     c1, pad, out, B, HW, out_amax);
  return check_launch("rac_tilecat_fwd");
}

int rac_absmax_rows(const float* x, int64_t rows, int64_t row_len, uint32_t* amax, void* stream) {
  RAC_REQUIRE(x && amax && rows > 0 && rows < 0x7FFFFFFFL && row_len > 0, "rac_absmax_rows: bad args");
  RAC_REQUIRE(row_len % 4 == 0 && aligned16(x), "rac_absmax_rows: row_len % 4 == 0, 16-byte aligned rows");
  hipLaunchKernelGGL(absmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, ST(stream), (const f32x4*)x,
                     (long)(row_len / 4), amax);
  return check_launch("rac_absmax_rows");
}

int rac_pad_rows(const float* src, int32_t C, float* dst, int32_t Cpad, int64_t R, void* stream) {
  RAC_REQUIRE(src && dst && C > 0 && Cpad >= C && R > 0, "rac_pad_rows: bad args");
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for((long)R * Cpad)), dim3(256), 0, ST(stream), src, C, dst, Cpad,
                     (long)R);
  return check_launch("rac_pad_rows");
}

int rac_unpad_add(const float* src, int32_t Cpad, float* dst, int32_t C, int64_t R, void* stream) {
  RAC_REQUIRE(src && dst && C > 0 && Cpad >= C && R > 0, "rac_unpad_add: bad args");
  hipLaunchKernelGGL(unpad_add_kernel, dim3(grid_for((long)R * C)), dim3(256), 0, ST(stream), src, Cpad, dst, C,
                     (long)R);
  return check_launch("rac_unpad_add");
}

int rac_slice_channels(const float* src, int32_t Csrc, int32_t off, int32_t nc, float* dst, int64_t M, void* stream) {
  RAC_REQUIRE(src && dst && M > 0 && nc > 0 && off >= 0 && off + nc <= Csrc, "rac_slice_channels: bad args");
  hipLaunchKernelGGL(slice_channels_kernel, dim3(grid_for((long)M * nc)), dim3(256), 0, ST(stream), src, Csrc, off, nc,
                     dst, (long)M);
  return check_launch("rac_slice_channels");
}

int rac_cat2_channels(const float* a, int32_t Ca, const float* b, int32_t Cb, float* dst, int64_t M, uint32_t* out_amax,
                      void* stream) {
  RAC_REQUIRE(dst && M > 0 && Ca > 0 && Cb > 0, "rac_cat2_channels: bad args");
  hipLaunchKernelGGL(cat2_channels_kernel, dim3(grid_for_amax((long)M * (Ca + Cb), out_amax)), dim3(256), 0, ST(stream),
                     a, Ca, b, Cb, dst, (long)M, out_amax);
  return check_launch("rac_cat2_channels");
}

int64_t rac_colsum_blocks(int64_t M, int32_t C) {
  if (M <= 0 || C <= 0) return 0;
  int rpb;
  return (int64_t)reduce_grid(M, C, &rpb).x;
}

int rac_colsum_steps(const float* const* xs, int32_t T, float* out, float* parts, int64_t M, int32_t C, void* stream) {
  RAC_REQUIRE(xs && out && T >= 1 && T <= RAC_WGRAD_MAX_STEPS && M > 0 && C > 0, "rac_colsum_steps: bad args");
  ColsumSteps p{};
  p.T = T;
  for (int t = 0; t < T; ++t) {
    RAC_REQUIRE(xs[t], "rac_colsum_steps: null step");
    p.x[t] = xs[t];
  }
  int rpb;
  dim3 grid = reduce_grid(M, C, &rpb);
  hipLaunchKernelGGL(colsum_steps_kernel, grid, dim3(256), 0, ST(stream), p, out, parts, (long)M, C, rpb);
  if (parts)
    hipLaunchKernelGGL(colsum_parts_add_kernel, dim3(cdiv(C, 64)), dim3(256), 0, ST(stream), parts, (int)grid.x, out, C);
  return check_launch("rac_colsum_steps");
}

int rac_colsum_acc(const float* x, float* out, float* parts, int64_t M, int32_t C, void* stream) {
  RAC_REQUIRE(x && out && M > 0 && C > 0, "rac_colsum_acc: bad args");
  int rpb;
  dim3 grid = reduce_grid(M, C, &rpb);
  hipLaunchKernelGGL(colsum_acc_kernel, grid, dim3(256), 0, ST(stream), x, out, parts, (long)M, C, rpb);
  if (parts)
    hipLaunchKernelGGL(colsum_parts_add_kernel, dim3(cdiv(C, 64)), dim3(256), 0, ST(stream), parts, (int)grid.x, out, C);
  return check_launch("rac_colsum_acc");
}

int rac_slab_reduce(const float* slabs, int32_t n_slabs, int64_t slab_stride, const float* bias, float* out,
                    int64_t n, int32_t N, uint32_t* out_amax, void* stream) {
  RAC_REQUIRE(slabs && out && n_slabs >= 1 && n > 0 && N > 0, "rac_slab_reduce: bad args");
  if (n % 4 == 0 && N % 4 == 0 && slab_stride % 4 == 0 && aligned16(slabs) && aligned16(out) && (!bias || aligned16(bias))) {
    hipLaunchKernelGGL(slab_reduce_kernel4, dim3(grid_for_amax(n / 4, out_amax)), dim3(256), 0, ST(stream),
                       (const f32x4*)slabs, n_slabs, (long)(slab_stride / 4), (const f32x4*)bias, (f32x4*)out, (long)(n / 4),
                       N / 4, out_amax);
    return check_launch("rac_slab_reduce");
  }
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for_amax(n, out_amax)), dim3(256), 0, ST(stream), slabs, n_slabs,
                     (long)slab_stride, bias, out, (long)n, N, out_amax);
  return check_launch("rac_slab_reduce");
}

int rac_slab_reduce2(const float* slabs, int32_t n_slabs, int64_t slab_stride, const float* bias, float* out0,
                     float* out1, int64_t M, int32_t N, int32_t o_split, uint32_t* out0_amax, uint32_t* out1_amax,
                     void* stream) {
  RAC_REQUIRE(slabs && out0 && out1 && n_slabs >= 1 && M > 0 && N > 0 && o_split > 0 && o_split < N,
              "rac_slab_reduce2: bad args");
  if (N % 4 == 0 && o_split % 4 == 0 && slab_stride % 4 == 0 && aligned16(slabs) && aligned16(out0) && aligned16(out1) &&
      (!bias || aligned16(bias))) {
    hipLaunchKernelGGL(slab_reduce2_kernel4, dim3(grid_for_amax((long)M * N / 4, out0_amax ? out0_amax : out1_amax)),
                       dim3(256), 0, ST(stream), (const f32x4*)slabs, n_slabs, (long)(slab_stride / 4), (const f32x4*)bias,
                       (f32x4*)out0, (f32x4*)out1, (long)M, N / 4, o_split / 4, out0_amax, out1_amax);
    return check_launch("rac_slab_reduce2");
  }
  hipLaunchKernelGGL(slab_reduce2_kernel, dim3(grid_for_amax((long)M * N, out0_amax ? out0_amax : out1_amax)), dim3(256),
                     0, ST(stream), slabs, n_slabs, (long)slab_stride, bias, out0, out1, (long)M, N, o_split, out0_amax,
                     out1_amax);
  return check_launch("rac_slab_reduce2");
}

static bool bn_rows_form(int C, long Mg, int groups, int max_blocks, int* bpg, int* rows_per_block);

int rac_slab_reduce_stats(const float* slabs, int32_t n_slabs, int64_t slab_stride, float* out, double* stats, int64_t M,
                          int32_t C, int32_t groups, uint32_t* out_amax, void* stream) {
  RAC_REQUIRE(slabs && out && stats && n_slabs >= 1 && M > 0 && C > 0 && groups >= 1 && M % groups == 0,
              "rac_slab_reduce_stats: bad args");
  int bpg, rpb, c4s;
  // (the statistics' fp64 atomics bound this pass too: see reduce_plan; bytes = the slabs read + the map written)
  RAC_REQUIRE(reduce_plan(C, M / groups, groups, (long)(n_slabs + 1) * M * C * 4, &bpg, &rpb, &c4s) && slab_stride % 4 == 0 &&
                  aligned16(slabs) && aligned16(out),
              "rac_slab_reduce_stats: C must be 4 * 2^k <= 1024, 16-byte aligned slabs (use rac_slab_reduce + rac_col_stats)");
  hipLaunchKernelGGL(slab_reduce_stats_rows_kernel, dim3(bpg * groups, C / 4 / c4s), dim3(256), 0, ST(stream),
                     (const f32x4*)slabs, n_slabs, (long)(slab_stride / 4), (f32x4*)out, stats, (long)(M / groups), C / 4, c4s,
                     rpb, bpg, out_amax);
  return check_launch("rac_slab_reduce_stats");
}

int rac_col_stats(const float* x, double* stats, int64_t M, int32_t C, int32_t groups, void* stream) {
  RAC_REQUIRE(x && stats && M > 0 && C > 0 && groups >= 1 && M % groups == 0, "rac_col_stats: bad args");
  int rpb;
  const long Mg = M / groups;
  dim3 grid = reduce_grid(Mg, C, &rpb);
  const int bpg = (int)grid.x;
  grid.x *= groups;
  hipLaunchKernelGGL(col_stats_kernel, grid, dim3(256), 0, ST(stream), x, stats, Mg, C, rpb, bpg);
  return check_launch("rac_col_stats");
}

int rac_act_bwd(const float* dy, const float* y, int32_t act, float* dx, int64_t n, void* stream) {
  RAC_REQUIRE(dy && y && dx && n > 0, "rac_act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), dy, y, act, dx, (long)n);
  return check_launch("rac_act_bwd");
}

int rac_lstm_cell_fwd(const float* gate_slabs, int32_t n_slabs, int64_t slab_stride, const float* bias,
                      const float* c_prev, float* h_out, float* c_out, float* act_out, int64_t M, int32_t g,
                      void* stream) {
  RAC_REQUIRE(gate_slabs && bias && c_prev && h_out && c_out && M > 0 && g > 0 && n_slabs >= 1,
              "rac_lstm_cell_fwd: bad args");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(grid_for((long)M * g)), dim3(256), 0, ST(stream), gate_slabs, n_slabs,
                     (long)slab_stride, bias, c_prev, h_out, c_out, act_out, (long)M, g);
  return check_launch("rac_lstm_cell_fwd");
}

int rac_lstm_cell_bwd(const float* dh, const float* dc_next, const float* act, const float* c_prev,
                      const float* c_new, float* dgates, float* dc_prev, int64_t M, int32_t g, uint32_t* dgates_amax,
                      void* stream) {
  RAC_REQUIRE(act && c_prev && c_new && dgates && dc_prev && M > 0 && g > 0, "rac_lstm_cell_bwd: bad args");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid_for_amax((long)M * g, dgates_amax)), dim3(256), 0, ST(stream), dh, dc_next, act,
                     c_prev, c_new, dgates, dc_prev, (long)M, g, dgates_amax);
  return check_launch("rac_lstm_cell_bwd");
}

int rac_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, int64_t n, void* stream) {
  RAC_REQUIRE(mu && logvar && eps && z && n > 0, "rac_reparam_fwd: bad args");
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), mu, logvar, eps, z, (long)n);
  return check_launch("rac_reparam_fwd");
}
int rac_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dlogvar, int64_t n, void* stream) {
  RAC_REQUIRE(dz && logvar && eps && dlogvar && n > 0, "rac_reparam_bwd: bad args");
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), dz, logvar, eps, dlogvar,
                     (long)n);
  return check_launch("rac_reparam_bwd");
}

static int pack_srcs(const rac_grad_src* srcs, int32_t n_srcs, int32_t C, GradSrcs* S, const char* who) {
  RAC_REQUIRE(n_srcs >= 0 && n_srcs <= 3 && (n_srcs == 0 || srcs), "%s: 0..3 gradient sources", who);
  S->n = n_srcs;
  for (int k = 0; k < 3; ++k) {
    S->p[k] = nullptr;
    S->slab_stride[k] = 0;
    S->n_slabs[k] = S->row_stride[k] = S->col_off[k] = 0;
  }
  for (int k = 0; k < n_srcs; ++k) {
    const rac_grad_src& q = srcs[k];
    RAC_REQUIRE(q.p && q.n_slabs >= 1 && q.col_off >= 0 && q.row_stride >= q.col_off + C, "%s: source %d out of its rows",
                who, k);
    RAC_REQUIRE((reinterpret_cast<uintptr_t>(q.p) & 3u) == 0, "%s: source %d is not a float pointer", who, k);
    S->p[k] = q.p, S->slab_stride[k] = (long)q.slab_stride, S->n_slabs[k] = q.n_slabs, S->row_stride[k] = q.row_stride,
    S->col_off[k] = q.col_off;
  }
  return RAC_OK;
}

int rac_grad_sum(const rac_grad_src* srcs, int32_t n_srcs, float* out, int64_t M, int32_t C, uint32_t* out_amax,
                 void* stream) {
  RAC_REQUIRE(out && M > 0 && C > 0 && C % 4 == 0 && n_srcs >= 1 && aligned16(out), "rac_grad_sum: bad args");
  GradSrcs S;
  if (int e = pack_srcs(srcs, n_srcs, C, &S, "rac_grad_sum")) return e;
  hipLaunchKernelGGL(grad_sum_kernel, dim3(grid_for_amax((long)M * C / 4, out_amax)), dim3(256), 0, ST(stream), S, out,
                     (long)M, C, out_amax);
  return check_launch("rac_grad_sum");
}

int rac_lstm_cell_bwd_srcs(const rac_grad_src* dh_srcs, int32_t n_srcs, const float* dc_next, const float* act,
                           const float* c_prev, const float* c_new, float* dgates, float* dc_prev, int64_t M, int32_t g,
                           uint32_t* dgates_amax, void* stream) {
  RAC_REQUIRE(act && c_prev && c_new && dgates && dc_prev && M > 0 && g > 0 && g % 4 == 0,
              "rac_lstm_cell_bwd_srcs: bad args");
  RAC_REQUIRE(aligned16(act) && aligned16(c_prev) && aligned16(c_new) && aligned16(dgates) && aligned16(dc_prev) &&
                  (!dc_next || aligned16(dc_next)),
              "rac_lstm_cell_bwd_srcs: 16-byte aligned maps");
  GradSrcs S;
  if (int e = pack_srcs(dh_srcs, n_srcs, g, &S, "rac_lstm_cell_bwd_srcs")) return e;
  hipLaunchKernelGGL(lstm_cell_bwd_srcs_kernel, dim3(grid_for_amax((long)M * g / 4, dgates_amax)), dim3(256), 0,
                     ST(stream), S, dc_next, act, c_prev, c_new, dgates, dc_prev, (long)M, g, dgates_amax);
  return check_launch("rac_lstm_cell_bwd_srcs");
}

int rac_reparam_head_bwd(const rac_grad_src* dz_srcs, int32_t n_srcs, const float* logvar, const float* eps,
                         const float* dmu_add, const float* dlogvar_add, float* dy, int64_t M, int32_t z,
                         uint32_t* dy_amax, void* stream) {
  RAC_REQUIRE(logvar && eps && dy && M > 0 && z > 0 && z % 4 == 0 && n_srcs >= 1, "rac_reparam_head_bwd: bad args");
  RAC_REQUIRE(aligned16(logvar) && aligned16(eps) && aligned16(dy) && (!dmu_add || aligned16(dmu_add)) &&
                  (!dlogvar_add || aligned16(dlogvar_add)),
              "rac_reparam_head_bwd: 16-byte aligned maps");
  GradSrcs S;
  if (int e = pack_srcs(dz_srcs, n_srcs, z, &S, "rac_reparam_head_bwd")) return e;
  hipLaunchKernelGGL(reparam_head_bwd_kernel, dim3(grid_for_amax((long)M * z / 4, dy_amax)), dim3(256), 0, ST(stream), S,
                     logvar, eps, dmu_add, dlogvar_add, dy, (long)M, z, dy_amax);
  return check_launch("rac_reparam_head_bwd");
}

int rac_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int32_t step, void* stream) {
  RAC_REQUIRE(p && g && m && v && n > 0 && step >= 1, "rac_adam_step: bad args");
  RAC_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "rac_adam_step: buffers must be 16-B aligned");
  double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  float step_size = (float)((double)lr / bc1);
  float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  long n4 = n / 4;
  int tail = (int)(n - n4 * 4);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(256), 0, ST(stream), (f32x4*)p,
                     (const f32x4*)g, (f32x4*)m, (f32x4*)v, n4, p + n4 * 4, g + n4 * 4, m + n4 * 4, v + n4 * 4, tail,
                     beta1, beta2, eps, step_size, inv_sqrt_bc2);
  return check_launch("rac_adam_step");
}

int rac_adam_ranges(float* p, const float* g, float* m, float* v, const rac_adam_range* ranges, int32_t n_ranges,
                    int64_t total_blocks, float lr, float beta1, float beta2, float eps, int32_t step, void* stream) {
  RAC_REQUIRE(p && g && m && v && ranges && n_ranges > 0 && total_blocks >= n_ranges && total_blocks < 0x7FFFFFFFL && step >= 1,
              "rac_adam_ranges: bad args");
  RAC_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "rac_adam_ranges: buffers must be 16-B aligned");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_ranges_kernel, dim3((unsigned)total_blocks), dim3(256), 0, ST(stream), (f32x4*)p, (const f32x4*)g,
                     (f32x4*)m, (f32x4*)v, ranges, n_ranges, beta1, beta2, eps, (float)((double)lr / bc1),
                     (float)(1.0 / sqrt(bc2)));
  return check_launch("rac_adam_ranges");
}

}  // extern "C"

// ------------------------------------------------------------------ GroupNorm / NormConvLSTMCell pieces
namespace rac {

__device__ __forceinline__ float block_sum256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// grid (G, B), 256 threads: one (sample, group) per workgroup; the group is HW rows of Cg contiguous channels.
__global__ void groupnorm_fwd_kernel(const float* x, const float* gamma, const float* beta, float* y, float* mean_o,
                                     float* rstd_o, int HW, int C, int G, float eps) {
  __shared__ float sh[4];
  const int g = blockIdx.x, b = blockIdx.y;
  const int Cg = C / G;
  const int n = HW * Cg;
  const float* xb = x + (long)b * HW * C + g * Cg;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += xb[(long)(i / Cg) * C + (i % Cg)];
  const float mean = block_sum256(s, sh) / n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float d = xb[(long)(i / Cg) * C + (i % Cg)] - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(block_sum256(q, sh) / n + eps);
  float* yb = y + (long)b * HW * C + g * Cg;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i % Cg;
    const long o = (long)(i / Cg) * C + c;
    yb[o] = (xb[o] - mean) * rstd * gamma[g * Cg + c] + beta[g * Cg + c];
  }
  if (threadIdx.x == 0) {
    mean_o[b * G + g] = mean;
    rstd_o[b * G + g] = rstd;
  }
}

// NormConvLSTMCell without a tape (lstm.py:174-198) in ONE launch: GroupNorm(16, 4g) of the two gate convs' outputs, the
// gate activations, c_raw = f c_prev + i g~, GroupNorm(16, g) of c_raw and h = o tanh(c) -- five launches and four more
// trips of the 4g-wide gate tensors through memory otherwise (23 + 7 + 1.5 ms of a 322 ms planner iteration on the
// deployed model).  The work decomposes exactly: quarter q of the channels, [q g/4, (q + 1) g/4), needs gate groups q,
// 4 + q, 8 + q, 12 + q of both convs (gate k of channel c is channel k g + c: group 4 k + q) and cell groups 4 q .. 4 q + 3,
// all of them whole.  grid (4, B), 256 threads: one (image, quarter) per workgroup, its 8 + 1 slabs of HW x g/4 values read
// three times (mean; centred squares, as groupnorm_fwd_kernel; gates) -- the first time from memory, then from the caches.
// An image's arithmetic touches nothing of another image: batch-invariant by construction.
__global__ __launch_bounds__(512) void norm_lstm_cell_fwd_kernel(const float* g_ih, const float* g_hh, const float* c_prev,
                                                                 const float* gam_ih, const float* bet_ih,
                                                                 const float* gam_hh, const float* bet_hh,
                                                                 const float* gam_c, const float* bet_c, float* h_out,
                                                                 float* c_out, float* act_out, float* craw_out,
                                                                 float* stat_ih, float* stat_hh, float* stat_c, int HW, int g,
                                                                 float eps) {
  // (training: act_out [B][HW][4g] = the activated gates, craw_out = the cell before its norm, stat_* [2][B][16] = mean and
  // 1 / std of every (image, group) -- what rac_lstm_out_bwd / rac_groupnorm_bwd / rac_lstm_core_bwd read; NULL: the frozen model)
  __shared__ float red[8][12];
  const int NT = blockDim.x, NW = NT >> 6;  // 256 threads, or 512 for small batches (few workgroups: more loads in flight each)
  const int q = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int Cq = g >> 2, Q4 = Cq >> 2;          // channels of the quarter, 16-byte vectors per pixel of it
  const int nq = HW * Q4;                       // vectors per slab
  const int cg = g >> 4;                        // channels per cell group
  const long row4 = 4L * g;                     // gate tensors: floats per pixel
  const float* ih = g_ih + (long)b * HW * row4 + q * Cq;
  const float* hh = g_hh + (long)b * HW * row4 + q * Cq;
  const float inv_n = 1.0f / (float)(HW * Cq);
  auto block8 = [&](float (&v)[8]) {            // sums of 8 values over the workgroup, in every thread
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
      for (int k = 0; k < 8; ++k) red[wv][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
      if (NW == 8) t += (red[4][k] + red[5][k]) + (red[6][k] + red[7][k]);
      v[k] = t;
    }
  };
  float mean[8], rstd[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) mean[k] = 0.f;
  for (int i = tid; i < nq; i += NT) {
    const int p = i / Q4, c4 = i - p * Q4;
    const long o = (long)p * row4 + 4 * c4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ih + o + (long)k * g);
      const f32x4 h = *reinterpret_cast<const f32x4*>(hh + o + (long)k * g);
      mean[k] += (a.x + a.y) + (a.z + a.w);
      mean[4 + k] += (h.x + h.y) + (h.z + h.w);
    }
  }
  block8(mean);
#pragma unroll
  for (int k = 0; k < 8; ++k) mean[k] *= inv_n, rstd[k] = 0.f;
  for (int i = tid; i < nq; i += NT) {
    const int p = i / Q4, c4 = i - p * Q4;
    const long o = (long)p * row4 + 4 * c4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ih + o + (long)k * g);
      const f32x4 h = *reinterpret_cast<const f32x4*>(hh + o + (long)k * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float da = a[e] - mean[k], dh = h[e] - mean[4 + k];
        rstd[k] += da * da;
        rstd[4 + k] += dh * dh;
      }
    }
  }
  block8(rstd);
#pragma unroll
  for (int k = 0; k < 8; ++k) rstd[k] = 1.0f / sqrtf(rstd[k] * inv_n + eps);
  if (stat_ih && tid == 0) {
    const int nB = gridDim.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      stat_ih[b * 16 + 4 * k + q] = mean[k], stat_ih[(nB + b) * 16 + 4 * k + q] = rstd[k];
      stat_hh[b * 16 + 4 * k + q] = mean[4 + k], stat_hh[(nB + b) * 16 + 4 * k + q] = rstd[4 + k];
    }
  }
  // gates and the raw cell; the quarter's four cell groups: a thread's vectors all lie in ONE group when Q4 divides 256
  // or 256 divides Q4's multiples -- the launcher's condition (Q4 a power of two <= 256), so one accumulator serves
  const float* cp = c_prev + (long)b * HW * g + q * Cq;
  float* co = c_out + (long)b * HW * g + q * Cq;
  float* ho = h_out + (long)b * HW * g + q * Cq;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < nq; i += NT) {
    const int p = i / Q4, c4 = i - p * Q4;
    const long o = (long)p * row4 + 4 * c4;
    f32x4 pre[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ih + o + (long)k * g);
      const f32x4 h = *reinterpret_cast<const f32x4*>(hh + o + (long)k * g);
      const int ch = k * g + q * Cq + 4 * c4;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gam_ih + ch), ba = *reinterpret_cast<const f32x4*>(bet_ih + ch);
      const f32x4 gh = *reinterpret_cast<const f32x4*>(gam_hh + ch), bh = *reinterpret_cast<const f32x4*>(bet_hh + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        pre[k][e] = ((a[e] - mean[k]) * rstd[k] * ga[e] + ba[e]) + ((h[e] - mean[4 + k]) * rstd[4 + k] * gh[e] + bh[e]);
    }
    const f32x4 cv = *reinterpret_cast<const f32x4*>(cp + (long)p * g + 4 * c4);
    f32x4 cr, gi, gf, gg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gi[e] = sigmoid_acc(pre[0][e]), gf[e] = sigmoid_acc(pre[1][e]), gg[e] = tanhf(pre[3][e]);
      cr[e] = gf[e] * cv[e] + gi[e] * gg[e];
    }
    *reinterpret_cast<f32x4*>(co + (long)p * g + 4 * c4) = cr;  // (raw; normalised below by the thread that wrote it)
    if (act_out) {
      float* ao = act_out + ((long)b * HW + p) * row4 + q * Cq + 4 * c4;
      f32x4 go;
#pragma unroll
      for (int e = 0; e < 4; ++e) go[e] = sigmoid_acc(pre[2][e]);
      *reinterpret_cast<f32x4*>(ao) = gi;
      *reinterpret_cast<f32x4*>(ao + g) = gf;
      *reinterpret_cast<f32x4*>(ao + 2L * g) = go;
      *reinterpret_cast<f32x4*>(ao + 3L * g) = gg;
      *reinterpret_cast<f32x4*>(craw_out + ((long)b * HW + p) * g + q * Cq + 4 * c4) = cr;
    }
    const int grp = (4 * c4) / cg;
    const float s4 = (cr.x + cr.y) + (cr.z + cr.w);
#pragma unroll
    for (int j = 0; j < 4; ++j) csum[j] += grp == j ? s4 : 0.f;
  }
  float m8[8] = {csum[0], csum[1], csum[2], csum[3], 0.f, 0.f, 0.f, 0.f};
  block8(m8);
  const float inv_c = 1.0f / (float)(HW * cg);
  float cmean[4], cr2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) cmean[j] = m8[j] * inv_c;
  for (int i = tid; i < nq; i += NT) {
    const int p = i / Q4, c4 = i - p * Q4;
    const f32x4 cr = *reinterpret_cast<const f32x4*>(co + (long)p * g + 4 * c4);
    const int grp = (4 * c4) / cg;
    float mg = cmean[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) mg = grp == j ? cmean[j] : mg;
    float d2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) d2 += (cr[e] - mg) * (cr[e] - mg);
#pragma unroll
    for (int j = 0; j < 4; ++j) cr2[j] += grp == j ? d2 : 0.f;
  }
  block8(cr2);
  float crstd[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) crstd[j] = 1.0f / sqrtf(cr2[j] * inv_c + eps);
  if (stat_c && tid == 0) {
    const int nB = gridDim.y;
#pragma unroll
    for (int j = 0; j < 4; ++j) stat_c[b * 16 + 4 * q + j] = cmean[j], stat_c[(nB + b) * 16 + 4 * q + j] = crstd[j];
  }
  for (int i = tid; i < nq; i += NT) {
    const int p = i / Q4, c4 = i - p * Q4;
    const long o = (long)p * row4 + 4 * c4;
    const f32x4 cr = *reinterpret_cast<const f32x4*>(co + (long)p * g + 4 * c4);
    const int grp = (4 * c4) / cg;
    float mg = cmean[0], rg = crstd[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) mg = grp == j ? cmean[j] : mg, rg = grp == j ? crstd[j] : rg;
    const int ch = q * Cq + 4 * c4;
    const f32x4 gc = *reinterpret_cast<const f32x4*>(gam_c + ch), bc = *reinterpret_cast<const f32x4*>(bet_c + ch);
    const f32x4 a = *reinterpret_cast<const f32x4*>(ih + o + 2L * g);
    const f32x4 h = *reinterpret_cast<const f32x4*>(hh + o + 2L * g);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gam_ih + 2 * g + ch), ba = *reinterpret_cast<const f32x4*>(bet_ih + 2 * g + ch);
    const f32x4 gh = *reinterpret_cast<const f32x4*>(gam_hh + 2 * g + ch), bh = *reinterpret_cast<const f32x4*>(bet_hh + 2 * g + ch);
    f32x4 cn, hn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cn[e] = (cr[e] - mg) * rg * gc[e] + bc[e];
      const float po = ((a[e] - mean[2]) * rstd[2] * ga[e] + ba[e]) + ((h[e] - mean[6]) * rstd[6] * gh[e] + bh[e]);
      hn[e] = sigmoid_acc(po) * tanhf(cn[e]);
    }
    *reinterpret_cast<f32x4*>(co + (long)p * g + 4 * c4) = cn;
    *reinterpret_cast<f32x4*>(ho + (long)p * g + 4 * c4) = hn;
  }
}

// Backward of the whole NormConvLSTMCell behind its gate convs (the adjoint of norm_lstm_cell_fwd_kernel) in ONE launch:
// h = o tanh(c), c = GroupNorm(16, g)(c_raw), c_raw = f c_prev + i g~, gates = act(GroupNorm(16, 4g)(g_ih) + GroupNorm(16, 4g)(g_hh)).
// In: dh, dc_in (gradients of h and of the normalised cell; either may be NULL), the forward pass's act / c / c_raw / stats.
// Out: dg_ih, dg_hh (gradients of the two convs' outputs, with their max |.| folded into two slots), dc_prev, and the
// three norms' affine gradients (+=: fp32 atomics, one per (workgroup, channel), as rac_groupnorm_bwd's).
// grid (4, B) as the forward kernel: quarter q of the channels owns gate groups q, 4 + q, 8 + q, 12 + q and cell groups
// 4 q .. 4 q + 3.  Pass 1: the cell norm's two group sums; pass 2: the gate pre-activation gradients (parked in dg_ih) and
// the gate norms' 16 group sums; pass 3: both conv-output gradients.
__global__ __launch_bounds__(512) void norm_lstm_cell_bwd_kernel(
    const float* dh, const float* dc_in, const float* act, const float* c_norm, const float* c_raw, const float* c_prev,
    const float* g_ih, const float* g_hh, const float* stat_ih, const float* stat_hh, const float* stat_c,
    const float* gam_ih, const float* gam_hh, const float* gam_c, float* dg_ih, float* dg_hh, float* dc_prev,
    float* dgam_ih, float* dbet_ih, float* dgam_hh, float* dbet_hh, float* dgam_c, float* dbet_c, unsigned* amax_ih,
    unsigned* amax_hh, int HW, int g) {
  __shared__ float red[8][16];
  const int NT = blockDim.x, NW = NT >> 6;
  // dgamma / dbeta: a thread's vectors all carry ONE channel quad (its index strides by 256, a multiple of Q4), so it sums
  // its 14 quads -- cell norm (dgamma, dbeta), gates k = 0..3 (dgamma_ih, dbeta, dgamma_hh) -- in registers; the 256 / Q4
  // threads of a quad meet in LDS ([14][256] vectors) once, at the end.  (LDS atomics per value: 107 us per launch.)
  extern __shared__ f32x4 chan4[];
  const int q = blockIdx.x, b = blockIdx.y, nB = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int Cq = g >> 2, Q4 = Cq >> 2, nq = HW * Q4, cg = g >> 4;
  const long row4 = 4L * g;
  const long base4 = (long)b * HW * row4 + q * Cq, base1 = (long)b * HW * g + q * Cq;
  auto block16 = [&](float (&v)[16]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
      for (int k = 0; k < 16; ++k) red[wv][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float t = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
      if (NW == 8) t += (red[4][k] + red[5][k]) + (red[6][k] + red[7][k]);
      v[k] = t;
    }
  };
  f32x4 acc[14];
#pragma unroll
  for (int v = 0; v < 14; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mu_c[4], rs_c[4], mu_g[8], rs_g[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) mu_c[j] = stat_c[b * 16 + 4 * q + j], rs_c[j] = stat_c[(nB + b) * 16 + 4 * q + j];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mu_g[k] = stat_ih[b * 16 + 4 * k + q], rs_g[k] = stat_ih[(nB + b) * 16 + 4 * k + q];
    mu_g[4 + k] = stat_hh[b * 16 + 4 * k + q], rs_g[4 + k] = stat_hh[(nB + b) * 16 + 4 * k + q];
  }
  __syncthreads();
  // the gradient reaching the normalised cell at one vector: dh o (1 - tanh^2 c) + dc_in
  auto dcn_of = [&](int p, int c4, f32x4& tc) {
    const long o1 = base1 + (long)p * g + 4 * c4;
    const f32x4 cn = *reinterpret_cast<const f32x4*>(c_norm + o1);
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) tc[e] = tanhf(cn[e]);
    if (dh) {
      const f32x4 dv = *reinterpret_cast<const f32x4*>(dh + o1);
      const f32x4 go = *reinterpret_cast<const f32x4*>(act + base4 + (long)p * row4 + 2L * g + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = dv[e] * go[e] * (1.f - tc[e] * tc[e]);
    }
    if (dc_in) {
      const f32x4 dv = *reinterpret_cast<const f32x4*>(dc_in + o1);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] += dv[e];
    }
    return d;
  };
  float s16[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) s16[k] = 0.f;
  for (int i = tid; i < nq; i += NT) {  // pass 1: sum(dcn gamma), sum(dcn gamma c^) per cell group; dgamma_c / dbeta_c
    const int p = i / Q4, c4 = i - p * Q4;
    f32x4 tc;
    const f32x4 d = dcn_of(p, c4, tc);
    const f32x4 cr = *reinterpret_cast<const f32x4*>(c_raw + base1 + (long)p * g + 4 * c4);
    const f32x4 gc = *reinterpret_cast<const f32x4*>(gam_c + q * Cq + 4 * c4);
    const int grp = (4 * c4) / cg;
    float mg = mu_c[0], rg = rs_c[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) mg = grp == j ? mu_c[j] : mg, rg = grp == j ? rs_c[j] : rg;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (cr[e] - mg) * rg, dg = d[e] * gc[e];
      a1 += dg, a2 += dg * xh;
      acc[0][e] += d[e] * xh;
      acc[1][e] += d[e];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s16[2 * j] += grp == j ? a1 : 0.f, s16[2 * j + 1] += grp == j ? a2 : 0.f;
  }
  block16(s16);
  float m1c[4], m2c[4];
  const float inv_c = 1.0f / (float)(HW * cg), inv_n = 1.0f / (float)(HW * Cq);
#pragma unroll
  for (int j = 0; j < 4; ++j) m1c[j] = s16[2 * j] * inv_c, m2c[j] = s16[2 * j + 1] * inv_c;
#pragma unroll
  for (int k = 0; k < 16; ++k) s16[k] = 0.f;
  for (int i = tid; i < nq; i += NT) {  // pass 2: dc_raw, dc_prev, the gate pre-activation gradients and their group sums
    const int p = i / Q4, c4 = i - p * Q4;
    const long o4 = base4 + (long)p * row4 + 4 * c4, o1 = base1 + (long)p * g + 4 * c4;
    f32x4 tc;
    const f32x4 d = dcn_of(p, c4, tc);
    const f32x4 cr = *reinterpret_cast<const f32x4*>(c_raw + o1);
    const f32x4 gc = *reinterpret_cast<const f32x4*>(gam_c + q * Cq + 4 * c4);
    const f32x4 cp = *reinterpret_cast<const f32x4*>(c_prev + o1);
    const int grp = (4 * c4) / cg;
    float mg = mu_c[0], rg = rs_c[0], m1 = m1c[0], m2 = m2c[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
      mg = grp == j ? mu_c[j] : mg, rg = grp == j ? rs_c[j] : rg, m1 = grp == j ? m1c[j] : m1, m2 = grp == j ? m2c[j] : m2;
    const f32x4 gi = *reinterpret_cast<const f32x4*>(act + o4), gf = *reinterpret_cast<const f32x4*>(act + o4 + g);
    const f32x4 go = *reinterpret_cast<const f32x4*>(act + o4 + 2L * g), gg = *reinterpret_cast<const f32x4*>(act + o4 + 3L * g);
    f32x4 dpre[4], dcp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (cr[e] - mg) * rg;
      const float dcr = rg * (d[e] * gc[e] - m1 - xh * m2);
      const float dov = dh ? dh[o1 + e] * tc[e] : 0.f;
      dpre[0][e] = dcr * gg[e] * gi[e] * (1.f - gi[e]);
      dpre[1][e] = dcr * cp[e] * gf[e] * (1.f - gf[e]);
      dpre[2][e] = dov * go[e] * (1.f - go[e]);
      dpre[3][e] = dcr * gi[e] * (1.f - gg[e] * gg[e]);
      dcp[e] = dcr * gf[e];
    }
    *reinterpret_cast<f32x4*>(dc_prev + o1) = dcp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      *reinterpret_cast<f32x4*>(dg_ih + o4 + (long)k * g) = dpre[k];  // (parked: pass 3 turns it into the conv's gradient)
      const f32x4 xa = *reinterpret_cast<const f32x4*>(g_ih + o4 + (long)k * g);
      const f32x4 xb = *reinterpret_cast<const f32x4*>(g_hh + o4 + (long)k * g);
      const int ch = k * g + q * Cq + 4 * c4;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gam_ih + ch), gb = *reinterpret_cast<const f32x4*>(gam_hh + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ha = (xa[e] - mu_g[k]) * rs_g[k], hb = (xb[e] - mu_g[4 + k]) * rs_g[4 + k];
        s16[2 * k] += dpre[k][e] * ga[e];
        s16[2 * k + 1] += dpre[k][e] * ga[e] * ha;
        s16[8 + 2 * k] += dpre[k][e] * gb[e];
        s16[8 + 2 * k + 1] += dpre[k][e] * gb[e] * hb;
        acc[2 + k][e] += dpre[k][e] * ha;
        acc[6 + k][e] += dpre[k][e];
        acc[10 + k][e] += dpre[k][e] * hb;
      }
    }
  }
  block16(s16);
#pragma unroll
  for (int k = 0; k < 16; ++k) s16[k] *= inv_n;
  unsigned mxa = 0, mxb = 0;
  for (int i = tid; i < nq; i += NT) {  // pass 3: the two convs' output gradients
    const int p = i / Q4, c4 = i - p * Q4;
    const long o4 = base4 + (long)p * row4 + 4 * c4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 dp = *reinterpret_cast<const f32x4*>(dg_ih + o4 + (long)k * g);
      const f32x4 xa = *reinterpret_cast<const f32x4*>(g_ih + o4 + (long)k * g);
      const f32x4 xb = *reinterpret_cast<const f32x4*>(g_hh + o4 + (long)k * g);
      const int ch = k * g + q * Cq + 4 * c4;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gam_ih + ch), gb = *reinterpret_cast<const f32x4*>(gam_hh + ch);
      f32x4 oa, ob;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ha = (xa[e] - mu_g[k]) * rs_g[k], hb = (xb[e] - mu_g[4 + k]) * rs_g[4 + k];
        oa[e] = rs_g[k] * (dp[e] * ga[e] - s16[2 * k] - ha * s16[2 * k + 1]);
        ob[e] = rs_g[4 + k] * (dp[e] * gb[e] - s16[8 + 2 * k] - hb * s16[8 + 2 * k + 1]);
        mxa = max(mxa, absbits(oa[e]));
        mxb = max(mxb, absbits(ob[e]));
      }
      *reinterpret_cast<f32x4*>(dg_ih + o4 + (long)k * g) = oa;
      *reinterpret_cast<f32x4*>(dg_hh + o4 + (long)k * g) = ob;
    }
  }
  if (amax_ih) amax_commit_block(mxa, amax_ih);
  if (amax_hh) {
    __syncthreads();
    amax_commit_block(mxb, amax_hh);
  }
  if (dgam_ih) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 14; ++v) chan4[v * NT + tid] = acc[v];
    __syncthreads();
    const int rpq = NT / Q4;  // threads per channel quad: tid = r * Q4 + c4
    for (int i = tid; i < 14 * Q4; i += NT) {
      const int v = i / Q4, c4 = i - v * Q4;
      f32x4 t = chan4[v * NT + c4];
      for (int r = 1; r < rpq; ++r) t += chan4[v * NT + r * Q4 + c4];
      float* dst;
      if (v == 0) dst = dgam_c + q * Cq;
      else if (v == 1) dst = dbet_c + q * Cq;
      else if (v < 6) dst = dgam_ih + (v - 2) * g + q * Cq;
      else if (v < 10) dst = dbet_ih + (v - 6) * g + q * Cq;
      else dst = dgam_hh + (v - 10) * g + q * Cq;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(dst + 4 * c4 + e, t[e]);
      if (v >= 6 && v < 10) {  // (dbeta of both gate norms is the same sum of the pre-activation gradients)
        float* d2 = dbet_hh + (v - 6) * g + q * Cq;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(d2 + 4 * c4 + e, t[e]);
      }
    }
  }
}

__global__ void groupnorm_bwd_kernel(const float* dy, const float* x, const float* gamma, const float* mean_i,
                                     const float* rstd_i, float* dx, float* dgamma, float* dbeta, int HW, int C,
                                     int G) {
  __shared__ float sh[4];
  extern __shared__ float chan[];  // [2][Cg] per-channel partials of dgamma / dbeta
  const int g = blockIdx.x, b = blockIdx.y;
  const int Cg = C / G;
  const int n = HW * Cg;
  const float mean = mean_i[b * G + g], rstd = rstd_i[b * G + g];
  const long base = (long)b * HW * C + g * Cg;
  for (int i = threadIdx.x; i < 2 * Cg; i += 256) chan[i] = 0.f;
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i % Cg;
    const long o = base + (long)(i / Cg) * C + c;
    const float xh = (x[o] - mean) * rstd;
    const float d = dy[o];
    const float dg = d * gamma[g * Cg + c];
    s1 += dg;
    s2 += dg * xh;
    if (dgamma) {
      atomicAdd(&chan[c], d * xh);
      atomicAdd(&chan[Cg + c], d);
    }
  }
  const float m1 = block_sum256(s1, sh) / n;
  const float m2 = block_sum256(s2, sh) / n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i % Cg;
    const long o = base + (long)(i / Cg) * C + c;
    const float xh = (x[o] - mean) * rstd;
    dx[o] = rstd * (dy[o] * gamma[g * Cg + c] - m1 - xh * m2);
  }
  if (dgamma) {
    __syncthreads();
    for (int c = threadIdx.x; c < Cg; c += 256) {
      atomicAdd(dgamma + g * Cg + c, chan[c]);
      atomicAdd(dbeta + g * Cg + c, chan[Cg + c]);
    }
  }
}

__global__ void lstm_out_fwd_kernel(const float* act, const float* c, float* h, long M, int g) {
  const long n = M * g;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m = i / g;
    int ch = (int)(i - m * g);
    h[i] = act[m * 4 * g + 2 * g + ch] * tanhf(c[i]);
  }
}
__global__ void lstm_out_bwd_kernel(const float* dh, const float* act, const float* c, float* d_act, float* dc, long M,
                                    int g) {
  const long n = M * g;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m = i / g;
    int ch = (int)(i - m * g);
    const float tc = tanhf(c[i]), o = act[m * 4 * g + 2 * g + ch], d = dh[i];
    float* da = d_act + m * 4 * g + ch;  // gradient w.r.t. the activated gates: only the o slot is non-zero
    da[0] = 0.f;
    da[g] = 0.f;
    da[2 * g] = d * tc;
    da[3 * g] = 0.f;
    dc[i] = d * o * (1.f - tc * tc);
  }
}
__global__ void lstm_core_bwd_kernel(const float* dc_raw, const float* d_act, const float* act, const float* c_prev,
                                     float* dgates, float* dc_prev, long M, int g) {
  const long n = M * g;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long m = i / g;
    int ch = (int)(i - m * g);
    const float* a = act + m * 4 * g + ch;
    const float gi = a[0], gf = a[g], go = a[2 * g], gg = a[3 * g];
    const float dc = dc_raw ? dc_raw[i] : 0.f;
    float di = dc * gg, df = dc * c_prev[i], dov = 0.f, dg = dc * gi;
    if (d_act) {  // direct gradients on the activated gates (the o slot from h = o * tanh(norm(c)))
      const float* e = d_act + m * 4 * g + ch;
      di += e[0], df += e[g], dov += e[2 * g], dg += e[3 * g];
    }
    float* d = dgates + m * 4 * g + ch;
    d[0] = di * gi * (1.f - gi);
    d[g] = df * gf * (1.f - gf);
    d[2 * g] = dov * go * (1.f - go);
    d[3 * g] = dg * (1.f - gg * gg);
    dc_prev[i] = dc * gf;
  }
}

}  // namespace rac

extern "C" {

int rac_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int32_t B, int32_t HW, int32_t C, int32_t G, float eps, void* stream) {
  RAC_REQUIRE(x && gamma && beta && y && mean && rstd && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0,
              "rac_groupnorm_fwd: bad args (C must be divisible by G)");
  hipLaunchKernelGGL(groupnorm_fwd_kernel, dim3(G, B), dim3(256), 0, ST(stream), x, gamma, beta, y, mean, rstd, HW, C,
                     G, eps);
  return check_launch("rac_groupnorm_fwd");
}

// threads per (image, quarter) workgroup of the fused NormConvLSTMCell kernels: 512 while the launch has fewer workgroups
// than two per CU (a training batch: 64) -- the passes are bound by what one workgroup keeps in flight -- else 256
static int norm_cell_threads(int B, int g) {
  static const int forced = [] { const char* e = getenv("RAC_NORM_CELL_THREADS"); return e ? atoi(e) : 0; }();
  if (forced == 256 || forced == 512) return forced;
  return (4 * B < 512 && g / 16 <= 256) ? 512 : 256;
}

int rac_norm_lstm_cell_fwd(const float* g_ih, const float* g_hh, const float* c_prev, const float* gamma_ih,
                           const float* beta_ih, const float* gamma_hh, const float* beta_hh, const float* gamma_c,
                           const float* beta_c, float* h, float* c, float* act, float* c_raw, float* stat_ih, float* stat_hh,
                           float* stat_c, int32_t B, int32_t HW, int32_t g, float eps, void* stream) {
  RAC_REQUIRE(g_ih && g_hh && c_prev && gamma_ih && beta_ih && gamma_hh && beta_hh && gamma_c && beta_c && h && c && B > 0 &&
                  HW > 0 && g > 0,
              "rac_norm_lstm_cell_fwd: bad args");
  RAC_REQUIRE((act != nullptr) == (c_raw != nullptr) && (act != nullptr) == (stat_ih != nullptr) &&
                  (act != nullptr) == (stat_hh != nullptr) && (act != nullptr) == (stat_c != nullptr) &&
                  (!act || (aligned16(act) && aligned16(c_raw))),
              "rac_norm_lstm_cell_fwd: act / c_raw / stat_ih / stat_hh / stat_c come together (training) or not at all");
  const int Q4 = g / 16;
  RAC_REQUIRE(g % 16 == 0 && Q4 >= 1 && Q4 <= 256 && (Q4 & (Q4 - 1)) == 0 && B <= 65535,
              "rac_norm_lstm_cell_fwd: g must be 16 * 2^k <= 4096 (GroupNorm(16, .) groups of whole 16-byte vectors)");
  RAC_REQUIRE(aligned16(g_ih) && aligned16(g_hh) && aligned16(c_prev) && aligned16(h) && aligned16(c) && aligned16(gamma_ih) &&
                  aligned16(beta_ih) && aligned16(gamma_hh) && aligned16(beta_hh) && aligned16(gamma_c) && aligned16(beta_c),
              "rac_norm_lstm_cell_fwd: 16-byte aligned operands");
  hipLaunchKernelGGL(norm_lstm_cell_fwd_kernel, dim3(4, B), dim3(norm_cell_threads(B, g)), 0, ST(stream), g_ih, g_hh, c_prev, gamma_ih, beta_ih,
                     gamma_hh, beta_hh, gamma_c, beta_c, h, c, act, c_raw, stat_ih, stat_hh, stat_c, HW, g, eps);
  return check_launch("rac_norm_lstm_cell_fwd");
}

int rac_norm_lstm_cell_bwd(const float* dh, const float* dc, const float* act, const float* c, const float* c_raw,
                           const float* c_prev, const float* g_ih, const float* g_hh, const float* stat_ih,
                           const float* stat_hh, const float* stat_c, const float* gamma_ih, const float* gamma_hh,
                           const float* gamma_c, float* dg_ih, float* dg_hh, float* dc_prev, float* dgamma_ih, float* dbeta_ih,
                           float* dgamma_hh, float* dbeta_hh, float* dgamma_c, float* dbeta_c, uint32_t* dg_ih_amax,
                           uint32_t* dg_hh_amax, int32_t B, int32_t HW, int32_t g, void* stream) {
  RAC_REQUIRE(act && c && c_raw && c_prev && g_ih && g_hh && stat_ih && stat_hh && stat_c && gamma_ih && gamma_hh && gamma_c &&
                  dg_ih && dg_hh && dc_prev && B > 0 && HW > 0 && g > 0,
              "rac_norm_lstm_cell_bwd: bad args");
  const int Q4 = g / 16;
  RAC_REQUIRE(g % 16 == 0 && Q4 >= 1 && Q4 <= 256 && (Q4 & (Q4 - 1)) == 0 && B <= 65535,
              "rac_norm_lstm_cell_bwd: g must be 16 * 2^k <= 4096");
  const bool aff = dgamma_ih != nullptr;
  RAC_REQUIRE(aff == (dbeta_ih != nullptr) && aff == (dgamma_hh != nullptr) && aff == (dbeta_hh != nullptr) &&
                  aff == (dgamma_c != nullptr) && aff == (dbeta_c != nullptr),
              "rac_norm_lstm_cell_bwd: the six affine gradients come together or not at all");
  RAC_REQUIRE((!dh || aligned16(dh)) && (!dc || aligned16(dc)) && aligned16(act) && aligned16(c) && aligned16(c_raw) &&
                  aligned16(c_prev) && aligned16(g_ih) && aligned16(g_hh) && aligned16(dg_ih) && aligned16(dg_hh) &&
                  aligned16(dc_prev) && aligned16(gamma_ih) && aligned16(gamma_hh) && aligned16(gamma_c),
              "rac_norm_lstm_cell_bwd: 16-byte aligned operands");
  const int nt = norm_cell_threads(B, g);
  const size_t lds = aff ? (size_t)14 * nt * 16 : 0;  // (57 / 115 KB: the one meeting of the per-thread affine sums)
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(norm_lstm_cell_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 14 * 512 * 16);
    RAC_REQUIRE(e == hipSuccess, "rac_norm_lstm_cell_bwd: %s", hipGetErrorString(e));
    attr = true;
  }
  hipLaunchKernelGGL(norm_lstm_cell_bwd_kernel, dim3(4, B), dim3(nt), lds, ST(stream), dh, dc, act, c, c_raw, c_prev, g_ih,
                     g_hh, stat_ih, stat_hh, stat_c, gamma_ih, gamma_hh, gamma_c, dg_ih, dg_hh, dc_prev, dgamma_ih, dbeta_ih,
                     dgamma_hh, dbeta_hh, dgamma_c, dbeta_c, dg_ih_amax, dg_hh_amax, HW, g);
  return check_launch("rac_norm_lstm_cell_bwd");
}

int rac_groupnorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                      float* dx, float* dgamma, float* dbeta, int32_t B, int32_t HW, int32_t C, int32_t G,
                      void* stream) {
  RAC_REQUIRE(dy && x && gamma && mean && rstd && dx && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0,
              "rac_groupnorm_bwd: bad args");
  RAC_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "rac_groupnorm_bwd: dgamma/dbeta must come together");
  const size_t lds = 2 * (size_t)(C / G) * sizeof(float);
  RAC_REQUIRE(lds <= 48 * 1024, "rac_groupnorm_bwd: group too wide");
  hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3(G, B), dim3(256), lds, ST(stream), dy, x, gamma, mean, rstd, dx,
                     dgamma, dbeta, HW, C, G);
  return check_launch("rac_groupnorm_bwd");
}

int rac_lstm_out_fwd(const float* act, const float* c, float* h, int64_t M, int32_t g, void* stream) {
  RAC_REQUIRE(act && c && h && M > 0 && g > 0, "rac_lstm_out_fwd: bad args");
  hipLaunchKernelGGL(lstm_out_fwd_kernel, dim3(grid_for((long)M * g)), dim3(256), 0, ST(stream), act, c, h, (long)M, g);
  return check_launch("rac_lstm_out_fwd");
}

int rac_lstm_out_bwd(const float* dh, const float* act, const float* c, float* d_act, float* dc, int64_t M, int32_t g,
                     void* stream) {
  RAC_REQUIRE(dh && act && c && d_act && dc && M > 0 && g > 0, "rac_lstm_out_bwd: bad args");
  hipLaunchKernelGGL(lstm_out_bwd_kernel, dim3(grid_for((long)M * g)), dim3(256), 0, ST(stream), dh, act, c, d_act, dc,
                     (long)M, g);
  return check_launch("rac_lstm_out_bwd");
}

int rac_lstm_core_bwd(const float* dc_raw, const float* d_act, const float* act, const float* c_prev, float* dgates,
                      float* dc_prev, int64_t M, int32_t g, void* stream) {
  RAC_REQUIRE(act && c_prev && dgates && dc_prev && M > 0 && g > 0, "rac_lstm_core_bwd: bad args");
  hipLaunchKernelGGL(lstm_core_bwd_kernel, dim3(grid_for((long)M * g)), dim3(256), 0, ST(stream), dc_raw, d_act, act,
                     c_prev, dgates, dc_prev, (long)M, g);
  return check_launch("rac_lstm_core_bwd");
}

}  // extern "C"

RAC_DEVICE_CODE_END
