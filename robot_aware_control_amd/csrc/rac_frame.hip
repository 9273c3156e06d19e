// Frame-level kernels at the model boundary (NCHW planes <-> NHWC maps):
// input packing with robot-region zeroing, mask compositing, the reconstruction
// losses (+ logging metrics) and the fused CEM step tail with fp64 cost sums.
// All of them stream 3-5 planes of H*W floats per frame: HBM-bound.
#include "rac_common.h"

namespace rac {

static inline int grid_for(long work_items) {
  long b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

__global__ void pack_input_kernel(const float* img, const float* zmask, const float* mask, int Cm, int pad,
                                  float* packed, int B, int HW) {
  const int C = 3 + Cm + pad;
  const long n = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / HW;
    int p = (int)(i - b * HW);
    const bool zero = zmask && zmask[i] != 0.f;
    float* o = packed + i * C;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = img[(b * 3 + c) * HW + p];
      o[c] = zero ? v * 0.f : v;
    }
    for (int c = 0; c < Cm; ++c) o[3 + c] = mask[(b * Cm + c) * HW + p];
    for (int c = 0; c < pad; ++c) o[3 + Cm + c] = 0.f;
  }
}

// First encoder layer of the frozen model straight from the NCHW planes (vgg_64.py:8-18 on dynamics.py:578-582's input):
// out[b][y][x][co] = act(scale[co] * sum_{tap, ci} in[b][ci][y + ky - 1][x + kx - 1] * w[co][tap][ci] + shift[co]), with
// in = [img * (zero_mask == 0) | mask planes].  3..8 input channels are 27..72 multiply-adds per output channel: a
// matrix-pipe conv would run on 32-channel chunks that are 3/4 padding (and first write the padded NHWC tensor).  Here a
// workgroup owns a 16 x 16 pixel tile (input halo tile and the [tap][ci][co] weights in LDS); a thread computes 4
// horizontally adjacent pixels x 16 output channels in fp32 FMAs, so every weight vector it reads feeds 4 pixels and
// the 4 threads of a pixel quad write 256 contiguous bytes per pixel.  (One pixel x 64 channels per thread is bound by
// its LDS weight reads: 0.74 ms at 1000 x 3 x 64 x 64 frames; weights through global loads 0.84.)
// Also leaves max |out| for the next conv's operand scale.
template <int CIN>
__global__ __launch_bounds__(256) void first_layer_kernel(const float* img, const float* zmask, const float* mask,
                                                          const float* w, const float* scale, const float* shift, int act,
                                                          float* out, unsigned* amax, int per_image, int H, int W) {
  constexpr int CO = 64, TP = 18, TPW = 20;  // halo tile 18 x 18, rows padded to 20 floats
  __shared__ float in_sh[CIN][TP * TPW];
  __shared__ __attribute__((aligned(16))) float w_sh[9 * CIN][CO];
  const int tid = threadIdx.x;
  const int cg = tid & 3, pq = tid >> 2;        // 16-channel group; pixel quad
  const int ty = pq >> 2, tx = (pq & 3) << 2;   // quad = pixels (ty, tx .. tx + 3) of the tile
  const int tiles_x = W >> 4;
  const int b = blockIdx.y, ty0 = (blockIdx.x / tiles_x) << 4, tx0 = (blockIdx.x % tiles_x) << 4;
  const long HW = (long)H * W;
  for (int i = tid; i < 9 * CIN * CO; i += 256) {  // w[co][tap][ci] -> w_sh[tap * CIN + ci][co]
    const int r = i / CO, co = i - r * CO;
    w_sh[r][co] = w[(long)co * 9 * CIN + r];
  }
  for (int i = tid; i < CIN * TP * TP; i += 256) {
    const int ci = i / (TP * TP), q = i - ci * TP * TP;
    const int qy = q / TP, qx = q - qy * TP;
    const int y = ty0 + qy - 1, x = tx0 + qx - 1;
    float v = 0.f;
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
      const long pix = (long)y * W + x;
      if (ci < 3) {
        v = img[((long)b * 3 + ci) * HW + pix];
        if (zmask && zmask[(long)b * HW + pix] != 0.f) v = v * 0.f;
      } else {
        v = mask[((long)b * (CIN - 3) + (ci - 3)) * HW + pix];
      }
    }
    in_sh[ci][qy * TPW + qx] = v;
  }
  __syncthreads();
  f32x4 acc[4][4];  // [pixel][4 channels]
#pragma unroll
  for (int px = 0; px < 4; ++px)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[px][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1  // (fully unrolled, the compiler hoists every weight vector into registers and spills)
  for (int tap = 0; tap < 9; ++tap) {
    const int q = (ty + tap / 3) * TPW + tx + tap % 3;
#pragma unroll 1
    for (int ci = 0; ci < CIN; ++ci) {
      const f32x4* wr = reinterpret_cast<const f32x4*>(&w_sh[tap * CIN + ci][cg * 16]);
      const f32x4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const float v = in_sh[ci][q + px];
        acc[px][0] += v * w0, acc[px][1] += v * w1, acc[px][2] += v * w2, acc[px][3] += v * w3;
      }
    }
  }
  const float slope = act == RAC_ACT_LEAKY02 ? 0.2f : 1.f;
  f32x4 sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc[j] = scale ? reinterpret_cast<const f32x4*>(scale + cg * 16)[j] : f32x4{1.f, 1.f, 1.f, 1.f};
    sh[j] = scale ? reinterpret_cast<const f32x4*>(shift + cg * 16)[j] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  unsigned mx = 0;
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    f32x4* o = reinterpret_cast<f32x4*>(out + (((long)b * H + ty0 + ty) * W + tx0 + tx + px) * CO + cg * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = acc[px][j] * sc[j] + sh[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = v[e] > 0.f ? v[e] : slope * v[e];
        mx = max(mx, absbits(v[e]));
      }
      o[j] = v;
    }
  }
  if (amax) amax_commit_block(mx, per_image ? amax + b : amax);
}

// Output head: ConvTranspose2d(64 -> 4, 3, 1, 1) + bias + Sigmoid (vgg_64.py:218-220) as exact-fp32 FMAs.
//   y[p][c] = sigmoid(bias[c] + sum_{ky, kx, ci} x[p - (ky - 1, kx - 1)][ci] * w[ky][kx][ci][c])
// On the matrix pipe this layer is a GEMM with N = 4: the 32-column tile of conv16_rows_kernel spends 7/8 of its MFMAs
// on padding and a whole tile's staging (halo, fp32 -> fp16 parts, barriers) on 18 K steps: 0.86-1.06 ms per 4 M pixels
// (0.02 of the pipe), against an HBM floor of 0.17 ms (1.05 GB in) and 0.27 ms of plain VALU time for its 9.4 G FMAs.
// Here: a workgroup owns an 8 x 32 pixel tile, one thread per pixel, 4 accumulators; the haloed input tile goes through
// LDS in two 32-channel halves ([pixel][36 floats]: a 144-byte pixel stride spreads the 8 lanes of a ds_read_b128 pass
// over all banks), every LDS vector feeds 16 FMAs, and the weights never touch LDS or VGPRs -- they are wave-uniform, so
// the compiler fetches them with scalar loads and the FMAs take them as SGPR operands.  LDS traffic 9 x the tile
// (2.3 KB per pixel, 9.4 GB per launch at 4 M pixels: ~0.16 ms), VALU 2304 FMAs per pixel: VALU-bound at ~0.3 ms.
// Exact fp32 (no operand split), a fixed order of sums per pixel: batch-invariant like the rest of the frozen path.
__global__ __launch_bounds__(256) void head_direct_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                          const float* __restrict__ bias, float* __restrict__ y, int H,
                                                          int W) {
  constexpr int TR = 8, TC = 32, PR = TR + 2, PC = TC + 2, CS = 36;
  __shared__ __attribute__((aligned(16))) float xs[PR * PC * CS];
  const int tid = threadIdx.x;
  const int py = tid >> 5, px = tid & 31;
  const int tiles_x = W / TC;
  const int b = blockIdx.y, ty0 = (blockIdx.x / tiles_x) * TR, tx0 = (blockIdx.x % tiles_x) * TC;
  const float* xb = x + (long)b * H * W * 64;
  float acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = bias[c];
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
    for (int i = tid; i < PR * PC * 8; i += 256) {
      const int pix = i >> 3, q = i & 7;
      const int row = pix / PC, col = pix - row * PC;
      const int gy = ty0 + row - 1, gx = tx0 + col - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(xb + ((long)gy * W + gx) * 64 + half * 32 + q * 4);
      *reinterpret_cast<f32x4*>(xs + pix * CS + q * 4) = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const float* xp = xs + ((py + 2 - ky) * PC + (px + 2 - kx)) * CS;
      const float* wp = wt + (tap * 64 + half * 32) * 4;  // wave-uniform: scalar loads
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + c4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] += xv[j] * wp[(c4 * 4 + j) * 4 + c];
      }
    }
  }
  f32x4 out;
#pragma unroll
  for (int c = 0; c < 4; ++c) out[c] = sigmoid_acc(acc[c]);
  *reinterpret_cast<f32x4*>(y + (((long)b * H + ty0 + py) * W + tx0 + px) * 4) = out;
}

// Weight gradient of a 3x3 conv between a wide (64-channel) and a thin (<= 8-channel) NHWC tensor:
//   out[w][tap][t] += sum_p wide[p][w] * thin[p + tap - (1, 1)][t]
// = the first encoder layer's dW (wide = dy, thin = the packed frame: vgg_64.py:8-18 backward) and the output head's
// (wide = x, thin = d(sigmoid): vgg_64.py:218-220 backward).  2 300 .. 4 600 sums over 327 680 pixels: ~1.5 GFLOP that
// the GEMM-shaped weight-gradient kernels spend 0.2-0.35 ms on (64 of their 128 rows and all but 4..8 of 32 columns
// are padding).  Here a workgroup owns a 16 x 16 pixel tile: the thin halo tile sits in LDS, thread (w = tid & 63,
// q = tid >> 6) walks the tile's pixels with wide[p][w] in a register and accumulates the (tap, t) pairs of its
// quarter, over all the tiles it is given; the workgroups' partial sums go to a workspace that rac_slab_accumulate adds
// in a fixed order (deterministic; one float atomic per output and tile instead measured 0.5 ms: 1 280 atomics per
// address across the XCDs).
template <int CT>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const float* wide, const float* thin, int ldt, float* part, int H,
                                                         int W, int n_tiles) {
  constexpr int TP = 18, NO = 9 * CT, NQ = (NO + 3) / 4;  // outputs per w; per thread
  __shared__ float th[TP * TP][CT];
  const int tid = threadIdx.x, w = tid & 63, q = tid >> 6;
  const int tiles_x = W >> 4, tiles_img = (H >> 4) * tiles_x;
  float acc[NQ];
  int toff[NQ];  // LDS offset of output j's (tap, t) relative to the pixel
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    acc[j] = 0.f;
    const int o = min(q * NQ + j, NO - 1), tap = o / CT;
    toff[j] = ((tap / 3) * TP + tap % 3) * CT + (o - tap * CT);
  }
  const float* thf = &th[0][0];
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {  // a workgroup keeps its sums over all its tiles
    const int b = tile / tiles_img, ti = tile - b * tiles_img;
    const int ty0 = (ti / tiles_x) << 4, tx0 = (ti % tiles_x) << 4;
    __syncthreads();
    for (int i = tid; i < TP * TP * CT; i += 256) {
      const int pix = i / CT, t = i - pix * CT;
      const int y = ty0 + pix / TP - 1, x = tx0 + pix % TP - 1;
      th[pix][t] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                       ? thin[(((long)b * H + y) * W + x) * ldt + t] : 0.f;
    }
    __syncthreads();
    const float* wp = wide + (((long)b * H + ty0) * W + tx0) * 64 + w;
    for (int py = 0; py < 16; ++py) {
      float v[16];  // one image row of the tile: 16 independent loads in flight
#pragma unroll
      for (int px = 0; px < 16; ++px) v[px] = wp[((long)py * W + px) * 64];
#pragma unroll
      for (int px = 0; px < 16; ++px) {
        const float* tp = thf + (py * TP + px) * CT;
#pragma unroll
        for (int j = 0; j < NQ; ++j) acc[j] += v[px] * tp[toff[j]];
      }
    }
  }
  float* dst = part + (long)blockIdx.x * 64 * NO + (long)w * NO;  // this workgroup's partial sums (plain stores)
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int o = q * NQ + j;
    if (o < NO) dst[o] = acc[j];
  }
}

__global__ void unpack_grad_kernel(const float* dpacked, int C, const float* zmask, float* dimg, int B, int HW) {
  const long n = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / HW;
    int p = (int)(i - b * HW);
    const bool zero = zmask && zmask[i] != 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) dimg[(b * 3 + c) * HW + p] = zero ? 0.f : dpacked[i * C + c];
  }
}

__global__ void zero_region_kernel(const float* img, const float* mask, float* out, int B, int HW) {
  const long n = (long)B * 3 * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long bc = i / HW;
    int p = (int)(i - bc * HW);
    long b = bc / 3;
    float v = img[i];
    out[i] = (mask[b * HW + p] != 0.f) ? v * 0.f : v;
  }
}

__global__ void composite_fwd_kernel(const f32x4* x4, const float* prev, float* out, int B, int HW) {
  const long n = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / HW;
    int p = (int)(i - b * HW);
    f32x4 v = x4[i];
    const float m = v.w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      long o = (b * 3 + c) * HW + p;
      out[o] = (1.f - m) * prev[o] + m * v[c];
    }
  }
}

__global__ void composite_bwd_kernel(const float* dout, const f32x4* x4, const float* prev, f32x4* dx4, float* dprev,
                                     int B, int HW) {
  const long n = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long b = i / HW;
    int p = (int)(i - b * HW);
    f32x4 v = x4[i], d;
    const float m = v.w;
    float dm = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      long o = (b * 3 + c) * HW + p;
      float g = dout[o];
      d[c] = g * m;
      dm += g * (v[c] - prev[o]);
      if (dprev) dprev[o] = g * (1.f - m);
    }
    d.w = dm;
    dx4[i] = d;
  }
}

// ---- reconstruction losses --------------------------------------------------
// per_sample[b][0..7]: 0 main sum, 1 #world values (3 per unmasked pixel), 2 sum robot d^2, 3 #robot values,
//                      4 sum world d^2
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[k];
  return t;
}

__global__ void recon_loss_sample_kernel(int kind, const float* pred, const float* target, const float* mask,
                                         float rw, float* per_sample, int HW) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  const long base = (long)b * 3 * HW;
  float s_main = 0.f, s_world_n = 0.f, s_rob2 = 0.f, s_rob_n = 0.f, s_world2 = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const bool rob = mask && mask[(long)b * HW + p] != 0.f;
    if (mask) {
      if (rob) s_rob_n += 3.f; else s_world_n += 3.f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float d = target[base + (long)c * HW + p] - pred[base + (long)c * HW + p];
      if (mask) {
        if (rob) s_rob2 += d * d; else s_world2 += d * d;
      }
      float dm = d;
      if ((kind == RAC_LOSS_DONTCARE_L1 || kind == RAC_LOSS_DONTCARE_MSE) && rob) dm = d * rw;
      s_main += (kind == RAC_LOSS_L1 || kind == RAC_LOSS_DONTCARE_L1) ? fabsf(dm) : dm * dm;
    }
  }
  float t0 = block_sum(s_main, sh), t1 = block_sum(s_world_n, sh), t2 = block_sum(s_rob2, sh),
        t3 = block_sum(s_rob_n, sh), t4 = block_sum(s_world2, sh);
  if (threadIdx.x == 0) {
    float* o = per_sample + b * 8;
    o[0] = t0, o[1] = t1, o[2] = t2, o[3] = t3, o[4] = t4;
  }
}

__global__ void recon_loss_final_kernel(int kind, const float* per_sample, const float* bw, bool has_mask, float* out,
                                        int B, int HW) {
  __shared__ float sh[16];
  float a = 0.f, r = 0.f, w = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float* s = per_sample + b * 8;
    float wt = bw ? bw[b] : 1.f;
    if (kind == RAC_LOSS_DONTCARE_L1 || kind == RAC_LOSS_DONTCARE_MSE)
      a += wt * s[0] / (s[1] + 1.f);
    else if (kind == RAC_LOSS_L1)
      a += wt * (s[0] / (3.f * HW));
    else
      a += s[0] / (3.f * HW);
    if (has_mask) {
      r += s[2] / (s[3] + 1.f);
      w += s[4] / (s[1] + 1.f);
    }
  }
  float ta = block_sum(a, sh), tr = block_sum(r, sh), tw = block_sum(w, sh);
  if (threadIdx.x == 0) {
    out[0] = ta / B;
    out[1] = tr / B;
    out[2] = tw / B;
  }
}

__global__ void recon_loss_bwd_kernel(int kind, const float* pred, const float* target, const float* mask, float rw,
                                      const float* bw, const float* per_sample, const float* gout, float* dpred,
                                      int B, int HW) {
  const long n = (long)B * 3 * HW;
  const float g = gout[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long bc = i / HW;
    int p = (int)(i - bc * HW);
    int b = (int)(bc / 3);
    float d = target[i] - pred[i];
    float wt = bw ? bw[b] : 1.f;
    float coef, dm = d, inner = 1.f;
    if (kind == RAC_LOSS_DONTCARE_L1 || kind == RAC_LOSS_DONTCARE_MSE) {
      coef = wt / ((per_sample[b * 8 + 1] + 1.f) * B);
      if (mask[(long)b * HW + p] != 0.f) {
        dm = d * rw;
        inner = rw;
      }
    } else {
      coef = ((kind == RAC_LOSS_L1) ? wt : 1.f) / (3.f * HW * B);
    }
    float dd;  // d loss_elem / d dm
    if (kind == RAC_LOSS_L1 || kind == RAC_LOSS_DONTCARE_L1)
      dd = (dm > 0.f) ? 1.f : ((dm < 0.f) ? -1.f : 0.f);
    else
      dd = 2.f * dm;
    dpred[i] = -g * coef * dd * inner;
  }
}

// ---- KL ---------------------------------------------------------------------
__global__ void kl_fwd_kernel(const float* mu1, const float* lv1, const float* mu2, const float* lv2, long n,
                              double* partial) {
  __shared__ float sh[16];
  float a = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s1 = expf(0.5f * lv1[i]), s2 = expf(0.5f * lv2[i]);
    float dmu = mu1[i] - mu2[i];
    a += logf(s2 / s1) + (expf(lv1[i]) + dmu * dmu) / (2.f * expf(lv2[i])) - 0.5f;
  }
  float t = block_sum(a, sh);
  if (threadIdx.x == 0) atomicAdd(partial, (double)t);
}
__global__ void kl_final_kernel(const double* partial, int bs, float* out) { out[0] = (float)(partial[0] / bs); }

__global__ void kl_bwd_kernel(const float* mu1, const float* lv1, const float* mu2, const float* lv2,
                              const float* gout, long n, int bs, float* dmu1, float* dlv1, float* dmu2, float* dlv2) {
  const float g = gout[0] / bs;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float e1 = expf(lv1[i]), e2 = expf(lv2[i]);
    float dmu = mu1[i] - mu2[i];
    float q = (e1 + dmu * dmu) / (2.f * e2);
    dmu1[i] = g * dmu / e2;
    dmu2[i] = -g * dmu / e2;
    dlv1[i] = g * (-0.5f + e1 / (2.f * e2));
    dlv2[i] = g * (0.5f - q);
  }
}

// ---- CEM step tail ------------------------------------------------------------
__global__ void cem_step_tail_kernel(const f32x4* x4, const float* curr, const float* next_mask, const float* goal,
                                     const float* cost_mask, const unsigned char* goal_mask, int kind, float weight,
                                     int add_cost, float* next_out, double* sum_cost, int HW) {
  __shared__ double shd[16];
  const int b = blockIdx.x;
  const long base = (long)b * 3 * HW;
  double acc = 0.0;
  float nworld = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    f32x4 v = x4[(long)b * HW + p];
    const float m = v.w;
    const bool zero = next_mask && next_mask[(long)b * HW + p] != 0.f;
    bool drop = false;
    if (kind == 1) {
      drop = (cost_mask && cost_mask[(long)b * HW + p] != 0.f) || (goal_mask && goal_mask[p] != 0);
      if (!drop) nworld += 1.f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      long o = base + (long)c * HW + p;
      float nx = (1.f - m) * curr[o] + m * v[c];
      if (zero) nx = nx * 0.f;
      next_out[o] = nx;
      float d = 255.f * (nx - goal[(long)c * HW + p]);
      float sq = d * d;
      if (!drop) acc += (double)sq;
    }
  }
  if (!add_cost) return;
  // block reduce (fp64)
  acc = wave_sum_d(acc);
  nworld = wave_sum(nworld);
  __shared__ float shn[16];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) {
    shd[w] = acc;
    shn[w] = nworld;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    float nw = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
      t += shd[k];
      nw += shn[k];
    }
    float dist = sqrtf((float)t);
    if (kind == 1) dist = dist / nw;
    sum_cost[b] += (double)(weight * (-dist));
  }
}

}  // namespace rac

using namespace rac;
#define ST(s) reinterpret_cast<hipStream_t>(s)


// ---- robot-aware CEM inputs on the device -----------------------------------------------------
// For every candidate n and step t: the robot state the reference's analytical models predict
// (src/dataset/wx250s/wx250s_model.py:57-80,121-163, locobot_model.py:50-116: the end effector moves by the planar
// action, height = push_height, rotation / gripper 0, re-normalised) and the robot mask of that end-effector position,
// looked up in an ATLAS of masks rendered once on a regular (x, y) grid of end-effector positions.
// One workgroup per (t, n): thread 0 re-accumulates the <= T actions, all threads copy the H*W mask bytes as floats.
struct RobotAtlasP {
  const float* actions;   // [T][N][A] (time first), world-frame displacements
  const float* start;     // [5] normalised start state (states[0] of every candidate), or [N][5] with per_sample
  const float* low;       // [5] / [N][5]
  const float* high;      // [5] / [N][5]
  const unsigned char* atlas;  // [ny][nx][HW]
  float* states;          // [T+1][N][5]
  float* masks;           // [T+1][N][HW]
  int T, N, A, HW, nx, ny;
  float x0, y0, inv_dx, inv_dy;  // grid: node (i, j) = (x0 + i / inv_dx, y0 + j / inv_dy) in the states' world frame
  float diff_x, diff_y, push_height;
  int per_sample;         // start / low / high hold one row per candidate (trainer windows: every sample has its own)
};

__global__ void robot_atlas_kernel(RobotAtlasP p) {
  __shared__ int node;
  const int n = blockIdx.x, t = blockIdx.y;
  if (threadIdx.x == 0) {
    const float* start = p.start + (p.per_sample ? (long)n * 5 : 0L);
    const float* low = p.low + (p.per_sample ? (long)n * 5 : 0L);
    const float* high = p.high + (p.per_sample ? (long)n * 5 : 0L);
    // denormalise the start, shift into the robot's own frame (float32, as the numpy / torch code does)
    float den[5];
    for (int k = 0; k < 5; ++k) den[k] = start[k] * (high[k] - low[k]) + low[k];
    float raw[5];
    if (t == 0) {
      raw[0] = (den[0] - p.diff_x) + p.diff_x, raw[1] = (den[1] - p.diff_y) + p.diff_y;
      raw[2] = den[2], raw[3] = den[3], raw[4] = den[4];
    } else {
      const float* a = p.actions + (long)n * p.A;
      const long st = (long)p.N * p.A;
      // the first step adds in float32 (both operands are), the later ones in float64 (numpy's promotion)
      double x = (double)((den[0] - p.diff_x) + a[0]), y = (double)((den[1] - p.diff_y) + a[1]);
      for (int s = 1; s < t; ++s) x += (double)a[s * st], y += (double)a[s * st + 1];
      raw[0] = (float)x + p.diff_x, raw[1] = (float)y + p.diff_y;
      raw[2] = p.push_height, raw[3] = 0.f, raw[4] = 0.f;
    }
    float* o = p.states + ((long)t * p.N + n) * 5;
    for (int k = 0; k < 5; ++k) o[k] = (raw[k] - low[k]) / (high[k] - low[k]);
    int ix = (int)rintf((raw[0] - p.x0) * p.inv_dx), iy = (int)rintf((raw[1] - p.y0) * p.inv_dy);
    ix = ix < 0 ? 0 : (ix >= p.nx ? p.nx - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= p.ny ? p.ny - 1 : iy);
    node = iy * p.nx + ix;
  }
  __syncthreads();
  const unsigned char* src = p.atlas + (long)node * p.HW;
  float* dst = p.masks + ((long)t * p.N + n) * p.HW;
  for (int i = threadIdx.x; i < p.HW; i += blockDim.x) dst[i] = src[i] ? 1.f : 0.f;
}

// Data gradient of the 64 -> 4 output head (vgg_64.py:218-220) w.r.t. its 64-channel input:
//   dx[b][y][x][ci] = sum_{ky, kx, co} d[b][y + ky - 1][x + kx - 1][co] * w[ci][ky][kx][co]        (zero outside the image)
// 4 channels in, 64 out, 36 products per output element: no shape for a GEMM tile (on the exact-fp32 implicit GEMM it
// ran at 0.07 of that pipe, 7x its HBM floor).  A thread owns one channel QUAD for the life of the workgroup -- its
// 4 x 9 weight float4s stay in registers (144 VGPRs) -- and walks pixels: per pixel nine float4 loads of d (the 16 lanes
// of a pixel read the same addresses: one request), 144 FMAs, one float4 store; the 64 lanes of a wave store four
// pixels' 64 channels = 1 KB contiguous.  Bound by the 256 B it writes per pixel.
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float4* __restrict__ d, const float4* __restrict__ w,
                                                         float4* __restrict__ dx, int H, int W, long n_pix, int per_block) {
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;  // channel quad, pixel lane (16 consecutive pixels of a row)
  float4 wv[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[c][t] = w[(4 * q + c) * 9 + t];
  const long first = (long)blockIdx.x * per_block;
  const long last = min(first + (long)per_block, n_pix);
  // the nine d vectors of a 16-pixel group (W % 16 == 0: one image row; its coordinates are workgroup-uniform); the next
  // group's are requested before this group's 144 FMAs: two waves per SIMD do not hide an L2 round trip by themselves
  // (buffer loads: a tap outside the image reads through an out-of-range offset, which returns zeros -- no branches)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(d), (short)0, (int)(n_pix * 16), 0x00020000);
  auto fetch = [&](long base, float4 (&dv)[9]) {
    const long row = base / W;
    const int x = (int)(base - row * W) + pl, y = (int)(row % H);
    const unsigned centre = (unsigned)(base + pl) * 16u;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const bool yok = (unsigned)(y + ky - 1) < (unsigned)H;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const bool ok = yok && (unsigned)(x + kx - 1) < (unsigned)W;
        const unsigned off = centre + (unsigned)(((ky - 1) * W + (kx - 1)) * 16);
        dv[ky * 3 + kx] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? off : 0xFFFFFFF0u), 0, 0));
      }
    }
  };
  float4 dv[9], dn[9];
  if (first < last) fetch(first, dv);
  for (long base = first; base < last; base += 16) {
    if (base + 16 < last) fetch(base + 16, dn);
    float acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        a = fmaf(dv[t].x, wv[c][t].x, a);
        a = fmaf(dv[t].y, wv[c][t].y, a);
        a = fmaf(dv[t].z, wv[c][t].z, a);
        a = fmaf(dv[t].w, wv[c][t].w, a);
      }
      acc[c] = a;
    }
    dx[(base + pl) * 16 + q] = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
    for (int t = 0; t < 9; ++t) dv[t] = dn[t];
  }
}

extern "C" {

int rac_pack_input(const float* img, const float* zmask, const float* mask, int32_t Cm, int32_t pad, float* packed,
                   int32_t B, int32_t HW, void* stream) {
  RAC_REQUIRE(img && packed && B > 0 && HW > 0 && Cm >= 0 && pad >= 0 && (Cm == 0 || mask), "rac_pack_input: bad args");
  hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, ST(stream), img, zmask, mask, Cm,
                     pad, packed, B, HW);
  return check_launch("rac_pack_input");
}

int rac_first_layer_fwd(const float* img, const float* zmask, const float* mask, int32_t Cm, const float* w,
                        const float* scale, const float* shift, int32_t act, float* out, uint32_t* out_amax,
                        int32_t amax_per_image, int32_t B, int32_t H, int32_t W, int32_t Cout, void* stream) {
  RAC_REQUIRE(img && w && out && B > 0 && H > 0 && W > 0 && Cm >= 0 && Cm <= 5 && (Cm == 0 || mask),
              "rac_first_layer_fwd: bad args (3 image planes + at most 5 mask / heatmap planes)");
  RAC_REQUIRE(Cout == 64 && H % 16 == 0 && W % 16 == 0 && (scale == nullptr) == (shift == nullptr) && aligned16(out) &&
                  (!scale || (aligned16(scale) && aligned16(shift))) && (act == RAC_ACT_NONE || act == RAC_ACT_LEAKY02),
              "rac_first_layer_fwd: Cout 64, H and W multiples of 16, act none / leaky");
  dim3 grid((H / 16) * (W / 16), B);
  typedef void (*fn_t)(const float*, const float*, const float*, const float*, const float*, const float*, int, float*,
                       unsigned*, int, int, int);
  static const fn_t fns[6] = {first_layer_kernel<3>, first_layer_kernel<4>, first_layer_kernel<5>,
                              first_layer_kernel<6>, first_layer_kernel<7>, first_layer_kernel<8>};
  hipLaunchKernelGGL(fns[Cm], grid, dim3(256), 0, ST(stream), img, zmask, mask, w, scale, shift, act, out, out_amax,
                     amax_per_image ? 1 : 0, H, W);
  return check_launch("rac_first_layer_fwd");
}

int rac_head_fwd(const float* x, const float* w_taps, const float* bias, float* y, int32_t B, int32_t H, int32_t W,
                 void* stream) {
  RAC_REQUIRE(x && w_taps && bias && y && B > 0 && H > 0 && W > 0, "rac_head_fwd: bad args");
  RAC_REQUIRE(H % 8 == 0 && W % 32 == 0 && aligned16(x) && aligned16(y) && aligned16(w_taps),
              "rac_head_fwd: H % 8 == 0, W % 32 == 0, 16-byte aligned buffers");
  hipLaunchKernelGGL(head_direct_kernel, dim3((H / 8) * (W / 32), B), dim3(256), 0, ST(stream), x, w_taps, bias, y, H, W);
  return check_launch("rac_head_fwd");
}

int rac_head_dgrad(const float* d, const float* w, float* dx, int32_t B, int32_t H, int32_t W, void* stream) {
  RAC_REQUIRE(d && w && dx && B > 0 && H > 0 && W > 0, "rac_head_dgrad: bad args");
  RAC_REQUIRE(W % 16 == 0 && aligned16(d) && aligned16(w) && aligned16(dx), "rac_head_dgrad: W % 16 == 0, 16-byte aligned buffers");
  const long n_pix = (long)B * H * W;
  // two workgroups per CU (the 144 weight registers allow two waves per SIMD), one round; whole 16-pixel groups each
  long per_block = ((n_pix + 511) / 512 + 15) / 16 * 16;
  if (per_block < 64) per_block = 64;
  const long blocks = (n_pix + per_block - 1) / per_block;
  RAC_REQUIRE(blocks < (1L << 31) && per_block < (1L << 31) && n_pix < (1L << 28) - 16, "rac_head_dgrad: too many pixels");
  hipLaunchKernelGGL(head_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), reinterpret_cast<const float4*>(d),
                     reinterpret_cast<const float4*>(w), reinterpret_cast<float4*>(dx), H, W, n_pix, (int)per_block);
  return check_launch("rac_head_dgrad");
}

int rac_thin_wgrad(const float* wide, const float* thin, int32_t thin_stride, int32_t Ct, float* parts, int32_t n_parts,
                   int32_t B, int32_t H, int32_t W, int32_t Cw, void* stream) {
  RAC_REQUIRE(wide && thin && parts && B > 0 && H > 0 && W > 0 && Ct >= 1 && Ct <= 8 && thin_stride >= Ct && n_parts >= 1,
              "rac_thin_wgrad: bad args (1 <= Ct <= 8)");
  RAC_REQUIRE(Cw == 64 && H % 16 == 0 && W % 16 == 0, "rac_thin_wgrad: Cw 64, H and W multiples of 16");
  const int n_tiles = B * (H / 16) * (W / 16);
  RAC_REQUIRE(n_parts <= n_tiles, "rac_thin_wgrad: more parts than 16 x 16 tiles");
  typedef void (*fn_t)(const float*, const float*, int, float*, int, int, int);
  static const fn_t fns[8] = {thin_wgrad_kernel<1>, thin_wgrad_kernel<2>, thin_wgrad_kernel<3>, thin_wgrad_kernel<4>,
                              thin_wgrad_kernel<5>, thin_wgrad_kernel<6>, thin_wgrad_kernel<7>, thin_wgrad_kernel<8>};
  hipLaunchKernelGGL(fns[Ct - 1], dim3(n_parts), dim3(256), 0, ST(stream), wide, thin, thin_stride, parts, H, W, n_tiles);
  return check_launch("rac_thin_wgrad");
}

int rac_unpack_grad(const float* dpacked, int32_t C, const float* zmask, float* dimg, int32_t B, int32_t HW,
                    void* stream) {
  RAC_REQUIRE(dpacked && dimg && C >= 3 && B > 0 && HW > 0, "rac_unpack_grad: bad args");
  hipLaunchKernelGGL(unpack_grad_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, ST(stream), dpacked, C, zmask,
                     dimg, B, HW);
  return check_launch("rac_unpack_grad");
}

int rac_zero_region(const float* img, const float* mask, float* out, int32_t B, int32_t HW, void* stream) {
  RAC_REQUIRE(img && mask && out && B > 0 && HW > 0, "rac_zero_region: bad args");
  hipLaunchKernelGGL(zero_region_kernel, dim3(grid_for((long)B * 3 * HW)), dim3(256), 0, ST(stream), img, mask, out, B,
                     HW);
  return check_launch("rac_zero_region");
}

int rac_composite_fwd(const float* x4, const float* prev, float* out, int32_t B, int32_t HW, void* stream) {
  RAC_REQUIRE(x4 && prev && out && B > 0 && HW > 0 && aligned16(x4), "rac_composite_fwd: bad args");
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, ST(stream), (const f32x4*)x4,
                     prev, out, B, HW);
  return check_launch("rac_composite_fwd");
}

int rac_composite_bwd(const float* dout, const float* x4, const float* prev, float* dx4, float* dprev, int32_t B,
                      int32_t HW, void* stream) {
  RAC_REQUIRE(dout && x4 && prev && dx4 && B > 0 && HW > 0 && aligned16(x4) && aligned16(dx4),
              "rac_composite_bwd: bad args");
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, ST(stream), dout,
                     (const f32x4*)x4, prev, (f32x4*)dx4, dprev, B, HW);
  return check_launch("rac_composite_bwd");
}

int rac_recon_loss_fwd(int32_t kind, const float* pred, const float* target, const float* mask, float robot_weight,
                       const float* batch_weight, float* per_sample, float* out, int32_t B, int32_t HW, void* stream) {
  RAC_REQUIRE(kind >= 0 && kind <= 3 && pred && target && per_sample && out && B > 0 && HW > 0,
              "rac_recon_loss_fwd: bad args");
  RAC_REQUIRE(!((kind == RAC_LOSS_DONTCARE_L1 || kind == RAC_LOSS_DONTCARE_MSE) && !mask),
              "rac_recon_loss_fwd: dontcare loss needs a mask");
  hipLaunchKernelGGL(recon_loss_sample_kernel, dim3(B), dim3(256), 0, ST(stream), kind, pred, target, mask,
                     robot_weight, per_sample, HW);
  hipLaunchKernelGGL(recon_loss_final_kernel, dim3(1), dim3(256), 0, ST(stream), kind, per_sample, batch_weight,
                     mask != nullptr, out, B, HW);
  return check_launch("rac_recon_loss_fwd");
}

int rac_recon_loss_bwd(int32_t kind, const float* pred, const float* target, const float* mask, float robot_weight,
                       const float* batch_weight, const float* per_sample, const float* gout, float* dpred, int32_t B,
                       int32_t HW, void* stream) {
  RAC_REQUIRE(kind >= 0 && kind <= 3 && pred && target && per_sample && gout && dpred && B > 0 && HW > 0,
              "rac_recon_loss_bwd: bad args");
  RAC_REQUIRE(!((kind == RAC_LOSS_DONTCARE_L1 || kind == RAC_LOSS_DONTCARE_MSE) && !mask),
              "rac_recon_loss_bwd: dontcare loss needs a mask");
  hipLaunchKernelGGL(recon_loss_bwd_kernel, dim3(grid_for((long)B * 3 * HW)), dim3(256), 0, ST(stream), kind, pred,
                     target, mask, robot_weight, batch_weight, per_sample, gout, dpred, B, HW);
  return check_launch("rac_recon_loss_bwd");
}

int rac_kl_fwd(const float* mu1, const float* lv1, const float* mu2, const float* lv2, int64_t n, int32_t bs,
               double* partial, float* out, void* stream) {
  RAC_REQUIRE(mu1 && lv1 && mu2 && lv2 && partial && out && n > 0 && bs > 0, "rac_kl_fwd: bad args");
  hipError_t e = hipMemsetAsync(partial, 0, sizeof(double), ST(stream));
  if (e != hipSuccess) {
    set_error("rac_kl_fwd: memset: %s", hipGetErrorString(e));
    return RAC_ELAUNCH;
  }
  long nb = (n + 255) / 256;
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3((int)nb), dim3(256), 0, ST(stream), mu1, lv1, mu2, lv2, (long)n, partial);
  hipLaunchKernelGGL(kl_final_kernel, dim3(1), dim3(1), 0, ST(stream), partial, bs, out);
  return check_launch("rac_kl_fwd");
}

int rac_kl_bwd(const float* mu1, const float* lv1, const float* mu2, const float* lv2, const float* gout, int64_t n,
               int32_t bs, float* dmu1, float* dlv1, float* dmu2, float* dlv2, void* stream) {
  RAC_REQUIRE(mu1 && lv1 && mu2 && lv2 && gout && dmu1 && dlv1 && dmu2 && dlv2 && n > 0 && bs > 0,
              "rac_kl_bwd: bad args");
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), mu1, lv1, mu2, lv2, gout, (long)n, bs,
                     dmu1, dlv1, dmu2, dlv2);
  return check_launch("rac_kl_bwd");
}

int rac_cem_step_tail(const float* x4, const float* curr, const float* next_mask, const float* goal_img,
                      const float* cost_mask, const uint8_t* goal_mask, int32_t kind, float weight, int32_t add_cost,
                      float* next_out, double* sum_cost, int32_t N, int32_t HW, void* stream) {
  RAC_REQUIRE(x4 && curr && goal_img && next_out && sum_cost && N > 0 && HW > 0 && aligned16(x4),
              "rac_cem_step_tail: bad args");
  RAC_REQUIRE(kind == 0 || kind == 1, "rac_cem_step_tail: kind must be 0 (l2) or 1 (dontcare)");
  hipLaunchKernelGGL(cem_step_tail_kernel, dim3(N), dim3(256), 0, ST(stream), (const f32x4*)x4, curr, next_mask,
                     goal_img, cost_mask, goal_mask, kind, weight, add_cost, next_out, sum_cost, HW);
  return check_launch("rac_cem_step_tail");
}

int rac_cem_robot_inputs(const float* actions, const float* start_state, const float* low, const float* high,
                         const uint8_t* atlas, int32_t nx, int32_t ny, float x0, float y0, float dx, float dy,
                         float diff_x, float diff_y, float push_height, float* states, float* masks, int32_t T,
                         int32_t N, int32_t A, int32_t HW, int32_t per_sample, void* stream) {
  RAC_REQUIRE(actions && start_state && low && high && atlas && states && masks, "rac_cem_robot_inputs: null pointer");
  RAC_REQUIRE(T >= 0 && N > 0 && A >= 2 && HW > 0 && nx > 0 && ny > 0 && dx > 0.f && dy > 0.f && T < 65535,
              "rac_cem_robot_inputs: bad sizes");
  RobotAtlasP p{actions, start_state, low, high, atlas, states, masks, T, N, A, HW, nx, ny, x0, y0, 1.f / dx, 1.f / dy,
                diff_x, diff_y, push_height, per_sample ? 1 : 0};
  hipLaunchKernelGGL(robot_atlas_kernel, dim3(N, T + 1), dim3(256), 0, ST(stream), p);
  return check_launch("rac_cem_robot_inputs");
}

}  // extern "C"

// ---- PSNR / SSIM of the eval path ---------------------------------------------------------------
namespace rac {

constexpr int SS_T = 16;             // output tile edge
constexpr int SS_R = 5;              // window radius (11x11)
constexpr int SS_E = SS_T + 2 * SS_R;  // staged edge incl. halo

struct SsimWin {
  float g[11];
};

// grid (tiles_x * tiles_y, 3, N), 256 threads: one 16x16 tile of one channel plane per workgroup.
__global__ void psnr_ssim_kernel(const float* a, const float* b, const float* mask, float* sq_err, float* ssim_sum,
                                 float* ssim_map, int H, int W, int tiles_x, SsimWin win) {
  __shared__ float sa[SS_E][SS_E + 1], sb[SS_E][SS_E + 1];
  __shared__ float red[8];
  const int n = blockIdx.z, c = blockIdx.y;
  const int ty0 = (blockIdx.x / tiles_x) * SS_T, tx0 = (blockIdx.x % tiles_x) * SS_T;
  const long plane = ((long)n * 3 + c) * H * W;
  for (int i = threadIdx.x; i < SS_E * SS_E; i += 256) {
    const int ly = i / SS_E, lx = i - ly * SS_E;
    const int y = ty0 + ly - SS_R, x = tx0 + lx - SS_R;
    float va = 0.f, vb = 0.f;
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
      const bool zero = mask && mask[(long)n * H * W + (long)y * W + x] != 0.f;
      va = a[plane + (long)y * W + x];
      vb = b[plane + (long)y * W + x];
      if (zero) {
        va *= 0.f;
        vb *= 0.f;
      }
    }
    sa[ly][lx] = va;
    sb[ly][lx] = vb;
  }
  __syncthreads();
  const int ly = threadIdx.x / SS_T, lx = threadIdx.x % SS_T;
  const int y = ty0 + ly, x = tx0 + lx;
  float se = 0.f, ss = 0.f;
  if (y < H && x < W) {
    float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        const float w = win.g[i] * win.g[j];
        const float p = sa[ly + i][lx + j], q = sb[ly + i][lx + j];
        m1 += w * p;
        m2 += w * q;
        s11 += w * (p * p);
        s22 += w * (q * q);
        s12 += w * (p * q);
      }
    }
    const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
    const float v1 = s11 - m1s, v2 = s22 - m2s, v12 = s12 - m12;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    ss = ((2.f * m12 + C1) * (2.f * v12 + C2)) / ((m1s + m2s + C1) * (v1 + v2 + C2));
    if (ssim_map) ssim_map[plane + (long)y * W + x] = ss;
    const float pa = fminf(fmaxf(sa[ly + SS_R][lx + SS_R], 0.f), 1.f);
    const float pb = fminf(fmaxf(sb[ly + SS_R][lx + SS_R], 0.f), 1.f);
    const float d = (pa + 1.f) * 0.5f - (pb + 1.f) * 0.5f;
    se = d * d;
  }
  se = wave_sum(se);
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = se;
    red[4 + (threadIdx.x >> 6)] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(sq_err + n, (red[0] + red[1]) + (red[2] + red[3]));
    atomicAdd(ssim_sum + n, (red[4] + red[5]) + (red[6] + red[7]));
  }
}

}  // namespace rac

extern "C" int rac_psnr_ssim(const float* a, const float* b, const float* mask, float* sq_err, float* ssim_sum,
                             float* ssim_map, int32_t N, int32_t H, int32_t W, void* stream) {
  using namespace rac;
  RAC_REQUIRE(a && b && sq_err && ssim_sum && N > 0 && H > 0 && W > 0, "rac_psnr_ssim: bad args");
  SsimWin win;
  double tot = 0.0;
  double gd[11];
  for (int i = 0; i < 11; ++i) {  // gaussian(11, 1.5) of src/utils/metrics.py:13-15, normalised in fp32 like torch
    gd[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
  }
  float gf[11], sum = 0.f;
  for (int i = 0; i < 11; ++i) {
    gf[i] = (float)gd[i];
    sum += gf[i];
  }
  (void)tot;
  for (int i = 0; i < 11; ++i) win.g[i] = gf[i] / sum;
  const int tx = cdiv(W, SS_T), ty = cdiv(H, SS_T);
  hipLaunchKernelGGL(psnr_ssim_kernel, dim3(tx * ty, 3, N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b,
                     mask, sq_err, ssim_sum, ssim_map, H, W, tx, win);
  return check_launch("rac_psnr_ssim");
}

RAC_DEVICE_CODE_END
