// Which LDS access patterns of the weight-gradient kernels' operand staging are bank-conflict free on gfx950?
// One kernel per pattern (so that rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE reports them apart); every kernel
// repeats its access `iters` times on a 16 KB [row][256 B] image.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_conflicts.hip -o build/lds_conflicts
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d out -o run --output-format csv -- build/lds_conflicts
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define LDSP(T, base, off) reinterpret_cast<__attribute__((address_space(3))) T*>((__attribute__((address_space(3))) unsigned char*)(base) + (off))

__device__ __forceinline__ int swz_old(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
__device__ __forceinline__ int swz_new(int r) { return (r & 3) | (((r ^ (r >> 3)) & 1) << 2); }

// PAT 0: x store, old swizzle   1: x store, new swizzle   2: dy store (two halves in order)   3: dy store, halves flipped
// by row parity   4: linear b128 store (lane l at 16 l)   5: transposing read, old swizzle   6: transposing read, new
// 7: linear b64 read (lane l at 8 l)   8: x store new swizzle, but b64 x 2   9: dy flipped + new swizzle
// 13: rows kernel's staging store into the padded image (W = 16)   14: linear store, second buffer one slot off   15: as 13,
// W = 32   16 / 17 / 18: the rows kernel's ds_read_b128 fragment reads (tap shift 0, (-1, -1) at W = 16, (+1, +1) at W = 32)
// 10: dy store, halves flipped where the (swizzled) segment index has bit 2 set   11: 64-co dy store (4 lanes per row),
// halves in order   12: the same, flipped by row parity
// Measured (profiles/r04_lds_conflicts.md): stores go 8 lanes x 16 B per clock into 32 banks (addresses mod 128 B), so
// patterns 2 / 3 / 9 / 11 are 2-way conflicts and 10 / 12 are free; the transposing reads are free with either swizzle.
template <int PAT>
__global__ __launch_bounds__(256) void pattern(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const int skk = tid >> 3, ssub = tid & 7;
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  float acc = 0.f;
  int a0 = 0, a1 = 0;
  if (PAT == 0 || PAT == 1 || PAT == 8) {
    const int sw = PAT == 0 ? swz_old(skk) : swz_new(skk);
    a0 = skk * 256 + (((ssub >> 1) ^ sw) * 32) + (ssub & 1) * 16;
    a1 = a0 ^ 128;
  } else if (PAT == 2 || PAT == 3 || PAT == 9) {
    const int sw = PAT == 9 ? swz_new(skk) : swz_old(skk);
    const int flip = PAT == 2 ? 0 : (skk & 1) * 16;
    a0 = skk * 256 + ((ssub ^ sw) * 32) + flip;
    a1 = a0 ^ 16;
  } else if (PAT == 10) {
    const int seg = ssub ^ swz_old(skk);
    a0 = skk * 256 + seg * 32 + ((seg >> 2) & 1) * 16;
    a1 = a0 ^ 16;
  } else if (PAT == 11 || PAT == 12) {
    const int row = (tid >> 2) & 31, sub = tid & 3;
    a0 = row * 256 + ((sub ^ swz_old(row)) * 32) + (PAT == 12 ? (row & 1) * 16 : 0);
    a1 = a0 ^ 16;
  } else if (PAT >= 13 && PAT <= 18) {
    // the rows kernel's padded image: row r of a W-pixel image row sits at slot (r / W) * (W + 2) + 1 + r % W (16 B slots)
    const int W = (PAT == 15 || PAT == 18) ? 32 : 16, WP = W + 2;
    if (PAT <= 15) {  // staging store: thread = row tid of 8-channel group 0 / 1 (planes 3072 B apart)
      a0 = ((tid / W) * WP + 1 + tid % W) * 16;
      a1 = a0 + 16 * 3072;
      if (PAT == 14) a0 = tid * 16, a1 = 4096 + tid * 16 + 16;  // linear, second one off by a slot
    } else {          // fragment read: lane (k-group lq, row lr) of block (wave) b, tap shift (ky - 1) * WP + (kx - 1)
      const int lq = lane >> 4, lr = lane & 15, b = tid >> 6;
      const int r0 = b * 16;
      const int sh = PAT == 16 ? 0 : (PAT == 17 ? -WP - 1 : WP + 1);
      a0 = lq * 3072 + ((1 + r0 / W) * WP + 1 + r0 % W + lr + sh) * 16;
      a1 = a0 + 4 * 3072;
    }
  } else if (PAT == 4) {
    a0 = tid * 16, a1 = 4096 + tid * 16;
  } else if (PAT == 5 || PAT == 6) {
    const int fg = lane >> 4, fq = (lane & 15) >> 2, fp = lane & 3;
    const int fsw = PAT == 5 ? swz_old(8 * fg + fq) : swz_new(8 * fg + fq);
    a0 = (8 * fg + fq) * 256 + fp * 8 + (((tid >> 6) ^ fsw) * 32);
    a1 = a0 + 1024;
  } else {
    a0 = tid * 8, a1 = 2048 + tid * 8;
  }
  for (int it = 0; it < iters; ++it) {
    if (PAT <= 4 || (PAT >= 9 && PAT <= 15)) {
      *reinterpret_cast<u32x4*>(lds + a0) = v;
      *reinterpret_cast<u32x4*>(lds + a1) = v;
      v.x += 1u;
    } else if (PAT == 8) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<u32x2*>(lds + a0) = u32x2{v.x, v.y};
      *reinterpret_cast<u32x2*>(lds + a0 + 8) = u32x2{v.z, v.w};
      *reinterpret_cast<u32x2*>(lds + a1) = u32x2{v.x, v.y};
      *reinterpret_cast<u32x2*>(lds + a1 + 8) = u32x2{v.z, v.w};
      v.x += 1u;
    } else if (PAT == 5 || PAT == 6) {
      const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDSP(h16x4, lds, a0));
      const h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDSP(h16x4, lds, a1));
      acc += (float)lo[0] + (float)hi[1];
    } else if (PAT >= 16) {
      const u32x4 x = *reinterpret_cast<volatile u32x4*>(lds + a0);
      const u32x4 y = *reinterpret_cast<volatile u32x4*>(lds + a1);
      acc += (float)(x.x + y.w);
    } else {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 x = *reinterpret_cast<volatile u32x2*>(lds + a0);
      const u32x2 y = *reinterpret_cast<volatile u32x2*>(lds + a1);
      acc += (float)(x.x + y.y);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  out[blockIdx.x * 256 + tid] = acc + (float)reinterpret_cast<unsigned*>(lds)[tid] + (float)v.x;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * sizeof(float));
  const int iters = 20000;
#define RUN(P) hipLaunchKernelGGL(pattern<P>, dim3(256), dim3(256), 0, 0, out, iters);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18)
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
