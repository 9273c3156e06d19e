// Standalone reproducer for the v_pk_fma_f32 question of DESIGN.md ("A correctness note"): does a packed fp32 FMA with an
// op_sel broadcast -- exactly the instruction forms the compiler emitted in first_layer_kernel -- return wrong sums when
// several processes share the GPU?  No library, no LDS (variants 0-2), no memory traffic inside the loop: every lane
// accumulates small integers (exact in fp32), checks its own result and counts mismatches per lane.
//   hipcc --offload-arch=gfx950 -O2 tools/repro/pk_fma_repro.hip -o build/pk_fma_repro
//   build/pk_fma_repro <variant> <launches> [iters]
// variant 0: v_pk_fma_f32 with op_sel broadcasts (op_sel_hi:[1,0,1] and op_sel:[0,1,0]), operands in registers
//         1: v_pk_fma_f32 without op_sel (plain packed)
//         2: four v_fma_f32 (the scalar form the library is built with)
//         3: as 0, operands re-read from LDS every iteration (ds_read_b64 / ds_read_b32 next to the FMAs, as in the kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void pk_kernel(unsigned* bad, int iters, float p0, float p1) {
  __shared__ f2 w_sh[256];
  __shared__ float px_sh[2];
  const int lane = threadIdx.x & 63;
  f2 w = {(float)(lane % 7 + 1), (float)(lane % 5 + 2)};
  f2 px = {p0, p1};
  w_sh[threadIdx.x] = w;
  if (threadIdx.x < 2) px_sh[threadIdx.x] = threadIdx.x ? p1 : p0;
  __syncthreads();
  f2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
    if (V == 3) {
      w = *(volatile f2*)&w_sh[threadIdx.x];
      px.x = *(volatile float*)&px_sh[0];
      px.y = *(volatile float*)&px_sh[1];
    }
    if (V == 0 || V == 3) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(lo) : "v"(w), "v"(px));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(hi) : "v"(w), "v"(px));
    } else if (V == 1) {
      f2 pa = {px.x, px.x}, pb = {px.y, px.y};
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(w), "v"(pa));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(w), "v"(pb));
    } else {
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo.x) : "v"(w.x), "v"(px.x));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo.y) : "v"(w.y), "v"(px.x));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi.x) : "v"(w.x), "v"(px.y));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi.y) : "v"(w.y), "v"(px.y));
    }
  }
  const float n = (float)iters;
  const f2 w0 = {(float)(lane % 7 + 1), (float)(lane % 5 + 2)};
  const bool ok = lo.x == n * w0.x * p0 && lo.y == n * w0.y * p0 && hi.x == n * w0.x * p1 && hi.y == n * w0.y * p1;
  if (!ok) {
    atomicAdd(&bad[lane], 1u);
    atomicAdd(&bad[64], 1u);
  }
}

int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0, launches = argc > 2 ? atoi(argv[2]) : 500;
  const int iters = argc > 3 ? atoi(argv[3]) : 32768;
  unsigned* bad;
  if (hipMalloc(&bad, 65 * 4) != hipSuccess || hipMemset(bad, 0, 65 * 4) != hipSuccess) return 2;
  unsigned host[65], prev = 0;
  int bad_launches = 0;
  for (int l = 0; l < launches; ++l) {
    switch (variant) {
      case 0: hipLaunchKernelGGL(pk_kernel<0>, dim3(2048), dim3(256), 0, 0, bad, iters, 1.0f, 2.0f); break;
      case 1: hipLaunchKernelGGL(pk_kernel<1>, dim3(2048), dim3(256), 0, 0, bad, iters, 1.0f, 2.0f); break;
      case 2: hipLaunchKernelGGL(pk_kernel<2>, dim3(2048), dim3(256), 0, 0, bad, iters, 1.0f, 2.0f); break;
      default: hipLaunchKernelGGL(pk_kernel<3>, dim3(2048), dim3(256), 0, 0, bad, iters, 1.0f, 2.0f); break;
    }
    if (hipMemcpy(host, bad, 65 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    if (host[64] != prev) ++bad_launches, prev = host[64];
  }
  unsigned q[4] = {0, 0, 0, 0};
  for (int i = 0; i < 64; ++i) q[i >> 4] += host[i];
  printf("variant %d pid %d: %d launches x 2048 x 256 threads x %d iterations: %d launches with wrong lanes; wrong results by "
         "lane quarter [0-15 16-31 32-47 48-63] = %u %u %u %u\n", variant, (int)getpid(), launches, iters, bad_launches, q[0], q[1],
         q[2], q[3]);
  return 0;
}
