// Which fp16 MFMA shape does the chip sustain better under its power cap?  Register-resident operands (random fp16
// values: toggling matters), no memory traffic: achieved TFLOP/s = what the clock it holds allows.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o build/mfma_power && build/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void burn(const f16x8* src, float* out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(tid * 8 + i) & 65535];
    b[i] = src[(tid * 8 + 4 + i) & 65535];
  }
  float sum = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      // the same FLOPs per iteration: 16 x (16x16x32) = 8 x (32x32x16)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 2 * r], b[j + 2 * r], acc[i * 2 + j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += acc[i][e];
  }
  out[tid] = sum;
}

int main() {
  const int blocks = 256 * 2, iters = 400000;
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  f16x8* src;
  float* out;
  hipMalloc(&src, h.size() * 2);
  hipMalloc(&out, blocks * 256 * 4);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int shape : {16, 32}) {
      hipEventRecord(e0);
      if (shape == 16) hipLaunchKernelGGL(burn<16>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      else hipLaunchKernelGGL(burn<32>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = 2.0 * 16 * 16 * 32 * 16 * (double)iters * blocks * 4;
      printf("%dx%d: %.2f ms  %.0f TFLOP/s (of 2500)\n", shape, shape, ms, flop / ms / 1e9);
    }
  return 0;
}
