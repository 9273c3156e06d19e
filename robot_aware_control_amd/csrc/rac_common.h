// Shared host/device helpers for librac_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "rac_hip.h"

// No v_pk_{mul,add,fma}_f32 in DEVICE code: the compiler's packed form of an fp32 inner product (broadcast operand through
// op_sel) was not reproducible when several processes shared the GPU (profiles/r03_pk_fma_experiment.md); the
// scalar forms measure the same on the train step and the planner.  A function attribute of the device pass only (the
// host pass of the same translation unit never sees an AMDGPU feature name); every .hip file ends with RAC_DEVICE_CODE_END.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#define RAC_DEVICE_CODE_END _Pragma("clang attribute pop")
#else
#define RAC_DEVICE_CODE_END
#endif

namespace rac {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return RAC_ELAUNCH;
  }
  return RAC_OK;
}

#define RAC_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      rac::set_error(__VA_ARGS__);  \
      return RAC_EINVAL;            \
    }                               \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// accurate forms (match libm to ~1 ulp; parity with the CPU path matters more than speed here)
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One Adam update (torch.optim.Adam semantics, reference trainer.py:109-110,461), shared by every kernel that applies it
// (adam_kernel, adam_ranges_kernel, adam_frag_multi_kernel) with floating-point contraction OFF: the same bits whichever
// kernel updates an element.  step_size = lr / (1 - beta1^t), inv_sqrt_bc2 = 1 / sqrt(1 - beta2^t).
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                            float step_size, float inv_sqrt_bc2) {
#pragma clang fp contract(off)
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + ((1.f - b2) * g) * g;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}

// max |v| bookkeeping for the split-precision convs (include/rac_hip.h, rac_absmax): a kernel that produces a
// tensor folds the bit pattern of its max |v| into a device slot -- wave reduce, then at most one atomic per wave and
// only if it can raise the slot.  Every lane of the wave must call amax_commit.
__device__ __forceinline__ unsigned absbits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7FFFFFFFu; }
__device__ __forceinline__ void amax_commit(unsigned m, unsigned* slot) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m);
}

// Workgroup form for the streaming kernels (every thread of the workgroup must call it): one conditional atomic per
// workgroup -- thousands of waves hitting ONE address cost tens of microseconds.
__device__ __forceinline__ void amax_commit_block(unsigned m, unsigned* slot) {
  __shared__ unsigned amax_sh[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) amax_sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) m = max(m, amax_sh[w]);
    if (m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m);
  }
}

}  // namespace rac
