// A compiled (non-Python) consumer of include/rac_hip.h: what a C++ host that binds librac_hip.so would do.
// Built by __graft_entry__.build():  hipcc -I include tests/abi_consumer.cpp -L robot_aware_control_amd -lrac_hip
//   ./abi_consumer        CPU: the library answers with the header's ABI version; prints the struct layouts this
//                         translation unit sees (tests compare them with the ctypes mirrors of _lib.py)
//   ./abi_consumer gpu    one rac_conv2d call (3x3 conv, 8x8 map, 8 -> 8 channels) checked against a host loop
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "rac_hip.h"

#define HIP_OK(e)                                                          \
  do {                                                                     \
    hipError_t err_ = (e);                                                 \
    if (err_ != hipSuccess) {                                              \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err_));           \
      return 2;                                                            \
    }                                                                      \
  } while (0)

static int gpu_conv_check() {
  const int B = 2, H = 8, W = 8, C = 8, N = 8, K = 3;
  std::vector<float> x(B * H * W * C), w(N * K * K * C), bias(N), want(B * H * W * N), got(want.size());
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto& v : x) v = rnd();
  for (auto& v : w) v = rnd() * 0.5f;
  for (auto& v : bias) v = rnd();
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
      for (int xx = 0; xx < W; ++xx)
        for (int n = 0; n < N; ++n) {
          double acc = bias[n];
          for (int ky = 0; ky < K; ++ky)
            for (int kx = 0; kx < K; ++kx) {
              const int yy = y + ky - 1, xs = xx + kx - 1;
              if (yy < 0 || yy >= H || xs < 0 || xs >= W) continue;
              for (int c = 0; c < C; ++c)
                acc += (double)x[((b * H + yy) * W + xs) * C + c] * w[((n * K + ky) * K + kx) * C + c];
            }
          want[((b * H + y) * W + xx) * N + n] = (float)acc;
        }
  float *dx, *dw, *db, *dy;
  HIP_OK(hipMalloc(&dx, x.size() * 4));
  HIP_OK(hipMalloc(&dw, w.size() * 4));
  HIP_OK(hipMalloc(&db, bias.size() * 4));
  HIP_OK(hipMalloc(&dy, got.size() * 4));
  HIP_OK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  rac_conv_args a;
  memset(&a, 0, sizeof a);
  a.mode = RAC_CONV_FWD, a.B = B, a.H = H, a.W = W, a.ksize = K, a.Cin = C, a.Cout = N, a.act = RAC_ACT_NONE;
  a.split_k = 1, a.a_split = C, a.a0 = dx, a.w = dw, a.out0 = dy, a.bias = db;
  const int rc = rac_conv2d(&a, stream);
  if (rc != RAC_OK) {
    fprintf(stderr, "rac_conv2d failed (%d): %s\n", rc, rac_last_error());
    return 3;
  }
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(got.data(), dy, got.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (size_t i = 0; i < got.size(); ++i) {
    worst = fmax(worst, fabs((double)got[i] - want[i]));
    scale = fmax(scale, fabs((double)want[i]));
  }
  printf("conv_max_rel_err %.3e\n", worst / scale);
  // a bad argument comes back as RAC_EINVAL with a message, not as a crash
  a.ksize = 4;
  const int bad = rac_conv2d(&a, stream);
  printf("bad_ksize_rc %d\n", bad);
  (void)hipFree(dx), (void)hipFree(dw), (void)hipFree(db), (void)hipFree(dy);
  return (worst / scale < 1e-5 && bad == RAC_EINVAL && strlen(rac_last_error()) > 0) ? 0 : 4;
}

int main(int argc, char** argv) {
  printf("rac_version %d\n", rac_version());
  printf("RAC_ABI_VERSION %d\n", RAC_ABI_VERSION);
  printf("arch %s\n", rac_device_arch());
  printf("sizeof_rac_conv_args %zu\n", sizeof(rac_conv_args));
  printf("offsetof_rac_conv_args_slab_stride %zu\n", offsetof(rac_conv_args, slab_stride));
  printf("offsetof_rac_conv_args_a0 %zu\n", offsetof(rac_conv_args, a0));
  printf("offsetof_rac_conv_args_stats_rows %zu\n", offsetof(rac_conv_args, stats_rows));
  printf("offsetof_rac_conv_args_a0_up %zu\n", offsetof(rac_conv_args, a0_up));
  printf("offsetof_rac_conv_args_amax_per_image %zu\n", offsetof(rac_conv_args, amax_per_image));
  printf("sizeof_rac_wgrad_args %zu\n", sizeof(rac_wgrad_args));
  printf("offsetof_rac_wgrad_args_dy %zu\n", offsetof(rac_wgrad_args, dy));
  printf("offsetof_rac_wgrad_args_dw %zu\n", offsetof(rac_wgrad_args, dw));
  printf("offsetof_rac_wgrad_args_presplit %zu\n", offsetof(rac_wgrad_args, presplit));
  printf("sizeof_rac_absmax_job %zu\n", sizeof(rac_absmax_job));
  printf("sizeof_rac_frag_job %zu\n", sizeof(rac_frag_job));
  printf("offsetof_rac_frag_job_block_begin %zu\n", offsetof(rac_frag_job, block_begin));
  printf("sizeof_rac_grad_src %zu\n", sizeof(rac_grad_src));
  printf("offsetof_rac_grad_src_n_slabs %zu\n", offsetof(rac_grad_src, n_slabs));
  printf("offsetof_rac_grad_src_col_off %zu\n", offsetof(rac_grad_src, col_off));
  if (rac_version() != RAC_ABI_VERSION || strcmp(rac_device_arch(), "gfx950") != 0) return 1;
  if (argc > 1 && strcmp(argv[1], "gpu") == 0) return gpu_conv_check();
  return 0;
}
