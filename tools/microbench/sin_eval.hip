// Device evaluator of the student's sine variants over caller-given arguments (tools/sin_cliff.py; tuning aid).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I talking-head-anime-4-demo_amd/csrc tools/microbench/sin_eval.hip -o tools/microbench/libsin_eval.so
// variant 0: the shipped 12-op polynomial sin_u (siren_kernels.h); 1: v_sin_f32 behind the 2-term Cody-Waite by 2 pi
// (the -DTHA4_HW_SIN body of sin_u); 2: v_sin_f32(u / 2 pi) without reduction.
#include <hip/hip_runtime.h>
#include "siren_kernels.h"
using namespace tha4;
__device__ float sin_u_hw(float u) {
  const float th = fmaf(u, 0x1.45f306p-3f, 12582912.0f);
  const float kh = th - 12582912.0f;
  float rh = fmaf(-kh, 6.28125f, u);
  rh = fmaf(-kh, 0x1.fb5444p-10f, rh);
  return __builtin_amdgcn_sinf(rh * 0x1.45f306p-3f);
}
__global__ void k(const float* u, float* o, long n, int variant) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = u[i];
  o[i] = variant == 0 ? sin_u(x) : variant == 1 ? sin_u_hw(x) : __builtin_amdgcn_sinf(x * 0x1.45f306p-3f);
}
extern "C" int sin_eval(const float* host_u, float* host_out, long n, int variant) {
  float *du = nullptr, *dout = nullptr;
  if (hipMalloc(&du, n * 4) != hipSuccess || hipMalloc(&dout, n * 4) != hipSuccess) return -1;
  if (hipMemcpy(du, host_u, n * 4, hipMemcpyHostToDevice) != hipSuccess) return -2;
  hipLaunchKernelGGL(k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, du, dout, n, variant);
  if (hipMemcpy(host_out, dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  (void)hipFree(du); (void)hipFree(dout);
  return 0;
}
