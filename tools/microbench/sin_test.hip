// Micro-test (GPU box): accuracy of candidate sin(30 z) implementations against fp64.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../talking-head-anime-4-demo_amd/csrc/siren_kernels.h"
using namespace tha4;
__device__ float sin_hw(float z) {           // v_sin_f32 takes revolutions
  const float t = z * 4.774648292756860f;   // 30 / (2 pi)
  const float r = t - rintf(t);
  return __builtin_amdgcn_sinf(r);
}
__device__ float sin_hw2(float z) {          // reference rounding of u = 30 z first, then 2-term reduction in revolutions
  const float u = 30.0f * z;
  const float k = rintf(u * 0.15915494309189535f);
  float r = fmaf(-k, 6.28125f, u);           // 2 pi split: 6.28125 + 1.9353071795864769e-3
  r = fmaf(-k, 1.9353071795864769e-3f, r);
  return __builtin_amdgcn_sinf(r * 0.15915494309189535f);
}
__global__ void k(const float* z, float* a, float* b, float* c, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { a[i] = sin_omega(z[i]); b[i] = sin_hw(z[i]); c[i] = sin_hw2(z[i]); }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> z(n), a(n), b(n), c(n);
  for (int i = 0; i < n; ++i) z[i] = -1.4f + 2.8f * (float)i / n;
  float *dz, *da, *db, *dc;
  hipMalloc(&dz, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(dz, z.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dz, da, db, dc, n);
  hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double ea = 0, eb = 0, ec = 0, ta = 0, tb = 0, tc = 0;
  for (int i = 0; i < n; ++i) {
    const double ref = std::sin((double)(30.0f * z[i])), tru = std::sin(30.0 * (double)z[i]);
    ea = std::fmax(ea, std::fabs(a[i] - ref)); eb = std::fmax(eb, std::fabs(b[i] - ref)); ec = std::fmax(ec, std::fabs(c[i] - ref));
    ta = std::fmax(ta, std::fabs(a[i] - tru)); tb = std::fmax(tb, std::fabs(b[i] - tru)); tc = std::fmax(tc, std::fabs(c[i] - tru));
  }
  printf("max|err| vs sin(fl(30z)) : poly %.3e  hw(fused) %.3e  hw(cody-waite) %.3e\n", ea, eb, ec);
  printf("max|err| vs sin(30z) exact: poly %.3e  hw(fused) %.3e  hw(cody-waite) %.3e\n", ta, tb, tc);
  return 0;
}
