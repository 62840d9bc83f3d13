// Micro-test (GPU box): accuracy of sin_u (generation 2: pre-scaled argument) - the shipped polynomial against a v_sin_f32
// variant - over argument ranges the student reaches.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../talking-head-anime-4-demo_amd/csrc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "siren_kernels.h"
using namespace tha4;
__device__ float sin_u_hw(float u) {
  const float th = fmaf(u, 0x1.45f306p-3f, 12582912.0f);
  const float kh = th - 12582912.0f;
  float rh = fmaf(-kh, 6.28125f, u);
  rh = fmaf(-kh, 0x1.fb5444p-10f, rh);
  return __builtin_amdgcn_sinf(rh * 0x1.45f306p-3f);
}
__device__ float sin_raw(float u) { return __builtin_amdgcn_sinf(u * 0x1.45f306p-3f); }
__global__ void k(const float* z, float* a, float* b, float* c, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { a[i] = sin_u(z[i]); b[i] = sin_u_hw(z[i]); c[i] = sin_raw(z[i]); }
}
int main() {
  const int n = 1 << 22;
  for (float range : {3.0f, 50.0f, 500.0f, 5000.0f}) {
    std::vector<float> z(n), a(n), b(n), c(n);
    for (int i = 0; i < n; ++i) z[i] = -range + 2 * range * (float)i / n;
    float *dz, *da, *db, *dc;
    hipMalloc(&dz, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dz, z.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dz, da, db, dc, n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double ea = 0, eb = 0, ec = 0; int ib = 0;
    for (int i = 0; i < n; ++i) {
      const double ref = std::sin((double)z[i]);
      ea = std::fmax(ea, std::fabs(a[i] - ref));
      if (std::fabs(b[i] - ref) > eb) { eb = std::fabs(b[i] - ref); ib = i; }
      ec = std::fmax(ec, std::fabs(c[i] - ref));
    }
    printf("|u| <= %6.0f : max|err| poly %.3e   v_sin + Cody-Waite %.3e (worst at u = %.6f: %.7f vs %.7f)   v_sin(u / 2pi) %.3e\n", range, ea, eb, z[ib], b[ib],
           std::sin((double)z[ib]), ec);
    hipFree(dz); hipFree(da); hipFree(db); hipFree(dc);
  }
  return 0;
}
