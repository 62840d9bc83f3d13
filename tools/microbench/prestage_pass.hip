// What a pre-staged operand pass would cost (round 4, profiles/r04_full_conv_tile_reading.md §4): the fused "finalize + activate + split" kernel the
// round-3 review proposed - read an fp32 C16 tensor, apply per-channel scale / shift + SiLU, split into fp16 hi + fp16 lo, write the two
// operand planes (4 + 4 bytes per element, the fp32 tensor's own size) - timed on the tensor shapes of the full model at batch 1 and 8.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/prestage_pass.hip -o tools/microbench/prestage_pass && tools/microbench/prestage_pass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) prestage(const f32x4* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh, f16x4* __restrict__ hi,
                                                 f16x4* __restrict__ lo, size_t quads, int px, int cb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i & 3);                          // [n][cb][pixel][16]: 4 quads of 4 channels per pixel
    const int c = (int)((i >> 2) / px % cb) * 16 + q * 4;
    f32x4 v = x[i];
    f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = v[j] * sc[c + j] + sh[c + j];
      t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));      // SiLU
      h[j] = (_Float16)t;
      l[j] = (_Float16)(t - (float)h[j]);
    }
    hi[i] = h;
    lo[i] = l;
  }
}

int main() {
  struct Shape { const char* name; int n, c, hw; } shapes[] = {
      {"batch 1, 512x512 x 32 ch", 1, 32, 512}, {"batch 1, 256x256 x 128 ch", 1, 128, 256}, {"batch 1, 128x128 x 256 ch", 1, 256, 128},
      {"batch 1, 64x64 x 256 ch", 1, 256, 64},  {"batch 1, 16x16 x 512 ch", 1, 512, 16},
      {"batch 8, 512x512 x 32 ch", 8, 32, 512}, {"batch 8, 256x256 x 128 ch", 8, 128, 256}, {"batch 8, 128x128 x 256 ch", 8, 256, 128}, {"batch 8, 64x64 x 256 ch", 8, 256, 64}};
  float *sc, *sh;
  hipMalloc(&sc, 4096); hipMalloc(&sh, 4096);
  hipMemset(sc, 0, 4096); hipMemset(sh, 0, 4096);
  for (auto& s : shapes) {
    const size_t elems = (size_t)s.n * s.c * s.hw * s.hw, quads = elems / 4;
    f32x4* x; f16x4 *hi, *lo;
    hipMalloc(&x, elems * 4); hipMalloc(&hi, elems * 2); hipMalloc(&lo, elems * 2);
    hipMemset(x, 0, elems * 4);
    const int grid = (int)((quads + 255) / 256 < 256 * 16 ? (quads + 255) / 256 : 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(prestage, dim3(grid), dim3(256), 0, 0, x, sc, sh, hi, lo, quads, s.hw * s.hw, s.c / 16);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(prestage, dim3(grid), dim3(256), 0, 0, x, sc, sh, hi, lo, quads, s.hw * s.hw, s.c / 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / reps;
    std::printf("%-28s %8.1f M elements  %7.1f us per pass  %6.2f TB/s (8 B per element)\n", s.name, elems / 1e6, us, elems * 8.0 / (us * 1e-6) / 1e12);
    hipFree(x); hipFree(hi); hipFree(lo);
  }
  return 0;
}
