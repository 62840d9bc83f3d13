// Machine-model microbenchmarks for gfx950 (tuning aid): shader clock under load, VALU issue rate for scalar / packed
// fp32 FMA, v_sin_f32, conversions, and MFMA co-issue, at 1/2/4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int ITERS = 4096, UNROLL = 16;

template <int MODE>
__global__ void __launch_bounds__(1024) rate_kernel(float* out, long long* clk) {
  float a[UNROLL];
  f32x2 p[UNROLL];
  f32x4 acc[4] = {};
  f16x8 ha, hb;
  for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)(threadIdx.x * 0.001f + j); hb[j] = (_Float16)(j * 0.5f); }
  for (int i = 0; i < UNROLL; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
  const float m = 1.0001f, c = 0.5f;
  const f32x2 m2 = {m, m}, c2 = {c, c};
  long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) a[i] = __builtin_fmaf(a[i], m, c);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) a[i] = __builtin_amdgcn_sinf(a[i]);
    } else if (MODE == 3) {   // MFMA only
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i & 3], 0, 0, 0);
    } else if (MODE == 4) {   // MFMA + 4 pk_fma each
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i & 3], 0, 0, 0);
        p[i] = __builtin_elementwise_fma(p[i], m2, c2);
        p[(i + 5) & 15] = __builtin_elementwise_fma(p[(i + 5) & 15], m2, c2);
        p[(i + 9) & 15] = __builtin_elementwise_fma(p[(i + 9) & 15], m2, c2);
      }
    } else if (MODE >= 10 && MODE < 30) {   // 1 MFMA + (MODE-10) scalar fma
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i & 3], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < MODE - 10; ++k) a[(i + 3 * k) & 15] = __builtin_fmaf(a[(i + 3 * k) & 15], m, c);
      }
    } else if (MODE == 30) {   // subnormal check: not a timing mode
    } else if (MODE == 5) {   // rndne + cvt
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) a[i] = __builtin_rintf(a[i] * m);
    } else if (MODE == 6) {   // cvt f32->f16->f32
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) a[i] = (float)(_Float16)a[i] + c;
    }
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i][0] + p[i][1];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, int insts_per_iter, int waves_per_simd) {
  float* out; long long* clk;
  const int threads = 64 * 4 * waves_per_simd;       // one workgroup per CU
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<MODE><<<256, threads>>>(out, clk);
  hipEventRecord(e0);
  rate_kernel<MODE><<<256, threads>>>(out, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double wall_s = h[1] / 100e6;                  // wall_clock64: 100 MHz
  const double insts = (double)ITERS * insts_per_iter * waves_per_simd;   // per SIMD
  const double clock_hz = h[0] / wall_s;
  printf("%-22s waves/SIMD %d: %.1f us, shader clock %.0f MHz, %.2f cycles per wave-instr per SIMD (kernel time x clock / instrs per SIMD)\n",
         name, waves_per_simd, ms * 1e3, clock_hz / 1e6, ms * 1e-3 * clock_hz / insts);
  hipFree(out); hipFree(clk);
}

__global__ void subnormal_kernel(float* out) {
  // A[i][k] = 1 for k == 0 else 0 ; B[0][j] = fp16 subnormal 2^-20 (j+1)  ->  D[i][j] = 2^-20 (j+1) if MFMA keeps fp16 denormals
  const int lane = threadIdx.x;
  f16x8 a = {}, b = {};
  if ((lane >> 4) == 0) { a[0] = (_Float16)1.0f; b[0] = (_Float16)(9.5367431640625e-07f * ((lane & 15) + 1)); }
  f32x4 d = {};
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
  out[lane] = d[0];
  // conversion: fp32 -> fp16 subnormal -> fp32
  const float x = 3.0e-6f * (lane + 1);
  out[64 + lane] = (float)(_Float16)x;
  // fma_mix style: fp16(v - (float)hi)
  const float v = 0.01f * (lane + 1) + 1.234e-5f;
  const _Float16 hi = (_Float16)v;
  out[128 + lane] = (float)(_Float16)(v - (float)hi);
  out[192 + lane] = v - (float)hi;
}

void subnormal_check() {
  float* out; hipMalloc(&out, 256 * 4);
  subnormal_kernel<<<1, 64>>>(out);
  float h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("MFMA fp16 subnormal inputs: D[0][j] =");
  for (int j = 0; j < 4; ++j) printf(" %.6g (want %.6g)", h[j], 9.5367431640625e-07 * (j + 1));
  printf("\ncvt f32->f16 subnormal:");
  for (int j = 0; j < 4; ++j) printf(" %.6g (in %.6g)", h[64 + j], 3.0e-6 * (j + 1));
  printf("\nlo = f16(v - hi):");
  for (int j = 0; j < 4; ++j) printf(" %.6g (exact %.6g)", h[128 + j], h[192 + j]);
  printf("\n");
  hipFree(out);
}

int main() {
  subnormal_check();
  for (int w : {1, 2, 4}) {
    if (w == 1) continue;
    run<0>("v_fma_f32", UNROLL, w);
    run<1>("v_pk_fma_f32", UNROLL, w);
    run<2>("v_sin_f32", UNROLL, w);
    run<3>("mfma_16x16x32_f16", UNROLL, w);
    run<4>("mfma + 3 pk_fma", UNROLL * 4, w);
    run<10>("mfma + 0 fma", UNROLL * 1, w);
    run<12>("mfma + 2 fma", UNROLL * 3, w);
    run<14>("mfma + 4 fma", UNROLL * 5, w);
    run<18>("mfma + 8 fma", UNROLL * 9, w);
    run<22>("mfma + 12 fma", UNROLL * 13, w);
  }
  return 0;
}
