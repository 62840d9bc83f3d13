// Does VALU work of one wave hide under the MFMAs of ANOTHER wave of the same SIMD on gfx950?  (tuning aid)
// Waves of a workgroup are dealt to the 4 SIMDs round-robin, so with 8 waves per workgroup waves w and w+4 share a
// SIMD.  Roles: 'M' = a chain of independent v_mfma_f32_16x16x32_f16, 'V' = a chain of independent v_fma_f32.
//   MM : both waves of a SIMD run M      VV : both run V      MV : one runs M, the other V
// If the matrix pipe and the vector pipe overlap across waves, t(MV) ~ max(t(MM), t(VV)) / 2 .. max(tM, tV);
// if they serialise, t(MV) ~ (t(MM) + t(VV)) / 2.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap.hip -o tools/microbench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int ITERS = 2048;

__device__ __forceinline__ float role_m(int lane) {
  f16x8 ha, hb;
  for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)(lane * 0.001f + j); hb[j] = (_Float16)(j * 0.5f); }
  f32x4 acc[8] = {};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i & 7], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  return s;
}

template <int NV>   // NV fmas per iteration (16 MFMAs = 256 cycles per iteration in role M)
__device__ __forceinline__ float role_v(int lane) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = lane * 0.001f + i;
  const float m = 1.0001f, c = 0.5f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i & 15] = __builtin_fmaf(a[i & 15], m, c);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  return s;
}

// intra-wave: the same wave alternates 1 MFMA with NV/16 independent fmas
template <int NV>
__device__ __forceinline__ float role_both(int lane) {
  f16x8 ha, hb;
  for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)(lane * 0.001f + j); hb[j] = (_Float16)(j * 0.5f); }
  f32x4 acc[8] = {};
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = lane * 0.001f + i;
  const float m = 1.0001f, c = 0.5f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i & 7], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV / 16; ++k) a[(i + k * 5) & 15] = __builtin_fmaf(a[(i + k * 5) & 15], m, c);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += a[i];
  return s;
}

// pattern: bit w of `mask` set -> wave pair member w (0: waves 0-3, 1: waves 4-7, ...) runs M, else V; mode 2 = both in one wave
template <int NV>
__global__ void __launch_bounds__(1024) k(float* out, unsigned mask, int both) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int member = wave >> 2;
  float s;
  if (both) s = role_both<NV>(lane);
  else if ((mask >> member) & 1u) s = role_m(lane);
  else s = role_v<NV>(lane);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
float run(const char* name, int waves_per_simd, unsigned mask, int both) {
  float* out;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 256 * waves_per_simd;
  k<NV><<<256, threads>>>(out, mask, both);
  hipEventRecord(e0);
  k<NV><<<256, threads>>>(out, mask, both);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("NV=%3d %-34s waves/SIMD %d: %8.1f us\n", NV, name, waves_per_simd, ms * 1e3);
  hipFree(out);
  return ms;
}

template <int NV>
void suite() {
  const float m1 = run<NV>("M alone (1 wave/SIMD)", 1, 1u, 0);
  const float v1 = run<NV>("V alone (1 wave/SIMD)", 1, 0u, 0);
  const float mm = run<NV>("MM", 2, 3u, 0);
  const float vv = run<NV>("VV", 2, 0u, 0);
  const float mv = run<NV>("MV (cross-wave)", 2, 1u, 0);
  const float b1 = run<NV>("M+V interleaved in ONE wave (1/SIMD)", 1, 0u, 1);
  const float b2 = run<NV>("M+V interleaved, 2 waves/SIMD", 2, 0u, 1);
  const float mmvv = run<NV>("MMVV", 4, 3u, 0);
  printf("  -> cross-wave overlap: t(MV) = %.1f us vs max(M,V) = %.1f, sum = %.1f   |  intra-wave: %.1f vs max %.1f, sum %.1f | 2 waves interleaved %.1f (serial: %.1f) | MMVV %.1f (max(MM,VV) %.1f, sum %.1f)\n",
         mv * 1e3, (m1 > v1 ? m1 : v1) * 1e3, (m1 + v1) * 1e3, b1 * 1e3, (m1 > v1 ? m1 : v1) * 1e3, (m1 + v1) * 1e3, b2 * 1e3, 2 * (m1 + v1) * 1e3,
         mmvv * 1e3, (mm > vv ? mm : vv) * 1e3, (mm + vv) * 1e3);
}

int main() {
  suite<64>();    // V ~ 64 x 2.2 = 140 cycles per 256 MFMA cycles
  suite<112>();   // V ~ 250 cycles: balanced
  suite<192>();   // V-heavy
  return 0;
}
