// Round 3: does VALU work hide under MFMAs on a gfx950 SIMD when the MFMA is the 8-pass v_mfma_f32_32x32x16_f16 (32 cycles)
// instead of the 4-pass v_mfma_f32_16x16x32_f16 (16 cycles) that mfma_valu_overlap.hip measured (answer there: no)?
// Every instruction of the timed loops is inline asm (the compiler can neither reorder MFMAs and FMAs nor fuse FMAs into
// v_pk_fma_f32); accumulators and FMA registers are all independent, so no software hazard nops are needed inside the loop.
//   role M : 8 MFMAs per iteration              role V : 8 x NV v_fma_f32 per iteration
//   role B : 8 x (1 MFMA + NV v_fma_f32), one wave           cross : waves 0-3 run M, waves 4-7 run V (pairs share a SIMD)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap2.hip -o tools/microbench/mfma_valu_overlap2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int ITERS = 4096;

#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define MFMA16(acc) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ha), "v"(hb))
#define MFMA32(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ha), "v"(hb))

// KIND 0: 16x16x32 (two per slot, so that a slot is 32 matrix cycles for both kinds), 1: 32x32x16.  ROLE 0 M, 1 V, 2 both.
// VK 0: v_fma_f32, 1: v_exp_f32 (transcendental)
template <int KIND, int NV, int ROLE, int VK>
__device__ __forceinline__ float body(int lane) {
  f16x8 ha, hb;
  for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)(lane * 0.001f + j); hb[j] = (_Float16)(j * 0.5f); }
  f32x16 a32[4] = {};
  f32x4 a16[8] = {};
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = lane * 0.001f + i;
  const float m = 1.0001f, c = 0.5f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (ROLE != 1) {
        if (KIND == 1) MFMA32(a32[s & 3]);
        else { MFMA16(a16[(2 * s) & 7]); if (NV <= 8) { /* second MFMA after half of the FMAs, below */ } }
      }
      if (ROLE != 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          if (KIND == 0 && ROLE != 1 && k == NV / 2) MFMA16(a16[(2 * s + 1) & 7]);
          if (VK == 0) FMA(a[(s * NV + k) & 15]); else EXP(a[(s * NV + k) & 15]);
        }
        if (KIND == 0 && ROLE != 1 && NV == 0) MFMA16(a16[(2 * s + 1) & 7]);
      } else if (KIND == 0) {
        MFMA16(a16[(2 * s + 1) & 7]);
      }
    }
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  float r = 0;
  for (int i = 0; i < 4; ++i) r += a32[i][0] + a32[i][15];
  for (int i = 0; i < 8; ++i) r += a16[i][0] + a16[i][3];
  for (int i = 0; i < 16; ++i) r += a[i];
  return r;
}

// mode 0: all waves M; 1: all V; 2: all B (interleaved in each wave); 3: cross (waves 0-3 M, 4-7 V; needs 512 threads);
// 4: cross with s_setprio 1 on the M waves; 5: cross with s_setprio 1 on the V waves
template <int KIND, int NV, int VK>
__global__ void __launch_bounds__(512) k(float* out, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float r;
  if (mode == 0) r = body<KIND, NV, 0, VK>(lane);
  else if (mode == 1) r = body<KIND, NV, 1, VK>(lane);
  else if (mode == 2) r = body<KIND, NV, 2, VK>(lane);
  else {
    if (wave < 4) { if (mode == 4) __builtin_amdgcn_s_setprio(1); r = body<KIND, NV, 0, VK>(lane); }
    else { if (mode == 5) __builtin_amdgcn_s_setprio(1); r = body<KIND, NV, 1, VK>(lane); }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int NV, int VK>
float run(int waves_per_simd, int mode) {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 256 * waves_per_simd;
  k<KIND, NV, VK><<<256, threads>>>(out, mode);
  hipEventRecord(e0);
  k<KIND, NV, VK><<<256, threads>>>(out, mode);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e3f;
}

template <int KIND, int NV, int VK>
void suite() {
  const float m1 = run<KIND, NV, VK>(1, 0), v1 = run<KIND, NV, VK>(1, 1), b1 = run<KIND, NV, VK>(1, 2);
  const float m2 = run<KIND, NV, VK>(2, 0), v2 = run<KIND, NV, VK>(2, 1), b2 = run<KIND, NV, VK>(2, 2);
  const float x = run<KIND, NV, VK>(2, 3), xm = run<KIND, NV, VK>(2, 4), xv = run<KIND, NV, VK>(2, 5);
  const float cyc = 2.4e3f / (ITERS * 8);      // us -> cycles per slot at 2.4 GHz (nominal)
  printf("%s %s x%2d per 32 matrix cycles | 1 wave/SIMD: M %6.1f V %6.1f both-in-one-wave %6.1f (max %6.1f sum %6.1f) | 2 waves/SIMD: MM %6.1f VV %6.1f BB %6.1f | "
         "cross M|V %6.1f (prio M %6.1f, prio V %6.1f; max(M,V) %6.1f sum %6.1f)  [cycles/slot: M %.1f V %.1f B %.1f cross %.1f]\n",
         KIND ? "32x32x16" : "16x16x32", VK ? "v_exp" : "v_fma", NV, m1, v1, b1, m1 > v1 ? m1 : v1, m1 + v1, m2, v2, b2, x, xm, xv, m1 > v1 ? m1 : v1, m1 + v1,
         m1 * cyc, v1 * cyc, b1 * cyc, x * cyc);
}

int main() {
  suite<0, 4, 0>(); suite<0, 8, 0>(); suite<0, 12, 0>(); suite<0, 16, 0>();
  suite<1, 4, 0>(); suite<1, 8, 0>(); suite<1, 12, 0>(); suite<1, 16, 0>();
  suite<0, 8, 1>(); suite<1, 8, 1>();
  return 0;
}
