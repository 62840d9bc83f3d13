// What bounds the weight ring of the student's streamed kernels (round-5 review, task 1)?  Two readings were on the table:
//   (A) "the L2 -> LDS stream of the chip is saturated at ~8 TB/s" (round 5),
//   (B) "one chunk in flight per CU x the LDS-DMA latency" (the review): bytes in flight, not bandwidth.
// Every workgroup (one per CU) streams the SAME L2-resident weight image (1008 KiB = level 0 of the student, REPS times) through an LDS
// ring with global_load_lds_dwordx4, exactly like gemm16_stream: the fetch of chunk c + DEPTH is issued when chunk c's barrier is
// passed, a counted s_waitcnt vmcnt in front of the barrier leaves the younger chunks in flight, and every wave "consumes" a chunk with
// the ds_read_b128 / MFMA mix of the front kernel (per 24 KiB: 8 fragment reads + 9 MFMAs per wave at 16 waves).
//   mode 0: ring, SLOTS = DEPTH + 1, one barrier per chunk           (DEPTH = 1 is the shipped two-slot ring)
//   mode 1: no consumption, no barrier: every wave keeps KEEP copies in flight  -> the ceiling of the LDS-DMA path
// Prints us per pass and GB/s per CU for each configuration.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_stream.hip -o tools/microbench/lds_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if constexpr (N == 21) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  else if constexpr (N == 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else static_assert(N < 0, "add the case");
}

// mode 0.  CHUNK_KB must be a multiple of WAVES (every wave issues CPW = CHUNK_KB / WAVES copies per chunk).
// MF: MFMAs per wave per chunk, RD: ds_read_b128 pairs (hi + lo) per wave per chunk.
template <int WAVES, int CHUNK_KB, int DEPTH, int MF, int RD>
__global__ void __launch_bounds__(WAVES * 64) ring_kernel(const char* w, int total_kb, int reps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SLOTS = DEPTH + 1, CPW = CHUNK_KB / WAVES, CHUNK = CHUNK_KB * 1024;
  static_assert(CHUNK_KB % WAVES == 0, "chunk must be a whole number of copies per wave");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nc = total_kb / CHUNK_KB;                       // chunks per pass
  const int NC = nc * reps;
  auto fetch = [&](int c) {
    const char* g = w + (size_t)(c % nc) * CHUNK;
    char* l = smem + (c % SLOTS) * CHUNK;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int pc = i * WAVES + wave;
      glds16(g + pc * 1024 + lane * 16, l + pc * 1024);
    }
  };
  f32x4 acc[4] = {};
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d);
#pragma unroll 1
  for (int c = 0; c < NC; ++c) {
    // chunk c has landed (everything older than the DEPTH - 1 younger chunks of this wave) for every wave, and every wave is done with chunk c - 1
    wait_vm<(DEPTH - 1) * CPW>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (c + DEPTH < NC) fetch(c + DEPTH);
    else {                                                   // keep the counted waits exact at the tail: dummy copies into the same slot
      fetch(c + DEPTH);
    }
    const char* s = smem + (c % SLOTS) * CHUNK + lane * 16;
    f16x8 a[2 * RD];
#pragma unroll
    for (int r = 0; r < 2 * RD; ++r) a[r] = *reinterpret_cast<const f16x8*>(s + ((wave * 2 * RD + r) % CHUNK_KB) * 1024);
#pragma unroll
    for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m % (2 * RD)], a[(m + 1) % (2 * RD)], acc[m & 3], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink) sink[blockIdx.x * WAVES * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

// mode 1: every wave streams its share with KEEP copies in flight, nothing else
template <int WAVES, int KEEP>
__global__ void __launch_bounds__(WAVES * 64) flood_kernel(const char* w, int total_kb, int reps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per_wave = total_kb / WAVES;                    // copies per wave per pass
  const int N = per_wave * reps;
  char* l = smem + wave * KEEP * 1024;
#pragma unroll 1
  for (int i = 0; i < N; i += KEEP) {
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
      const int pc = ((i + k) % per_wave) * WAVES + wave;
      glds16(w + (size_t)pc * 1024 + lane * 16, l + k * 1024);
    }
    wait_vm<KEEP / 2>();                                     // half of them stay in flight while the next batch is issued
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && lane == 0 && wave == 0) sink[blockIdx.x] = (float)smem[17];
}

static float time_launches(void (*launch)(hipStream_t), int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch(0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch(0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.0f / iters;
}

static char* g_w;
static float* g_sink;
static int g_total_kb = 1008, g_reps = 4, g_grid = 256;

template <int WAVES, int CHUNK_KB, int DEPTH, int MF, int RD>
static void run_ring(const char* label) {
  constexpr int lds = (DEPTH + 1) * CHUNK_KB * 1024;
  auto k = ring_kernel<WAVES, CHUNK_KB, DEPTH, MF, RD>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  static auto kk = k;
  kk = k;
  auto launch = [](hipStream_t s) { hipLaunchKernelGGL(kk, dim3(g_grid), dim3(WAVES * 64), (DEPTH + 1) * CHUNK_KB * 1024, s, g_w, g_total_kb - g_total_kb % CHUNK_KB, g_reps, nullptr); };
  const float us = time_launches(launch, 20);
  const double kb = (double)(g_total_kb - g_total_kb % CHUNK_KB) * g_reps;
  printf("ring  %-28s waves %2d chunk %2d KiB depth %d (ring %3d KiB) mfma/chunk/wave %2d: %8.1f us/launch  %6.2f us/pass  %6.1f GB/s/CU  %5.2f TB/s chip  %5.0f ns/chunk\n", label, WAVES,
         CHUNK_KB, DEPTH, lds / 1024, MF, us, us / g_reps, kb * 1024 / us / 1e3, kb * 1024 * g_grid / us / 1e6, us * 1e3 / (kb / CHUNK_KB));
}

template <int WAVES, int KEEP>
static void run_flood() {
  auto k = flood_kernel<WAVES, KEEP>;
  constexpr int lds = WAVES * KEEP * 1024;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  static auto kk = k;
  kk = k;
  auto launch = [](hipStream_t s) { hipLaunchKernelGGL(kk, dim3(g_grid), dim3(WAVES * 64), WAVES * KEEP * 1024, s, g_w, g_total_kb - g_total_kb % WAVES, g_reps, nullptr); };
  const float us = time_launches(launch, 20);
  const double kb = (double)(g_total_kb - g_total_kb % WAVES) * g_reps;
  printf("flood waves %2d keep %2d (%3d KiB in flight per CU): %8.1f us/launch  %6.1f GB/s/CU  %5.2f TB/s chip\n", WAVES, KEEP, WAVES * KEEP, us, kb * 1024 / us / 1e3,
         kb * 1024 * g_grid / us / 1e6);
}

int main(int argc, char** argv) {
  if (argc > 1) g_grid = atoi(argv[1]);
  hipMalloc(&g_w, 2 << 20);
  hipMemset(g_w, 0x3c, 2 << 20);
  hipMalloc(&g_sink, 1 << 22);
  printf("grid %d workgroups, %d KiB per pass, %d passes per launch\n", g_grid, g_total_kb, g_reps);
  // the ceiling of the path
  run_flood<4, 4>();
  run_flood<4, 8>();
  run_flood<4, 16>();
  run_flood<8, 4>();
  run_flood<8, 8>();
  run_flood<16, 2>();
  run_flood<16, 4>();
  run_flood<16, 8>();
  // the shipped front kernel's shape: 16 waves, 2-slot ring.  24 KiB chunks are 1.5 copies per wave, so 16- and 32-KiB chunks bracket it
  // (MFMAs per wave and chunk scaled with the chunk: 9 per 24 KiB -> 6 per 16 KiB, 12 per 32 KiB)
  run_ring<16, 16, 1, 6, 3>("2 slots (shipped form)");
  run_ring<16, 32, 1, 12, 4>("2 slots (shipped form)");
  run_ring<16, 16, 2, 6, 3>("3 slots");
  run_ring<16, 16, 3, 6, 3>("4 slots (fits the front LDS)");
  run_ring<16, 16, 4, 6, 3>("5 slots");
  run_ring<16, 16, 6, 6, 3>("7 slots");
  run_ring<16, 32, 2, 12, 4>("3 slots");
  run_ring<16, 32, 3, 12, 4>("4 slots");
  // no matrix work at all: the ring protocol alone
  run_ring<16, 16, 1, 0, 3>("2 slots, no MFMA");
  run_ring<16, 16, 3, 0, 3>("4 slots, no MFMA");
  // the register-resident form: 8 or 4 waves, LDS = ring only; per 24 KiB chunk a wave owns ALL 12 blocks x PG pixel groups: 36 (PG 1) / 72 (PG 2) MFMAs
  run_ring<8, 24, 1, 36, 4>("8 waves PG1 2 slots");
  run_ring<8, 24, 2, 36, 4>("8 waves PG1 3 slots");
  run_ring<8, 24, 4, 36, 4>("8 waves PG1 5 slots");
  run_ring<8, 24, 5, 36, 4>("8 waves PG1 6 slots");
  run_ring<8, 24, 2, 72, 4>("8 waves PG2 3 slots");
  run_ring<8, 24, 5, 72, 4>("8 waves PG2 6 slots");
  run_ring<4, 24, 2, 36, 4>("4 waves PG1 3 slots");
  run_ring<4, 24, 5, 36, 4>("4 waves PG1 6 slots");
  run_ring<4, 24, 5, 72, 4>("4 waves PG2 6 slots");
  run_ring<8, 16, 4, 24, 4>("8 waves PG1 16K 5 slots");
  run_ring<8, 16, 8, 24, 4>("8 waves PG1 16K 9 slots");
  return 0;
}
