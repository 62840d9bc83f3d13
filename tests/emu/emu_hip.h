// Fiber-based SIMT emulator for the CPU unit tests (TEST INFRASTRUCTURE ONLY - never part of the
// product library).  It lets the *unmodified* kernel source of
// talking-head-anime-4-demo_amd/csrc/siren_kernels.h be compiled as host C++ (clang++) and executed
// one workgroup at a time with HIP semantics:
//   * every thread of a workgroup is a ucontext fiber; __syncthreads() and the wave-collective
//     operations (MFMA, lane reads) are rendezvous points that yield to the round-robin scheduler;
//   * v_mfma_f32_16x16x4_f32 follows the gfx950 fragment layout (cdna_hip_programming.md §3):
//       A[i][k] from lane 16k+i, B[k][j] from lane 16k+j, D[i][j] in lane 16(i/4)+j, register i%4,
//     accumulated as a k-ordered fp32 fmaf chain (the hardware's numerics);
//   * global_load_lds_dwordx4 copies 16 B per lane to (wave-uniform LDS base + 16*lane).
// Single OS thread; launch geometry is 1-D/2-D grids of 1-D blocks.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__

struct emu_uint3 { unsigned x = 0, y = 0, z = 0; };
struct uchar4 { unsigned char x, y, z, w; };      // HIP's vector type, as far as the image kernels use it
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace emu {

inline emu_uint3 g_threadIdx, g_blockIdx;
inline dim3 g_blockDim, g_gridDim;
inline char* g_lds = nullptr;

struct WaveState {
  float a[64], b[64], c[64][4], d[64][4];
  float a8[64][8], b8[64][8];
  float sh_in[64], sh_out[64];
  int sh_src[64];
  int arrived = 0;
  unsigned gen = 0;
};

struct Fiber {
  ucontext_t ctx;
  std::unique_ptr<char[]> stack;
  bool done = false;
  emu_uint3 tid;
};

struct BlockState {
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  ucontext_t sched;
  int current = -1;
  int bar_arrived = 0;
  unsigned bar_gen = 0;
  std::function<void()> body;
};

inline BlockState* g_blk = nullptr;

inline void yield() {
  BlockState* B = g_blk;
  Fiber& f = B->fibers[B->current];
  swapcontext(&f.ctx, &B->sched);
}

inline void syncthreads() {
  BlockState* B = g_blk;
  const unsigned my = B->bar_gen;
  if (++B->bar_arrived == (int)B->fibers.size()) {
    B->bar_arrived = 0;
    B->bar_gen++;
    return;
  }
  while (B->bar_gen == my) yield();
}

inline WaveState& my_wave() { return g_blk->waves[g_threadIdx.x >> 6]; }

template <class Compute>
inline void wave_rendezvous(WaveState& W, Compute compute_all) {
  const unsigned my = W.gen;
  if (++W.arrived == 64) {
    compute_all();
    W.arrived = 0;
    W.gen++;
    return;
  }
  while (W.gen == my) yield();
}

inline void mfma_f32_16x16x4(float a, float b, float (&cd)[4]) {
  WaveState& W = my_wave();
  const int lane = g_threadIdx.x & 63;
  W.a[lane] = a;
  W.b[lane] = b;
  for (int r = 0; r < 4; ++r) W.c[lane][r] = cd[r];
  wave_rendezvous(W, [&W]() {
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        const int dl = 16 * (i / 4) + j, dr = i % 4;
        float acc = W.c[dl][dr];
        for (int k = 0; k < 4; ++k) acc = std::fmaf(W.a[16 * k + i], W.b[16 * k + j], acc);
        W.d[dl][dr] = acc;
      }
  });
  for (int r = 0; r < 4; ++r) cd[r] = W.d[lane][r];
}

// v_mfma_f32_16x16x32_f16: A[i][k] = lane 16*(k/8)+i element k%8, B[k][j] = lane 16*(k/8)+j element k%8,
// D[i][j] in lane 16*(i/4)+j register i%4; fp16 products are exact in fp32, accumulated in fp32.
inline void mfma_f32_16x16x32(const float (&a)[8], const float (&b)[8], float (&cd)[4]) {
  WaveState& W = my_wave();
  const int lane = g_threadIdx.x & 63;
  for (int j = 0; j < 8; ++j) { W.a8[lane][j] = a[j]; W.b8[lane][j] = b[j]; }
  for (int r = 0; r < 4; ++r) W.c[lane][r] = cd[r];
  wave_rendezvous(W, [&W]() {
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        const int dl = 16 * (i / 4) + j, dr = i % 4;
        float acc = W.c[dl][dr];
        for (int k = 0; k < 32; ++k) acc = std::fmaf(W.a8[16 * (k / 8) + i][k % 8], W.b8[16 * (k / 8) + j][k % 8], acc);
        W.d[dl][dr] = acc;
      }
  });
  for (int r = 0; r < 4; ++r) cd[r] = W.d[lane][r];
}

inline float shfl(float v, int src) {
  WaveState& W = my_wave();
  const int lane = g_threadIdx.x & 63;
  W.sh_in[lane] = v;
  W.sh_src[lane] = src & 63;
  wave_rendezvous(W, [&W]() {
    for (int l = 0; l < 64; ++l) W.sh_out[l] = W.sh_in[W.sh_src[l]];
  });
  return W.sh_out[lane];
}

inline void glds16(const void* gsrc_lane, void* lds_base_uniform) {
  const int lane = g_threadIdx.x & 63;
  std::memcpy(static_cast<char*>(lds_base_uniform) + 16 * lane, gsrc_lane, 16);
}

inline void fiber_entry() {
  BlockState* B = g_blk;
  B->body();
  B->fibers[B->current].done = true;
  swapcontext(&B->fibers[B->current].ctx, &B->sched);
}

// Run ONE workgroup (blockIdx = bid) of `kernel(args...)` with `threads` threads and `lds_bytes` of LDS.
template <class K, class... Args>
inline void run_block(K kernel, dim3 grid, dim3 bid, int threads, size_t lds_bytes, Args... args) {
  constexpr size_t kStack = 256 * 1024;
  BlockState B;
  B.fibers.resize(threads);
  B.waves.resize((threads + 63) / 64);
  std::vector<char> lds(lds_bytes + 64, 0);
  char* lds_aligned = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(lds.data()) + 63) & ~uintptr_t(63));
  B.body = [&]() { kernel(args...); };
  g_blk = &B;
  g_lds = lds_aligned;
  g_gridDim = grid;
  g_blockDim = dim3(threads);
  g_blockIdx.x = bid.x; g_blockIdx.y = bid.y; g_blockIdx.z = bid.z;
  for (int t = 0; t < threads; ++t) {
    Fiber& f = B.fibers[t];
    f.stack.reset(new char[kStack]);
    f.tid.x = t;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.get();
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  int remaining = threads;
  while (remaining > 0) {
    for (int t = 0; t < threads; ++t) {
      Fiber& f = B.fibers[t];
      if (f.done) continue;
      B.current = t;
      g_threadIdx = f.tid;
      swapcontext(&B.sched, &f.ctx);
      if (f.done) --remaining;
    }
  }
  g_blk = nullptr;
  g_lds = nullptr;
}

}  // namespace emu

#define threadIdx emu::g_threadIdx
#define blockIdx emu::g_blockIdx
#define blockDim emu::g_blockDim
#define gridDim emu::g_gridDim
inline void __syncthreads() { emu::syncthreads(); }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
using std::min;
using std::max;
