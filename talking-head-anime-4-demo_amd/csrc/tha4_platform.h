// Platform glue for the THA4 HIP kernels.
//
// Product build: hipcc --offload-arch=gfx950 (CDNA4 only; no other backend exists).
// THA4_EMU build: the SAME kernel source compiled as host C++ against tests/emu/emu_hip.h,
// a fiber-based SIMT emulator used by the CPU unit tests only (never shipped, never linked
// into libtha4_hip.so).
#pragma once

#ifdef THA4_EMU
#include "emu_hip.h"
#define THA4_DYN_LDS(name) char* name = emu::g_lds
#else
#include <hip/hip_runtime.h>
#define THA4_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

#include <type_traits>

#define THA4_DEV __device__ __forceinline__

// instruction-scheduling fence: nothing moves across it (software pipelining by hand, and caps on loads in flight)
#if !defined(THA4_EMU) && !defined(THA4_NO_PIPELINE)
#define THA4_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define THA4_SCHED_FENCE()
#endif

// Timing ablations and hazard-hunt switches (every one of them produces WRONG results by construction) live in tha4_tuning.h and
// exist in tuning builds only (-DTHA4_TUNING_BUILD, tools/sweep.py): the product build refuses them.  The kernels call the
// THA4_HOOK_* macros below; without the tuning header each hook is the plain operation.
#if defined(THA4_TUNING_BUILD)
#include "tha4_tuning.h"
#elif defined(THA4_ABLATE_MFMA) || defined(THA4_ABLATE_SIN) || defined(THA4_ABLATE_FETCH) || defined(THA4_ABLATE_BARRIER) ||       \
    defined(THA4_ABLATE_ZLOAD) || defined(THA4_ABLATE_TILE_STAGE_VALU) || defined(THA4_ABLATE_TILE_WINDOW) ||                       \
    defined(THA4_ABLATE_TILE_EPILOGUE) || defined(THA4_HUNT_FENCE_LGKM) || defined(THA4_HUNT_FENCE_VM) ||                            \
    defined(THA4_HUNT_WAIT_BEFORE_STORES) || defined(THA4_HUNT_WAIT_TOP)
#error "THA4_ABLATE_* / THA4_HUNT_* switches produce wrong results by construction: tuning builds only (-DTHA4_TUNING_BUILD, csrc/tha4_tuning.h)"
#endif
#ifndef THA4_HOOK_MFMA16H                  // the matrix instruction of the fp16 hi/lo contraction
#define THA4_HOOK_MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif
#ifndef THA4_HOOK_SIN_BYPASS               // 1: the SIREN sine returns its argument
#define THA4_HOOK_SIN_BYPASS 0
#endif
#ifndef THA4_HOOK_FETCH                    // one weight piece global -> LDS
#define THA4_HOOK_FETCH(stmt) stmt
#endif
#ifndef THA4_HOOK_CHUNK_BARRIER            // the per-chunk workgroup barrier of the streamed SIREN layers
#define THA4_HOOK_CHUNK_BARRIER() __syncthreads()
#endif
#ifndef THA4_HOOK_POSE_ROW                 // row of the pose-weight matrix the fold reads for pose parameter k
#define THA4_HOOK_POSE_ROW(k) (k)
#endif
#ifndef THA4_HOOK_ZLOAD                    // one z tap of the x2 upsample
#define THA4_HOOK_ZLOAD(ptr, instead) (*reinterpret_cast<const f32x4*>(ptr))
#endif
#ifndef THA4_HOOK_BEFORE_STORES
#define THA4_HOOK_BEFORE_STORES()
#endif
#ifndef THA4_HOOK_STRIP_TOP
#define THA4_HOOK_STRIP_TOP()
#endif
#ifndef THA4_HOOK_TILE_STAGE_VALU_BYPASS   // 1: conv_tile_kernel writes the raw loaded bits into its window (no normalise / activate / split)
#define THA4_HOOK_TILE_STAGE_VALU_BYPASS 0
#endif
#ifndef THA4_HOOK_TILE_WINDOW_BYPASS       // 1: conv_tile_kernel neither loads nor writes its window (weights + MFMA + barriers only)
#define THA4_HOOK_TILE_WINDOW_BYPASS 0
#endif
#ifndef THA4_HOOK_TILE_EPILOGUE_BYPASS     // 1: conv_tile_kernel stores nothing
#define THA4_HOOK_TILE_EPILOGUE_BYPASS 0
#endif

// wave-level ordering point for wave-PRIVATE LDS traffic (one lane writes, another lane of the same wave reads): the
// hardware executes a wave's DS instructions in program order, so nothing is needed but a fence for the compiler's
// scheduler; the fiber emulator runs lanes one after the other and needs a real rendezvous
#ifdef THA4_EMU
#define THA4_WAVE_SYNC() ((void)emu::shfl(0.0f, 0))
#else
#define THA4_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

// Workgroup barrier that lets the newest `keep` global_load_lds / global loads of this wave stay in flight (counted
// s_waitcnt vmcnt): `__syncthreads()` drains every LDS-DMA before its s_barrier, which defeats a weight ring deeper than two
// slots.  Everything OLDER than the newest `keep` VMEM operations of the wave is complete when the barrier is passed; LDS
// traffic of the wave is complete (lgkmcnt(0)).  One asm block with a memory clobber: nothing is moved across it.
#ifdef THA4_EMU
#define THA4_BARRIER_KEEP(keep) __syncthreads()
#else
#define THA4_BARRIER_KEEP_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
#define THA4_BARRIER_KEEP(keep)                                                                                   \
  do {                                                                                                            \
    switch (keep) {                                                                                               \
      THA4_BARRIER_KEEP_CASE(1) THA4_BARRIER_KEEP_CASE(2) THA4_BARRIER_KEEP_CASE(3) THA4_BARRIER_KEEP_CASE(4)     \
      THA4_BARRIER_KEEP_CASE(5) THA4_BARRIER_KEEP_CASE(6) THA4_BARRIER_KEEP_CASE(7) THA4_BARRIER_KEEP_CASE(8)     \
      THA4_BARRIER_KEEP_CASE(9) THA4_BARRIER_KEEP_CASE(10) THA4_BARRIER_KEEP_CASE(11) THA4_BARRIER_KEEP_CASE(12)  \
      THA4_BARRIER_KEEP_CASE(13) THA4_BARRIER_KEEP_CASE(14) THA4_BARRIER_KEEP_CASE(15) THA4_BARRIER_KEEP_CASE(16) \
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;                  \
    }                                                                                                             \
  } while (0)
#endif

// The same with the wave's own LDS reads left in flight (no lgkmcnt wait): for rings whose slots are only rewritten a whole chunk after their last
// read was CONSUMED (an MFMA needed it), so that outstanding reads at the barrier can only belong to chunks nobody overwrites yet.
#ifdef THA4_EMU
#define THA4_BARRIER_KEEP_VM(keep) __syncthreads()
#else
#define THA4_BARRIER_KEEP_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory"); break;
#define THA4_BARRIER_KEEP_VM(keep)                                                                                            \
  do {                                                                                                                        \
    switch (keep) {                                                                                                           \
      THA4_BARRIER_KEEP_VM_CASE(1) THA4_BARRIER_KEEP_VM_CASE(2) THA4_BARRIER_KEEP_VM_CASE(3) THA4_BARRIER_KEEP_VM_CASE(4)     \
      THA4_BARRIER_KEEP_VM_CASE(5) THA4_BARRIER_KEEP_VM_CASE(6) THA4_BARRIER_KEEP_VM_CASE(7) THA4_BARRIER_KEEP_VM_CASE(8)     \
      THA4_BARRIER_KEEP_VM_CASE(9) THA4_BARRIER_KEEP_VM_CASE(10) THA4_BARRIER_KEEP_VM_CASE(11) THA4_BARRIER_KEEP_VM_CASE(12)  \
      THA4_BARRIER_KEEP_VM_CASE(13) THA4_BARRIER_KEEP_VM_CASE(14) THA4_BARRIER_KEEP_VM_CASE(15) THA4_BARRIER_KEEP_VM_CASE(16) \
      default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;                                          \
    }                                                                                                                         \
  } while (0)
#endif

// Workgroup barrier for LDS traffic only: the wave's LDS operations are complete, its global loads stay in flight
// (`__syncthreads()` carries a release fence that drains vmcnt).  Global stores before it are NOT ordered by it.
#ifdef THA4_EMU
#define THA4_BARRIER_LDS() __syncthreads()
#else
#define THA4_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// Issue priority of a wave's current phase (round 3, tools/microbench/mfma_valu_overlap2.hip): two waves share a SIMD; by default
// a wave issuing back-to-back MFMAs starves its partner's VALU instructions and the two phases take the SUM of their times,
// with the VALU-phase wave at s_setprio 1 they overlap (cross-wave 16x16x32: 1059 -> 766 us for 494 us of MFMA + 565 us of FMA).
// THA4_PRIO_VALU() opens a VALU-heavy phase (sine / staging epilogues), THA4_PRIO_MFMA() a matrix phase.
#ifndef THA4_PHASE_PRIO
#define THA4_PHASE_PRIO 1      // round 4: on (it is what makes the two workgroups per CU of conv_tile_kernel<..., NW = 4> overlap; neutral elsewhere)
#endif
#if !defined(THA4_EMU) && THA4_PHASE_PRIO
#define THA4_PRIO_VALU() __builtin_amdgcn_s_setprio(1)
#define THA4_PRIO_MFMA() __builtin_amdgcn_s_setprio(0)
#else
#define THA4_PRIO_VALU()
#define THA4_PRIO_MFMA()
#endif

namespace tha4 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#ifndef THA4_EMU
// 16-byte async global -> LDS copy (global_load_lds_dwordx4).  The LDS destination is
// wave-uniform base + lane*16; the global source is per lane.
THA4_DEV void glds16(const void* gsrc_lane, void* lds_base_uniform) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc_lane,
      (__attribute__((address_space(3))) void*)lds_base_uniform, 16, 0, 0);
}
THA4_DEV f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_f16: A[i][k] = lane 16*(k/8)+i element k%8, B[k][j] = lane 16*(k/8)+j element k%8
THA4_DEV f32x4 mfma16h(f16x8 a, f16x8 b, f32x4 c) {
  return THA4_HOOK_MFMA16H(a, b, c);
}
THA4_DEV int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// the next value of a workgroup-shared LDS counter, fetched by lane 0 and broadcast to the wave
THA4_DEV int wave_take_ticket(int* lds_counter, int lane) {
  int v = 0;
  if (lane == 0) v = atomicAdd(lds_counter, 1);
  return __builtin_amdgcn_readfirstlane(v);
}
THA4_DEV float lane_read(float v, int src_lane) { return __shfl(v, src_lane, 64); }
// device-scope 64-bit integer add without a returned value (global_atomic_add_x2): integer sums are independent of the order of arrival
THA4_DEV void atomic_add_i64(long long* p, long long v) {
  __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15 = the 16 pixels a lane group holds of one channel), left in every lane of
// the row: four v_add_f32 with a row_ror DPP operand (rotations by 8, 4, 2, 1 - a fixed, direction-independent order).  Round 4:
// the per-channel statistics of the convolution epilogues spent 8 ds_bpermute_b32 per value (128 LDS round trips per workgroup
// tile); DPP needs no LDS at all.
template <int N>
THA4_DEV float row16_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
THA4_DEV float row16_sum(float v, int) {
  v += row16_ror<8>(v);
  v += row16_ror<4>(v);
  v += row16_ror<2>(v);
  v += row16_ror<1>(v);
  return v;
}
// value held by lane `src_lane` (wave-uniform index) as a scalar: v_readlane_b32 - a per-tap table kept in one VGPR
// (lane t holds entry t) replaces a scalar memory load + s_waitcnt lgkmcnt(0) inside the MFMA loops
THA4_DEV int lane_pick(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
#else
THA4_DEV void atomic_add_i64(long long* p, long long v) { *p += v; }      // (the emulator runs the workgroups one after the other)
THA4_DEV void glds16(const void* gsrc_lane, void* lds_base_uniform) { emu::glds16(gsrc_lane, lds_base_uniform); }
THA4_DEV f32x4 mfma16(float a, float b, f32x4 c) {
  float r[4] = {c[0], c[1], c[2], c[3]};
  emu::mfma_f32_16x16x4(a, b, r);
  f32x4 o = {r[0], r[1], r[2], r[3]};
  return o;
}
THA4_DEV f32x4 mfma16h(f16x8 a, f16x8 b, f32x4 c) {
  float af[8], bf[8], r[4] = {c[0], c[1], c[2], c[3]};
  for (int j = 0; j < 8; ++j) { af[j] = (float)a[j]; bf[j] = (float)b[j]; }
  emu::mfma_f32_16x16x32(af, bf, r);
  return f32x4{r[0], r[1], r[2], r[3]};
}
THA4_DEV int uniform_i32(int v) { return v; }
THA4_DEV int wave_take_ticket(int* lds_counter, int lane) {      // fibers run one at a time: a plain read-modify-write
  float v = 0.f;
  if (lane == 0) { v = (float)*lds_counter; *lds_counter += 1; }
  return (int)emu::shfl(v, 0);                                   // tickets are small integers: exact in fp32
}
THA4_DEV float lane_read(float v, int src_lane) { return emu::shfl(v, src_lane); }
THA4_DEV float row16_sum(float v, int lane) {                   // the device's order: rotations by 8, 4, 2, 1 inside the 16-lane row
  for (int m = 8; m >= 1; m >>= 1) v += emu::shfl(v, (lane & ~15) | ((lane + m) & 15));
  return v;
}
THA4_DEV int lane_pick(int v, int src_lane) { return (int)emu::shfl((float)v, src_lane); }       // |v| < 2^24: exact in fp32
#endif

// Kernel arguments: the argument block of a launch sits in device memory the host has just written - nothing on the chip holds it.  The compiler
// reads a by-value argument struct where it first needs each field: fifteen `s_load ... s_waitcnt lgkmcnt(0)` rounds in conv_small_kernel's prologue,
// seven of them the first touch of a 64-byte line, i.e. seven DEPENDENT cold misses (in-kernel stamps: 1.4 - 3.2 us between a workgroup's
// entry and its first vector-memory request; profiles/r04_full_conv_tile_reading.md, section 10).  warm_kernarg() touches every line of the block at the
// top of the kernel - one scalar load per line, all in flight together, one wait: ONE miss latency, after which every argument read hits the scalar
// cache.  No effect on results.  THA4_NO_KERNARG_WARM builds without it (A/B).
template <int BYTES>
THA4_DEV void warm_kernarg() {
#if !defined(THA4_EMU) && !defined(THA4_NO_KERNARG_WARM)
  typedef const int __attribute__((address_space(4))) * kargp;
  const kargp kp = (kargp)__builtin_amdgcn_kernarg_segment_ptr();
  int t = kp[(BYTES - 1) / 4];
#pragma unroll
  for (int i = 0; i < (BYTES + 63) / 64; ++i) t |= kp[i * 16];
  asm volatile("" ::"s"(t));
#endif
}

// fp16 hi/lo split of a PAIR of fp32 values: v = hi + lo with hi = fp16(v), lo = fp16(v - hi).  One v_cvt_pk_f16_f32 for both hi halves
// and one v_fma_mix per lo half (the f16 -> f32 conversion of hi rides inside the FMA): 3-4 instructions per pair where the plain
// expression `lo = fp16(v - float(hi))` compiles to 8 (two v_cvt_f16_f32, two v_cvt_f32_f16, two v_sub_f32, two packing converts).  The
// multiplier -1 has to come from a register the optimiser cannot see through - fma(hi, -1, v) is otherwise folded to v - hi and the mix
// form is lost.  Same bits either way: v - hi is exact in fp32 and rounded to fp16 once.
THA4_DEV float split_minus_one() {
  float m1 = -1.0f;
#ifndef THA4_EMU
  asm("" : "+s"(m1));
#endif
  return m1;
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
THA4_DEV void split_pair(float a, float b, float m1, _Float16& ha, _Float16& hb, _Float16& la, _Float16& lb) {
#ifdef THA4_PLAIN_SPLIT   // A/B build: the expression rounds 1-2 and most of round 3 used (8 instructions per pair)
  (void)m1;
  ha = (_Float16)a;
  hb = (_Float16)b;
  la = (_Float16)__builtin_fmaf(-1.0f, (float)ha, a);
  lb = (_Float16)__builtin_fmaf(-1.0f, (float)hb, b);
#else
  const f16x2_t h = __builtin_convertvector((f32x2_t){a, b}, f16x2_t);
  ha = h[0];
  hb = h[1];
  la = (_Float16)__builtin_fmaf((float)h[0], m1, a);
  lb = (_Float16)__builtin_fmaf((float)h[1], m1, b);
#endif
}

}  // namespace tha4

// ---- hand-counted LDS fragment reads --------------------------------------------------------------------------------------------------
// With an LDS-DMA (global_load_lds) in flight, or behind an inline-asm barrier, the compiler's s_waitcnt insertion waits for EVERY outstanding LDS
// read in front of the first use of ANY of them (lgkmcnt(0) where lgkmcnt(8) was meant; /tmp-sized reproducer in profiles/r06_student_b1_reading.md):
// a register-level look-ahead of the A fragments then hides nothing.  lds_read16 issues the ds_read_b128 from inline asm - the compiler neither
// tracks nor waits for it - and lds_wait<N>() is the explicit `s_waitcnt lgkmcnt(N)`: LDS operations of a wave return in order, so "at most N newer
// reads outstanding" means every older one has landed.  OFF must be a constant expression (static_for below).  Compiler-issued LDS operations
// around these stay correct: their own waits can only become stricter by the extra outstanding reads.
#ifdef THA4_EMU
template <int OFF>
THA4_DEV void lds_read16(tha4::f16x8& dst, const char* base) { dst = *reinterpret_cast<const tha4::f16x8*>(base + OFF); }
template <int N>
THA4_DEV void lds_wait(tha4::f16x8&, tha4::f16x8&) {}
template <int N>
THA4_DEV void lds_wait(tha4::f16x8&, tha4::f16x8&, tha4::f16x8&, tha4::f16x8&) {}
template <int N>
THA4_DEV void lds_wait(tha4::f16x8&, tha4::f16x8&, tha4::f16x8&, tha4::f16x8&, tha4::f16x8&, tha4::f16x8&) {}
#else
template <int OFF>
THA4_DEV void lds_read16(tha4::f16x8& dst, const char* base) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 takes a 16-bit offset");
  const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)base;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// (the fragments the wait is FOR are in/out operands: whatever consumes them is ordered behind the wait by data dependence, not by the goodwill of a scheduler)
template <int N>
THA4_DEV void lds_wait(tha4::f16x8& a, tha4::f16x8& b) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
THA4_DEV void lds_wait(tha4::f16x8& a, tha4::f16x8& b, tha4::f16x8& c, tha4::f16x8& d) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
THA4_DEV void lds_wait(tha4::f16x8& a, tha4::f16x8& b, tha4::f16x8& c, tha4::f16x8& d, tha4::f16x8& e, tha4::f16x8& f) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}
#endif
// 16-byte WRITE-THROUGH store (global_store_dwordx4 ... sc1): the line leaves the XCD's L2 as it is written instead of waiting dirty for the kernel
// boundary.  For hand-off images whose reader is the NEXT kernel on other XCDs (z1 / z2 of the student, 12.6 / 25.2 MB per frame): a plain store leaves
// them dirty and the boundary then costs bytes / ~6 TB/s on top of its ~1.5 us (MI355X_MICROARCH.md, "boundary" and "publish-large" rows).
#ifndef THA4_Z_WRITE_THROUGH
#define THA4_Z_WRITE_THROUGH 1
#endif
THA4_DEV void store16_wt(void* p, const tha4::f32x4& v) {
#if defined(THA4_EMU) || !THA4_Z_WRITE_THROUGH
  *reinterpret_cast<tha4::f32x4*>(p) = v;
#else
  // (s_nop 1: a store of more than 8 bytes reads its data registers a few cycles after issue - the compiler pads its own stores against a following
  //  overwrite of those registers, but cannot see this one)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}
// The same switch for the full model's large feature maps (conv_tile_kernel's output stores, 17 - 34 MB per launch at 256x256 / 512x512): the
// norm_finalize launches behind them wait 10 - 12 us for the L2 write-back of those lines at the kernel boundary (profiles/r05_full_b1_reading.md).
// Same-box A/B (tools/runs_r06/gpu_r06_c15.sh, two rounds): steady 184.6 -> 187.3 frames/s, cold 161.4 -> 163.5, batch 8 321.7 -> 324.3: on.
#ifndef THA4_TILE_OUT_WT
#define THA4_TILE_OUT_WT 1
#endif
#ifndef THA4_POINT_OUT_WT
#define THA4_POINT_OUT_WT 1      // the same for conv_point_kernel's outputs (1x1 convolutions on 64x64 .. 512x512 maps): +0.3 % steady on both rounds of gpu_r06_c16.sh
#endif
template <bool WT = (THA4_TILE_OUT_WT != 0)>
THA4_DEV void store16_out(float* p, const tha4::f32x4& v) {
#if defined(THA4_EMU)
  *reinterpret_cast<tha4::f32x4*>(p) = v;
#else
  if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *reinterpret_cast<tha4::f32x4*>(p) = v;
#endif
}
// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N - 1 (the index is a constant expression inside f)
template <int I, int N, class F>
THA4_DEV void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}
