// Timing ablations and hazard-hunt switches of the tuning builds (tools/sweep.py, tools/sin_cliff.py, tools/runs_r0*/).
//
// EVERYTHING in this file changes what the kernels compute or adds waits nobody needs: a library built with any of these
// switches is for timing / fault isolation only and its results are wrong by construction.  The product build
// (tha4_amd/_build.py) never defines THA4_TUNING_BUILD, tha4_platform.h refuses the switches without it, and
// tests/test_api_surface.py checks both.  The kernels reach this file only through the THA4_HOOK_* macros of tha4_platform.h.
#pragma once
#ifndef THA4_TUNING_BUILD
#error "tha4_tuning.h is for tuning builds only (-DTHA4_TUNING_BUILD): its switches produce wrong results by construction"
#endif

// ---- timing ablations ---------------------------------------------------------------------------------------------
#ifdef THA4_ABLATE_MFMA                    // no matrix instruction (one scalar FMA keeps the operands alive)
#define THA4_HOOK_MFMA16H(a, b, c) ((c)[0] += (float)(a)[0] * (float)(b)[0], (c))
#endif
#ifdef THA4_ABLATE_SIN                     // the SIREN sine returns its argument
#define THA4_HOOK_SIN_BYPASS 1
#endif
#ifdef THA4_ABLATE_FETCH                   // no weight fetch global -> LDS in the streamed SIREN layers
#define THA4_HOOK_FETCH(stmt)
#endif
#ifdef THA4_ABLATE_BARRIER                 // no per-chunk workgroup barrier in the streamed SIREN layers
#define THA4_HOOK_CHUNK_BARRIER()
#endif
#ifdef THA4_ABLATE_ZLOAD                   // no z-tap loads of the x2 upsample
#define THA4_HOOK_ZLOAD(ptr, instead) (instead)
#endif
#ifdef THA4_ABLATE_POSEFOLD                // the pose fold of the register kernels' prologues reads 4 of its 45 weight rows (what a precomputed pose bias could save at most)
#define THA4_HOOK_POSE_ROW(k) ((k) & 3)
#endif
#ifdef THA4_ABLATE_TILE_STAGE_VALU         // conv_tile_kernel: raw bits into the window (loads + LDS writes stay, the staging VALU goes)
#define THA4_HOOK_TILE_STAGE_VALU_BYPASS 1
#endif
#ifdef THA4_ABLATE_TILE_WINDOW             // conv_tile_kernel: no window at all (the ceiling of pre-staged operands / async fills)
#define THA4_HOOK_TILE_WINDOW_BYPASS 1
#endif
#ifdef THA4_ABLATE_TILE_EPILOGUE           // conv_tile_kernel: no output stores / statistics
#define THA4_HOOK_TILE_EPILOGUE_BYPASS 1
#endif

// ---- hazard hunt (profiles/r03_sin_cliff.md) ------------------------------------------------------------------------
#ifndef THA4_EMU
#if defined(THA4_HUNT_FENCE_LGKM)          // drain one memory counter at every scheduling fence
#undef THA4_SCHED_FENCE
#define THA4_SCHED_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#elif defined(THA4_HUNT_FENCE_VM)
#undef THA4_SCHED_FENCE
#define THA4_SCHED_FENCE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#ifdef THA4_HUNT_WAIT_BEFORE_STORES        // level 2: no load outstanding when a store issues
#define THA4_HOOK_BEFORE_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#ifdef THA4_HUNT_WAIT_TOP                  // level 2: the previous strip's stores are complete before this strip's loads
#define THA4_HOOK_STRIP_TOP() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#endif
