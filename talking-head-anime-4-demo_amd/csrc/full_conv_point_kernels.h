// 1x1 ("pointwise") convolution on maps above 32x32: the U-Net skip / projection convolutions of the full THA4 system.
//
// Round 1-2 ran these on conv_mfma_kernel (exact-fp32 v_mfma_f32_16x16x4_f32, one 16-channel operand step in flight): its
// operand loads sit under control flow (source kind, optional scale/shift, padding), so the compiler can only wait with
// vmcnt(0) and every step pays a full memory round trip - 30-50 TFLOP/s, 64x64 512->256 took 36 us for 1.3 us of matrix
// work (profiles/r02_full_b1_reading.md).  A 1x1 convolution needs no window: the B operand of a lane IS two 16-byte C16
// loads (channels 4g..4g+3 of quads 2Q and 2Q+1 at its pixel).  Here
//   * a workgroup = 4 waves x PG pixel groups (16 consecutive pixels of the flattened map) x TMB output blocks;
//   * the K loop is STRAIGHT-LINE: no memory operation under control flow, so every wait is a counted vmcnt.  Per chunk of
//     D = 4 K groups (128 channels): one LDS-only barrier (THA4_BARRIER_LDS: s_waitcnt lgkmcnt(0) + s_barrier - the
//     activation loads in flight are not drained), the loads of the NEXT chunk's weight pieces into registers, D steps of
//     { normalise + activate + split stage d into fp16 hi/lo, re-request stage d for the next chunk into the same
//     registers, 3*TMB*PG v_mfma_f32_16x16x32_f16 }, then the weight registers go to the other ring slot.  Past the last
//     K group the loads re-read the last group (clamped addresses) and the operand is zeroed by a select;
//   * per-channel scale/shift come from an LDS table filled in the prologue - copied from the vectors norm_finalize_kernel
//     wrote, or reduced from the producer's per-tile moments (fused_norm_table) - while the first loads are in flight;
//   * 1-D grid: all output-channel tiles of one pixel tile run on ONE XCD (its L2 serves the re-reads of the activations).
// Same weight image (pack_conv_weight16, one tap) and the same fp16 hi/lo numerics as conv_tile_kernel.
#pragma once
#include "full_conv16_kernels.h"

namespace tha4 {

constexpr int kPointWaves = 4;
constexpr int kPointThreads = kPointWaves * 64;
constexpr int kPointD = 4;             // K groups per streamed weight chunk = activation stages in flight per wave

constexpr size_t point_slot_bytes(int TMB) { return (size_t)kPointD * TMB * 2048; }
// dynamic LDS: [scale | shift table] [2 ring slots] [per-wave moments]
inline size_t point_lds_bytes(int TMB, int cbtot) {
  return (((size_t)2 * cbtot * 16 * sizeof(float) + 127) & ~(size_t)127) + 2 * point_slot_bytes(TMB) + (size_t)kPointWaves * TMB * 16 * 2 * sizeof(float);
}

inline bool point_act_supported(int act) { return act == ACT_NONE || act == ACT_RELU || act == ACT_SILU; }

template <int TMB, int PG>
__global__ void __launch_bounds__(kPointThreads) conv_point_kernel(ConvArgs a) {
  warm_kernarg<(int)sizeof(ConvArgs)>();
  constexpr int D = kPointD;
  constexpr int SLOT = D * TMB * 2048;
  constexpr int GL = SLOT / 1024 / kPointWaves;            // global_load_lds instructions per wave and chunk
  THA4_DYN_LDS(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform_i32(tid >> 6);
  const int p = lane & 15, g4 = (lane >> 4) * 4;
  const float m1 = split_minus_one();
  const int px = a.tile_h * a.tile_w;                      // input = output pixels per frame
  constexpr int WGPX = kPointWaves * PG * 16;
  const int tiles_per_frame = (px + WGPX - 1) / WGPX;
  const int mtiles = a.nb / TMB;
  const int T = a.batch * tiles_per_frame;
  int tl, mtile;
  if ((T & 7) == 0) {                                      // workgroups are dealt round-robin to the 8 XCDs
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    tl = (slot / mtiles) * 8 + xcd;
    mtile = slot % mtiles;
  } else {
    tl = blockIdx.x / mtiles;
    mtile = blockIdx.x % mtiles;
  }
  const int n = tl / tiles_per_frame;
  const int tile = tl % tiles_per_frame;

  const int cb0 = a.src[0].cb, cb1 = a.nsrc > 1 ? a.src[1].cb : 0;
  const int cbtot = cb0 + cb1;
  const int NQ = (cbtot + 1) >> 1;                         // 32-channel K groups
  const int nchunks = (NQ + D - 1) / D;
  const int ctot = cbtot * 16;
  float* tab_sc = reinterpret_cast<float*>(smem);
  float* tab_sh = tab_sc + ctot;
  char* ring = smem + ((2 * ctot * 4 + 127) & ~127);
  float* red = reinterpret_cast<float*>(ring + 2 * SLOT);  // [wave][TMB*16][2]
  const char* gw = reinterpret_cast<const char*>(a.w16) + (size_t)mtile * NQ * TMB * 2048 + lane * 16;
  const int kb_last = NQ * TMB * 2 - 1;                    // last 1 KiB half-piece of this output tile's weights

  // weight chunk c: global -> registers now, registers -> ring slot at the end of the previous chunk's steps (plain loads and
  // ds_writes, not global_load_lds: the compiler cannot tell an LDS read from an LDS-DMA write in flight and drains the DMA
  // - vmcnt(0) - in front of every LDS read that follows one).  Past the end: re-reads of the last piece, never multiplied
  struct WRegs { f32x4 v[GL]; };
  auto wload = [&](int c, WRegs& w) {
#pragma unroll
    for (int k = 0; k < GL; ++k) {
      const int kb = min(c * (SLOT / 1024) + wave + kPointWaves * k, kb_last);
      w.v[k] = *reinterpret_cast<const f32x4*>(gw + (size_t)kb * 1024);
    }
  };
  auto wstore = [&](int slot, const WRegs& w) {
#pragma unroll
    for (int k = 0; k < GL; ++k) *reinterpret_cast<f32x4*>(ring + slot * SLOT + (wave + kPointWaves * k) * 1024 + lane * 16) = w.v[k];
  };

  int pix[PG];
  bool inside[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    const int i = ((tile * kPointWaves + wave) * PG + pg) * 16 + p;
    inside[pg] = i < px;
    pix[pg] = min(i, px - 1);
  }
  const float* d0 = a.src[0].data + (size_t)n * cb0 * px * 16 + g4;
  const float* d1 = a.nsrc > 1 ? a.src[1].data + (size_t)n * cb1 * px * 16 + g4 : d0;
  const int act0 = a.src[0].act, act1 = a.nsrc > 1 ? a.src[1].act : ACT_NONE;
  auto quad_ptr = [&](int q) -> const float* {
    q = min(q, cbtot - 1);                                 // phantom quad of an odd block count: its weights are zero
    return q >= cb0 ? d1 + (size_t)(q - cb0) * px * 16 : d0 + (size_t)q * px * 16;
  };
  struct Stage { f32x4 va[PG], vb[PG]; };
  auto request = [&](int Q, Stage& s) {
    Q = min(Q, NQ - 1);
    const float* pa = quad_ptr(2 * Q);
    const float* pb = quad_ptr(2 * Q + 1);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      s.va[pg] = *reinterpret_cast<const f32x4*>(pa + (size_t)pix[pg] * 16);
      s.vb[pg] = *reinterpret_cast<const f32x4*>(pb + (size_t)pix[pg] * 16);
    }
  };

  f32x4 acc[TMB][PG];
#pragma unroll
  for (int b = 0; b < TMB; ++b)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) acc[b][pg] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand of K group Q from its raw stage: scale/shift, activation, zero past the last group, fp16 hi/lo split
  struct Frag { f16x8 h[PG], l[PG]; };
  auto prepare = [&](int Q, const Stage& cur) -> Frag {
    const bool valid = Q < NQ;
    const int qa = min(2 * Q, cbtot - 1), qb = min(2 * Q + 1, cbtot - 1);
    const f32x4 sca = *reinterpret_cast<const f32x4*>(tab_sc + qa * 16 + g4), sha = *reinterpret_cast<const f32x4*>(tab_sh + qa * 16 + g4);
    const f32x4 scb = *reinterpret_cast<const f32x4*>(tab_sc + qb * 16 + g4), shb = *reinterpret_cast<const f32x4*>(tab_sh + qb * 16 + g4);
    const int acta = qa >= cb0 ? act1 : act0, actb = qb >= cb0 ? act1 : act0;
    Frag f;
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      const f32x4 xa = apply_act4(cur.va[pg], sca, sha, acta, !valid);
      const f32x4 xb = apply_act4(cur.vb[pg], scb, shb, actb, !valid);
      float ua[4], ub[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { ua[j] = xa[j]; ub[j] = xb[j]; }
      _Float16 h[8], l[8];
      split_pair(ua[0], ua[1], m1, h[0], h[1], l[0], l[1]);
      split_pair(ua[2], ua[3], m1, h[2], h[3], l[2], l[3]);
      split_pair(ub[0], ub[1], m1, h[4], h[5], l[4], l[5]);
      split_pair(ub[2], ub[3], m1, h[6], h[7], l[6], l[7]);
#pragma unroll
      for (int j = 0; j < 8; ++j) { f.h[pg][j] = h[j]; f.l[pg][j] = l[j]; }
    }
    return f;
  };
  auto multiply = [&](int d, int slot, const Frag& f) {
    const char* wp = ring + slot * SLOT + d * TMB * 2048 + lane * 16;
#pragma unroll
    for (int b = 0; b < TMB; ++b) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(wp + b * 2048);
      const f16x8 al = *reinterpret_cast<const f16x8*>(wp + b * 2048 + 1024);
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) {
        acc[b][pg] = mfma16h(ah, f.h[pg], acc[b][pg]);
        acc[b][pg] = mfma16h(ah, f.l[pg], acc[b][pg]);
        acc[b][pg] = mfma16h(al, f.h[pg], acc[b][pg]);
      }
    }
  };

  // ---- prologue: first weight chunk and the first D activation stages go out before the table is built -------------
  WRegs wr;
  wload(0, wr);
  Stage st[D];
#pragma unroll
  for (int d = 0; d < D; ++d) request(d, st[d]);
  if (a.fnorm.enabled) {
    fused_norm_table(a, n, tid, kPointThreads, tab_sc, tab_sh, reinterpret_cast<double*>(ring + SLOT));   // scratch: ring slot 1, not in use yet
  } else {
    for (int c = tid; c < ctot; c += kPointThreads) {
      const int s = c >= cb0 * 16 ? 1 : 0;
      const ConvSrc& S = a.src[s];
      const int cl = c - (s ? cb0 * 16 : 0);
      tab_sc[c] = S.scale ? S.scale[(size_t)n * S.cb * 16 + cl] : 1.0f;
      tab_sh[c] = S.shift ? S.shift[(size_t)n * S.cb * 16 + cl] : 0.0f;
    }
  }
  wstore(0, wr);

  // ---- K loop: one LDS-only barrier per chunk of D K groups, no memory operation under control flow ------------------
  for (int c = 0; c < nchunks; ++c) {
    // chunk c is in its ring slot for every wave, and every wave is done reading the other slot (chunk c - 1); global loads
    // in flight (the activation stages) are NOT waited for
    THA4_BARRIER_LDS();
    wload(c + 1, wr);                                      // unconditional (clamped)
#pragma unroll
    for (int d = 0; d < D; ++d) {
      // the raw stage dies in prepare(): the re-request lands in the same registers (no copies, no drain at the back edge)
      THA4_PRIO_VALU();
      const Frag f = prepare(c * D + d, st[d]);
      THA4_PRIO_MFMA();
      THA4_SCHED_FENCE();
      request((c + 1) * D + d, st[d]);
      multiply(d, c & 1, f);
    }
    wstore((c + 1) & 1, wr);
  }

  // ---- epilogue: 1/scale, bias, residual, activation, store, deterministic per-tile moments ----------------------------
  // three passes like conv_tile_kernel's epilogue (round 4): residual loads together, values, then the stores back to back - a load between
  // two stores makes its s_waitcnt vmcnt wait for the older store as well (gfx9 counts both in vmcnt)
  float ssum[TMB][4], ssq[TMB][4];
  size_t offs[TMB][PG];
  f32x4 res[TMB][PG];
#pragma unroll
  for (int b = 0; b < TMB; ++b)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      offs[b][pg] = (((size_t)n * a.nb + mtile * TMB + b) * px + pix[pg]) * 16 + g4;
      res[b][pg] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.residual && inside[pg]) res[b][pg] = *reinterpret_cast<const f32x4*>(a.residual + offs[b][pg]);
    }
#pragma unroll
  for (int b = 0; b < TMB; ++b) {
    const int bo = mtile * TMB + b;
    f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + bo * 16 + g4);
    int codes[4] = {0, 0, 0, 0};
    if (a.act_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j) codes[j] = a.act_out[bo * 16 + g4 + j];
    }
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      f32x4 v = acc[b][pg] * a.w16_inv_scale + bias;
      if (a.residual) v = v + res[b][pg];
      if (a.act_out) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], codes[j]);
      }
      acc[b][pg] = v;
    }
  }
#pragma unroll
  for (int b = 0; b < TMB; ++b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[b][j] = 0.f; ssq[b][j] = 0.f; }
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      if (!inside[pg]) continue;                           // ragged last tile
      const f32x4 v = acc[b][pg];
      store16_out<(THA4_POINT_OUT_WT != 0)>(a.out + offs[b][pg], v);
#pragma unroll
      for (int j = 0; j < 4; ++j) { ssum[b][j] += v[j]; ssq[b][j] = fmaf(v[j], v[j], ssq[b][j]); }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int b = 0; b < TMB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = ssum[b][j], q = ssq[b][j];
        s = row16_sum(s, lane);
        q = row16_sum(q, lane);
        if (p == 0) {
          red[((wave * TMB + b) * 16 + g4 + j) * 2 + 0] = s;
          red[((wave * TMB + b) * 16 + g4 + j) * 2 + 1] = q;
        }
      }
    THA4_BARRIER_LDS();        // LDS only: the output stores above stay in flight (a full __syncthreads would wait for every store's acknowledgement)
    for (int i = tid; i < TMB * 16; i += kPointThreads) {
      float s = 0.f, q = 0.f;
      for (int w2 = 0; w2 < kPointWaves; ++w2) {
        s += red[((w2 * TMB) * 16 + i) * 2 + 0];
        q += red[((w2 * TMB) * 16 + i) * 2 + 1];
      }
      float* dst = a.stats + ((((size_t)n * a.stats_tiles + a.stats_tile0 + tile) * a.nb + mtile * TMB) * 16 + i) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

}  // namespace tha4
