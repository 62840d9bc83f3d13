// Small-map convolution (16x16 .. 64x64 maps, 128-768 input channels): ONE launch per layer.
//
// These layers are ~40 % of a frame's launches and none of them has enough output tiles to fill 256 CUs; in round 1
// they ran as  conv_tile_kernel phase 1 (K split over blockIdx.z, fp32 partials to a workspace)  ->  phase 2 (reduce +
// epilogue)  ->  norm_finalize_kernel : three launches of 5-9 us each around ~0.5 us of matrix work.  Here
//   * a workgroup owns 16*PG output positions x ONE output block (16 channels) and its 8 waves split K BETWEEN THEM:
//     wave w takes the work units u = w, w+8, ... where a unit is a 32-channel K group (or, when there are fewer than
//     8 K groups, a tap range of one).  Each wave stages the input window of its own K group into a wave-private LDS
//     region - already normalised, activated, resampled, zero padded and split into fp16 hi + fp16 lo exactly like
//     conv_tile_kernel does - and reads its weights straight from L2 into registers (fragment-linear pieces, one
//     coalesced 16-byte load per lane and half), two 3-tap chunks in flight.  No workgroup barrier inside the K loop;
//   * the eight partial accumulators are combined through LDS in wave order (fixed order: deterministic) and wave 0
//     runs the usual epilogue (1/scale, bias, residual, activation, store, per-tile moments);
//   * the normalisation that precedes the layer is folded in (fused_norm_table): every workgroup reduces the
//     producer's per-tile moments itself - fp64, fixed order, Instance/GroupNorm(32) over a concatenation with both
//     FiLM stages, the same arithmetic as norm_finalize_kernel - into an LDS table of per-channel scale/shift while
//     its first loads are in flight.  The standalone finalize launch disappears for every tensor with few tiles.
// Weights use the image of conv_tile_kernel<1,...> (full_layout.h pack_conv_weight16 with TMB = 1); numerics are
// the same fp16 hi/lo split (three v_mfma_f32_16x16x32_f16 per 32-deep k step, fp32 accumulate).
#pragma once
#include "full_conv16_kernels.h"

namespace tha4 {

constexpr int kSmallWaves = 8;
constexpr int kSmallThreads = kSmallWaves * 64;
constexpr int kSmallMaxItems = 7;     // staging items (pixel, lane group) per lane and K group: window <= 112 pixels
constexpr int kSmallTapChunk = 3;     // taps per register-resident weight chunk (two chunks in flight)

#ifndef THA4_SMALL_PREFETCH_EPI
#define THA4_SMALL_PREFETCH_EPI(PG, POOL) (!((PG) == 4 && (POOL)))     // (the instantiation without spare registers requests them after the exchange)
#endif
template <int PG, int INMODE>
__global__ void __launch_bounds__(kSmallThreads) conv_small_kernel(ConvArgs a) {
  warm_kernarg<(int)sizeof(ConvArgs)>();
  constexpr bool kPool = INMODE == IN_POOL2;
  constexpr int KI = kSmallMaxItems, TC = kSmallTapChunk;
  THA4_DYN_LDS(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform_i32(tid >> 6);
  const int g = lane >> 4, g4 = g * 4;
  const float m1 = split_minus_one();
#if defined(THA4_PHASE_TIMING) && !defined(THA4_EMU)
  long long* stamps = a.dbg ? a.dbg + ((size_t)blockIdx.x * kSmallWaves + wave) * 64 : nullptr;
  int nstamp = 0;
#define THA4_SSTAMP() do { if (stamps && lane == 0 && nstamp < 64) stamps[nstamp] = clock64(); ++nstamp; } while (0)
#else
#define THA4_SSTAMP()
#endif
  THA4_SSTAMP();                                           // (phase-timing builds) S0: wave start (argument block warm)

  // ---- tile decomposition (16*PG output positions, TH x 2^twl) ------------------------------------
  // (every divisor below is a launch constant with a host-computed reciprocal - FastDiv, full_kernels.h: the eight run-time integer divisions of this
  // prologue were 2.3 k cycles of scalar work per wave)
  const int twl = a.wg_tw_log2, TWW = 1 << twl, TWH = (PG * 16) >> twl;
  const int tiles_x = a.tiles_x;
  const int tiles_per_frame = a.tiles_per_frame;
  // 1-D grid of batch * tiles * nb workgroups.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2):
  // all pixel tiles of one output block share that block's weights, so they are mapped to ONE XCD (block bo -> XCD
  // bo % 8) and every weight is fetched from HBM / MALL once per frame instead of once per XCD
  const int G = a.batch * tiles_per_frame;                 // workgroups per output block
  int cls = 0, bx = (int)blockIdx.x;                        // merged transposed convolution: the parity class is the slowest grid dimension
  if (a.nclass > 1) { cls = fast_div(bx, a.d_class); bx -= cls * a.d_class.d; }
  const ConvClass cg = conv_class(a, cls, tiles_per_frame);
  int bo, tl;
  if ((a.nb & 7) == 0) {
    const int xcd = bx & 7, slot = bx >> 3, sq = fast_div(slot, a.d_group);
    bo = xcd + 8 * sq;
    tl = slot - sq * G;
  } else {
    bo = fast_div(bx, a.d_group);
    tl = bx - bo * G;
  }
  const int n = fast_div(tl, a.d_tpf);
  const int tile = tl - n * tiles_per_frame;
  const int tile_row = fast_div(tile, a.d_tiles_x);
  const int tile_y0 = tile_row * TWH, tile_x0 = (tile - tile_row * tiles_x) << twl;
  const int vh = INMODE == IN_UP2 ? a.in_h * 2 : (kPool ? a.in_h / 2 : a.in_h);
  const int vw = INMODE == IN_UP2 ? a.in_w * 2 : (kPool ? a.in_w / 2 : a.in_w);
  const int in_px = a.in_h * a.in_w;
  const int WW = a.win_w, NPX = a.win_h * a.win_w;
  const int PLANE = tile_plane_bytes(NPX);
  const int vy0 = tile_y0 * a.in_stride + cg.win_dy0, vx0 = tile_x0 * a.in_stride + cg.win_dx0;

  int cbtot = 0;
  for (int s = 0; s < a.nsrc; ++s) cbtot += a.src[s].cb;
  const int NQ = (cbtot + 1) >> 1;                         // 32-channel K groups
  const int upq = a.units_per_q;                           // tap ranges per K group (1 unless NQ < 8)
  const int tpu = a.taps_per_unit;                         // taps per unit = ceil(ntaps / upq)
  const int nunits = NQ * upq;

  // ---- LDS: [scale | shift table] [8 waves x (4 hi + 4 lo planes)]; the reduction buffer aliases the windows -----
  const int tabf = fused_table_floats(a);
  float* tab_sc = reinterpret_cast<float*>(smem);
  float* tab_sh = tab_sc + (tabf >> 1);
  char* wins = smem + ((tabf * 4 + 127) & ~127);
  char* win_hi = wins + wave * 8 * PLANE;
  char* win_lo = win_hi + 4 * PLANE;
  const char* gw = reinterpret_cast<const char*>(a.w16) + (size_t)cls * a.w16_class_bytes + (size_t)bo * NQ * a.ntaps * 2048 + lane * 16;

  // weights of one chunk of <= TC taps of K group Q, straight from L2: fragment-linear pieces [Q][tap][hi 1 KiB | lo 1 KiB]
  struct WChunk { f16x8 h[TC], l[TC]; };
  auto load_weights = [&](int Q, int t0, int t1, WChunk& w) {
#pragma unroll
    for (int i = 0; i < TC; ++i) {
      const int t = min(t0 + i, t1 - 1);                    // past the range: re-read the last tap (never multiplied)
      const char* pc = gw + ((size_t)Q * a.ntaps + t) * 2048;
      w.h[i] = *reinterpret_cast<const f16x8*>(pc);
      w.l[i] = *reinterpret_cast<const f16x8*>(pc + 1024);
    }
  };
  // three register-resident weight chunks (3 taps each: a whole 3x3 K group) are in flight per wave; longer tap ranges
  // (4x4 stride 2: 16 taps) roll through the three slots
  WChunk w0, w1, w2;
  auto load_unit_head = [&](int Q, int t0, int t1) {       // the first three chunks of a unit
    load_weights(Q, t0, t1, w0);
    if (t0 + TC < t1) load_weights(Q, t0 + TC, t1, w1);
    if (t0 + 2 * TC < t1) load_weights(Q, t0 + 2 * TC, t1, w2);
  };
  THA4_SSTAMP();                                           // S1: tile decomposition + LDS carve-up done (scalar work)
  // The first unit's weights - the largest request of the prologue, and one that needs nothing but the grid position - go out HERE, in front
  // of the per-lane pixel / staging-item set-up (a few hundred instructions, fetched cold): they used to land ~1.2 k cycles after the window
  // (in-kernel stamps), now the set-up runs under their round trip
  int u = wave;
  if (u < nunits) {
    const int uq = fast_div(u, a.d_upq), t0 = (u - uq * upq) * tpu;
    load_unit_head(uq, t0, min(a.ntaps, t0 + tpu));
  }

  THA4_SSTAMP();                                           // S2: first weights requested
  // ---- per-lane output pixels ------------------------------------------------------------------
  // the pixel of the group this MFMA column computes (conflict-free window reads: full_kernels.h).  Any assignment is correct; the PG = 4 instantiations sit at
  // 256 VGPRs and one more live value spills, so they keep pixel = column (these launches are latency-, not LDS-bound: profiles/r05_boundaries_reading.md)
  const int pcol = PG == 4 ? (lane & 15) : pixel_of_column(a, lane & 15);
  int ly[PG], lx[PG], boff[PG];
  bool inside[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    const int i = pg * 16 + pcol;
    ly[pg] = i >> twl;
    lx[pg] = i & (TWW - 1);
    boff[pg] = ((ly[pg] * a.in_stride) * WW + lx[pg] * a.in_stride) * 16 + g * PLANE;
    inside[pg] = tile_y0 + ly[pg] < a.tile_h && tile_x0 + lx[pg] < a.tile_w;
  }

  // ---- staging items of this lane (geometry is the same for every K group) ----------------------
  const int sg = lane & 3;
  const int nitems = NPX * 4;
  struct Offsets { int v[KI]; } go;
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int item = lane + k * 64;
    const int px = item >> 2;
    const int wy = fast_div(px, a.d_win_w), wx = px - wy * WW;
    const int vy = vy0 + wy, vx = vx0 + wx;
    const bool ok = item < nitems && (unsigned)vy < (unsigned)vh && (unsigned)vx < (unsigned)vw;
    int o;
    if (INMODE == IN_DIRECT) o = vy * a.in_w + vx;
    else if (INMODE == IN_UP2) o = (vy >> 1) * a.in_w + (vx >> 1);
    else o = (2 * vy) * a.in_w + 2 * vx;
    go.v[k] = ok ? o * 64 + sg * 16 : -1;
  }

  struct QuadCtx { const char* base; f32x4 sc, sh; int act; int kind; bool fused; int tab; };
  // the two source descriptors as plain scalars (constant indices): indexing a.src[] with a run-time index costs a chain of
  // dependent scalar loads from the argument block at the top of every K group
  const float *s0_data = a.src[0].data, *s0_scale = a.src[0].scale, *s0_shift = a.src[0].shift;
  const float *s1_data = a.src[1].data, *s1_scale = a.src[1].scale, *s1_shift = a.src[1].shift;
  const int s0_cb = a.src[0].cb, s0_kind = a.src[0].kind, s0_act = a.src[0].act;
  const int s1_cb = a.src[1].cb, s1_kind = a.src[1].kind, s1_act = a.src[1].act;
  auto quad_ctx = [&](int q) -> QuadCtx {
    QuadCtx c;
    c.base = nullptr; c.act = ACT_NONE; c.kind = SRC_TENSOR; c.fused = false; c.tab = 0;
    c.sc = f32x4{1.f, 1.f, 1.f, 1.f};
    c.sh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q >= cbtot) return c;                              // phantom quad of an odd channel-block count
    const bool second = a.nsrc > 1 && q >= s0_cb;
    const int ql = second ? q - s0_cb : q;
    const float* const S_data = second ? s1_data : s0_data;
    const float* const S_scale = second ? s1_scale : s0_scale;
    const float* const S_shift = second ? s1_shift : s0_shift;
    const int S_cb = second ? s1_cb : s0_cb, S_kind = second ? s1_kind : s0_kind, S_act = second ? s1_act : s0_act;
    c.act = S_act; c.kind = S_kind;
    if (a.fnorm.enabled && S_kind == SRC_TENSOR) {
      c.fused = true;
      c.tab = q * 16 + sg * 4;                             // table index = padded channel of the concatenation
    } else if (S_scale) {
      c.sc = *reinterpret_cast<const f32x4*>(S_scale + ((size_t)n * S_cb + ql) * 16 + sg * 4);
      c.sh = *reinterpret_cast<const f32x4*>(S_shift + ((size_t)n * S_cb + ql) * 16 + sg * 4);
    }
    c.base = reinterpret_cast<const char*>(S_kind == SRC_VECTOR ? S_data + ((size_t)n * S_cb + ql) * 16
                                                                : S_data + ((size_t)n * S_cb + ql) * (size_t)in_px * 16);
    return c;
  };
  auto bind_table = [&](QuadCtx& c) {                      // after the table barrier
    if (!c.fused) return;
    c.sc = *reinterpret_cast<const f32x4*>(tab_sc + c.tab);
    c.sh = *reinterpret_cast<const f32x4*>(tab_sh + c.tab);
  };
  auto activate = [&](const f32x4& r, const QuadCtx& c) -> f32x4 { return apply_act4(r, c.sc, c.sh, c.act); };

  f32x4 rawA[KI], rawB[KI];
  QuadCtx cA, cB;
  // raw loads of one K group's window (activation is applied when the window is written: the table may not exist yet)
  auto load_window = [&](int Q) {
    cA = quad_ctx(2 * Q);
    cB = quad_ctx(2 * Q + 1);
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int o = go.v[k];
      rawA[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      rawB[k] = rawA[k];
      if (o < 0) continue;
      if (cA.base) rawA[k] = cA.kind == SRC_VECTOR ? *reinterpret_cast<const f32x4*>(cA.base + sg * 16) : *reinterpret_cast<const f32x4*>(cA.base + (unsigned)o);
      if (cB.base) rawB[k] = cB.kind == SRC_VECTOR ? *reinterpret_cast<const f32x4*>(cB.base + sg * 16) : *reinterpret_cast<const f32x4*>(cB.base + (unsigned)o);
    }
  };
  // IN_POOL2: the three other samples of the 2x2 window are fetched and activated here (AvgPool2d of the ACTIVATED tensor, unet.py:58)
  auto pooled = [&](const QuadCtx& c, int o, const f32x4& v00) -> f32x4 {
    const float* ptr = reinterpret_cast<const float*>(c.base + (unsigned)o);
    const f32x4 a00 = activate(v00, c);
    const f32x4 a01 = activate(*reinterpret_cast<const f32x4*>(ptr + 16), c);
    const f32x4 a10 = activate(*reinterpret_cast<const f32x4*>(ptr + (size_t)a.in_w * 16), c);
    const f32x4 a11 = activate(*reinterpret_cast<const f32x4*>(ptr + (size_t)a.in_w * 16 + 16), c);
    return ((a00 + a01) + (a10 + a11)) * 0.25f;
  };
  auto write_window = [&]() {
    bind_table(cA);
    bind_table(cB);
#ifdef THA4_PHASE_WINDOW
    THA4_SSTAMP();
#endif
#pragma unroll
    for (int k = 0; k < KI; ++k) {
#ifdef THA4_PHASE_WINDOW
      if (k > 0) THA4_SSTAMP();
#endif
      const int item = lane + k * 64;
      if (item >= nitems) continue;
      f32x4 va = rawA[k], vb = rawB[k];
      const int o = go.v[k];
      if (o < 0) {                                         // zero padding is applied AFTER normalisation + activation
        va = f32x4{0.f, 0.f, 0.f, 0.f};
        vb = va;
      } else {
        if (cA.base) va = (kPool && cA.kind == SRC_TENSOR) ? pooled(cA, o, va) : activate(va, cA);
        if (cB.base) vb = (kPool && cB.kind == SRC_TENSOR) ? pooled(cB, o, vb) : activate(vb, cB);
      }
      f16x8 hi, lo;
{
        _Float16 h[8], l[8];
        split_pair(va[0], va[1], m1, h[0], h[1], l[0], l[1]);
        split_pair(va[2], va[3], m1, h[2], h[3], l[2], l[3]);
        split_pair(vb[0], vb[1], m1, h[4], h[5], l[4], l[5]);
        split_pair(vb[2], vb[3], m1, h[6], h[7], l[6], l[7]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = h[j]; lo[j] = l[j]; }
      }
      const int off = sg * PLANE + (item >> 2) * 16;
      *reinterpret_cast<f16x8*>(win_hi + off) = hi;
      *reinterpret_cast<f16x8*>(win_lo + off) = lo;
    }
  };

  f32x4 acc[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) acc[pg] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tapl = lane & (kMaxTaps - 1);                  // tap offsets in lane t of one VGPR (see conv_tile_kernel)
  const int my_toff = ((conv_tap_dy(a, cls, tapl) - cg.win_dy0) * WW + (conv_tap_dx(a, cls, tapl) - cg.win_dx0)) * 16;
  auto mac_chunk = [&](int t0, int t1, const WChunk& w) {
#pragma unroll
    for (int i = 0; i < TC; ++i) {
      const int t = t0 + i;
      if (t >= t1) break;
      const int toff = lane_pick(my_toff, t);
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) {
        const f16x8 bh = *reinterpret_cast<const f16x8*>(win_hi + boff[pg] + toff);
        const f16x8 bl = *reinterpret_cast<const f16x8*>(win_lo + boff[pg] + toff);
        acc[pg] = mfma16h(w.h[i], bh, acc[pg]);
        acc[pg] = mfma16h(w.h[i], bl, acc[pg]);
        acc[pg] = mfma16h(w.l[i], bh, acc[pg]);
      }
    }
  };

  THA4_SSTAMP();                                           // S3 (was 0): per-lane set-up done
  // ---- first unit's loads go out before the normalisation table is built -------------------------
  int curQ = -1;
  if (u < nunits) {
    curQ = fast_div(u, a.d_upq);
    load_window(curQ);
  }
  // wave 0 runs the epilogue: its residual values are requested now, not after the reduction
  const int out_px = a.out_h * a.out_w;
  constexpr bool kPrefetchRes = !(PG == 4 && kPool);       // (the one instantiation without a spare register keeps the load in the epilogue)
  auto load_residual = [&](int pg) -> f32x4 {
    const int oy = (tile_y0 + ly[pg]) * a.out_sy + cg.out_oy, ox = (tile_x0 + lx[pg]) * a.out_sx + cg.out_ox;
    if (a.res_mode == IN_DIRECT)
      return *reinterpret_cast<const f32x4*>(a.residual + (((size_t)n * a.nb + bo) * out_px + (size_t)oy * a.out_w + ox) * 16 + g4);
    if (a.res_mode == IN_UP2) {                            // ResBlock x_resample = Upsample (unet.py:46): nearest
      const int rw = a.out_w >> 1, rpx = out_px >> 2;
      return *reinterpret_cast<const f32x4*>(a.residual + (((size_t)n * a.nb + bo) * rpx + (size_t)(oy >> 1) * rw + (ox >> 1)) * 16 + g4);
    }
    const int rw = a.out_w * 2;                            // x_resample = Downsample = AvgPool2d(2,2) (unet.py:58)
    const float* r0 = a.residual + (((size_t)n * a.nb + bo) * ((size_t)out_px * 4) + (size_t)(2 * oy) * rw + 2 * ox) * 16 + g4;
    return ((*reinterpret_cast<const f32x4*>(r0) + *reinterpret_cast<const f32x4*>(r0 + 16)) +
            (*reinterpret_cast<const f32x4*>(r0 + (size_t)rw * 16) + *reinterpret_cast<const f32x4*>(r0 + (size_t)rw * 16 + 16))) * 0.25f;
  };
  f32x4 resv[kPrefetchRes ? PG : 1];
  if (kPrefetchRes) {
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      resv[pg] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (wave == 0 && a.residual && inside[pg]) resv[pg] = load_residual(pg);
    }
  }
  // ... and so are its bias and output-activation codes: requested after the partial-sum exchange they were FIVE dependent memory round
  // trips at the end of every launch (bias, then one code per channel in front of apply_act's branches: ~3 k of a 26 k-cycle launch)
  constexpr bool kPrefetchEpi = THA4_SMALL_PREFETCH_EPI(PG, kPool);
  f32x4 bias_pre = f32x4{0.f, 0.f, 0.f, 0.f};
  int codes_pre[4] = {ACT_NONE, ACT_NONE, ACT_NONE, ACT_NONE};
  if (kPrefetchEpi && wave == 0) {
    if (a.bias) bias_pre = *reinterpret_cast<const f32x4*>(a.bias + bo * 16 + g4);
    if (a.act_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j) codes_pre[j] = a.act_out[bo * 16 + g4 + j];
    }
  }
  THA4_SSTAMP();                                           // 1: first unit's loads issued
#if defined(THA4_PHASE_TIMING) && !defined(THA4_EMU)
  if (a.dbg) {                                             // tuning aid: when do the window (requested first) and the weights land?
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the weights are requested first since round 4: they are older than the window)
    THA4_SSTAMP();                                         // 2: window landed
    THA4_SSTAMP();                                         // 3: weights landed
  }
#endif
  if (a.fnorm.enabled) {
    fused_norm_table<false>(a, n, tid, kSmallThreads, tab_sc, tab_sh, reinterpret_cast<double*>(wins));      // (per-tile moments only: full_conv16_kernels.h)
    __syncthreads();                                       // table complete; scratch (aliasing the windows) no longer read
  }
  THA4_SSTAMP();                                           // 2: normalisation table built
  bool staged = false;
  for (; u < nunits; u += kSmallWaves) {
    const int Q = fast_div(u, a.d_upq);
    const int t0 = (u - Q * upq) * tpu, t1 = min(a.ntaps, t0 + tpu);
    if (Q != curQ) { load_window(Q); curQ = Q; staged = false; }
    if (!staged) {                                         // wave-private LDS: program order, no workgroup barrier
      THA4_PRIO_VALU();
      write_window();
      THA4_PRIO_MFMA();
      staged = true;
      THA4_WAVE_SYNC();
    }
    THA4_SSTAMP();                                         // 3 + 2i: unit i's window written (its loads have arrived)
    for (int t = t0; t < t1; t += 3 * TC) {                // slots are refilled as soon as their MFMAs are issued
      mac_chunk(t, t1, w0);
      if (t + 3 * TC < t1) load_weights(Q, t + 3 * TC, t1, w0);
      if (t + TC < t1) {
        mac_chunk(t + TC, t1, w1);
        if (t + 4 * TC < t1) load_weights(Q, t + 4 * TC, t1, w1);
      }
      if (t + 2 * TC < t1) {
        mac_chunk(t + 2 * TC, t1, w2);
        if (t + 5 * TC < t1) load_weights(Q, t + 5 * TC, t1, w2);
      }
    }
    THA4_SSTAMP();                                         // 4 + 2i: unit i's MFMAs issued
    // the next unit's window + weights are requested before this wave idles: one memory round trip per unit
    const int un = u + kSmallWaves;
    if (un < nunits) {
      const int Qn = fast_div(un, a.d_upq);
      if (Qn != curQ) { load_window(Qn); curQ = Qn; staged = false; }
      const int tn = (un - Qn * upq) * tpu;
      load_unit_head(Qn, tn, min(a.ntaps, tn + tpu));
    }
  }

  // ---- combine the eight K slices in wave order, epilogue on wave 0 -----------------------------------
  THA4_SSTAMP();                                           // K loop left
  __syncthreads();                                         // every wave is done with its window: the region is reused
  f32x4* red = reinterpret_cast<f32x4*>(wins);             // [wave][pg][64]
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) red[(wave * PG + pg) * 64 + lane] = acc[pg];
  __syncthreads();
  THA4_SSTAMP();                                           // partial sums exchanged
  if (wave != 0) return;
  // every operand of the epilogue is requested (or already here) before anything is consumed; the stores go out LAST, behind the DPP
  // reduction of the moments, so that no instruction waits for a store's acknowledgement (same operations on the same values as the
  // one-position-at-a-time form: same bits)
  f32x4 bias = bias_pre;
  int codes[4] = {codes_pre[0], codes_pre[1], codes_pre[2], codes_pre[3]};
  if (!kPrefetchEpi) {
    if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + bo * 16 + g4);
    if (a.act_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j) codes[j] = a.act_out[bo * 16 + g4 + j];
    }
  }
  f32x4 resl[kPrefetchRes ? 1 : PG];
  if (!kPrefetchRes) {
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      resl[pg] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.residual && inside[pg]) resl[pg] = load_residual(pg);
    }
  }
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 vout[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    f32x4 s = red[(0 * PG + pg) * 64 + lane];
#pragma unroll
    for (int w2i = 1; w2i < kSmallWaves; ++w2i) s = s + red[(w2i * PG + pg) * 64 + lane];
    vout[pg] = s;
    if (!inside[pg]) continue;                             // ragged tile: position outside the map
    f32x4 v = s * a.w16_inv_scale + bias;
    if (a.residual) v = v + (kPrefetchRes ? resv[kPrefetchRes ? pg : 0] : resl[kPrefetchRes ? 0 : pg]);
    if (a.act_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], codes[j]);
    }
    vout[pg] = v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[j] += v[j]; ssq[j] = fmaf(v[j], v[j], ssq[j]); }
  }
  float rs[4], rq[4];
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rs[j] = row16_sum(ssum[j], lane);
      rq[j] = row16_sum(ssq[j], lane);
    }
  }
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    if (!inside[pg]) continue;
    const int oy = (tile_y0 + ly[pg]) * a.out_sy + cg.out_oy, ox = (tile_x0 + lx[pg]) * a.out_sx + cg.out_ox;
    const size_t off = (((size_t)n * a.nb + bo) * out_px + (size_t)oy * a.out_w + ox) * 16 + g4;
    *reinterpret_cast<f32x4*>(a.out + off) = vout[pg];
  }
  if (a.stats && (lane & 15) == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* dst = a.stats + ((((size_t)n * a.stats_tiles + cg.stats_tile0 + tile) * a.nb + bo) * 16 + g4 + j) * 2;
      dst[0] = rs[j];
      dst[1] = rq[j];
      // (many-tile outputs - the transposed convolutions of the encoder-decoders - also feed the tensor's moment accumulators: full_kernels.h MomentAcc)
      if (a.stats_acc) moment_acc_add(a.stats_acc + (((size_t)n * kMomentShards + ((int)blockIdx.x & (kMomentShards - 1))) * a.nb + bo) * 16 + g4 + j, rs[j], rq[j], a.acc_fault);
    }
  }
  THA4_SSTAMP();                                           // wave 0: epilogue issued
}

}  // namespace tha4
