// Tiled implicit-GEMM convolution, second generation: every k > 1 convolution of the full THA4 system.
//
// conv_mfma_kernel (full_kernels.h) re-forms its im2col operand for EVERY tap and output tile - scale/shift,
// activation, padding select and 64-bit address arithmetic per 16-byte load - and multiplies with exact-fp32
// v_mfma_f32_16x16x4_f32 (32 cycles per 4-deep k step).  Here
//   * a workgroup (8 waves) owns a TH x TW block of output positions and TMB output blocks; per 32-channel K group
//     it stages the (TH*s + halo) x (TW*s + halo) input window ONCE into LDS, already normalised, activated
//     (SiLU on v_exp_f32 / v_rcp_f32), resampled (nearest-up / avg-pool), zero padded and split into fp16 hi + fp16 lo
//     (v = hi + lo, 22 bits);
//   * all taps then read their B fragments from that window at a constant LDS offset per tap; the weights stream
//     through a 2..4-slot LDS ring (as deep as the LDS left by the window allows without costing occupancy) as fp16
//     hi/lo fragment pieces, pre-scaled by a power of two so that small weights keep their low half out of the fp16
//     subnormal range; every 32-deep k step costs three v_mfma_f32_16x16x32_f16 (hi*hi + hi*lo + lo*hi, 48 cycles
//     instead of 256) into one fp32 accumulator;
//   * the window of K group Q+1 is loaded into registers under the MFMAs of group Q;
//   * small maps (16x16 .. 48x48 with 256-512 channels) do not have enough output tiles to fill 256 CUs, and a
//     pixel-tiled grid would re-read every weight once per tile: there blockIdx.z splits the K groups instead
//     (phase 1: every workgroup writes its fp32 partial fragments to a workspace indexed by output block) and a second
//     launch of the same kernel with ONE output block per workgroup (phase 2) adds the partials in split order and
//     runs the epilogue.  Every weight is read exactly once per frame.  (A single-launch "last workgroup reduces"
//     variant is far slower on this chip: tools/microbench/last_arrival.hip.)  Not compiled into the PG = 4 kernel;
//   * tile grids need not divide the map: positions outside it are computed on zero padding and masked at the store.
// Numerics: products are exact in fp32; dropped lo*lo terms and the rounding of the lo halves are ~2^-22 relative,
// four times the fp32 rounding the reference itself commits per product (tests: 3e-5 abs on O(1) outputs).
// tools/phase_timing_full.py (-DTHA4_PHASE_TIMING) prints the cycle budget of a K group from in-kernel stamps.
//
// K-slot permutation: lane group g = lane>>4 holds, in its 8 k-slots, channels 4g..4g+3 of quad 2Q (j<4) and of quad
// 2Q+1 (j>=4), i.e. exactly the two 16-byte C16 loads a staging thread makes for one (pixel, g); the weight
// pieces use the same permutation (full_layout.h pack_conv_weight16).
#pragma once
#include "full_kernels.h"

namespace tha4 {

#ifndef THA4_TILE_COUNTED_WAIT
// 1: the chunk barriers leave the weight fetches of the chunks after the next one in flight (counted s_waitcnt vmcnt + raw
// s_barrier, so that a ring deeper than two slots hides fetch latency).  Measured (round 2, tools/gpu_call19.sh): parity-clean but
// 1 % SLOWER (146.5 vs 148.3 fps; batch 8 263 vs 268) - the barriers wait for wave skew, not for the fetches - so it stays off.
#define THA4_TILE_COUNTED_WAIT 0
#endif
constexpr int kTileWaves = 8;
constexpr int kTileThreads = kTileWaves * 64;
constexpr int kTileMaxItems = 5;     // staging items (pixel, g) per thread and K group: window <= 640 pixels
constexpr int kTileWavesHalf = 4;    // the four-wave form (NW = 4, round 4): half the pixel tile, TWO workgroups per CU
constexpr int kTileMaxItemsHalf = 6; // its staging items per thread: window <= 384 pixels (16 x 16 outputs + halo)

// (tile_plane_bytes - the bytes of one lane-group plane of the window image - lives in full_kernels.h next to the pixel permutation that depends on it)

// Per-channel scale/shift of the normalisation that precedes a convolution, computed by the consumer itself from the
// producer's per-tile moments.  Mirrors norm_finalize_kernel (same fp64 arithmetic); all `nthreads` threads of the
// workgroup must call it; `scratch` holds 2*ctot doubles and may alias memory that is not in use yet.  The caller
// synchronises the workgroup afterwards before reading the table.
// ACC = false compiles the moment-accumulator route out (conv_small_kernel: its prologue already holds a unit's weights and window in registers - the 64
// more the accumulator loads take spill - and the planner never gives it one: small maps have few tiles).
template <bool ACC = true>
THA4_DEV void fused_norm_table(const ConvArgs& a, int n, int tid, int nthreads, float* tab_sc, float* tab_sh, double* scratch) {
  const FusedNorm& f = a.fnorm;
  const int c0 = a.src[0].cb * 16;
  const int ctot = c0 + (a.nsrc > 1 && a.src[1].kind == SRC_TENSOR ? a.src[1].cb * 16 : 0);     // <= 2 * nthreads (host-checked)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int MAXC = 2, TB = 8;
  // ONE memory round trip: every global value this thread's channels need - the moments of all tiles (TB loads in flight
  // per batch), gamma, beta and both FiLM rows - is requested before anything is consumed
  float gam[MAXC], bet[MAXC], s0v[MAXC], b0v[MAXC], s1v[MAXC], b1v[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * nthreads;
    gam[i] = 0.f; bet[i] = 0.f; s0v[i] = 0.f; b0v[i] = 0.f; s1v[i] = 0.f; b1v[i] = 0.f;
    if (c < ctot && c < f.channels) {
      gam[i] = f.gamma[c];
      bet[i] = f.beta[c];
      if (f.film0) { s0v[i] = f.film0[c]; b0v[i] = f.film0[f.channels + c]; }
      if (f.film1) { s1v[i] = f.film1[(size_t)n * f.film1_stride + c]; b1v[i] = f.film1[(size_t)n * f.film1_stride + f.channels + c]; }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * nthreads;
    if (c >= ctot) continue;
    const int s = c < c0 ? 0 : 1;
    const int cl = c - (s ? c0 : 0);
    const int cw = a.src[s].cb * 16;
    const int nt = f.tiles[s];
    const float* ps = f.stats[s] + ((size_t)n * nt * cw + cl) * 2;
    double su = 0.0, sq = 0.0;
    if (ACC && f.acc) {                                      // the producers' moment accumulators: kMomentShards (hi, lo) integer pairs per channel, all requested at once
      const MomentAcc* pa = reinterpret_cast<const MomentAcc*>(f.stats[s]) + ((size_t)n * kMomentShards * cw + cl);
      double vs[kMomentShards], vq[kMomentShards];
#pragma unroll
      for (int u2 = 0; u2 < kMomentShards; ++u2) moment_acc_read(pa + (size_t)u2 * cw, vs[u2], vq[u2]);
#pragma unroll
      for (int u2 = 0; u2 < kMomentShards; ++u2) { su += vs[u2]; sq += vq[u2]; }      // fixed shard order
    } else
    for (int t0 = 0; t0 < nt; t0 += TB) {
      f32x2 v[TB];
#pragma unroll
      for (int u2 = 0; u2 < TB; ++u2) v[u2] = *reinterpret_cast<const f32x2*>(ps + (size_t)min(t0 + u2, nt - 1) * cw * 2);
#pragma unroll
      for (int u2 = 0; u2 < TB; ++u2) {                    // fixed order; tiles past the last one are re-reads, dropped by the select
        const bool keep = t0 + u2 < nt;
        su += keep ? (double)v[u2][0] : 0.0;
        sq += keep ? (double)v[u2][1] : 0.0;
      }
    }
    scratch[c] = su;
    scratch[ctot + c] = sq;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = tid + i * nthreads;
    if (c >= ctot) continue;
    float sc = 0.f, sh = 0.f;
    if (c < f.channels) {
      double mean, var;
      if (f.groups == 0) {
        mean = scratch[c] * f.inv_count;
        var = scratch[ctot + c] * f.inv_count - mean * mean;
      } else {
        const int gs = f.channels / f.groups;
        const int gi = c / gs;
        double su = 0.0, sq = 0.0;
        for (int k = gi * gs; k < (gi + 1) * gs; ++k) { su += scratch[k]; sq += scratch[ctot + k]; }
        mean = su * f.inv_count / gs;
        var = sq * f.inv_count / gs - mean * mean;
      }
      const double rstd = 1.0 / sqrt(fmax(var, 0.0) + (double)f.eps);
      double k = (double)gam[i] * rstd;
      double b = (double)bet[i] - mean * k;
      if (f.film0) { k *= (1.0 + (double)s0v[i]); b = b * (1.0 + (double)s0v[i]) + (double)b0v[i]; }
      if (f.film1) { k *= (1.0 + (double)s1v[i]); b = b * (1.0 + (double)s1v[i]) + (double)b1v[i]; }
      sc = (float)k;
      sh = (float)b;
      report_fault_unless_finite(f.fault, sc, sh);
    }
    tab_sc[c] = sc;
    tab_sh[c] = sh;
  }
}

THA4_DEV int fused_table_floats(const ConvArgs& a) {      // 2 x padded concatenated channels, 0 when nothing is fused
  if (!a.fnorm.enabled) return 0;
  int c = a.src[0].cb * 16;
  if (a.nsrc > 1 && a.src[1].kind == SRC_TENSOR) c += a.src[1].cb * 16;
  return 2 * c;
}

// MSW = 2 (round 3): SIXTEEN waves on the same workgroup tile - the TMB output blocks are split over two halves of the workgroup
// (wave w: pixel slot w & 7, block half w >> 3), so the window, the weight ring and every byte of traffic stay what they are while each
// wave carries half the accumulators and a SIMD hosts four waves instead of two (what the student kernels gained 5-10 % from).
// NW = 4 (round 4): FOUR pixel-slot waves per workgroup = half the pixel tile (64 PG positions), at most 80 KiB of LDS and the register
// budget of two waves per SIMD, so that TWO workgroups share a CU: one workgroup's prologue (norm table, first fetch, first window),
// staging phases and epilogue (stores, statistics) run under the other one's MFMAs instead of leaving the matrix pipe idle - the
// lock-step of the 8-wave form has every wave of the CU in the same phase.  It needs the staging phase at raised issue priority
// (THA4_PHASE_PRIO: a VALU wave only hides under a partner's MFMAs at s_setprio 1, profiles/r03_machine_model.md) and pays on grids of
// several rounds, where the workgroups of a CU drift out of phase by themselves: measured +4-7 % at batch 8 for the <4,4> / <4,2> /
// <2,4> classes, nothing at batch 1 where the two workgroups of a CU start and end together - a start delay of the one in the odd
// slot (HW_ID.TG_ID, 3k-40k cycles) measured neutral to negative and is not in the code (profiles/r04_full_conv_tile_reading.md).
template <int TMB, int PG, int INMODE, int MSW = 1, int NW = kTileWaves>
__global__ void __launch_bounds__(64 * NW * MSW, NW == kTileWaves ? 2 * MSW : 2) conv_tile_kernel(ConvArgs a) {      // (threads, waves per SIMD)
  warm_kernarg<(int)sizeof(ConvArgs)>();
  static_assert(MSW == 1 || (MSW == 2 && TMB % 2 == 0), "the block split needs an even block count");
  static_assert(NW == kTileWaves || (NW == kTileWavesHalf && MSW == 1), "four or eight pixel-slot waves");
  constexpr bool kPool = INMODE == IN_POOL2;
  constexpr int kWaves = NW * MSW, kThreads = 64 * kWaves;                     // waves / threads of this instantiation
  constexpr int TMBW = TMB / MSW;                                              // output blocks per wave
  constexpr int KI = NW == kTileWavesHalf ? kTileMaxItemsHalf : MSW == 2 ? (kTileMaxItems + 1) / 2 : kTileMaxItems;       // staging items per thread
  THA4_DYN_LDS(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform_i32(tid >> 6);
  const int pw = wave & (NW - 1), mh = wave / NW;                      // pixel slot and block half of this wave
  const int p = lane & 15, g = lane >> 4, g4 = g * 4;
  const float m1 = split_minus_one();

  // ---- tile decomposition -------------------------------------------------------------------
  const int twl = a.wg_tw_log2, TWW = 1 << twl, TWH = (NW * PG * 16) >> twl;
  // (every divisor below is a launch constant with a host-computed reciprocal - FastDiv, full_kernels.h)
  const int tiles_x = a.tiles_x;
  const int tiles_per_frame = a.tiles_per_frame;
  int cls = 0, bx = (int)blockIdx.x, mtile = (int)blockIdx.y;   // merged transposed convolution: the parity class is the slowest part of the x index
  if (a.xcd_remap) {                                        // 1-D grid: the output-channel tiles of one pixel tile are consecutive workgroups of one XCD (ConvArgs::xcd_remap)
    const int xcd = bx & 7, s = bx >> 3;
    const int ts = fast_div(s, a.d_mtiles);
    mtile = s - ts * a.d_mtiles.d;
    bx = ts * 8 + xcd;
  }
  if (a.nclass > 1) { cls = fast_div(bx, a.d_class); bx -= cls * a.d_class.d; }
  const ConvClass cg = conv_class(a, cls, tiles_per_frame);
  const int n = fast_div(bx, a.d_tpf);
  const int tile = bx - n * tiles_per_frame;
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
  const int NQ = (cbtot + 1) >> 1;                       // 32-channel K groups
  const int ksplit = a.phase == 0 ? 1 : a.ksplit, ks = a.phase == 1 ? (int)blockIdx.z : 0;
  const int q_per = a.phase == 0 ? NQ : a.q_per;         // ceil(NQ / ksplit)
  int q_begin = ks * q_per;
  const int q_end = min(NQ, q_begin + q_per);              // this workgroup's K groups
  const int ntc = a.ntc;                                  // weight chunks per K group = ceil(ntaps / taps_per_chunk) (the last one may be shorter: 9 = 5 + 4)
  const int slot_bytes = a.taps_per_chunk * TMB * 2048;
  char* win_hi = smem;
  char* win_lo = smem + 4 * PLANE;
  const int WB = a.win_buffers == 2 ? 8 * PLANE : 0;      // byte distance of the second window buffer (0: single-buffered)
  char* ring = smem + 8 * PLANE + WB;
  const int D = a.ring_slots;                                            // ring depth (2..4)
  float* red = reinterpret_cast<float*>(ring + D * slot_bytes);          // [NW waves][TMB*16][2]
  // normalisation folded into this kernel (FusedNorm): per-channel scale | shift table behind `red`
  float* tab_sc = red + NW * TMB * 16 * 2;
  float* tab_sh = tab_sc + (fused_table_floats(a) >> 1);
  const char* gw = reinterpret_cast<const char*>(a.w16) + (size_t)cls * a.w16_class_bytes + (size_t)mtile * NQ * a.ntaps * TMB * 2048;
  auto fetch = [&](int chunk, int slot) {                  // chunk = K group * ntc + chunk of the group
    const int cq = fast_div(chunk, a.d_ntc), ct = chunk - cq * ntc;
    const int t0 = ct * a.taps_per_chunk, nt = min(a.taps_per_chunk, a.ntaps - t0);
    const char* src = gw + ((size_t)cq * a.ntaps + t0) * TMB * 2048;
    char* dst = ring + slot * slot_bytes;
    const int pieces = nt * TMB * 2;
    for (int pc = wave; pc < pieces; pc += kWaves) glds16(src + pc * 1024 + (unsigned)(lane * 16), dst + pc * 1024);
  };
  // The first weight chunks - the largest request of the prologue, and one that needs nothing but the grid position - go out HERE, in front of
  // the per-lane pixel / staging-item set-up (a few hundred instructions, fetched cold): the set-up runs under their round trip
  int chunk = q_begin * ntc;
  const int nchunks = q_end * ntc;
  const bool reduce_phase = a.phase == 2;
  if (reduce_phase) q_begin = q_end;                       // nothing to multiply: partials come from the workspace
  int issued = chunk, islot = 0;                           // next chunk to fetch and the slot it goes to
  if (q_begin < q_end) {
    for (int i = 0; i < D - 1 && issued < nchunks; ++i) {  // D-1 chunks ahead; the D-th slot is the one being read
      fetch(issued++, islot);
      islot = islot + 1 == D ? 0 : islot + 1;
    }
  }

  // ---- per-lane output pixels ------------------------------------------------------------------
  // (column p of a 16-pixel group computes pixel pcol: the host picks the assignment that makes the B-fragment reads conflict-free, full_kernels.h)
  const int pcol = pixel_of_column(a, p);
  int ly[PG], lx[PG], boff[PG];
  bool inside[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    const int i = (pw * PG + pg) * 16 + pcol;
    ly[pg] = i >> twl;
    lx[pg] = i & (TWW - 1);
    boff[pg] = ((ly[pg] * a.in_stride) * WW + lx[pg] * a.in_stride) * 16 + g * PLANE;
    inside[pg] = tile_y0 + ly[pg] < a.tile_h && tile_x0 + lx[pg] < a.tile_w;
  }

  // ---- staging items of this thread (geometry is the same for every K group) --------------------
  const int sg = tid & 3;                                 // lane group plane this thread stages
  const int nitems = NPX * 4;
  struct Offsets { int v[KI]; } go;                        // byte offset of the item inside a quad's plane; < 0: zero padding
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int item = tid + k * kThreads;
    const int px = item >> 2;
    const int wy = fast_div(px, a.d_win_w), wx = px - wy * WW;
    const int vy = vy0 + wy, vx = vx0 + wx;
    const bool ok = item < nitems && (unsigned)vy < (unsigned)vh && (unsigned)vx < (unsigned)vw;
    int o;
    if (INMODE == IN_DIRECT) o = vy * a.in_w + vx;
    else if (INMODE == IN_UP2) o = (vy >> 1) * a.in_w + (vx >> 1);
    else o = (2 * vy) * a.in_w + 2 * vx;
    go.v[k] = ok ? o * 64 + sg * 16 : -1;                  // BYTE offset from the (wave-uniform) quad base: saddr + voffset loads
  }

  // glds instructions EVERY wave issues per chunk (wave w issues ceil((pieces - w) / 8)): the lower bound the counted barrier uses
  const int keep_per_chunk = ((a.ntaps - (ntc - 1) * a.taps_per_chunk) * TMB * 2) / kWaves;     // (of the shortest chunk)

  // one quad of one K group: which source, its scale/shift/activation for lane group sg
  struct QuadCtx { const char* base; f32x4 sc, sh; int act; int kind; };   // base: wave-uniform
  // the two source descriptors as plain scalars (constant indices): indexing a.src[] with a run-time index costs a chain of
  // dependent scalar loads from the argument block at the top of every K group
  const float *s0_data = a.src[0].data, *s0_scale = a.src[0].scale, *s0_shift = a.src[0].shift;
  const float *s1_data = a.src[1].data, *s1_scale = a.src[1].scale, *s1_shift = a.src[1].shift;
  const int s0_cb = a.src[0].cb, s0_kind = a.src[0].kind, s0_act = a.src[0].act;
  const int s1_cb = a.src[1].cb, s1_kind = a.src[1].kind, s1_act = a.src[1].act;
  auto quad_ctx = [&](int q) -> QuadCtx {
    QuadCtx c;
    c.base = nullptr; c.act = ACT_NONE; c.kind = SRC_TENSOR;
    c.sc = f32x4{1.f, 1.f, 1.f, 1.f};
    c.sh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q >= cbtot) return c;                             // phantom quad of an odd channel-block count
    const bool second = a.nsrc > 1 && q >= s0_cb;
    const int ql = second ? q - s0_cb : q;
    const float* const S_data = second ? s1_data : s0_data;
    const float* const S_scale = second ? s1_scale : s0_scale;
    const float* const S_shift = second ? s1_shift : s0_shift;
    const int S_cb = second ? s1_cb : s0_cb, S_kind = second ? s1_kind : s0_kind, S_act = second ? s1_act : s0_act;
    c.act = S_act; c.kind = S_kind;
    if (a.fnorm.enabled && S_kind == SRC_TENSOR) {          // table index = padded channel of the concatenation
      c.sc = *reinterpret_cast<const f32x4*>(tab_sc + q * 16 + sg * 4);
      c.sh = *reinterpret_cast<const f32x4*>(tab_sh + q * 16 + sg * 4);
    } else if (S_scale) {
      c.sc = *reinterpret_cast<const f32x4*>(S_scale + ((size_t)n * S_cb + ql) * 16 + sg * 4);
      c.sh = *reinterpret_cast<const f32x4*>(S_shift + ((size_t)n * S_cb + ql) * 16 + sg * 4);
    }
    c.base = reinterpret_cast<const char*>(S_kind == SRC_VECTOR ? S_data + ((size_t)n * S_cb + ql) * 16
                                                                : S_data + ((size_t)n * S_cb + ql) * (size_t)in_px * 16);
    return c;
  };
  auto activate = [&](const f32x4& r, const QuadCtx& c) -> f32x4 { return apply_act4(r, c.sc, c.sh, c.act); };
  // raw (IN_DIRECT / IN_UP2) or finished (IN_POOL2: the 2x2 mean of the activated samples) values of one item
  auto load_quad = [&](const QuadCtx& c, int o) -> f32x4 {
    if (!c.base) return f32x4{0.f, 0.f, 0.f, 0.f};          // wave-uniform: phantom quad
    if (c.kind == SRC_VECTOR) return *reinterpret_cast<const f32x4*>(c.base + sg * 16);
    // padding items (o < 0) read pixel 0 instead of branching per lane: their value is discarded by write_window, and ten
    // divergent branches per K group cost more than ten redundant loads
    const float* ptr = reinterpret_cast<const float*>(c.base + (unsigned)max(o, 0));
    if (!kPool) return *reinterpret_cast<const f32x4*>(ptr);
    const f32x4 v00 = activate(*reinterpret_cast<const f32x4*>(ptr), c);
    const f32x4 v01 = activate(*reinterpret_cast<const f32x4*>(ptr + 16), c);
    const f32x4 v10 = activate(*reinterpret_cast<const f32x4*>(ptr + (size_t)a.in_w * 16), c);
    const f32x4 v11 = activate(*reinterpret_cast<const f32x4*>(ptr + (size_t)a.in_w * 16 + 16), c);
    return ((v00 + v01) + (v10 + v11)) * 0.25f;           // AvgPool2d(2,2) of the activated tensor (unet.py:58)
  };

  f32x4 rawA[KI], rawB[KI];
  QuadCtx cA, cB;
  auto load_window = [&](int Q, const Offsets gofs) {
    if (THA4_HOOK_TILE_WINDOW_BYPASS) return;
    cA = quad_ctx(2 * Q);
    cB = quad_ctx(2 * Q + 1);
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      rawA[k] = load_quad(cA, gofs.v[k]);
      rawB[k] = load_quad(cB, gofs.v[k]);
    }
  };
  auto write_window = [&](const Offsets gofs, int wofs) {   // wofs: byte offset of the window buffer written
    if (THA4_HOOK_TILE_WINDOW_BYPASS) return;
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int item = tid + k * kThreads;
      if (item >= nitems) continue;
      f32x4 va = rawA[k], vb = rawB[k];
      if (THA4_HOOK_TILE_STAGE_VALU_BYPASS) {                // tuning builds only: the loaded bits as they are
        const int off = sg * PLANE + (item >> 2) * 16;
        *reinterpret_cast<f32x4*>(win_hi + wofs + off) = va;
        *reinterpret_cast<f32x4*>(win_lo + wofs + off) = vb;
        continue;
      }
      const bool pad = gofs.v[k] < 0;                      // zero padding is applied AFTER normalisation + activation
      if (!kPool) {                                        // (wave-uniform conditions only: no per-lane branch; pooled samples were activated at load)
        if (cA.base) va = apply_act4(va, cA.sc, cA.sh, cA.act);
        if (cB.base) vb = apply_act4(vb, cB.sc, cB.sh, cB.act);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {                        // (merged into the ReLU select this costs 28-30 VGPRs and spills the <4,4> tile: kept apart)
        va[j] = pad ? 0.0f : va[j];
        vb[j] = pad ? 0.0f : vb[j];
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
      *reinterpret_cast<f16x8*>(win_hi + wofs + off) = hi;
      *reinterpret_cast<f16x8*>(win_lo + wofs + off) = lo;
    }
  };

  f32x4 acc[TMBW][PG];
#pragma unroll
  for (int b = 0; b < TMBW; ++b)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) acc[b][pg] = f32x4{0.f, 0.f, 0.f, 0.f};
  // LDS offset of tap t inside the window, kept in lane t of one VGPR: the MFMA loop picks it with v_readlane instead of two
  // scalar loads from the argument block and an s_waitcnt lgkmcnt(0) (which also drains the LDS reads in flight) per tap
  const int tl = lane & (kMaxTaps - 1);
  const int my_toff = ((conv_tap_dy(a, cls, tl) - cg.win_dy0) * WW + (conv_tap_dx(a, cls, tl) - cg.win_dx0)) * 16;

#if defined(THA4_PHASE_TIMING) && !defined(THA4_EMU)
  long long* stamps = (a.dbg && a.phase != 2 && blockIdx.z == 0) ? a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWaves + wave) * 64 : nullptr;
  int nstamp = 0;
#define THA4_CSTAMP() do { if (stamps && lane == 0 && nstamp < 64) stamps[nstamp] = clock64(); ++nstamp; } while (0)
#else
#define THA4_CSTAMP()
#endif
  THA4_CSTAMP();                                           // 0: entry (after index set-up)
  int slot = 0;
  if (q_begin < q_end) {
    if (a.fnorm.enabled) {                                 // scale/shift table from the producer's moments (scratch: the window region)
      fused_norm_table(a, n, tid, kThreads, tab_sc, tab_sh, reinterpret_cast<double*>(smem));
      __syncthreads();
    }
    load_window(q_begin, go);
    THA4_CSTAMP();                                         // 1: first window loads issued
    write_window(go, 0);
    THA4_CSTAMP();                                         // 2: first window written
  }
  __syncthreads();
  THA4_CSTAMP();                                           // 3: prologue barrier passed
  for (int Q = q_begin; Q < q_end; ++Q) {
    if (Q + 1 < q_end) load_window(Q + 1, go);
    const int rd = ((Q - q_begin) & 1) ? WB : 0;          // window buffer of this K group
    for (int tc = 0; tc < ntc; ++tc) {
      if (issued < nchunks) {                              // refill the slot freed by the previous barrier
        fetch(issued++, islot);
        islot = islot + 1 == D ? 0 : islot + 1;
      }
      const char* wsl = ring + slot * slot_bytes + lane * 16 + (size_t)(mh * TMBW) * 2048;      // this wave's blocks of every tap
      const int taps_here = min(a.taps_per_chunk, a.ntaps - tc * a.taps_per_chunk);
      // (Register double-buffering across the taps of a chunk - the fragments of tap tt + 1 requested under the MFMAs of tap tt, +38-64 VGPRs, no
      // spill - was measured in round 3 and is NOT it: 161.2 -> 160.1 frames/s steady, 150.0 -> 149.2 cold, tools/runs_r03/gpu_r03_c49.sh.  The second
      // wave of the SIMD already covers the LDS round trip of a tap.)
      for (int tt = 0; tt < taps_here; ++tt) {
        const int t = tc * a.taps_per_chunk + tt;
        const int toff = lane_pick(my_toff, t);
        f16x8 bh[PG], bl[PG];
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
          bh[pg] = *reinterpret_cast<const f16x8*>(win_hi + rd + boff[pg] + toff);
          bl[pg] = *reinterpret_cast<const f16x8*>(win_lo + rd + boff[pg] + toff);
        }
        if (TMBW % 4 == 0) {
          // all A fragments of (four blocks of) the tap are requested together with the B fragments: one LDS wait per tap instead of one per
          // output block (the compiler otherwise reuses one register pair and waits lgkmcnt(0) before every block).  Only for
          // the four- and eight-block tiles: with two blocks the extra registers cost the <2,4> kernels their third wave per SIMD
#pragma unroll
          for (int hb = 0; hb < TMBW; hb += 4) {
            f16x8 ah[4], al[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              ah[b] = *reinterpret_cast<const f16x8*>(wsl + (size_t)(tt * TMB + hb + b) * 2048);
              al[b] = *reinterpret_cast<const f16x8*>(wsl + (size_t)(tt * TMB + hb + b) * 2048 + 1024);
            }
            THA4_SCHED_FENCE();
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
              for (int pg = 0; pg < PG; ++pg) acc[hb + b][pg] = mfma16h(ah[b], bh[pg], acc[hb + b][pg]);
#pragma unroll
              for (int pg = 0; pg < PG; ++pg) acc[hb + b][pg] = mfma16h(ah[b], bl[pg], acc[hb + b][pg]);
#pragma unroll
              for (int pg = 0; pg < PG; ++pg) acc[hb + b][pg] = mfma16h(al[b], bh[pg], acc[hb + b][pg]);
            }
          }
        } else {
#pragma unroll
          for (int b = 0; b < TMBW; ++b) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(wsl + (size_t)(tt * TMB + b) * 2048);
            const f16x8 al = *reinterpret_cast<const f16x8*>(wsl + (size_t)(tt * TMB + b) * 2048 + 1024);
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) acc[b][pg] = mfma16h(ah, bh[pg], acc[b][pg]);
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) acc[b][pg] = mfma16h(ah, bl[pg], acc[b][pg]);
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) acc[b][pg] = mfma16h(al, bh[pg], acc[b][pg]);
          }
        }
      }
      THA4_CSTAMP();                                       // chunk MFMAs issued
      // every wave must (a) be done reading this slot and (b) have ITS pieces of the NEXT chunk in LDS; the chunks fetched
      // after that one (ring deeper than two slots) may stay in flight: at least keep_per_chunk glds per younger chunk.
      // Double-buffered window: the LAST chunk barrier of the K group is deferred behind the window write below
      if (!(WB && tc == ntc - 1)) {
        const int younger = issued - chunk - 2;            // chunks requested after chunk + 1
        const int keep = THA4_TILE_COUNTED_WAIT && younger > 0 ? min(8, younger * keep_per_chunk) : 0;
        THA4_BARRIER_KEEP(keep);
      }
      THA4_CSTAMP();                                       // chunk barrier passed
      slot = slot + 1 == D ? 0 : slot + 1;
      ++chunk;
    }
    // single window: every wave has finished reading window Q (last chunk barrier), K group Q+1 overwrites it behind its own
    // barrier.  Double-buffered: Q+1 goes into the OTHER buffer (nobody reads it during K group Q) before the deferred last
    // chunk barrier, which then also publishes it - one barrier per K group less, and the write overlaps the other waves'
    // last MFMAs.  (ONE call site of write_window: a second one inside the chunk loop cost 90 VGPRs and 17 % fps.)
    THA4_PRIO_VALU();
    if (Q + 1 < q_end) write_window(go, WB ? WB - rd : 0);
    THA4_PRIO_MFMA();
    THA4_CSTAMP();                                         // next window written
    if (WB || Q + 1 < q_end) __syncthreads();
    THA4_CSTAMP();                                         // window barrier passed
  }

  // ---- split-K: phase 1 publishes the partial fragments, phase 2 adds them in split order ----------------
  if (PG < 4 && a.phase != 0) {      // the planner never splits K with the largest tile (its registers are all spoken for)
    const size_t frag = (size_t)NW * PG * 64;                                 // f32x4 fragments per output block (NW pixel slots)
    // partial layout [split][frame][output block][tile][fragment]: indexed by the OUTPUT BLOCK, not by (m-tile, b), so
    // that phase 2 can run with one output block per workgroup (TMB = 1: 4x the workgroups, a quarter of the serial
    // load rounds each) on partials written by a TMB = 4 phase 1
    const size_t stride_split = (size_t)a.batch * a.nb * tiles_per_frame * frag;      // wave-uniform strides (f32x4 units)
    const size_t stride_block = (size_t)tiles_per_frame * frag;
    f32x4* part = reinterpret_cast<f32x4*>(a.partial) + (size_t)cls * a.ksplit * stride_split +      // (merged classes: one partial image per class)
                  (((size_t)n * a.nb + (size_t)mtile * TMB + mh * TMBW) * tiles_per_frame + tile) * frag +
                  (size_t)(pw * PG) * 64 + lane;        // this lane's fragment of this wave's first output block, split 0, pixel group 0
    if (a.phase == 1) {
#pragma unroll
      for (int b = 0; b < TMBW; ++b)
#pragma unroll
        for (int pg = 0; pg < PG; ++pg)
          part[(size_t)ks * stride_split + (size_t)b * stride_block + (size_t)pg * 64] = acc[b][pg];
      THA4_CSTAMP();                                       // partials written
      return;
    }
#pragma unroll
    for (int b = 0; b < TMBW; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) {
        // up to 16 splits: all loads of a fragment in flight at once, added in split order
        f32x4 v[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
          // branch-free: splits past the last one re-read it and are zeroed by a select, so no load waits on a branch
          const int kc = min(k2, a.ksplit - 1);
          const f32x4 ld = part[(size_t)kc * stride_split + (size_t)b * stride_block + (size_t)pg * 64];
          const float keep = k2 < a.ksplit ? 1.0f : 0.0f;
          v[k2] = ld * keep;
        }
        f32x4 s = v[0];
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2) s = s + v[k2];
        acc[b][pg] = s;
        THA4_SCHED_FENCE();                                  // one fragment's 16 loads in flight, not TMB*PG of them
      }
  }

  THA4_CSTAMP();                                           // K loop (and split-K reduction) done
  // ---- epilogue: 1/scale, bias, residual, activation, store, deterministic per-tile statistics ----
  // Three passes (round 4): every residual load of the tile goes out first, then the values are finished in the accumulators, then ALL
  // stores are issued back to back.  The one-fragment-at-a-time form (rounds 1-3) put a load - residual, activation codes - between
  // consecutive stores, and on gfx9 a load's s_waitcnt vmcnt also waits for every older STORE: the 16 fragments of a <4,4> tile went out as
  // 16 dependent store round trips (32.8 k cycles between the last MFMA and the last store in the in-kernel stamps).  Same arithmetic
  // per element, same order of the statistics' additions: bit-identical outputs.
  const int out_px = a.out_h * a.out_w;
  float ssum[TMBW][4], ssq[TMBW][4];
  size_t offs[TMBW][PG];
#pragma unroll
  for (int b = 0; b < TMBW; ++b) {
    const int bo = mtile * TMB + mh * TMBW + b;
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      const int oy = (tile_y0 + ly[pg]) * a.out_sy + cg.out_oy, ox = (tile_x0 + lx[pg]) * a.out_sx + cg.out_ox;
      offs[b][pg] = (((size_t)n * a.nb + bo) * out_px + (size_t)oy * a.out_w + ox) * 16 + g4;
    }
  }
  if (a.residual) {                                        // pass 1: residual values of every fragment requested together
    f32x4 res[TMBW][PG];
#pragma unroll
    for (int b = 0; b < TMBW; ++b) {
      const int bo = mtile * TMB + mh * TMBW + b;
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) {
        res[b][pg] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!inside[pg]) continue;                          // ragged tile: position outside the map
        const int oy = (tile_y0 + ly[pg]) * a.out_sy + cg.out_oy, ox = (tile_x0 + lx[pg]) * a.out_sx + cg.out_ox;
        if (a.res_mode == IN_DIRECT) {
          res[b][pg] = *reinterpret_cast<const f32x4*>(a.residual + offs[b][pg]);
        } else if (a.res_mode == IN_UP2) {      // ResBlock x_resample = Upsample (unet.py:46): nearest
          const int rw = a.out_w >> 1, rpx = out_px >> 2;
          res[b][pg] = *reinterpret_cast<const f32x4*>(a.residual + (((size_t)n * a.nb + bo) * rpx + (size_t)(oy >> 1) * rw + (ox >> 1)) * 16 + g4);
        } else {                                // x_resample = Downsample = AvgPool2d(2,2) (unet.py:58)
          const int rw = a.out_w * 2;
          const float* r0 = a.residual + (((size_t)n * a.nb + bo) * ((size_t)out_px * 4) + (size_t)(2 * oy) * rw + 2 * ox) * 16 + g4;
          res[b][pg] = ((*reinterpret_cast<const f32x4*>(r0) + *reinterpret_cast<const f32x4*>(r0 + 16)) +
                        (*reinterpret_cast<const f32x4*>(r0 + (size_t)rw * 16) + *reinterpret_cast<const f32x4*>(r0 + (size_t)rw * 16 + 16))) * 0.25f;
        }
      }
    }
#pragma unroll
    for (int b = 0; b < TMBW; ++b) {
      const int bo = mtile * TMB + mh * TMBW + b;
      f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + bo * 16 + g4);
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[b][pg] = (acc[b][pg] * a.w16_inv_scale + bias) + res[b][pg];
    }
  } else {
#pragma unroll
    for (int b = 0; b < TMBW; ++b) {
      const int bo = mtile * TMB + mh * TMBW + b;
      f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + bo * 16 + g4);
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[b][pg] = acc[b][pg] * a.w16_inv_scale + bias;
    }
  }
  if (a.act_out) {                                         // head blocks only: per-channel output activations (codes depend on the block, not on the pixel)
#pragma unroll
    for (int b = 0; b < TMBW; ++b) {
      const int bo = mtile * TMB + mh * TMBW + b;
      int codes[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) codes[j] = a.act_out[bo * 16 + g4 + j];
#pragma unroll
      for (int pg = 0; pg < PG; ++pg)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[b][pg][j] = apply_act(acc[b][pg][j], codes[j]);
    }
  }
#pragma unroll
  for (int b = 0; b < TMBW; ++b) {                         // pass 3: stores back to back + the statistics of the positions inside the map
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[b][j] = 0.f; ssq[b][j] = 0.f; }
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      if (!inside[pg]) continue;                            // ragged tile: position outside the map
      if (THA4_HOOK_TILE_EPILOGUE_BYPASS && acc[b][pg][0] != 1.2345e33f) continue;     // tuning builds only
      const f32x4 v = acc[b][pg];
      store16_out(a.out + offs[b][pg], v);
#pragma unroll
      for (int j = 0; j < 4; ++j) { ssum[b][j] += v[j]; ssq[b][j] = fmaf(v[j], v[j], ssq[b][j]); }
    }
  }
  THA4_CSTAMP();                                           // output stores issued
  if (a.stats) {
#pragma unroll
    for (int b = 0; b < TMBW; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = ssum[b][j], q = ssq[b][j];
        s = row16_sum(s, lane);
        q = row16_sum(q, lane);
        if (p == 0) {
          red[((pw * TMB + mh * TMBW + b) * 16 + g4 + j) * 2 + 0] = s;
          red[((pw * TMB + mh * TMBW + b) * 16 + g4 + j) * 2 + 1] = q;
        }
      }
    THA4_BARRIER_LDS();        // LDS only: the output stores above stay in flight (a full __syncthreads would wait for every store's acknowledgement)
    for (int i = tid; i < TMB * 16; i += kThreads) {
      float s = 0.f, q = 0.f;
      for (int wv2 = 0; wv2 < NW; ++wv2) {
        s += red[((wv2 * TMB) * 16 + i) * 2 + 0];
        q += red[((wv2 * TMB) * 16 + i) * 2 + 1];
      }
      float* dst = a.stats + ((((size_t)n * a.stats_tiles + cg.stats_tile0 + tile) * a.nb + mtile * TMB) * 16 + i) * 2;
      dst[0] = s;
      dst[1] = q;
      // ... and into the tensor's moment accumulators (full_kernels.h MomentAcc): the consumer's normalisation needs no finalize launch.  The shard is a
      // function of the workgroup id only, so each shard's (integer) sum is the same whatever the order of arrival
      if (a.stats_acc) moment_acc_add(a.stats_acc + (((size_t)n * kMomentShards + ((int)blockIdx.x & (kMomentShards - 1))) * a.nb + mtile * TMB) * 16 + i, s, q, a.acc_fault);
    }
  }
  THA4_CSTAMP();                                           // statistics written: end of the kernel
}

}  // namespace tha4
