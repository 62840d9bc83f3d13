// Host-side layout helpers for the full-model kernels (plain C++, shared with the CPU unit tests).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

namespace tha4 {

// Launch-plan knobs (THA4_WANT_WGS, THA4_KSPLIT_MAX, THA4_TILE_TAPS_MAX, ...) are tuning aids of tools/sweep_plan_knobs.sh: they are
// honoured only when THA4_TUNING is set, so that a stray variable in a production environment cannot change the plan that the
// parity tests pinned.  (Diagnostics that only print - THA4_DUMP_SCHEDULE, THA4_DBG_CONV - read the environment directly.)
inline const char* tune_env(const char* name) {
  static const bool on = std::getenv("THA4_TUNING") != nullptr;
  return on ? std::getenv(name) : nullptr;
}

constexpr int kMaxTapsHost = 16;

// One launch of conv_mfma_kernel in "tile coordinates": which taps it visits and how tile
// coordinates map to input / output pixels.
struct ConvGeom {
  int ntaps = 0;
  int ky[kMaxTapsHost], kx[kMaxTapsHost];   // kernel element used by the tap
  int dy[kMaxTapsHost], dx[kMaxTapsHost];   // virtual input coord = tile*in_stride + d
  int in_stride = 1;
  int out_sy = 1, out_sx = 1, out_oy = 0, out_ox = 0;
};

// Conv2d(k, stride 1, padding k/2)            (conv.py:33-41 conv3, conv1)
inline ConvGeom geom_conv_same(int k) {
  ConvGeom g;
  for (int y = 0; y < k; ++y)
    for (int x = 0; x < k; ++x) {
      g.ky[g.ntaps] = y; g.kx[g.ntaps] = x; g.dy[g.ntaps] = y - k / 2; g.dx[g.ntaps] = x - k / 2; ++g.ntaps;
    }
  return g;
}

// Conv2d(4, stride 2, padding 1)              (conv.py:127-147)
inline ConvGeom geom_conv4_s2() {
  ConvGeom g;
  g.in_stride = 2;
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x) {
      g.ky[g.ntaps] = y; g.kx[g.ntaps] = x; g.dy[g.ntaps] = y - 1; g.dx[g.ntaps] = x - 1; ++g.ntaps;
    }
  return g;
}

// ConvTranspose2d(4, stride 2, padding 1) (conv.py:164-177) as four 2x2 gather convolutions, one per
// output parity class (py, px): out(2i+py, 2j+px) = sum_{a,b} in(i+dy_a, j+dx_b) W[ci][co][ky_a][kx_b]
//   parity 0: (k=1, d=0), (k=3, d=-1)      parity 1: (k=0, d=+1), (k=2, d=0)
inline ConvGeom geom_convT4_s2(int py, int px) {
  static const int kk[2][2] = {{1, 3}, {0, 2}};
  static const int dd[2][2] = {{0, -1}, {1, 0}};
  ConvGeom g;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      g.ky[g.ntaps] = kk[py][a]; g.kx[g.ntaps] = kk[px][b];
      g.dy[g.ntaps] = dd[py][a]; g.dx[g.ntaps] = dd[px][b];
      ++g.ntaps;
    }
  g.out_sy = 2; g.out_sx = 2; g.out_oy = py; g.out_ox = px;
  return g;
}

struct ChannelSegment {   // a run of input channels of the weight tensor mapped to one conv source
  int offset;             // first input channel in the weight tensor
  int count;              // real channels (padded up to a multiple of 16 in the packed image)
};

// Pack a convolution weight for conv_mfma_kernel<TMB,...>:  P[mtile][q][tap][b<TMB][lane][j]
//   = W[o = 16*(mtile*TMB+b) + (lane&15)][i = seg channel 16*ql + 4*(lane>>4) + j][ky][kx]
// W is [cout][cin][kh][kw] (Conv2d) or, when transposed, [cin][cout][kh][kw] (ConvTranspose2d).
inline std::vector<float> pack_conv_weight(const float* W, int cout, int cin, int kh, int kw, bool transposed,
                                           const ConvGeom& g, const std::vector<ChannelSegment>& segs, int TMB) {
  const int nb = (cout + 15) / 16;
  const int mtiles = (nb + TMB - 1) / TMB;
  int cbtot = 0;
  for (auto& s : segs) cbtot += (s.count + 15) / 16;
  std::vector<float> P((size_t)mtiles * cbtot * g.ntaps * TMB * 256, 0.f);
  for (int mt = 0; mt < mtiles; ++mt) {
    int q = 0;
    for (auto& s : segs) {
      const int cb = (s.count + 15) / 16;
      for (int ql = 0; ql < cb; ++ql, ++q)
        for (int t = 0; t < g.ntaps; ++t)
          for (int b = 0; b < TMB; ++b)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 4; ++j) {
                const int o = 16 * (mt * TMB + b) + (lane & 15);
                const int il = 16 * ql + 4 * (lane >> 4) + j;
                float v = 0.f;
                if (o < cout && il < s.count) {
                  const int i = s.offset + il;
                  const size_t idx = transposed ? (((size_t)i * cout + o) * kh + g.ky[t]) * kw + g.kx[t]
                                                : (((size_t)o * cin + i) * kh + g.ky[t]) * kw + g.kx[t];
                  v = W[idx];
                }
                P[(((((size_t)mt * cbtot + q) * g.ntaps + t) * TMB + b) * 64 + lane) * 4 + j] = v;
              }
    }
  }
  return P;
}

// fp16 hi/lo weight image for conv_tile_kernel<TMB,...>: 2 KiB pieces [mtile][Q][tap][b<TMB][hi: lane x 8 | lo: lane x 8],
// K group Q = quads 2Q, 2Q+1 of the concatenated sources; k-slot j of lane group g = lane>>4 is channel
// 4g + (j&3) of quad 2Q + (j>>2).  Weights are multiplied by a power of two S (max |W| S in [8192, 16384)) before the
// split so that the low halves of small weights stay clear of the fp16 subnormal range; *inv_scale = 1/S.
inline std::vector<char> pack_conv_weight16(const float* W, int cout, int cin, int kh, int kw, bool transposed, const ConvGeom& g,
                                            const std::vector<ChannelSegment>& segs, int TMB, float* inv_scale) {
  const int nb = (cout + 15) / 16;
  const int mtiles = (nb + TMB - 1) / TMB;
  std::vector<std::pair<int, int>> quads;     // (segment, local quad)
  for (size_t s = 0; s < segs.size(); ++s)
    for (int ql = 0; ql < (segs[s].count + 15) / 16; ++ql) quads.push_back({(int)s, ql});
  const int NQ = ((int)quads.size() + 1) / 2;
  float mx = 0.f;
  for (size_t i = 0; i < (size_t)cout * cin * kh * kw; ++i) mx = std::max(mx, std::fabs(W[i]));
  int e = 0;
  if (mx > 0.f) { std::frexp(16384.0f / mx, &e); e -= 1; }     // 2^e <= 16384 / mx < 2^(e+1)
  e = std::min(std::max(e, -24), 24);
  const float S = std::ldexp(1.0f, e);
  *inv_scale = std::ldexp(1.0f, -e);
  std::vector<char> P((size_t)mtiles * NQ * g.ntaps * TMB * 2048, 0);
  for (int mt = 0; mt < mtiles; ++mt)
    for (int Q = 0; Q < NQ; ++Q)
      for (int t = 0; t < g.ntaps; ++t)
        for (int b = 0; b < TMB; ++b) {
          _Float16* hi = reinterpret_cast<_Float16*>(P.data() + ((((size_t)mt * NQ + Q) * g.ntaps + t) * TMB + b) * 2048);
          _Float16* lo = hi + 512;
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int q = 2 * Q + (j >> 2);
              float v = 0.f;
              if (q < (int)quads.size()) {
                const ChannelSegment& sg = segs[quads[q].first];
                const int il = 16 * quads[q].second + 4 * (lane >> 4) + (j & 3);
                const int o = 16 * (mt * TMB + b) + (lane & 15);
                if (o < cout && il < sg.count) {
                  const int i = sg.offset + il;
                  const size_t idx = transposed ? (((size_t)i * cout + o) * kh + g.ky[t]) * kw + g.kx[t]
                                                : (((size_t)o * cin + i) * kh + g.ky[t]) * kw + g.kx[t];
                  v = W[idx] * S;
                }
              }
              const _Float16 h = (_Float16)v;
              hi[lane * 8 + j] = h;
              lo[lane * 8 + j] = (_Float16)(v - (float)h);
            }
        }
  return P;
}

// Workgroup tiling of conv_tile_kernel<TMB, PG>: 8 waves x PG pixel groups of 16 positions, tile = th x 2^tw_log2
// positions; the tile grid may overhang the map (masked), `efficiency` is the fraction of computed positions kept.
struct TileGeom {
  bool ok = false;
  int tw_log2 = 4, th = 0;
  int tiles = 0;                 // tiles per frame (and per parity class)
  float efficiency = 0.f;
  int win_h = 0, win_w = 0, dy0 = 0, dx0 = 0;
  int taps_per_chunk = 1;
  int ring_slots = 2;            // LDS weight ring depth: as deep (<= 4) as the LDS left by the window allows
  int win_buffers = 1;           // 2: the window is double-buffered (one barrier per K group less)
  size_t lds = 0;
};
// nw: pixel-slot waves of the workgroup - 8 (one workgroup per CU, up to 160 KiB of LDS) or 4 (conv_tile_kernel<..., NW = 4>: half the
// tile, at most 80 KiB so that two workgroups share a CU)
inline TileGeom tile_geom(const ConvGeom& g, int tile_h, int tile_w, int PG, int TMB, int tw_log2, size_t extra_lds = 0, int nw = 8) {
  TileGeom t;
  const int px = nw * PG * 16;
  const size_t lds_cap = nw == 8 ? 160 * 1024 : 80 * 1024;
  t.tw_log2 = tw_log2;
  const int tw = 1 << tw_log2;
  t.th = px / tw;
  if (t.th < 1) return t;
  t.tiles = ((tile_h + t.th - 1) / t.th) * ((tile_w + tw - 1) / tw);
  t.efficiency = (float)(tile_h * tile_w) / (float)(t.tiles * px);
  int dy_lo = g.dy[0], dy_hi = g.dy[0], dx_lo = g.dx[0], dx_hi = g.dx[0];
  for (int i = 1; i < g.ntaps; ++i) {
    dy_lo = std::min(dy_lo, g.dy[i]); dy_hi = std::max(dy_hi, g.dy[i]);
    dx_lo = std::min(dx_lo, g.dx[i]); dx_hi = std::max(dx_hi, g.dx[i]);
  }
  t.dy0 = dy_lo; t.dx0 = dx_lo;
  t.win_h = (t.th - 1) * g.in_stride + (dy_hi - dy_lo) + 1;
  t.win_w = (tw - 1) * g.in_stride + (dx_hi - dx_lo) + 1;
  const int npx = t.win_h * t.win_w;
  if (npx * 4 > (nw == 8 ? 5 * 512 : 6 * 256)) return t;      // kTileMaxItems (kTileMaxItemsHalf) staging items per thread
  t.taps_per_chunk = 1;
  // One chunk barrier per streamed weight chunk: whole K groups (9 taps of a 3x3, 8 of the 16 of a 4x4) per chunk where two
  // ring slots of that size fit beside the window, else up to 4 taps / 32 KiB (measured: 155.9 -> 158.4 fps, batch 8 294.7 -> 297.3)
  const int taps_max = tune_env("THA4_TILE_TAPS_MAX") ? std::atoi(tune_env("THA4_TILE_TAPS_MAX")) : 9;            // tuning aid
  const int slot_max = tune_env("THA4_TILE_SLOT_MAX_KB") ? std::atoi(tune_env("THA4_TILE_SLOT_MAX_KB")) * 1024 : 72 * 1024;
  const size_t plane = (size_t)(npx * 16 + 127) / 128 * 128 + 32;
  const size_t red = (size_t)nw * TMB * 16 * 2 * sizeof(float);
  // the largest chunk that still leaves room for two ring slots; chunks need not divide the taps (9 = 5 + 4: two barriers
  // instead of three for the four-block tiles), but a chunk size is only taken if it lowers the chunk count
  const bool uneven = tune_env("THA4_TILE_EVEN_CHUNKS") == nullptr;     // tuning aid (measured: 157.15 -> 157.75 fps with 5 + 4)
  int best_chunks = g.ntaps + 1;
  for (int d = 1; d <= g.ntaps; ++d) {
    const int chunks = (g.ntaps + d - 1) / d;
    if ((uneven || g.ntaps % d == 0) && d <= taps_max && (size_t)d * TMB * 2048 <= (size_t)slot_max && chunks < best_chunks &&
        (nw == 8 || 8 * plane + 2 * (size_t)d * TMB * 2048 + red + extra_lds <= lds_cap) &&
        ((d <= 4 && (size_t)d * TMB * 2048 <= 32 * 1024) || 8 * plane + 2 * (size_t)d * TMB * 2048 + red + extra_lds <= 160 * 1024)) {
      t.taps_per_chunk = d;
      best_chunks = chunks;
    }
  }
  const size_t slot = (size_t)t.taps_per_chunk * TMB * 2048;
  // deeper while it costs no occupancy: never push a workgroup that fits twice on a CU (<= 80 KiB) over that line
  t.ring_slots = 2;
  // second window buffer where it fits beside two ring slots (PG <= 2 tiles)
  t.win_buffers = (!tune_env("THA4_NO_DOUBLE_WINDOW") && 16 * plane + 2 * slot + red + extra_lds <= lds_cap) ? 2 : 1;
  const size_t win = 8 * plane * t.win_buffers;
  const size_t base = win + 2 * slot + red + extra_lds;           // extra_lds: scale/shift table of a fused normalisation
  const size_t cap = base <= 80 * 1024 ? 80 * 1024 : lds_cap;
  while (t.ring_slots < 4 && win + (t.ring_slots + 1) * slot + red + extra_lds <= cap) ++t.ring_slots;
  t.lds = win + t.ring_slots * slot + red + extra_lds;
  t.ok = t.lds <= lds_cap;
  return t;
}

// Launch plan: pixel groups per wave, tile shape and K split for one convolution (all parity classes share it).
//   want_wgs: workgroups needed to fill the chip; nq: 32-channel K groups; mtiles: output-channel tiles
struct TilePlan {
  bool ok = false;
  int pg = 1, ksplit = 1;
  TileGeom geom;
};
// `frames`: frames per call the schedule is built for (max_batch): every frame brings its own tiles, so a batched call fills
// the chip without a K split where a single frame needs one
inline TilePlan plan_tile_conv(const ConvGeom& g, int tile_h, int tile_w, int TMB, int mtiles, int nq, int want_wgs = 256, int frames = 1) {
  TilePlan best;
  if (tune_env("THA4_WANT_WGS")) want_wgs = std::atoi(tune_env("THA4_WANT_WGS"));   // tuning aid
  const int kmax = tune_env("THA4_KSPLIT_MAX") ? std::atoi(tune_env("THA4_KSPLIT_MAX")) : 16;   // tuning aid
  const int min_nq = tune_env("THA4_KSPLIT_MIN_NQ") ? std::atoi(tune_env("THA4_KSPLIT_MIN_NQ")) : 0;   // tuning aid
  float best_eff = 0.f;
  for (int pg : {4, 2, 1})
    for (int twl : {5, 4, 3}) {
      const TileGeom t = tile_geom(g, tile_h, tile_w, pg, TMB, twl);
      if (t.ok) best_eff = std::max(best_eff, t.efficiency);
    }
  if (best_eff == 0.f) return best;
  // 1) no K split: the largest tile that still gives every CU a workgroup, else the smallest tile if that at least
  //    half-fills the chip (a K split costs ksplit x the output in partial traffic plus a second launch)
  auto pick = [&](int pg) {
    for (int twl : {4, 5, 3}) {
      const TileGeom t = tile_geom(g, tile_h, tile_w, pg, TMB, twl);
      if (t.ok && t.efficiency >= 0.9f * best_eff) { best.ok = true; best.pg = pg; best.geom = t; return true; }
    }
    return false;
  };
  const long F = frames < 1 ? 1 : frames;
  for (int pg : {4, 2, 1})
    if (pick(pg) && best.geom.tiles * mtiles * F >= want_wgs) return best;
  if (pick(1) && best.geom.tiles * mtiles * F >= want_wgs / 2) return best;
  if (pick(2) && best.geom.tiles * mtiles * F >= want_wgs / 2) return best;
  // 2) small maps: the largest tile whose K groups can still be spread over the chip
  best.ok = false;
  for (int pg : {2, 1})            // K split is compiled out of the PG = 4 kernel
    if (pick(pg) && (long)best.geom.tiles * mtiles * F * std::min(nq, kmax) >= want_wgs) break;
  if (!best.ok) return best;
  const int wgs = (int)(best.geom.tiles * mtiles * F);
  int ksplit = 1;
  if (wgs < want_wgs / 2 && nq >= min_nq && best.pg < 4) {
    const int want = std::min(std::min(nq, kmax), (want_wgs + wgs - 1) / wgs);
    const int per = (nq + want - 1) / want;
    ksplit = (nq + per - 1) / per;                     // every split gets at least one K group
  }
  best.ksplit = ksplit;
  return best;
}

// Launch plan of conv_point_kernel<TMB, PG> (full_conv_point_kernels.h): a workgroup = 4 waves x PG pixel groups of 16
// consecutive pixels x TMB output blocks.  The largest register tile per wave (fewest LDS reads per MFMA) whose grid still
// gives every CU a workgroup; with a folded normalisation the reduction scratch (2 doubles per padded input channel) must
// fit one ring slot and fused_norm_table() handles at most 2 channels per thread.
struct PointPlan {
  bool ok = false;
  int tmb = 1, pg = 1, tiles = 0;
};
inline PointPlan plan_point_conv(int px, int nb, int cbtot, int frames, bool fused, int want_wgs = 256) {
  PointPlan best;
  const long F = frames < 1 ? 1 : frames;
  long best_wgs = -1;
  const int cand[6][2] = {{4, 2}, {2, 2}, {4, 1}, {2, 1}, {1, 2}, {1, 1}};
  for (auto& c : cand) {
    const int tmb = c[0], pg = c[1];
    if (nb % tmb) continue;
    if (fused && ((size_t)cbtot * 16 * 16 > (size_t)4 * tmb * 2048 || cbtot * 16 > 512)) continue;
    const int tiles = (px + 64 * pg - 1) / (64 * pg);
    const long wgs = (long)tiles * (nb / tmb) * F;
    if (wgs >= want_wgs) { best.ok = true; best.tmb = tmb; best.pg = pg; best.tiles = tiles; return best; }
    if (wgs > best_wgs) { best_wgs = wgs; best.ok = true; best.tmb = tmb; best.pg = pg; best.tiles = tiles; }
  }
  return best;
}

// Launch plan of conv_small_kernel<PG> (full_conv_small_kernels.h): a workgroup = 16*PG output positions (TH x 2^twl) x one
// output block, its 8 waves split the K groups (x tap ranges when there are fewer than 8 groups).
struct SmallPlan {
  bool ok = false;
  int pg = 1, tw_log2 = 4, th = 0;
  int tiles = 0;                 // pixel tiles per frame (and per parity class)
  float efficiency = 0.f;
  int win_h = 0, win_w = 0, dy0 = 0, dx0 = 0;
  int units_per_q = 1;
  size_t lds = 0;                // without the fused-norm table
};
inline SmallPlan small_geom(const ConvGeom& g, int tile_h, int tile_w, int pg, int tw_log2) {
  SmallPlan t;
  t.pg = pg; t.tw_log2 = tw_log2;
  const int tw = 1 << tw_log2;
  t.th = 16 * pg / tw;
  if (t.th < 1 || t.th * tw != 16 * pg) return t;
  t.tiles = ((tile_h + t.th - 1) / t.th) * ((tile_w + tw - 1) / tw);
  t.efficiency = (float)(tile_h * tile_w) / (float)(t.tiles * 16 * pg);
  int dy_lo = g.dy[0], dy_hi = g.dy[0], dx_lo = g.dx[0], dx_hi = g.dx[0];
  for (int i = 1; i < g.ntaps; ++i) {
    dy_lo = std::min(dy_lo, g.dy[i]); dy_hi = std::max(dy_hi, g.dy[i]);
    dx_lo = std::min(dx_lo, g.dx[i]); dx_hi = std::max(dx_hi, g.dx[i]);
  }
  t.dy0 = dy_lo; t.dx0 = dx_lo;
  t.win_h = (t.th - 1) * g.in_stride + (dy_hi - dy_lo) + 1;
  t.win_w = (tw - 1) * g.in_stride + (dx_hi - dx_lo) + 1;
  const int npx = t.win_h * t.win_w;
  if (npx * 4 > 7 * 64) return t;                      // kSmallMaxItems staging items per lane
  const size_t plane = (size_t)(npx * 16 + 127) / 128 * 128 + 32;
  t.lds = 64 * plane;                                  // 8 waves x (4 hi + 4 lo planes); the 8*PG KiB reduction buffer aliases them
  t.ok = true;
  return t;
}
inline SmallPlan plan_small_conv(const ConvGeom& g, int tile_h, int tile_w, int nb, int nq, int max_wgs = 256, int frames = 1) {
  SmallPlan best;
  float best_eff = 0.f;
  for (int pg : {4, 2, 1})
    for (int twl : {4, 3, 2}) {
      const SmallPlan t = small_geom(g, tile_h, tile_w, pg, twl);
      if (t.ok) best_eff = std::max(best_eff, t.efficiency);
    }
  if (best_eff == 0.f) return best;
  // One workgroup per CU is all this kernel's registers allow, so a grid larger than the chip runs in rounds (288
  // workgroups cost twice 256): take the tile size that comes closest to `max_wgs` from below; maps that cannot be
  // covered in one round stay on conv_tile_kernel's two-launch K split (measured: profiles/r02_full_b1_reading.md)
  auto pick = [&](int pg) {
    SmallPlan b;
    for (int twl : {4, 3, 2}) {
      const SmallPlan t = small_geom(g, tile_h, tile_w, pg, twl);
      if (t.ok && t.efficiency >= 0.9f * best_eff && (!b.ok || t.win_h * t.win_w < b.win_h * b.win_w)) b = t;
    }
    return b;
  };
  const int cap = tune_env("THA4_SMALL_MAX_WGS") ? std::atoi(tune_env("THA4_SMALL_MAX_WGS")) : max_wgs;   // tuning aid
  const bool big_first = tune_env("THA4_SMALL_BIG_FIRST") != nullptr;                                        // tuning aid
  for (int i = 0; i < 3; ++i) {
    const int pg = big_first ? (4 >> i) : (1 << i);
    const SmallPlan t = pick(pg);
    if (!t.ok || (long)t.tiles * nb * (frames < 1 ? 1 : frames) > cap) continue;
    best = t;
    break;                                             // pg ascending = workgroup count descending: the first fit is the largest
  }
  if (!best.ok) return best;
  int upq = 1;
  while (nq * upq < 8 && upq * 2 <= g.ntaps) upq *= 2;  // few K groups: split each one's taps over several waves
  best.units_per_q = upq;
  return best;
}

// NCHW [c][h*w] <-> C16 [cb][h*w][16] for one frame
inline void nchw_to_c16(const float* src, int c, int px, float* dst) {
  const int cb = (c + 15) / 16;
  std::memset(dst, 0, sizeof(float) * (size_t)cb * px * 16);
  for (int ch = 0; ch < c; ++ch)
    for (int i = 0; i < px; ++i) dst[((size_t)(ch >> 4) * px + i) * 16 + (ch & 15)] = src[(size_t)ch * px + i];
}
inline void c16_to_nchw(const float* src, int c, int px, float* dst) {
  for (int ch = 0; ch < c; ++ch)
    for (int i = 0; i < px; ++i) dst[(size_t)ch * px + i] = src[((size_t)(ch >> 4) * px + i) * 16 + (ch & 15)];
}

}  // namespace tha4
