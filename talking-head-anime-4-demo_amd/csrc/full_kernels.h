// gfx950 kernels for the full THA4 system (reference mode_07: conv encoder-decoders + diffusion-style U-Nets).
//
// Feature maps live in HBM in the "C16" layout  T[n][cb][pixel][16]  (cb = channel block of 16,
// pixel = row-major y*W+x, zero padded channels).  A lane's 16-byte load at (cb, pixel, 4g) is exactly
// the B fragment of v_mfma_f32_16x16x4_f32 for pixel column p = lane&15, k-steps 4g..4g+3, so the
// implicit-GEMM convolution reads its im2col operand straight from L1/L2 with no LDS staging, applying
// the producer's normalisation (per-(n,c) scale/shift), activation and x2 resampling on the fly.
// Weights use the same fragment-linear image as the student (siren_layout.h), streamed through a 2-slot
// LDS ring with global_load_lds.  Everything is exact fp32 (v_mfma_f32_16x16x4_f32).
//
// Reference ops covered (paths relative to /root/reference/src/tha4/nn):
//   conv.py:103-177            conv3 / conv4 s2 / convT4 s2 (+ InstanceNorm + ReLU applied by the CONSUMER)
//   resnet_block.py:52-67      residual add in the epilogue
//   common/unet.py:33-62,154-165  nearest-up / avg-pool on both ResBlock branches, GroupNorm+FiLM+SiLU on load
//   common/unet.py:192-239     attention (qkv/proj are 1-tap convs; core in attention_kernel)
#pragma once
#include "tha4_platform.h"

namespace tha4 {

enum : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_SIGMOID = 3, ACT_TANH = 4 };
enum : int { IN_DIRECT = 0, IN_UP2 = 1, IN_POOL2 = 2 };
enum : int { SRC_TENSOR = 0, SRC_VECTOR = 1 };

constexpr int kMaxTaps = 16;

struct ConvSrc {
  const float* data;    // SRC_TENSOR: C16 [n][cb][in_h*in_w][16];  SRC_VECTOR: [n][cb*16] (spatially constant)
  const float* scale;   // [n][cb*16] or null (identity)
  const float* shift;   // [n][cb*16] or null
  int cb;               // channel blocks of this source
  int kind;
  int act;              // activation applied after scale/shift (before zero padding / pooling)
};

// Numeric-fault detection (tha4_hip.h THA4_ERR_NUMERIC_RANGE).  The convolutions stage their operands as UNSCALED fp16 hi + lo
// halves: a normalised + activated value of |v| >= 65520 becomes hi = inf, lo = -inf and every accumulator it touches NaN.
// Such a fault (or a NaN / inf coming from the weights) always reaches the moments of the next normalisation or a network's
// head block, so THOSE are checked - per channel in the normalisation arithmetic, per pixel in the image tails - and the hot
// loops carry no range check.  The flag lives in pinned host memory mapped into the device: set once, read by the host at
// the next tha4_full_pose / tha4_full_numeric_status.
THA4_DEV void report_fault_unless_finite(int* fault, float a, float b) {
  if (fault && !(fabsf(a) < __builtin_inff() && fabsf(b) < __builtin_inff())) *fault = 1;
}

// Normalisation folded into the consumer (full_conv_small_kernels.h fused_norm_table): instead of reading per-(n, c)
// scale/shift vectors produced by norm_finalize_kernel, the convolution reduces the producer's per-tile moments itself.
struct FusedNorm {
  const float* stats[2];   // per source: partial sums [n][tiles][cb*16][2] written by the producer's epilogue
  int tiles[2];
  int channels;            // real channels of the (concatenated) normalised tensor
  int groups;              // 0: InstanceNorm2d, > 0: GroupNorm(groups)
  float inv_count;         // 1 / pixels per channel
  float eps;
  const float* gamma;      // [channels]
  const float* beta;
  const float* film0;      // [2*channels] constant (scale | shift) or null
  const float* film1;      // [n][film1_stride] per-frame (scale | shift) or null
  long long film1_stride;
  int enabled;
  int acc;                 // 1: stats[] are MomentAcc accumulators [n][kMomentShards][cb*16] filled by the producers' atomics (tiles[] = kMomentShards), not per-tile floats
  int* fault;              // sticky numeric-fault flag of the handle (a non-finite scale/shift sets it), or null
};

// Moments of a whole tensor accumulated by its PRODUCER with integer atomics (round 5): the per-(frame, channel) sums of x and x^2 that a normalisation needs used
// to be per-tile partial sums reduced by a separate norm_finalize_kernel launch whenever a tensor has more than 64 tiles (48 launches of 5.5 us on the
// dependency chain of a batch-1 frame).  Here every workgroup of the producing convolution adds its tile's sums to one of kMomentShards accumulators per
// channel, and the CONSUMER folds the 8 shards into its scale / shift table like it folds per-tile moments (fused_norm_table).  A sum travels as a pair of
// 64-bit integers - hi = trunc(x), lo = rint((x - hi) 2^32) - so that the result does not depend on the order in which workgroups arrive (integer addition
// is associative: a frame's bytes stay reproducible) and small tensors keep a 2^-32 absolute resolution per tile sum while large ones have the range of an
// int64 (|sum| < 9e15 is checked: beyond it - or NaN - the numeric-fault flag is raised like for a non-finite scale / shift).
constexpr int kMomentShards = 8;
struct MomentAcc { long long s_hi, s_lo, q_hi, q_lo; };      // 32 bytes per (frame, shard, channel)
THA4_DEV void moment_acc_add(MomentAcc* slot, float s, float q, int* fault) {
  const double ds = (double)s, dq = (double)q;
  if (!(fabs(ds) < 9.0e15 && fabs(dq) < 9.0e15)) {         // NaN / inf / out of the accumulator's range
    if (fault) *fault = 1;
    return;
  }
  const double hs = trunc(ds), hq = trunc(dq);
  long long* w = reinterpret_cast<long long*>(slot);
  atomic_add_i64(w + 0, (long long)hs);
  atomic_add_i64(w + 1, (long long)rint((ds - hs) * 4294967296.0));
  atomic_add_i64(w + 2, (long long)hq);
  atomic_add_i64(w + 3, (long long)rint((dq - hq) * 4294967296.0));
}
THA4_DEV void moment_acc_read(const MomentAcc* slot, double& s, double& q) {
  const long long* w = reinterpret_cast<const long long*>(slot);
  s = (double)w[0] + (double)w[1] * (1.0 / 4294967296.0);
  q = (double)w[2] + (double)w[3] * (1.0 / 4294967296.0);
}

// Division by a launch constant: q = (x * m) >> 42 with m = ceil(2^42 / d) - exact for 0 <= x < 2^22 and 1 <= d < 2^20 (x * (m d - 2^42) < x d < 2^42; x * m < 2^64), a handful of
// instructions where the compiler's sequence for a run-time divisor is ~30.  In-kernel stamps (profiles/r04_raw/c37_phase_prologue.txt) put 2.3 k cycles of
// SCALAR work - the decomposition of blockIdx.x, eight such divisions - in front of a conv_small_kernel wave's first request, another ~2 k of per-lane divisions
// behind it: every divisor of the prologues is a constant of the launch, computed on the host (finish_conv_args / finish_conv_batch).  The range covers the largest
// batch-sized divisor of the documented handle limit (max_batch = 256 x 1024 tiles of a 512x512 map = 2^18; round 4's 2^40 / d < 2^18 form refused exactly that plan).
constexpr int kFastDivShift = 42;
struct FastDiv { unsigned long long m; int d; int pad_; };
inline bool fastdiv_make(FastDiv& f, long long d) {
  if (d < 1 || d >= (1 << 20)) return false;
  f.d = (int)d; f.pad_ = 0;
  f.m = ((1ull << kFastDivShift) + (unsigned long long)d - 1) / (unsigned long long)d;
  return true;
}
THA4_DEV int fast_div(int x, const FastDiv& f) { return (int)(((unsigned long long)(unsigned)x * f.m) >> kFastDivShift); }

struct ConvArgs {
  ConvSrc src[2];
  int nsrc;
  int in_h, in_w;        // stored spatial size of the tensor sources
  int in_mode;           // IN_DIRECT | IN_UP2 (virtual 2h x 2w, nearest) | IN_POOL2 (virtual h/2 x w/2, mean of 2x2); must equal the template INMODE
  int ntaps;
  signed char tap_dy[kMaxTaps], tap_dx[kMaxTaps];   // virtual input coord = tile coord * in_stride + d (bytes since round 5: the argument block must stay within
                                                    // the lines one batch of scalar loads warms at kernel entry - warm_kernarg, tests/test_api_surface.py)
  int in_stride;
  int tile_h, tile_w;    // grid of output positions computed by this launch (per frame)
  int out_h, out_w;      // stored output size; output coord = tile * out_s + out_o
  int out_sy, out_sx, out_oy, out_ox;
  const float* w;        // packed [mtile][q][tap][TMB][64][4]
  const float* bias;     // [nb*16] or null
  const float* residual; // C16 [n][nb][res_h*res_w][16] added before the output activation, or null
  int res_mode;          // IN_DIRECT: same size as out | IN_UP2: nearest x2 of a half-size tensor | IN_POOL2: 2x2 mean of a double-size tensor
  const int* act_out;    // per-output-channel activation codes [nb*16] or null (none)
  float* out;            // C16 [n][nb][out_h*out_w][16]
  float* stats;          // partial sums [n][stats_tiles][nb*16][2] (sum, sum of squares) or null
  MomentAcc* stats_acc;  // conv_tile_kernel / conv_small_kernel: moment accumulators [n][kMomentShards][nb*16] this launch adds its tiles' sums to (MomentAcc above), or null
  int* acc_fault;        // numeric-fault flag for sums the accumulators cannot hold
  int stats_tiles;       // tiles per frame in the stats buffer (several launches may fill one buffer: convT parity classes)
  int stats_tile0;       // first tile index written by this launch
  int nb;                // output channel blocks
  int chunk_quads;       // input quads per streamed weight chunk
  int batch;
  // conv_tile_kernel (full_conv16_kernels.h) only
  const void* w16;       // fp16 hi/lo pieces [mtile][K group][tap][TMB][hi 1 KiB | lo 1 KiB] (full_layout.h pack_conv_weight16)
  float w16_inv_scale;   // 1 / (power-of-two scale folded into the fp16 weights)
  int wg_tw_log2;        // log2 of the workgroup tile width (tile positions)
  int win_h, win_w;      // staged input window (virtual input pixels)
  int win_dy0, win_dx0;  // window origin relative to (tile origin * in_stride)
  int taps_per_chunk;    // taps per streamed weight chunk (divides ntaps)
  int ring_slots;        // depth of the LDS weight ring (2..4)
  int win_buffers;       // 1: one LDS window, rewritten behind its own barrier; 2: double-buffered (the write of K group Q+1 precedes Q's last chunk barrier)
  float* partial;        // split-K workspace [ksplit][batch][mtile][tile][TMB][8*PG][64] f32x4, or null
  int ksplit;            // K splits (phase 1 grid z)
  int phase;             // 0: whole convolution; 1: partial products of K split blockIdx.z only; 2: reduce partials + epilogue
  FusedNorm fnorm;       // conv_tile_kernel / conv_small_kernel: normalisation of the tensor sources computed in the prologue
  int units_per_q;       // conv_small_kernel: tap ranges per K group (1, 2, 4 or 8: spreads few K groups over the 8 waves)
  // conv_tile_kernel / conv_small_kernel: the four output-parity classes of a ConvTranspose2d(4, stride 2, padding 1) in ONE launch
  // (round 4).  nclass = 4: the class is the slowest part of blockIdx.x; its taps, window origin, output offsets and statistics tiles
  // follow from the class id (conv_class below = full_layout.h geom_convT4_s2), its weights are `w16_class_bytes` further on.  nclass <= 1:
  // one class, geometry from the fields above.
  int nclass;
  long long w16_class_bytes;
  // launch constants of the tile / small kernels' prologues, computed on the host (finish_conv_args: plan time; finish_conv_batch: per call)
  int tiles_x, tiles_per_frame;   // pixel tiles per row / per frame (and parity class)
  int taps_per_unit;               // conv_small_kernel: ceil(ntaps / units_per_q)
  int q_per, ntc;                  // conv_tile_kernel: K groups per split, weight chunks per K group
  FastDiv d_tiles_x, d_tpf, d_upq, d_win_w, d_ntc;
  FastDiv d_group;                 // batch * tiles_per_frame
  FastDiv d_class;                 // workgroups of one parity class: conv_small batch * tiles_per_frame * nb, conv_tile batch * tiles_per_frame
  // conv_tile_kernel, XCD-aware workgroup order (round 5, finish_conv_remap): 1 = a 1-D grid of gx * mtiles workgroups in which the `mtiles` output-channel
  // tiles of one pixel tile are CONSECUTIVE workgroups of ONE XCD (workgroups are dealt round-robin to the 8 XCDs: id & 7), so that the fp32 window lines the
  // first of them fetches are L2 hits for the others; 0 = the (gx, mtiles) grid of rounds 2-4 (same XCD when gx % 8 == 0, but a whole grid row apart in time)
  int xcd_remap;
  FastDiv d_mtiles;
  // conv_tile_kernel / conv_small_kernel: which pixel of a 16-pixel group each MFMA column (lane & 15) computes - sixteen 4-bit entries, chosen on the host so
  // that the B-fragment reads of the window are free of LDS bank conflicts (pixel_permutation below; identity = 0x76543210 / 0xfedcba98)
  unsigned pix_perm_lo, pix_perm_hi;
#ifdef THA4_PHASE_TIMING
  long long* dbg;        // tuning aid: s_memtime stamps [workgroup][wave][64] of ONE selected convolution, else null
#endif
};

// bytes of one lane-group plane of the LDS window image of conv_tile_kernel / conv_small_kernel; planes are skewed by 32 B so that the staging writes
// (ds_write_b128 is serviced in contiguous 8-lane groups = 2 pixels x 4 planes, bank = (a / 4) mod 32) hit distinct banks
constexpr int tile_plane_bytes(int win_px) { return (win_px * 16 + 127) / 128 * 128 + 32; }

// ---- pixel <-> MFMA-column assignment (round 5) ----------------------------------------------------------------------------------------------
// A wave reads the B fragment of one tap for a 16-pixel group with ONE ds_read_b128: lane (p = lane & 15, g = lane >> 4) reads the 16 bytes of pixel p in
// plane g.  The LDS services that instruction in four 16-lane groups - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -
// one cycle per group when its 16 addresses fall into 16 distinct 16-byte slots of the 256-byte bank row.  With pixel = p, a 16-wide tile and the 32-byte plane
// skew the staging writes need, every group mixes lanes 12-15 of plane g with lanes 4-11 of plane g + 1 shifted by 32 bytes: two slots collide, five cycles
// instead of four - the 22-30 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the conv_tile classes in profiles/r04_full_b1_profile.md (37-49 % on the narrower
// tiles of conv_small_kernel).  Which pixel a column computes is free (the epilogue stores by pixel): lanes 4-11 take the EVEN pixels of the group and lanes
// 0-3 / 12-15 the odd ones, and a shift by an even number of slots maps evens to evens - conflict-free for every 16-wide stride-1 geometry.  For the other
// geometries (8- and 4-pixel-wide tiles, stride-2 windows) the host searches a permutation with the same cost function.
inline int window_read_conflicts(const int perm[16], int twl, int win_w, int in_stride, int plane_bytes) {
  static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  const int tw = 1 << twl;
  int extra = 0;                                           // LDS cycles beyond the conflict-free four
  for (int gi = 0; gi < 4; ++gi) {
    int addr[16], worst = 1;
    for (int k = 0; k < 16; ++k) {
      const int lane = groups[gi][k], i = perm[lane & 15];
      addr[k] = (lane >> 4) * (plane_bytes / 16) + ((i >> twl) * in_stride * win_w + (i & (tw - 1)) * in_stride);     // in 16-byte slots
    }
    for (int slot = 0; slot < 16; ++slot) {                // distinct addresses on one slot of the bank row serialise; equal addresses broadcast
      int distinct = 0;
      for (int k = 0; k < 16; ++k) {
        if ((addr[k] & 15) != slot) continue;
        bool seen = false;
        for (int j = 0; j < k; ++j) seen = seen || addr[j] == addr[k];
        distinct += !seen;
      }
      worst = distinct > worst ? distinct : worst;
    }
    extra += worst - 1;
  }
  return extra;
}
// deterministic search: identity, the even / odd assignment, then hill climbing by pair swaps (fixed LCG); identity is kept unless something is strictly better
inline void pixel_permutation(int twl, int win_w, int in_stride, int plane_bytes, int out[16], int* cost_identity = nullptr, int* cost_best = nullptr) {
  int best[16], cur[16];
  for (int p = 0; p < 16; ++p) best[p] = p;
  int cbest = window_read_conflicts(best, twl, win_w, in_stride, plane_bytes);
  if (cost_identity) *cost_identity = cbest;
  for (int p = 0; p < 16; ++p) cur[p] = (p >= 4 && p < 12) ? 2 * (p - 4) : (p < 4 ? 2 * p + 1 : 2 * (p - 8) + 1);
  int ccur = window_read_conflicts(cur, twl, win_w, in_stride, plane_bytes);
  if (ccur < cbest) { cbest = ccur; for (int p = 0; p < 16; ++p) best[p] = cur[p]; }
  unsigned long long rng = 0x9E3779B97F4A7C15ull;
  for (int restart = 0; restart < 8 && cbest > 0; ++restart) {
    for (int p = 0; p < 16; ++p) cur[p] = best[p];
    ccur = cbest;
    for (int it = 0; it < 3000 && ccur > 0; ++it) {
      rng = rng * 6364136223846793005ull + 1442695040888963407ull;
      const int x = (int)((rng >> 33) & 15), y = (int)((rng >> 41) & 15);
      if (x == y) continue;
      int t = cur[x]; cur[x] = cur[y]; cur[y] = t;
      const int c = window_read_conflicts(cur, twl, win_w, in_stride, plane_bytes);
      if (c <= ccur) ccur = c;
      else { t = cur[x]; cur[x] = cur[y]; cur[y] = t; }
    }
    if (ccur < cbest) { cbest = ccur; for (int p = 0; p < 16; ++p) best[p] = cur[p]; }
  }
  for (int p = 0; p < 16; ++p) out[p] = best[p];
  if (cost_best) *cost_best = cbest;
}
THA4_DEV int pixel_of_column(const ConvArgs& a, int p) { return (int)(((p < 8 ? a.pix_perm_lo : a.pix_perm_hi) >> ((p & 7) * 4)) & 15u); }

// Host: the launch constants that do not depend on the batch.  `px_per_wg` = output positions of a workgroup tile (16 * PG * pixel-slot waves), `nq` = 32-channel
// K groups of the convolution (conv_tile_kernel's K split).  False when a divisor is out of FastDiv's range.
inline bool finish_conv_args(ConvArgs& a, int px_per_wg, int nq) {
  const int twl = a.wg_tw_log2, tww = 1 << twl, twh = px_per_wg >> twl;
  if (twh < 1) return false;
  a.tiles_x = (a.tile_w + tww - 1) >> twl;
  a.tiles_per_frame = a.tiles_x * ((a.tile_h + twh - 1) / twh);
  const int upq = a.units_per_q > 0 ? a.units_per_q : 1;
  a.taps_per_unit = (a.ntaps + upq - 1) / upq;
  const int ksplit = a.ksplit > 0 ? a.ksplit : 1;
  a.q_per = (nq + ksplit - 1) / ksplit;
  const int tpc = a.taps_per_chunk > 0 ? a.taps_per_chunk : 1;
  a.ntc = (a.ntaps + tpc - 1) / tpc;
  {                                                         // pixel <-> column assignment of the B-fragment reads (above)
    int perm[16];
    pixel_permutation(twl, a.win_w > 0 ? a.win_w : 1, a.in_stride > 0 ? a.in_stride : 1, tile_plane_bytes(a.win_h * a.win_w), perm);
#ifdef THA4_IDENTITY_PIXELS
    for (int p = 0; p < 16; ++p) perm[p] = p;               // A/B build: the pixel = column assignment of rounds 2-4
#endif
    a.pix_perm_lo = a.pix_perm_hi = 0u;
    for (int p = 0; p < 8; ++p) { a.pix_perm_lo |= (unsigned)perm[p] << (4 * p); a.pix_perm_hi |= (unsigned)perm[p + 8] << (4 * p); }
  }
  return fastdiv_make(a.d_tiles_x, a.tiles_x) && fastdiv_make(a.d_tpf, a.tiles_per_frame) && fastdiv_make(a.d_upq, upq) &&
         fastdiv_make(a.d_win_w, a.win_w > 0 ? a.win_w : 1) && fastdiv_make(a.d_ntc, a.ntc);
}
// Host, per call: the divisors that carry the batch.  `grid_x` = blockIdx.x range of the launch (the dividend bound).
inline bool finish_conv_batch(ConvArgs& a, bool small, long long grid_x) {
  const long long g = (long long)a.batch * a.tiles_per_frame;
  return grid_x < (1 << 22) && fastdiv_make(a.d_group, g) && fastdiv_make(a.d_class, small ? g * a.nb : g);
}

// Host, per call: the XCD-aware order applies when the x extent deals whole pixel tiles to XCDs (gx % 8 == 0) and there is more than one output-channel tile.
inline bool finish_conv_remap(ConvArgs& a, int mtiles, long long gx) {
  a.xcd_remap = 0;
  if (mtiles > 1 && gx > 0 && (gx & 7) == 0 && gx * mtiles < (1 << 22) && fastdiv_make(a.d_mtiles, mtiles)) a.xcd_remap = 1;
  return a.xcd_remap != 0;
}

// Geometry of one launch class.  Merged transposed convolution (nclass = 4): class (py, px), tap t = 2a + b reads input (i + dd[py][a],
// j + dd[px][b]) with dd = {{0, -1}, {1, 0}} and writes output (2i + py, 2j + px) - geom_convT4_s2 (full_layout.h, conv.py:164-177).
struct ConvClass { int win_dy0, win_dx0, out_oy, out_ox, stats_tile0; };
THA4_DEV int convt_delta(int parity, int ab) { return parity ? (ab ? 0 : 1) : (ab ? -1 : 0); }
THA4_DEV ConvClass conv_class(const ConvArgs& a, int cls, int tiles_per_class) {
  ConvClass c;
  c.win_dy0 = a.win_dy0; c.win_dx0 = a.win_dx0; c.out_oy = a.out_oy; c.out_ox = a.out_ox; c.stats_tile0 = a.stats_tile0;
  if (a.nclass == 4) {
    const int py = cls >> 1, px = cls & 1;
    c.win_dy0 = py ? 0 : -1; c.win_dx0 = px ? 0 : -1;
    c.out_oy = py; c.out_ox = px;
    c.stats_tile0 = cls * tiles_per_class;
  }
  return c;
}
THA4_DEV int conv_tap_dy(const ConvArgs& a, int cls, int t) { return a.nclass == 4 ? convt_delta(cls >> 1, (t >> 1) & 1) : a.tap_dy[t]; }
THA4_DEV int conv_tap_dx(const ConvArgs& a, int cls, int t) { return a.nclass == 4 ? convt_delta(cls & 1, t & 1) : a.tap_dx[t]; }

// 1 / (1 + e^-v) on the hardware transcendentals: v_exp_f32 (2^x, ~1 ulp) and v_rcp_f32 (1 ulp) - libm's expf plus an
// IEEE division cost ~40 VALU instructions per element, which made the SiLU operand staging of the U-Net convolutions
// (8 elements per staged item) the largest single cost of those kernels; relative error ~2e-7, parity unaffected.
THA4_DEV float fast_sigmoid(float v) {
#ifdef THA4_EMU
  return 1.0f / (1.0f + expf(-v));
#else
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
#endif
}

// e^v on v_exp_f32 (2^x): ~2 ulp relative; the softmax weights of attention_kernel (libm's expf is ~20 instructions)
THA4_DEV float fast_exp(float v) {
#ifdef THA4_EMU
  return expf(v);
#else
  return __builtin_amdgcn_exp2f(v * 1.4426950408889634f);
#endif
}

// tanh on the same two transcendentals: sign(v) (1 - e) / (1 + e), e = 2^(-2 |v| log2 e).  Absolute error ~1e-7 (the
// cancellation in 1 - e near 0 costs relative, not absolute, accuracy; outputs are O(1) colour changes).  libm's tanhf
// is ~150 instructions and was inlined at every activation site - four per 16-byte item, in the staging loops of every
// convolution kernel even though no layer has a tanh INPUT activation: the small-map kernels spent more cycles fetching
// that code than executing anything (profiles/r02_full_b1_reading.md).
THA4_DEV float fast_tanh(float v) {
#ifdef THA4_EMU
  return tanhf(v);
#else
  const float e = __builtin_amdgcn_exp2f(fabsf(v) * -2.8853900817779268f);
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, v);
#endif
}

THA4_DEV float apply_act(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  if (act == ACT_SILU) return v * fast_sigmoid(v);
  if (act == ACT_SIGMOID) return fast_sigmoid(v);
  if (act == ACT_TANH) return fast_tanh(v);
  return v;
}

// INPUT-side activation of the convolution kernels: act(x * sc + sh) on four values, `act` in {none, ReLU, SiLU} and wave-uniform (sigmoid / tanh only
// occur as OUTPUT activations of the head blocks - apply_act - and FullNet::conv rejects them on the input side: their per-element dispatch used to
// turn the staging code into a maze of scalar branches).  `zero` (per lane) forces the result to 0 - the zero padding of the window, which applies AFTER
// normalisation + activation - and shares ONE select per value with the ReLU clamp (compare, scalar mask logic, v_cndmask).  NaN-transparent like
// torch.relu: t < 0 is false for a NaN, so a NaN produced upstream reaches the moments of the next normalisation (the numeric-range guard).
THA4_DEV f32x4 apply_act4(const f32x4& x, const f32x4& sc, const f32x4& sh, int act, bool zero = false) {
  f32x4 o;
  const float lim = act == ACT_RELU ? 0.0f : -__builtin_inff();      // clamp threshold as a scalar FLOAT: no second lane mask to keep alive
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t = fmaf(x[j], sc[j], sh[j]);
    o[j] = (zero || t < lim) ? 0.0f : t;
  }
  if (act == ACT_SILU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = o[j] * fast_sigmoid(o[j]);      // 0 * sigmoid(0) = 0: padding stays zero
  }
  return o;
}

// raw operand of one (quad, tap) step for one pixel group (NV = 4 samples of the 2x2 window when pooling)
template <int NV>
struct RawFrag {
  f32x4 v[NV];
  bool valid;   // false: the tap falls into the zero padding
};

// Implicit-GEMM convolution.  Workgroup = 4 waves; each wave owns PG pixel groups (16 consecutive
// pixels of the tile grid each) and computes TMB output blocks for them; blockIdx.x = pixel tile
// (over the batch), blockIdx.y = output-channel tile.
template <int TMB, int PG, int INMODE>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs a) {
  warm_kernarg<(int)sizeof(ConvArgs)>();
  constexpr int NV = INMODE == IN_POOL2 ? 4 : 1;
  using Raw = RawFrag<NV>;
  THA4_DYN_LDS(smem);
  const int lane = threadIdx.x & 63;
  const int wave = uniform_i32(threadIdx.x >> 6);
  const int p = lane & 15, g4 = (lane >> 4) * 4;
  const int tile_px = a.tile_h * a.tile_w;
  const int pgs_per_frame = tile_px / 16;
  const int wg_pgs = 4 * PG;
  const int tiles_per_frame = pgs_per_frame / wg_pgs;
  const int n = blockIdx.x / tiles_per_frame;
  const int tile = blockIdx.x % tiles_per_frame;
  const int mtile = blockIdx.y;
  const int vh = INMODE == IN_UP2 ? a.in_h * 2 : (INMODE == IN_POOL2 ? a.in_h / 2 : a.in_h);
  const int vw = INMODE == IN_UP2 ? a.in_w * 2 : (INMODE == IN_POOL2 ? a.in_w / 2 : a.in_w);
  const int in_px = a.in_h * a.in_w;

  int ty[PG], tx[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    const int pix = ((tile * 4 + wave) * PG + pg) * 16 + p;
    ty[pg] = pix / a.tile_w;
    tx[pg] = pix % a.tile_w;
  }
  int cbtot = 0;
  for (int s = 0; s < a.nsrc; ++s) cbtot += a.src[s].cb;
  const int piece_per_quad = a.ntaps * TMB;                         // KiB of weights per input quad for this m-tile
  const int slot_bytes = a.chunk_quads * piece_per_quad * 1024;
  char* ring = smem;
  float* red = reinterpret_cast<float*>(smem + 2 * slot_bytes);      // [4 waves][TMB*16][2]
  const char* gw = reinterpret_cast<const char*>(a.w) + (size_t)mtile * cbtot * piece_per_quad * 1024;
  const int nchunks = (cbtot + a.chunk_quads - 1) / a.chunk_quads;

  auto fetch = [&](int chunk, int slot) {
    const int q0 = chunk * a.chunk_quads;
    const int nq = min(a.chunk_quads, cbtot - q0);
    const int pieces = nq * piece_per_quad;
    const char* g = gw + (size_t)q0 * piece_per_quad * 1024;
    char* l = ring + slot * slot_bytes;
    for (int pc = wave; pc < pieces; pc += 4) glds16(g + pc * 1024 + (unsigned)(lane * 16), l + pc * 1024);
  };

  // operand fetch for global quad q (over the concatenated sources) and tap t
  auto load_raw = [&](int q, int t, Raw (&r)[PG], f32x4& sc, f32x4& sh, int& act, bool new_quad) {
    int s = 0, ql = q;
    if (a.nsrc > 1 && q >= a.src[0].cb) { s = 1; ql = q - a.src[0].cb; }
    const ConvSrc& S = a.src[s];
    act = S.act;
    if (new_quad) {
      if (S.scale) {
        sc = *reinterpret_cast<const f32x4*>(S.scale + ((size_t)n * S.cb + ql) * 16 + g4);
        sh = *reinterpret_cast<const f32x4*>(S.shift + ((size_t)n * S.cb + ql) * 16 + g4);
      } else {
        sc = f32x4{1.f, 1.f, 1.f, 1.f};
        sh = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      const int vy = ty[pg] * a.in_stride + a.tap_dy[t];
      const int vx = tx[pg] * a.in_stride + a.tap_dx[t];
      const bool ok = (unsigned)vy < (unsigned)vh && (unsigned)vx < (unsigned)vw;
      const int cy = min(max(vy, 0), vh - 1), cx = min(max(vx, 0), vw - 1);
      r[pg].valid = ok;
      if (S.kind == SRC_VECTOR) {
#pragma unroll
        for (int i = 0; i < NV; ++i) r[pg].v[i] = *reinterpret_cast<const f32x4*>(S.data + ((size_t)n * S.cb + ql) * 16 + g4);
      } else {
        const float* base = S.data + (((size_t)n * S.cb + ql) * in_px) * 16 + g4;
        if (INMODE == IN_DIRECT) {
          r[pg].v[0] = *reinterpret_cast<const f32x4*>(base + ((size_t)cy * a.in_w + cx) * 16);
        } else if (INMODE == IN_UP2) {
          r[pg].v[0] = *reinterpret_cast<const f32x4*>(base + ((size_t)(cy >> 1) * a.in_w + (cx >> 1)) * 16);
        } else {
          const float* b00 = base + ((size_t)(2 * cy) * a.in_w + 2 * cx) * 16;
          r[pg].v[0] = *reinterpret_cast<const f32x4*>(b00);
          r[pg].v[NV > 1 ? 1 : 0] = *reinterpret_cast<const f32x4*>(b00 + 16);
          r[pg].v[NV > 1 ? 2 : 0] = *reinterpret_cast<const f32x4*>(b00 + (size_t)a.in_w * 16);
          r[pg].v[NV > 1 ? 3 : 0] = *reinterpret_cast<const f32x4*>(b00 + (size_t)a.in_w * 16 + 16);
        }
      }
    }
  };

  auto finish = [&](const Raw& r, const f32x4& sc, const f32x4& sh, int act_in) -> f32x4 {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v;
      if (NV == 4) {   // AvgPool2d(2,2) of the activated tensor (unet.py:58; ATen sums the window then divides)
        v = ((apply_act(fmaf(r.v[0][j], sc[j], sh[j]), act_in) + apply_act(fmaf(r.v[NV > 1 ? 1 : 0][j], sc[j], sh[j]), act_in)) +
             (apply_act(fmaf(r.v[NV > 1 ? 2 : 0][j], sc[j], sh[j]), act_in) + apply_act(fmaf(r.v[NV > 1 ? 3 : 0][j], sc[j], sh[j]), act_in))) * 0.25f;
      } else {
        v = apply_act(fmaf(r.v[0][j], sc[j], sh[j]), act_in);
      }
      o[j] = r.valid ? v : 0.0f;
    }
    return o;
  };

  f32x4 acc[TMB][PG];
#pragma unroll
  for (int b = 0; b < TMB; ++b)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) acc[b][pg] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch(0, 0);
  Raw nxt[PG];
  f32x4 sc_n, sh_n;
  int act_n = ACT_NONE;
  load_raw(0, 0, nxt, sc_n, sh_n, act_n, true);
  __syncthreads();
  int slot = 0;
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) fetch(c + 1, slot ^ 1);
    const int q0 = c * a.chunk_quads;
    const int nq = min(a.chunk_quads, cbtot - q0);
    const f32x4* wv = reinterpret_cast<const f32x4*>(ring + slot * slot_bytes) + lane;
    // blocked summation: each streamed chunk (~128-256 k-terms) accumulates into a fresh fragment that is then
    // added to the running sum, so rounding error grows like sqrt(chunk)+sqrt(#chunks) instead of sqrt(K)
    // (K is up to 4716 in the 512-channel bottlenecks; ATen/oneDNN also sums in vector-register blocks)
    f32x4 part[TMB][PG];
#pragma unroll
    for (int b = 0; b < TMB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) part[b][pg] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int qq = 0; qq < nq; ++qq) {
      for (int t = 0; t < a.ntaps; ++t) {
        Raw cur[PG];
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) cur[pg] = nxt[pg];
        const f32x4 sc = sc_n, sh = sh_n;
        const int act_c = act_n;
        // prefetch the operand of the next (quad, tap) step under this step's MFMAs
        int nt = t + 1, nqg = q0 + qq;
        bool newq = false;
        if (nt == a.ntaps) { nt = 0; nqg += 1; newq = true; }
        if (nqg < cbtot) load_raw(nqg, nt, nxt, sc_n, sh_n, act_n, newq);
        f32x4 bf[PG];
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) bf[pg] = finish(cur[pg], sc, sh, act_c);
        const f32x4* wt = wv + (size_t)((qq * a.ntaps + t) * TMB) * 64;
#pragma unroll
        for (int b = 0; b < TMB; ++b) {
          const f32x4 av = wt[b * 64];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) part[b][pg] = mfma16(av[j], bf[pg][j], part[b][pg]);
        }
      }
    }
#pragma unroll
    for (int b = 0; b < TMB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[b][pg] = acc[b][pg] + part[b][pg];
    __syncthreads();
    slot ^= 1;
  }

  // epilogue: bias, residual, activation, store, deterministic per-tile statistics
  const int out_px = a.out_h * a.out_w;
  float ssum[TMB][4], ssq[TMB][4];
#pragma unroll
  for (int b = 0; b < TMB; ++b) {
    const int bo = mtile * TMB + b;
    f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + bo * 16 + g4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[b][j] = 0.f; ssq[b][j] = 0.f; }
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      const int oy = ty[pg] * a.out_sy + a.out_oy, ox = tx[pg] * a.out_sx + a.out_ox;
      const size_t off = (((size_t)n * a.nb + bo) * out_px + (size_t)oy * a.out_w + ox) * 16 + g4;
      f32x4 v = acc[b][pg] + bias;
      if (a.residual) {
        if (a.res_mode == IN_DIRECT) {
          v = v + *reinterpret_cast<const f32x4*>(a.residual + off);
        } else if (a.res_mode == IN_UP2) {      // ResBlock x_resample = Upsample (unet.py:46): nearest
          const int rw = a.out_w >> 1, rpx = out_px >> 2;
          v = v + *reinterpret_cast<const f32x4*>(a.residual + (((size_t)n * a.nb + bo) * rpx + (size_t)(oy >> 1) * rw + (ox >> 1)) * 16 + g4);
        } else {                                // x_resample = Downsample = AvgPool2d(2,2) (unet.py:58)
          const int rw = a.out_w * 2;
          const float* r0 = a.residual + (((size_t)n * a.nb + bo) * ((size_t)out_px * 4) + (size_t)(2 * oy) * rw + 2 * ox) * 16 + g4;
          const f32x4 r = ((*reinterpret_cast<const f32x4*>(r0) + *reinterpret_cast<const f32x4*>(r0 + 16)) +
                           (*reinterpret_cast<const f32x4*>(r0 + (size_t)rw * 16) + *reinterpret_cast<const f32x4*>(r0 + (size_t)rw * 16 + 16))) * 0.25f;
          v = v + r;
        }
      }
      if (a.act_out) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], a.act_out[bo * 16 + g4 + j]);
      }
      *reinterpret_cast<f32x4*>(a.out + off) = v;
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
    __syncthreads();
    for (int i = threadIdx.x; i < TMB * 16; i += 256) {
      float s = 0.f, q = 0.f;
      for (int wv2 = 0; wv2 < 4; ++wv2) {
        s += red[((wv2 * TMB) * 16 + i) * 2 + 0];
        q += red[((wv2 * TMB) * 16 + i) * 2 + 1];
      }
      float* dst = a.stats + ((((size_t)n * a.stats_tiles + a.stats_tile0 + tile) * a.nb + mtile * TMB) * 16 + i) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Split-K convolution for the small feature maps (16x16 .. 32x32, 256-512 channels): these layers hold
// >25% of a frame's launches but only 16..64 pixel groups, so the pixel-tiled kernel above leaves most
// CUs idle and each wave latency-bound.  Here a workgroup owns ONE pixel group and TMB output blocks; its
// 4 waves take the (quad, tap) steps s = wave, wave+4, ... and read BOTH operands straight from L2
// (weights are 1 KiB fragment-linear pieces, one coalesced load per lane), prefetching the next step under
// the current MFMAs; partial fragments are combined through LDS in a fixed order (deterministic) and wave 0
// runs the same epilogue.  grid = (pixel groups over the batch, output tiles).
// ---------------------------------------------------------------------------------------------
template <int TMB, int INMODE>
__global__ void __launch_bounds__(256) conv_splitk_kernel(ConvArgs a) {
  warm_kernarg<(int)sizeof(ConvArgs)>();
  constexpr int NV = INMODE == IN_POOL2 ? 4 : 1;
  THA4_DYN_LDS(smem);
  const int lane = threadIdx.x & 63;
  const int wave = uniform_i32(threadIdx.x >> 6);
  const int p = lane & 15, g4 = (lane >> 4) * 4;
  const int pgs_per_frame = a.tile_h * a.tile_w / 16;
  const int n = blockIdx.x / pgs_per_frame;
  const int tile = blockIdx.x % pgs_per_frame;
  const int mtile = blockIdx.y;
  const int vh = INMODE == IN_UP2 ? a.in_h * 2 : (INMODE == IN_POOL2 ? a.in_h / 2 : a.in_h);
  const int vw = INMODE == IN_UP2 ? a.in_w * 2 : (INMODE == IN_POOL2 ? a.in_w / 2 : a.in_w);
  const int in_px = a.in_h * a.in_w;
  const int pix = tile * 16 + p;
  const int ty = pix / a.tile_w, tx = pix % a.tile_w;
  int cbtot = 0;
  for (int s = 0; s < a.nsrc; ++s) cbtot += a.src[s].cb;
  const int nsteps = cbtot * a.ntaps;
  const float* gw = a.w + (size_t)mtile * cbtot * a.ntaps * TMB * 256;

  struct Step { f32x4 b[NV]; f32x4 sc, sh; f32x4 w[TMB]; bool valid; int act; };
  auto load_step = [&](int s, Step& st) {
    const int q = s / a.ntaps, t = s - q * a.ntaps;
    int si = 0, ql = q;
    if (a.nsrc > 1 && q >= a.src[0].cb) { si = 1; ql = q - a.src[0].cb; }
    const ConvSrc& S = a.src[si];
    st.act = S.act;
    if (S.scale) {
      st.sc = *reinterpret_cast<const f32x4*>(S.scale + ((size_t)n * S.cb + ql) * 16 + g4);
      st.sh = *reinterpret_cast<const f32x4*>(S.shift + ((size_t)n * S.cb + ql) * 16 + g4);
    } else {
      st.sc = f32x4{1.f, 1.f, 1.f, 1.f};
      st.sh = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int vy = ty * a.in_stride + a.tap_dy[t], vx = tx * a.in_stride + a.tap_dx[t];
    st.valid = (unsigned)vy < (unsigned)vh && (unsigned)vx < (unsigned)vw;
    const int cy = min(max(vy, 0), vh - 1), cx = min(max(vx, 0), vw - 1);
    if (S.kind == SRC_VECTOR) {
#pragma unroll
      for (int i = 0; i < NV; ++i) st.b[i] = *reinterpret_cast<const f32x4*>(S.data + ((size_t)n * S.cb + ql) * 16 + g4);
    } else {
      const float* base = S.data + (((size_t)n * S.cb + ql) * in_px) * 16 + g4;
      if (INMODE == IN_DIRECT) {
        st.b[0] = *reinterpret_cast<const f32x4*>(base + ((size_t)cy * a.in_w + cx) * 16);
      } else if (INMODE == IN_UP2) {
        st.b[0] = *reinterpret_cast<const f32x4*>(base + ((size_t)(cy >> 1) * a.in_w + (cx >> 1)) * 16);
      } else {
        const float* b00 = base + ((size_t)(2 * cy) * a.in_w + 2 * cx) * 16;
        st.b[0] = *reinterpret_cast<const f32x4*>(b00);
        st.b[NV > 1 ? 1 : 0] = *reinterpret_cast<const f32x4*>(b00 + 16);
        st.b[NV > 1 ? 2 : 0] = *reinterpret_cast<const f32x4*>(b00 + (size_t)a.in_w * 16);
        st.b[NV > 1 ? 3 : 0] = *reinterpret_cast<const f32x4*>(b00 + (size_t)a.in_w * 16 + 16);
      }
    }
    const f32x4* wp = reinterpret_cast<const f32x4*>(gw + (size_t)s * TMB * 256) + lane;
#pragma unroll
    for (int b = 0; b < TMB; ++b) st.w[b] = wp[b * 64];
  };
  auto operand = [&](const Step& st) -> f32x4 {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v;
      if (NV == 4) {
        v = ((apply_act(fmaf(st.b[0][j], st.sc[j], st.sh[j]), st.act) + apply_act(fmaf(st.b[NV > 1 ? 1 : 0][j], st.sc[j], st.sh[j]), st.act)) +
             (apply_act(fmaf(st.b[NV > 1 ? 2 : 0][j], st.sc[j], st.sh[j]), st.act) + apply_act(fmaf(st.b[NV > 1 ? 3 : 0][j], st.sc[j], st.sh[j]), st.act))) * 0.25f;
      } else {
        v = apply_act(fmaf(st.b[0][j], st.sc[j], st.sh[j]), st.act);
      }
      o[j] = st.valid ? v : 0.0f;
    }
    return o;
  };

  f32x4 acc[TMB];
#pragma unroll
  for (int b = 0; b < TMB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  Step nxt;
  if (wave < nsteps) load_step(wave, nxt);
  for (int s = wave; s < nsteps; s += 4) {
    const Step cur = nxt;
    if (s + 4 < nsteps) load_step(s + 4, nxt);
    const f32x4 bf = operand(cur);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < TMB; ++b) acc[b] = mfma16(cur.w[b][j], bf[j], acc[b]);
  }
  // fixed-order combination of the four K slices
  f32x4* red = reinterpret_cast<f32x4*>(smem);      // [4][TMB][64]
#pragma unroll
  for (int b = 0; b < TMB; ++b) red[(wave * TMB + b) * 64 + lane] = acc[b];
  __syncthreads();
  if (wave != 0) return;
  const int out_px = a.out_h * a.out_w;
  const int oy = ty * a.out_sy + a.out_oy, ox = tx * a.out_sx + a.out_ox;
#pragma unroll
  for (int b = 0; b < TMB; ++b) {
    const int bo = mtile * TMB + b;
    f32x4 v = (red[(0 * TMB + b) * 64 + lane] + red[(1 * TMB + b) * 64 + lane]) + (red[(2 * TMB + b) * 64 + lane] + red[(3 * TMB + b) * 64 + lane]);
    if (a.bias) v = v + *reinterpret_cast<const f32x4*>(a.bias + bo * 16 + g4);
    const size_t off = (((size_t)n * a.nb + bo) * out_px + (size_t)oy * a.out_w + ox) * 16 + g4;
    if (a.residual) {
      if (a.res_mode == IN_DIRECT) {
        v = v + *reinterpret_cast<const f32x4*>(a.residual + off);
      } else if (a.res_mode == IN_UP2) {
        const int rw = a.out_w >> 1, rpx = out_px >> 2;
        v = v + *reinterpret_cast<const f32x4*>(a.residual + (((size_t)n * a.nb + bo) * rpx + (size_t)(oy >> 1) * rw + (ox >> 1)) * 16 + g4);
      } else {
        const int rw = a.out_w * 2;
        const float* r0 = a.residual + (((size_t)n * a.nb + bo) * ((size_t)out_px * 4) + (size_t)(2 * oy) * rw + 2 * ox) * 16 + g4;
        v = v + ((*reinterpret_cast<const f32x4*>(r0) + *reinterpret_cast<const f32x4*>(r0 + 16)) +
                 (*reinterpret_cast<const f32x4*>(r0 + (size_t)rw * 16) + *reinterpret_cast<const f32x4*>(r0 + (size_t)rw * 16 + 16))) * 0.25f;
      }
    }
    if (a.act_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], a.act_out[bo * 16 + g4 + j]);
    }
    *reinterpret_cast<f32x4*>(a.out + off) = v;
    if (a.stats) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float su = v[j], sq = v[j] * v[j];
        su = row16_sum(su, lane);
        sq = row16_sum(sq, lane);
        if (p == 0) {
          float* dst = a.stats + ((((size_t)n * a.stats_tiles + a.stats_tile0 + tile) * a.nb + bo) * 16 + g4 + j) * 2;
          dst[0] = su;
          dst[1] = sq;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ResnetBlock output (resnet_block.py:63-67): out = actA(A*sa+ha) + (B*sb+hb) on C16 tensors of equal
// shape; scale/shift are per (n, channel) vectors or null (identity).  One thread per 16-byte quad.
// ---------------------------------------------------------------------------------------------
struct FusedInstanceNorm {   // InstanceNorm2d(affine) scale/shift computed by the consumer from the producer's per-tile moments
  const float* stats;        // [n][tiles][cb*16][2] or null (not fused)
  int tiles;
  int channels;              // real channels: gamma / beta hold this many floats, padded channels get scale = shift = 0
  int* fault;                // sticky numeric-fault flag of the handle, or null
  float inv_count, eps;
  const float* gamma;
  const float* beta;
};

struct AffineAddArgs {
  const float* a; const float* sa; const float* ha; int act_a;
  const float* b; const float* sb; const float* hb;
  float* out;
  int cb, px;
  FusedInstanceNorm fa, fb;  // when .stats is set it replaces sa/ha (sb/hb)
};

// scale/shift of channel c of frame n from per-tile moments (same fp64 arithmetic as norm_finalize_kernel, groups == 0)
THA4_DEV void instance_norm_from_moments(const FusedInstanceNorm& f, int n, int cw, int c, float& sc, float& sh) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if (c >= f.channels) { sc = 0.f; sh = 0.f; return; }     // padded channel of the last block (as norm_finalize_kernel / fused_norm_table)
  const float* ps = f.stats + ((size_t)n * f.tiles * cw + c) * 2;
  const float gam = f.gamma[c], bet = f.beta[c];           // requested together with the moments: one memory round trip
  double su = 0.0, sq = 0.0;
  for (int t0 = 0; t0 < f.tiles; t0 += 8) {                // eight tile loads in flight, added in tile order
    f32x2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x2*>(ps + (size_t)min(t0 + u, f.tiles - 1) * cw * 2);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool keep = t0 + u < f.tiles;
      su += keep ? (double)v[u][0] : 0.0;
      sq += keep ? (double)v[u][1] : 0.0;
    }
  }
  const double mean = su * f.inv_count;
  const double var = sq * f.inv_count - mean * mean;
  const double rstd = 1.0 / sqrt(fmax(var, 0.0) + (double)f.eps);
  const double kk = (double)gam * rstd;
  sc = (float)kk;
  sh = (float)((double)bet - mean * kk);
  report_fault_unless_finite(f.fault, sc, sh);
}

__global__ void __launch_bounds__(256) affine_add_kernel(AffineAddArgs k) {
  warm_kernarg<(int)sizeof(AffineAddArgs)>();
  THA4_DYN_LDS(smem);                                      // 256 B: fused path: scale_a | shift_a | scale_b | shift_b of this workgroup's channel block
  float (*tab)[16] = reinterpret_cast<float (*)[16]>(smem);
  const size_t quads = (size_t)k.cb * k.px * 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  const bool fused = k.fa.stats || k.fb.stats;             // the host guarantees px*4 % 256 == 0: one channel block per workgroup
  if (fused) {
    const int blk = (int)(((size_t)blockIdx.x * 256) / ((size_t)k.px * 4));
    const int t = threadIdx.x;
    if (t < 32) {
      const int c = blk * 16 + (t & 15);
      float sc = 1.f, sh = 0.f;
      const FusedInstanceNorm& f = t < 16 ? k.fa : k.fb;
      if (f.stats) instance_norm_from_moments(f, n, k.cb * 16, c, sc, sh);
      tab[t < 16 ? 0 : 2][t & 15] = sc;
      tab[t < 16 ? 1 : 3][t & 15] = sh;
    }
    __syncthreads();
  }
  if (i >= quads) return;
  const int c4 = (int)(i / ((size_t)k.px * 4)) * 16 + (int)(i & 3) * 4;     // first channel of this quad
  const size_t off = ((size_t)n * quads + i) * 4;
  f32x4 va = *reinterpret_cast<const f32x4*>(k.a + off);
  f32x4 vb = *reinterpret_cast<const f32x4*>(k.b + off);
  if (k.fa.stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j) va[j] = fmaf(va[j], tab[0][(c4 & 15) + j], tab[1][(c4 & 15) + j]);
  } else if (k.sa) {
    const f32x4 s = *reinterpret_cast<const f32x4*>(k.sa + (size_t)n * k.cb * 16 + c4);
    const f32x4 h = *reinterpret_cast<const f32x4*>(k.ha + (size_t)n * k.cb * 16 + c4);
#pragma unroll
    for (int j = 0; j < 4; ++j) va[j] = fmaf(va[j], s[j], h[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) va[j] = apply_act(va[j], k.act_a);
  if (k.fb.stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j) vb[j] = fmaf(vb[j], tab[2][(c4 & 15) + j], tab[3][(c4 & 15) + j]);
  } else if (k.sb) {
    const f32x4 s = *reinterpret_cast<const f32x4*>(k.sb + (size_t)n * k.cb * 16 + c4);
    const f32x4 h = *reinterpret_cast<const f32x4*>(k.hb + (size_t)n * k.cb * 16 + c4);
#pragma unroll
    for (int j = 0; j < 4; ++j) vb[j] = fmaf(vb[j], s[j], h[j]);
  }
  *reinterpret_cast<f32x4*>(k.out + off) = va + vb;
}

// ---------------------------------------------------------------------------------------------
// normalisation finalize: per-tile partial sums -> per-(n, channel) scale/shift for the consumer.
//   groups == 0 : InstanceNorm2d(affine, eps)            (normalization.py:90-95)
//   groups  > 0 : GroupNorm(groups, eps) over the CONCATENATION of up to two tensors (unet.py:65-66),
//                 optionally followed by two FiLM stages  h*(1+s)+b  (unet.py:90-97,157-163)
// One workgroup per frame; fixed summation order, fp64 moments.
// ---------------------------------------------------------------------------------------------
struct NormArgs {
  const float* stats[2];   // partial sums [n][tiles][cb*16][2]
  int tiles[2], cb[2];
  int nsrc;
  int channels;            // real channel count of the concatenation (<= total padded)
  int groups;
  float inv_count;         // 1 / (pixels per channel)
  float eps;
  const float* gamma;      // [channels]
  const float* beta;
  const float* film0;      // [2*channels] (scale | shift) per frame with stride film0_stride (0: shared constant), or null
  const float* film1;
  long long film0_stride, film1_stride;
  float* scale[2];         // outputs per source, [n][cb*16]
  float* shift[2];
  int cpb;                 // padded channels per workgroup (a multiple of the group size; blockIdx.y selects the range)
  int* fault;              // sticky numeric-fault flag of the handle, or null
};

constexpr int kNormThreads = 1024;
#ifndef THA4_NORM_LOADS_IN_FLIGHT
#define THA4_NORM_LOADS_IN_FLIGHT 4      // independent 8-byte moment loads per thread and round (a power of two).  MEASURED: eight is 0.3-0.7 % SLOWER on the
                                         // batch-1 frame than four, with either channel split (same-box A/Bs, profiles/r05_raw/c4_ab_norm.txt, c5_ab_norm.txt)
#endif
// channels per workgroup: whole GroupNorm groups, at least 32 channels, about an eighth of the tensor - a large map has
// 256-512 tiles per channel, and the more tile slices a workgroup's 1024 threads form the fewer dependent load rounds each
// thread pays (one workgroup per frame: 8-16 rounds; 4-8 workgroups: 2)
inline int norm_channels_per_block(int ctot, int channels, int groups, int tiles = 0) {
  const int gs = groups > 0 ? channels / groups : 1;
  int want = ctot / 8 > 32 ? ctot / 8 : 32;
  // `tiles` > 0 (tuning aid, THA4_NORM_TILE_SPLIT): fewer channels per workgroup when a channel has many tiles (a 512x512 map: 1024+), so that a thread walks
  // about one round of loads instead of 8+ (never below one GroupNorm group, never below 4 channels).  MEASURED NEUTRAL (profiles/r05_raw/c5_ab_norm.txt:
  // 183.57 vs 183.57 frames/s): the 10-12 us finalize launches behind the 256x256 / 512x512 convolutions are not their own load rounds - the kernel boundary
  // behind a convolution that left 17-34 MB of dirty lines in the L2s waits for their write-back (bytes / ~6 TB/s), whatever the next kernel is
  if (tiles > 0) {
    int cap = kNormThreads * THA4_NORM_LOADS_IN_FLIGHT / tiles;
    cap = cap < 4 ? 4 : cap;
    want = want < cap ? want : cap;
  }
  int cpb = (want + gs - 1) / gs * gs;
  return cpb < ctot ? cpb : ctot;
}
__global__ void __launch_bounds__(kNormThreads) norm_finalize_kernel(NormArgs a) {
  warm_kernarg<(int)sizeof(NormArgs)>();
  THA4_DYN_LDS(smem);
  const int n = blockIdx.x;
  const int c0 = a.cb[0] * 16;
  const int ctot_all = c0 + (a.nsrc > 1 ? a.cb[1] * 16 : 0);
  const int cbeg = blockIdx.y * a.cpb;                       // this workgroup's channels [cbeg, cbeg + ctot)
  const int ctot = min(a.cpb, ctot_all - cbeg);
  // thread (slice, c): channel c = t % ctot sums tiles slice, slice+S, ... in fp64; consecutive threads read
  // consecutive channels of one tile row (coalesced).  Slices are then combined in a fixed order: deterministic.
  const int S = max(1, kNormThreads / ctot);                 // tile slices (ctot <= 1024)
  double* part = reinterpret_cast<double*>(smem);            // [S][ctot][2]
  double* csum = part + (size_t)S * ctot * 2;                // [ctot]
  double* csq = csum + ctot;
  // the affine / FiLM operands of this thread's channel (ctot <= kNormThreads: thread t finishes channel cbeg + t) are requested
  // here, together with the moments - not after the two reduction barriers, where they were one more dependent memory round trip
  // (same-box A/B: +0.45 % on the batch-1 frame, profiles/r04_raw/c25_ab.txt)
  // All six loads are unconditional (a clamped channel; an absent FiLM row reads gamma instead and is never used): conditional loads
  // are waited for at the end of their `if`, which made three dependent round trips of them
  const int cpre = min(cbeg + (int)threadIdx.x, a.channels - 1);
  const float* f0 = a.film0 ? a.film0 + (size_t)n * a.film0_stride : nullptr;
  const float* f1 = a.film1 ? a.film1 + (size_t)n * a.film1_stride : nullptr;
  const float p_gam = a.gamma[cpre], p_bet = a.beta[cpre];
  const float p_s0 = (f0 ? f0 : a.gamma)[cpre], p_b0 = (f0 ? f0 + a.channels : a.gamma)[cpre];
  const float p_s1 = (f1 ? f1 : a.gamma)[cpre], p_b1 = (f1 ? f1 + a.channels : a.gamma)[cpre];
  for (int cl0 = threadIdx.x % ctot, sl = threadIdx.x / ctot; sl < S && cl0 < ctot; sl += kNormThreads) {   // one pass (S*ctot <= threads)
    const int c = cbeg + cl0;
    const int s = c < c0 ? 0 : 1;
    const int cl = c - (s ? c0 : 0);
    // (the two-entry argument arrays are read with CONSTANT indices and selected: indexed by `s` they become vector loads from the
    // argument block, a dependent memory round trip in front of the moment loads - and two more in front of the final stores)
    const int cw = (s ? a.cb[1] : a.cb[0]) * 16;
    const int nt = s ? a.tiles[1] : a.tiles[0];
    const float* ps = (s ? a.stats[1] : a.stats[0]) + ((size_t)n * nt * cw + cl) * 2;
    // NP independent partial sums (fixed assignment t -> sum (t/S)%NP, combined pairwise in a fixed order): the loads of NP tiles are in flight at once
    // instead of one fp64 add chain waiting on each 8-byte load in turn
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int NP = THA4_NORM_LOADS_IN_FLIGHT;
    double su[NP], sq[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) { su[u] = 0.0; sq[u] = 0.0; }
    int t = sl;
    for (; t + (NP - 1) * S < nt; t += NP * S) {
      f32x2 v[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) v[u] = *reinterpret_cast<const f32x2*>(ps + (size_t)(t + u * S) * cw * 2);
#pragma unroll
      for (int u = 0; u < NP; ++u) { su[u] += (double)v[u][0]; sq[u] += (double)v[u][1]; }
    }
    if (t < nt) {                                          // at most NP - 1 left-over tiles: requested together, added in tile order (no left-over: no
      f32x2 v[NP - 1];                                     // extra dependent round - the unconditional form of this block cost 0.7 us on EVERY launch)
#pragma unroll
      for (int u = 0; u < NP - 1; ++u) v[u] = *reinterpret_cast<const f32x2*>(ps + (size_t)min(t + u * S, nt - 1) * cw * 2);
#pragma unroll
      for (int u = 0; u < NP - 1; ++u) {
        const bool keep = t + u * S < nt;
        su[u] += keep ? (double)v[u][0] : 0.0;
        sq[u] += keep ? (double)v[u][1] : 0.0;
      }
    }
#pragma unroll
    for (int w = 1; w < NP; w *= 2)                        // pairwise, fixed order: ((0+1) + (2+3)) + ((4+5) + (6+7))
#pragma unroll
      for (int u = 0; u + w < NP; u += 2 * w) { su[u] += su[u + w]; sq[u] += sq[u + w]; }
    part[((size_t)sl * ctot + cl0) * 2] = su[0];
    part[((size_t)sl * ctot + cl0) * 2 + 1] = sq[0];
  }
  __syncthreads();
  for (int cl0 = threadIdx.x; cl0 < ctot; cl0 += kNormThreads) {
    double su = 0.0, sq = 0.0;
    for (int sl = 0; sl < S; ++sl) { su += part[((size_t)sl * ctot + cl0) * 2]; sq += part[((size_t)sl * ctot + cl0) * 2 + 1]; }
    csum[cl0] = su;
    csq[cl0] = sq;
  }
  __syncthreads();
  // logical channel index of padded channel c: source 0 holds channels [0, C0real), source 1 the rest.
  // Both sources are exact multiples of 16 whenever two are concatenated (unet.py skip widths).
  for (int cl0 = threadIdx.x; cl0 < ctot; cl0 += kNormThreads) {
    const int c = cbeg + cl0;
    const int s = c < c0 ? 0 : 1;
    const int cl = c - (s ? c0 : 0);
    float sc = 0.f, sh = 0.f;
    if (c < a.channels) {
      double mean, var;
      if (a.groups == 0) {
        mean = csum[cl0] * a.inv_count;
        var = csq[cl0] * a.inv_count - mean * mean;
      } else {
        const int gs = a.channels / a.groups;
        const int gi = c / gs;                               // (the workgroup's range holds whole groups: cpb is a multiple of gs)
        double su = 0.0, sq = 0.0;
        for (int k = gi * gs; k < (gi + 1) * gs; ++k) { su += csum[k - cbeg]; sq += csq[k - cbeg]; }
        mean = su * a.inv_count / gs;
        var = sq * a.inv_count / gs - mean * mean;
      }
      const double rstd = 1.0 / sqrt(fmax(var, 0.0) + (double)a.eps);
      double k = (double)p_gam * rstd;                      // (cl0 == threadIdx.x: the loop runs once)
      double b = (double)p_bet - mean * k;
      if (a.film0) {
        const double s0 = p_s0, b0 = p_b0;
        k *= (1.0 + s0); b = b * (1.0 + s0) + b0;
      }
      if (a.film1) {
        const double s1 = p_s1, b1 = p_b1;
        k *= (1.0 + s1); b = b * (1.0 + s1) + b1;
      }
      sc = (float)k;
      sh = (float)b;
      report_fault_unless_finite(a.fault, sc, sh);
    }
    const size_t oidx = (size_t)n * (s ? a.cb[1] : a.cb[0]) * 16 + cl;
    (s ? a.scale[1] : a.scale[0])[oidx] = sc;
    (s ? a.shift[1] : a.shift[0])[oidx] = sh;
  }
}

// ---------------------------------------------------------------------------------------------
// small dense layers (cond/time embeddings, FiLM projections): y[n][r] = act_out(b[r] + W[r,:] . act_in(x[n,:]))
// one wave per output row (unet.py:137-146, 443-452)
// ---------------------------------------------------------------------------------------------
struct GemvArgs {
  const float* w;      // [rows][k] row-major
  const float* bias;   // [rows]
  const float* x;      // [n][k] with row stride x_stride
  float* y;            // [n][rows]
  int rows, k;
  long long x_stride;
  int act_in, act_out;
};

__global__ void __launch_bounds__(256) gemv_kernel(GemvArgs a) {
  warm_kernarg<(int)sizeof(GemvArgs)>();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.y;
  if (row >= a.rows) return;
  const float* w = a.w + (size_t)row * a.k;
  const float* x = a.x + (size_t)n * a.x_stride;
  float s = 0.f;
  for (int i = lane; i < a.k; i += 64) s = fmaf(w[i], apply_act(x[i], a.act_in), s);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += lane_read(s, lane ^ m);
  if (lane == 0) a.y[(size_t)n * a.rows + row] = apply_act(s + a.bias[row], a.act_out);
}

// ---------------------------------------------------------------------------------------------
// attention core (unet.py:192-202, new attention order): per (frame, head) workgroup, 256 tokens,
// head dim 32.  qkv in C16 [n][3C/16][L][16]; out C16 [n][C/16][L][16], L <= 256 a multiple of 32.  K/V of the head
// sit in LDS; 8 lanes share a query token.  (0.4 GFLOP per network: VALU is adequate.)
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
  const float* qkv;
  float* out;
  int channels;   // C (=256)
  int heads;      // 8
  int tokens;     // 256
};

constexpr int kAttnHeadDim = 32;     // 256 channels / 8 heads (mode_07.py:222-224,253-255)
#ifndef THA4_ATTN_SLICES
#define THA4_ATTN_SLICES 16
#endif
constexpr int kAttnSlices = THA4_ATTN_SLICES;    // key slices per query (consecutive lanes share a query): 16 -> 128 workgroups of 16
                                                  // queries for the 16x16 maps instead of 64 of 32 (13.8 -> ~10 us per launch)
constexpr int kAttnQueries = 256 / kAttnSlices;  // query tokens per workgroup
constexpr int kAttnRow = kAttnHeadDim / 4 + 1;   // f32x4 per K/V row in LDS (+1: consecutive rows, read by consecutive slices, on distinct banks)
constexpr int kAttnMaxKeys = 256 / kAttnSlices;  // keys per slice held in registers: tokens <= 256

// grid (heads, frames, tokens / kAttnQueries), 256 threads: thread (query ql = t / S, slice sl = t % S; S = kAttnSlices) scores keys sl, sl+S, ...
// against its query, takes the slice maximum, accumulates exp-weighted values, and the 8 slices of a query are merged
// with lane shuffles (max first, then one rescale per slice) - a fixed order.
__global__ void __launch_bounds__(256) attention_kernel(AttnArgs a) {
  warm_kernarg<(int)sizeof(AttnArgs)>();
  THA4_DYN_LDS(smem);
  constexpr int CH = kAttnHeadDim, Q4 = CH / 4;
  const int L = a.tokens;
  f32x4* ks = reinterpret_cast<f32x4*>(smem);   // [L][kAttnRow]
  f32x4* vs = ks + L * kAttnRow;                // [L][kAttnRow]
  const int n = blockIdx.y, h = blockIdx.x, t = threadIdx.x;
  const int ql = t / kAttnSlices, sl = t % kAttnSlices;
  const int tq = blockIdx.z * kAttnQueries + ql;
  const int cbq = a.channels / 16;
  const float scale = 1.0f / sqrtf(sqrtf((float)CH));
  // head h owns channels [h*32, h*32+32) of q, k and v = two C16 blocks each; a token's 16 channels are contiguous
  const float* qbase = a.qkv + (((size_t)n * 3 * cbq + 2 * h) * L) * 16;
  const float* kbase = a.qkv + (((size_t)n * 3 * cbq + cbq + 2 * h) * L) * 16;
  const float* vbase = a.qkv + (((size_t)n * 3 * cbq + 2 * cbq + 2 * h) * L) * 16;
  for (int tok = t; tok < L; tok += 256) {
#pragma unroll
    for (int i = 0; i < Q4; ++i) {
      const size_t off = (size_t)(i / 4) * L * 16 + (size_t)tok * 16 + (i % 4) * 4;   // block i/4, floats (i%4)*4.. of the token's 16
      ks[tok * kAttnRow + i] = *reinterpret_cast<const f32x4*>(kbase + off) * scale;
      vs[tok * kAttnRow + i] = *reinterpret_cast<const f32x4*>(vbase + off);
    }
  }
  f32x4 q[Q4];
#pragma unroll
  for (int i = 0; i < Q4; ++i)
    q[i] = *reinterpret_cast<const f32x4*>(qbase + (size_t)(i / 4) * L * 16 + (size_t)tq * 16 + (i % 4) * 4) * scale;
  __syncthreads();
  const int nk = L / kAttnSlices;               // keys of this slice (<= kAttnMaxKeys)
  float sc[kAttnMaxKeys];
  float m = -3.0e38f;
#pragma unroll
  for (int it = 0; it < kAttnMaxKeys; ++it) {
    float d = -3.0e38f;
    if (it < nk) {
      const f32x4* kr = ks + (size_t)(sl + it * kAttnSlices) * kAttnRow;
      d = 0.f;
#pragma unroll
      for (int i = 0; i < Q4; ++i) {
        const f32x4 k4 = kr[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) d = fmaf(q[i][j], k4[j], d);
      }
    }
    sc[it] = d;
    m = fmaxf(m, d);
  }
#pragma unroll
  for (int x = 1; x < kAttnSlices; x <<= 1) m = fmaxf(m, lane_read(m, (t & 63) ^ x));    // row maximum over the slices
  f32x4 o[Q4];
#pragma unroll
  for (int i = 0; i < Q4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float den = 0.f;
#pragma unroll
  for (int it = 0; it < kAttnMaxKeys; ++it) {
    if (it < nk) {
      const float e = fast_exp(sc[it] - m);
      den += e;
      const f32x4* vr = vs + (size_t)(sl + it * kAttnSlices) * kAttnRow;
#pragma unroll
      for (int i = 0; i < Q4; ++i) {
        const f32x4 v4 = vr[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i][j] = fmaf(e, v4[j], o[i][j]);
      }
    }
  }
#pragma unroll
  for (int x = 1; x < kAttnSlices; x <<= 1) {
    den += lane_read(den, (t & 63) ^ x);
#pragma unroll
    for (int i = 0; i < Q4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[i][j] += lane_read(o[i][j], (t & 63) ^ x);
  }
  if (sl != 0) return;
  const float inv = 1.0f / den;
  float* op = a.out + (((size_t)n * cbq + 2 * h) * L + tq) * 16;
#pragma unroll
  for (int i = 0; i < Q4; ++i) *reinterpret_cast<f32x4*>(op + (size_t)(i / 4) * L * 16 + (i % 4) * 4) = o[i] * inv;
}

}  // namespace tha4
