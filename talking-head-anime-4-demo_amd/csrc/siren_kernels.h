// Hand-written CDNA4 (gfx950) kernels for the THA4 distilled-student poser (reference mode_14).
//
// One frame = 5 launches on one stream:
//   posebias   fold the pose columns of every first layer into a per-frame bias vector
//   face       SirenFaceMorpher00.forward      (siren_face_morpher_00.py:34-51)   128x128 px
//   level0/1/2 SirenMorpher03.forward levels   (siren_morpher_03.py:107-123)      128^2 / 256^2 / 512^2 px
//              level2 also runs last_linear, the grid_sample warp and the alpha blend
//              (siren_morpher_03.py:125-131, image_processing_util.py:33-54) and reads the
//              face patch in place of the image inside rows 80:208 x cols 192:320 (mode_14.py:72-78).
//
// Design: every SIREN layer is a 1x1 conv = per-pixel GEMV, so a tile of pixels runs
// the WHOLE layer chain of its level inside one workgroup: activations never leave the CU (LDS,
// MFMA-fragment-linear image), weights stream HBM/L2 -> LDS with global_load_lds in 1 KiB pieces
// (double-buffered ring, one barrier per K chunk), contractions are exact-fp32
// v_mfma_f32_16x16x4_f32 (SIREN's sin(30 W x) amplifies 16-bit weight rounding to O(1) errors -
// SURVEY.md §0.4), sin() is an explicit Cody-Waite + degree-9 polynomial on the VALU.
// The bilinear x2 upsample between levels commutes with the (linear) feature part of the next
// level's first layer, so each level emits z = W_next[:, :C] h at ITS resolution and the next level
// only gathers 4 taps of z per pixel (4x fewer MACs for that layer, no 180/90-channel upsampled map).
//
// Workgroup geometry: NS pixel slots x MS row splits = NS*MS waves (8 = two per SIMD, so one
// wave's sin()/LDS/epilogue work hides under the other's MFMAs).  A pixel slot owns PG pixel
// groups of 16 px (a 16-px strip of one image row) and one activation image in LDS; with MS = 2
// two waves share a slot and each computes half of every layer's output blocks (this is what lets
// level 0 / face fill all 256 CUs with two waves per SIMD at batch 1, where there are only 1024
// pixel groups for 1024 SIMDs).
#pragma once
#include "tha4_platform.h"
#include "siren_layout.h"
#include "image_io_kernels.h"

namespace tha4 {

struct StudentDev {
  // packed parameters (device)
  const float *w_face, *w_l0, *w_l1, *w_l2;          // weight streams (siren_layout.h)
  const float *b_face, *b_l0, *b_l1, *b_l2;          // biases of the streamed layers
  const float *wx[4], *wy[4], *bias1[4], *wpose[4];  // first layers: 0 face, 1..3 body levels
  const float *pos128, *pos256, *pos512;             // affine_grid axes
  // workspace (device)
  float* pbias;   // [B][kPbStride]
  int front_l0_blocks;   // v2::front16_kernel: level-0 workgroups at the head of the merged face + level-0 grid
  float* z1;      // [B][kNB1][4][128*128][4]   (z_offset, siren_layout.h)
  float* z2;      // [B][kNB2][4][256*256][4]
  float* face;    // [B][4][128][128]
  // i/o (device)
  const float* image;        // [B or 1][4][512][512]
  long long image_stride;    // floats between consecutive frames' images (0: one image shared by the batch)
  const float* pose;         // [B][45]
  float* out_blended;        // [B][4][512][512]
  float* out_alpha;          // [B][1][512][512] or null
  float* out_color;          // [B][4][512][512] or null
  float* out_warped;         // [B][4][512][512] or null
  float* out_grid;           // [B][2][512][512] or null
  unsigned char* out_rgba8;  // [B][512][512][4] display epilogue of the posed frame (tha4_hip.h tha4_display) or null; with it
                             // out_blended may be null
  int rgba8_has_bg;          // blend over the opaque background colour rgba8_bg (sRGB-encoded, [0,1]); alpha becomes 1
  float rgba8_bg[3];
  int batch;
  // generation 2 (siren16_kernels.h): 1/S of every streamed layer in execution order; the first-layer tables (wx, wy),
  // the pose-folded biases and the z hand-off carry the sine's 30x (pb_scale = 30; generation 1: 1)
  const float *s_face, *s_l0, *s_l1, *s_l2;
  float pb_scale;
};

// Display epilogue of the posed frame, fused into the warp/blend tail (tha4_hip.h tha4_display): lane
// group g holds channel g of pixel p.  The four bytes of a pixel are collected in lane p (three lane reads) and written as ONE
// dword: 16 lanes cover 64 contiguous bytes of the HWC frame.  (One byte store per lane - 64 lanes, 64 contiguous bytes - was
// measured first: 269 us per frame instead of 140; byte-granular stores are read-modify-write traffic at the L2.)
// Must be called by all 64 lanes.
THA4_DEV void store_display(const StudentDev& d, int n, size_t pix, int g, int p, float blended) {
  float a01 = 0.0f;
  if (d.rgba8_has_bg) a01 = fminf(fmaxf((lane_read(blended, p + 48) + 1.0f) * 0.5f, 0.0f), 1.0f);     // wave-uniform branch
  const float bg = g == 0 ? d.rgba8_bg[0] : (g == 1 ? d.rgba8_bg[1] : d.rgba8_bg[2]);
  const float mine = (float)display_channel(blended, g, a01, d.rgba8_has_bg != 0, bg);               // 0..255, exact in fp32
  const unsigned r = (unsigned)mine, gch = (unsigned)lane_read(mine, p + 16), b = (unsigned)lane_read(mine, p + 32),
                 a = (unsigned)lane_read(mine, p + 48);
  if (g == 0) reinterpret_cast<unsigned*>(d.out_rgba8)[(size_t)n * (kImg * kImg) + pix] = r | (gch << 8) | (b << 16) | (a << 24);
}

// ---------------------------------------------------------------------------------------------
// sin(30 z) exactly as the reference rounds it: u = fl(30 z) (siren.py:39), then sin(u) to ~1.4e-7
// abs: k = rint(u/pi), r = u - k pi by 3-term Cody-Waite (k pi_A exact for |k| < 4096),
// sin(r) = r + r^3 q(r^2) on [-pi/2, pi/2] (degree-9 minimax), sign flipped for odd k.
// ---------------------------------------------------------------------------------------------
THA4_DEV float sin_omega(float z) {
  if (THA4_HOOK_SIN_BYPASS) return z;
  const float u = kOmega * z;
#if defined(THA4_HW_SIN) && !defined(THA4_EMU)
  // v_sin_f32 variant (input in revolutions): same reference rounding of u, 2-term Cody-Waite by 2 pi,
  // 3.8e-7 max abs error on the device (tools/microbench/sin_test.hip) instead of 1.3e-7, 10 issue slots instead of 15
  const float kr = rintf(u * 0x1.45f306p-3f);
  float rr = fmaf(-kr, 6.28125f, u);
  rr = fmaf(-kr, 0x1.fb5444p-10f, rr);
  return __builtin_amdgcn_sinf(rr * 0x1.45f306p-3f);
#endif
  const float k = rintf(u * 0x1.45f306p-2f);
  float r = fmaf(-k, 0x1.92p+1f, u);
  r = fmaf(-k, 0x1.fb4p-11f, r);
  r = fmaf(-k, 0x1.4442d2p-23f, r);
  const float r2 = r * r;
  float q = 0x1.5cf94cp-19f;
  q = fmaf(q, r2, -0x1.9f5ff6p-13f);
  q = fmaf(q, r2, 0x1.110e6ap-7f);
  q = fmaf(q, r2, -0x1.555548p-3f);
  const float s = fmaf(r * r2, q, r);
  const unsigned flip = ((unsigned)(int)k) << 31;
  return __uint_as_float(__float_as_uint(s) ^ flip);
}

// The sine of generation 2 for a pre-scaled argument.  Default (THA4_SIN_TURNS): the argument is in TURNS (omega_0 / 2 pi is
// folded into weights and biases, siren_layout.h) and the sine is ONE v_sin_f32, which reduces its argument itself (valid for
// |t| <= 256 turns; the shipped students stay below 7) - 3.8e-7 max abs error over every argument of a frame
// (tools/sin_cliff.py, profiles/r03_sin_cliff.md) against 1.4e-7 for the polynomial below, and the posed frame is as close to
// the reference (64-pose sweeps, both characters).  THA4_SIN_TURNS=0: radians; k = rint(u/pi) comes from
// adding 1.5*2^23 (valid for |u| < 2^22 pi; the low mantissa bit of the sum is the parity of k), r = u - k pi by a
// 2-term Cody-Waite (k * 3.140625 is exact for |k| < 2^16; total error <= |k| 6e-11), degree-9 polynomial:
// 12 VALU ops: fma, sub, 2 fma, mul, 3 fma, mul, fma, shift, xor.
THA4_DEV float sin_u(float u) {
  if (THA4_HOOK_SIN_BYPASS) return u;
#if THA4_SIN_TURNS
#ifdef THA4_EMU
  if (!(fabsf(u) <= 256.0f)) return 0.0f;          // the instruction's domain: 0 beyond 256 turns (and for NaN / inf inputs the device returns NaN; not modelled)
  return (float)sin(6.283185307179586476925 * (double)u);
#else
  return __builtin_amdgcn_sinf(u);
#endif
#endif
#if defined(THA4_HW_SIN) && !defined(THA4_EMU)
  // A/B variant (tools/sin_cliff.py, profiles/r03_sin_cliff.md; never shipped): k = rint(u / 2 pi) by the magic add, 2-term
  // Cody-Waite by 2 pi, v_sin_f32 (argument in revolutions) on the reduced argument.  7 issue slots instead of 12.
  {
    const float th = fmaf(u, 0x1.45f306p-3f, 12582912.0f);
    const float kh = th - 12582912.0f;
    float rh = fmaf(-kh, 6.28125f, u);
    rh = fmaf(-kh, 0x1.fb5444p-10f, rh);
#if defined(THA4_HW_SIN_NOP)
    float sv;                                              // the same instruction with wait states behind it (hazard experiment)
    asm volatile("v_sin_f32 %0, %1\n\ts_nop 7\n\ts_nop 7" : "=v"(sv) : "v"(rh * 0x1.45f306p-3f));
    return sv;
#elif defined(THA4_HW_SIN_INPLACE)
    float sv = rh * 0x1.45f306p-3f;                        // source == destination: nothing can overwrite the source early (hazard experiment)
    asm volatile("v_sin_f32 %0, %0\n\ts_nop 1" : "+v"(sv));
    return sv;
#else
    return __builtin_amdgcn_sinf(rh * 0x1.45f306p-3f);
#endif
  }
#endif
  const float t = fmaf(u, 0x1.45f306p-2f, 12582912.0f);
  const float k = t - 12582912.0f;
  float r = fmaf(-k, 0x1.92p+1f, u);
  r = fmaf(-k, 0x1.fb5444p-11f, r);
  const float r2 = r * r;
  float q = 0x1.5cf94cp-19f;
  q = fmaf(q, r2, -0x1.9f5ff6p-13f);
  q = fmaf(q, r2, 0x1.110e6ap-7f);
  q = fmaf(q, r2, -0x1.555548p-3f);
  const float s = fmaf(r * r2, q, r);
  return __uint_as_float(__float_as_uint(s) ^ (__float_as_uint(t) << 31));
}

// ---------------------------------------------------------------------------------------------
// workgroup geometry + LDS carve:  [ring slot 0][ring slot 1][act slot 0]..[act slot NS-1]
// ---------------------------------------------------------------------------------------------
template <int NS_, int MS_, int PG_, int ACTQ_, int SLOT_PIECES_>
struct Geo {
  static constexpr int NS = NS_, MS = MS_, PG = PG_, ACTQ = ACTQ_;
  static constexpr int WAVES = NS * MS, THREADS = WAVES * 64;
  static constexpr int SLOT = SLOT_PIECES_ * 1024;              // bytes of one ring slot
  static constexpr int ACT_BYTES = PG * ACTQ * 1024;            // one pixel slot's activation image
  static constexpr int LDS = 2 * SLOT + NS * ACT_BYTES;
  static constexpr int PX = NS * PG * 16;                       // pixels per workgroup
  static_assert(LDS <= 160 * 1024, "LDS budget exceeded");
  static_assert(THREADS <= 1024, "too many waves");
};

struct WaveCtx {
  int lane, wave, ns, ms;   // ns: pixel slot, ms: row split
  int blk, nblk;            // this workgroup's index in its kernel's tile sequence (blockIdx.x / gridDim.x unless several
                            // kernels share one launch: v2::front16_kernel)
};

template <class G>
THA4_DEV WaveCtx wave_ctx() {
  WaveCtx c;
  c.lane = threadIdx.x & 63;
  c.wave = uniform_i32(threadIdx.x >> 6);
  c.ns = c.wave % G::NS;
  c.ms = G::MS == 1 ? 0 : c.wave / G::NS;   // compile-time 0 keeps block indices static when rows are not split
  c.blk = blockIdx.x;
  c.nblk = gridDim.x;
  return c;
}

// weight stream: all waves of the workgroup copy PIECES x 1 KiB from global to an LDS ring slot
template <int PIECES, int WAVES>
THA4_DEV void fetch_pieces(const char* g, char* l, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < (PIECES + WAVES - 1) / WAVES; ++i) {
    const int pc = i * WAVES + wave;
    THA4_HOOK_FETCH(if (PIECES % WAVES == 0 || pc < PIECES) glds16(g + pc * 1024 + (unsigned)(lane * 16), l + pc * 1024));      // (no branch on the wave id where every wave copies)
  }
}


// blocks per software-pipeline group: the A fragments of group t+1 are read from LDS while the
// MFMAs of group t issue; inside a group MFMAs are ordered k-step-major so that consecutive
// instructions hit different accumulators (a dependent v_mfma_f32_16x16x4_f32 costs 40 cycles, an
// independent one 32).
constexpr int group_blocks(int nbw) { return nbw % 4 == 0 ? 4 : (nbw % 3 == 0 ? 3 : (nbw % 2 == 0 ? 2 : 1)); }

// One linear layer, this wave's share: acc[b][pg] += W-block (mbase+b) x act of pixel group pg.
//   NB   output blocks of the layer in the stream (all waves)   NBW  blocks computed by this wave
//   KQ   input quads   CQ quads per streamed chunk (KQ % CQ == 0)
//   NEXT_PIECES  KiB of the first chunk of whatever follows in the stream (0: nothing); prefetched
//                during this layer's last chunk so layer boundaries cost no load bubble.
// On entry chunk 0 of this layer is resident in ring slot `slot` (covered by the preceding barrier).
// `active` (wave-uniform) = false: take part in the fetches and barriers only.
template <class G, int NB, int NBW, int KQ, int CQ, int NEXT_PIECES>
THA4_DEV void gemm_stream(const char*& gw, char* ring, int& slot, const f32x4* actv, f32x4 (&acc)[NBW][G::PG],
                          const WaveCtx& w, int mbase, bool active) {
  static_assert(KQ % CQ == 0, "chunking must divide K");
  static_assert(CQ * NB * 1024 <= G::SLOT, "chunk exceeds ring slot");
  static_assert(NEXT_PIECES * 1024 <= G::SLOT, "next chunk exceeds ring slot");
  constexpr int NC = KQ / CQ;
  constexpr int CHUNK = CQ * NB * 1024;
  constexpr int PG = G::PG;
  constexpr int GB = group_blocks(NBW), NG = NBW / GB, T = CQ * NG;
#pragma unroll 1
  for (int c = 0; c < NC; ++c) {
    const int nslot = slot ^ 1;
    if (c + 1 < NC) {
      fetch_pieces<CQ * NB, G::WAVES>(gw + (size_t)(c + 1) * CHUNK, ring + nslot * G::SLOT, w.wave, w.lane);
    } else if (NEXT_PIECES > 0) {
      fetch_pieces<NEXT_PIECES, G::WAVES>(gw + (size_t)NC * CHUNK, ring + nslot * G::SLOT, w.wave, w.lane);
    }
    if (active) {
      const f32x4* wv = reinterpret_cast<const f32x4*>(ring + slot * G::SLOT) + (size_t)mbase * 64 + w.lane;
      const f32x4* av = actv + (size_t)c * CQ * 64 + w.lane;
      f32x4 abuf[2][GB], bbuf[2][PG];
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) bbuf[0][pg] = av[pg * G::ACTQ * 64];
#pragma unroll
      for (int b = 0; b < GB; ++b) abuf[0][b] = wv[b * 64];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int qq = t / NG, g = t % NG;
        // k-step 0 of this group first: the compiler's s_waitcnt for the group's fragments lands
        // here, BEFORE the next group's LDS reads are issued, so it never drains those
#pragma unroll
        for (int b = 0; b < GB; ++b)
#pragma unroll
          for (int pg = 0; pg < PG; ++pg)
            acc[g * GB + b][pg] = mfma16(abuf[t & 1][b][0], bbuf[qq & 1][pg][0], acc[g * GB + b][pg]);
        THA4_SCHED_FENCE();
        if (t + 1 < T) {   // LDS reads of the next group fly under the remaining 3/4 of this group's MFMAs
          const int nq = (t + 1) / NG, ng = (t + 1) % NG;
          if (ng == 0) {
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) bbuf[nq & 1][pg] = av[(pg * G::ACTQ + nq) * 64];
          }
#pragma unroll
          for (int b = 0; b < GB; ++b) abuf[(t + 1) & 1][b] = wv[(nq * NB + ng * GB + b) * 64];
        }
        THA4_SCHED_FENCE();
#pragma unroll
        for (int j = 1; j < 4; ++j)
#pragma unroll
          for (int b = 0; b < GB; ++b)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg)
              acc[g * GB + b][pg] = mfma16(abuf[t & 1][b][j], bbuf[qq & 1][pg][j], acc[g * GB + b][pg]);
        THA4_SCHED_FENCE();
      }
    }
    THA4_HOOK_CHUNK_BARRIER();   // next slot landed (vmcnt(0)) and every wave is done reading this one
    slot = nslot;
  }
  gw += (size_t)NC * CHUNK;
}

template <int NBW, int PG>
THA4_DEV void zero_acc(f32x4 (&acc)[NBW][PG]) {
#pragma unroll
  for (int b = 0; b < NBW; ++b)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) acc[b][pg] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// sine hidden layer: act <- sin(30 (W act + b)).  Output block b becomes input quad b of the next layer.
template <class G, int NB, int KQ, int CQ, int NEXT_PIECES>
THA4_DEV void sine_layer(const char*& gw, const float*& bias, char* ring, int& slot, f32x4* actv, const WaveCtx& w) {
  static_assert(NB % G::MS == 0, "row split must divide the block count");
  constexpr int NBW = NB / G::MS, PG = G::PG;
  const int mbase = w.ms * NBW;
  f32x4 acc[NBW][PG];
  zero_acc<NBW, PG>(acc);
  gemm_stream<G, NB, NBW, KQ, CQ, NEXT_PIECES>(gw, ring, slot, actv, acc, w, mbase, true);
  const int g4 = (w.lane >> 4) * 4;
#pragma unroll
  for (int b = 0; b < NBW; ++b) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + (mbase + b) * 16 + g4);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      f32x4 v = acc[b][pg] + bb;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = sin_omega(v[j]);
      actv[(pg * G::ACTQ + mbase + b) * 64 + w.lane] = v;
    }
  }
  bias += NB * 16;
  if (G::MS > 1) __syncthreads();   // the slot's other row-split wave reads these blocks next
}

// z layer: z = W act, written to global as z[n][b][g][pix][4] (one 256 B run per (block, lane group, pixel group))
template <class G, int NB, int KQ, int CQ>
THA4_DEV void z_layer(const char*& gw, char* ring, int& slot, const f32x4* actv, float* zframe, int npix,
                      const int (&pix0)[G::PG], const WaveCtx& w) {
  static_assert(NB % G::MS == 0, "row split must divide the block count");
  constexpr int NBW = NB / G::MS, PG = G::PG;
  const int mbase = w.ms * NBW;
  f32x4 acc[NBW][PG];
  zero_acc<NBW, PG>(acc);
  gemm_stream<G, NB, NBW, KQ, CQ, 0>(gw, ring, slot, actv, acc, w, mbase, true);
  const int p = w.lane & 15, g4 = (w.lane >> 4) * 4;
#pragma unroll
  for (int b = 0; b < NBW; ++b)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg)
      *reinterpret_cast<f32x4*>(zframe + z_offset(mbase + b, w.lane >> 4, pix0[pg] + p, npix)) = acc[b][pg];
}

// first layer from position only: act <- sin(30 (wx x + wy y + pb))      (pose folded into pb)
template <class G, int NB>
THA4_DEV void first_layer_pos(const float* wx, const float* wy, const float* pb, const float (&x)[G::PG],
                              const float (&y)[G::PG], f32x4* actv, const WaveCtx& w) {
  static_assert(NB % G::MS == 0, "row split must divide the block count");
  constexpr int NBW = NB / G::MS, PG = G::PG;
  const int g4 = (w.lane >> 4) * 4, mbase = w.ms * NBW;
#pragma unroll
  for (int bb = 0; bb < NBW; ++bb) {
    const int b = mbase + bb;
    const f32x4 vx = *reinterpret_cast<const f32x4*>(wx + b * 16 + g4);
    const f32x4 vy = *reinterpret_cast<const f32x4*>(wy + b * 16 + g4);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(pb + b * 16 + g4);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = sin_omega(fmaf(vx[j], x[pg], fmaf(vy[j], y[pg], vb[j])));
      actv[(pg * G::ACTQ + b) * 64 + w.lane] = v;
    }
  }
}

// bilinear x2 taps of F.interpolate(align_corners=False): src = max(0,(d+0.5)/2-0.5)
THA4_DEV void up2_taps(int d, int n, int& i0, int& i1, float& l0, float& l1) {
  const float src = fmaxf(0.0f, (d + 0.5f) * 0.5f - 0.5f);
  i0 = (int)src;
  i1 = min(i0 + 1, n - 1);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

// first layer of level 1/2: act <- sin(30 (upsample2x(z)[pixel] + wx x + wy y + pb))
template <class G, int NB>
THA4_DEV void first_layer_up(const float* zframe, int lowS, const float* wx, const float* wy, const float* pb,
                             const int (&X0)[G::PG], const int (&Y)[G::PG], const float (&x)[G::PG],
                             const float (&y)[G::PG], f32x4* actv, const WaveCtx& w) {
  static_assert(NB % G::MS == 0, "row split must divide the block count");
  constexpr int NBW = NB / G::MS, PG = G::PG;
  const int p = w.lane & 15, g4 = (w.lane >> 4) * 4, mbase = w.ms * NBW;
  const int npix = lowS * lowS;
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    int x0, x1, y0, y1;
    float lx0, lx1, ly0, ly1;
    up2_taps(X0[pg] + p, lowS, x0, x1, lx0, lx1);
    up2_taps(Y[pg], lowS, y0, y1, ly0, ly1);
    const float* z00 = zframe + z_offset(0, w.lane >> 4, y0 * lowS + x0, npix);
    const float* z01 = zframe + z_offset(0, w.lane >> 4, y0 * lowS + x1, npix);
    const float* z10 = zframe + z_offset(0, w.lane >> 4, y1 * lowS + x0, npix);
    const float* z11 = zframe + z_offset(0, w.lane >> 4, y1 * lowS + x1, npix);
#pragma unroll
    for (int bb = 0; bb < NBW; ++bb) {
      const int b = mbase + bb;
      const size_t off = (size_t)b * npix * 16;
      const f32x4 a = *reinterpret_cast<const f32x4*>(z00 + off);
      const f32x4 bq = *reinterpret_cast<const f32x4*>(z01 + off);
      const f32x4 c = *reinterpret_cast<const f32x4*>(z10 + off);
      const f32x4 d = *reinterpret_cast<const f32x4*>(z11 + off);
      const f32x4 vx = *reinterpret_cast<const f32x4*>(wx + b * 16 + g4);
      const f32x4 vy = *reinterpret_cast<const f32x4*>(wy + b * 16 + g4);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(pb + b * 16 + g4);
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float up = ly0 * (lx0 * a[j] + lx1 * bq[j]) + ly1 * (lx0 * c[j] + lx1 * d[j]);
        v[j] = sin_omega(up + fmaf(vx[j], x[pg], fmaf(vy[j], y[pg], vb[j])));
      }
      actv[(pg * G::ACTQ + b) * 64 + w.lane] = v;
    }
  }
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Remapped, XCD k works on the k-th contiguous
// eighth of the tile sequence, so vertically neighbouring strips - which share the rows of the x2-upsample taps -
// share an L2 instead of each fetching them from HBM (level 2: FETCH_SIZE 67 MB -> see profiles/).
THA4_DEV int xcd_tile(int block, int nblocks) {
  constexpr int kXcd = 8;
  return nblocks % kXcd == 0 ? (block % kXcd) * (nblocks / kXcd) + block / kXcd : block;
}

// pixel groups of this wave's slot: global id (over the batch) -> frame, strip origin, positions
template <class G, int S>
THA4_DEV int slot_pixels(const WaveCtx& w, const float* axis, int (&pix0)[G::PG], int (&X0)[G::PG], int (&Y)[G::PG],
                         float (&px)[G::PG], float (&py)[G::PG]) {
  constexpr int PGS = S * S / 16;
  static_assert(PGS % (G::NS * G::PG) == 0, "a workgroup must not straddle frames");
  const int pg_first = (xcd_tile(w.blk, w.nblk) * G::NS + w.ns) * G::PG;
#pragma unroll
  for (int pg = 0; pg < G::PG; ++pg) {
    pix0[pg] = ((pg_first + pg) % PGS) * 16;
    X0[pg] = pix0[pg] % S;
    Y[pg] = pix0[pg] / S;
    px[pg] = axis[X0[pg] + (w.lane & 15)];
    py[pg] = axis[Y[pg]];
  }
  return pg_first / PGS;   // frame index
}

// ---------------------------------------------------------------------------------------------
// kernel 0: pose-folded first-layer biases.  pb[n][c] = b[c] + sum_k Wpose[k][c] pose[n][k]
// grid (ceil(kPbStride/64), B), 64 threads: latency-bound, so spread over many CUs
// ---------------------------------------------------------------------------------------------
constexpr int kPoseBiasBlock = 64;
__global__ void __launch_bounds__(kPoseBiasBlock) posebias_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  const int idx = blockIdx.x * kPoseBiasBlock + threadIdx.x;
  const int n = blockIdx.y;
  if (idx >= kPbStride) return;
  int net, c, width;
  if (idx < kPbL0) { net = 0; c = idx - kPbFace; width = kNBF * 16; }
  else if (idx < kPbL1) { net = 1; c = idx - kPbL0; width = kNB0 * 16; }
  else if (idx < kPbL2) { net = 2; c = idx - kPbL1; width = kNB1 * 16; }
  else { net = 3; c = idx - kPbL2; width = kNB2 * 16; }
  // wpose is zero-padded to kPose rows for every net (the face net only sees pose[0:39], mode_14.py:66),
  // so all 45 loads are independent and issued up front: the kernel is pure load latency otherwise
  const float* wp = d.wpose[net] + c;
  const float* pose = d.pose + (size_t)n * kPose;
  float s[3] = {d.bias1[net][c], 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < kPose; ++k) s[k % 3] = fmaf(wp[(size_t)k * width], pose[k], s[k % 3]);
  d.pbias[(size_t)n * kPbStride + idx] = (s[0] + (s[1] + s[2])) * d.pb_scale;
}

// ---------------------------------------------------------------------------------------------
// kernel 1: face morpher (128x128): 41->128 (VALU, pose folded), 7 x 128->128, 128->4
// ---------------------------------------------------------------------------------------------
template <int NS, int MS, int PG, int CQ>
struct FaceCfg {
  static constexpr int kSlotPieces = CQ * kNBF > 8 ? CQ * kNBF : 8;
  using G = Geo<NS, MS, PG, kNBF, kSlotPieces>;
};

template <int NS, int MS, int PG, int CQ>
__global__ void __launch_bounds__(NS* MS * 64) face_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  using G = typename FaceCfg<NS, MS, PG, CQ>::G;
  constexpr int S = kFaceSize, NPIX = S * S;
  THA4_DYN_LDS(smem);
  const WaveCtx w = wave_ctx<G>();
  char* ring = smem;
  f32x4* actv = reinterpret_cast<f32x4*>(smem + 2 * G::SLOT + w.ns * G::ACT_BYTES);
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos128, pix0, X0, Y, px, py);
  const char* gw = reinterpret_cast<const char*>(d.w_face);
  const float* bias = d.b_face;
  int slot = 0;
  fetch_pieces<CQ * kNBF, G::WAVES>(gw, ring, w.wave, w.lane);
  first_layer_pos<G, kNBF>(d.wx[0], d.wy[0], d.pbias + (size_t)n * kPbStride + kPbFace, px, py, actv, w);
  __syncthreads();
#pragma unroll 1
  for (int l = 0; l < 6; ++l) sine_layer<G, kNBF, kNBF, CQ, CQ * kNBF>(gw, bias, ring, slot, actv, w);
  sine_layer<G, kNBF, kNBF, CQ, 8>(gw, bias, ring, slot, actv, w);
  // last_linear 128 -> 4 (no nonlinearity, siren.py:87-91): one block, rows 0..3 live in lane group 0
  f32x4 acc[1][PG];
  zero_acc<1, PG>(acc);
  gemm_stream<G, 1, 1, kNBF, kNBF, 0>(gw, ring, slot, actv, acc, w, 0, w.ms == 0);
  if (w.ms == 0 && w.lane < 16) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias);
    float* fo = d.face + (size_t)n * 4 * NPIX;
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      const f32x4 v = acc[0][pg] + bb;
#pragma unroll
      for (int j = 0; j < 4; ++j) fo[(size_t)j * NPIX + pix0[pg] + w.lane] = v[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// kernel 2: body level 0 (128x128): 47->360 (VALU, pose folded) ->360 ->180, then z1 = W10[:, :180] h0
// ---------------------------------------------------------------------------------------------
template <int NS, int MS, int PG, int CQB>
struct Level0Cfg {   // the 23-quad layers stream one quad per chunk (23 is prime); CQB: chunk of the 12-quad z layer
  static constexpr int kP1 = kNB0, kP2 = kNB1, kP3 = CQB * kNB1;
  static constexpr int kSlotPieces = kP1 > kP3 ? kP1 : kP3;
  using G = Geo<NS, MS, PG, kNB0, kSlotPieces>;
};

template <int NS, int MS, int PG, int CQB>
__global__ void __launch_bounds__(NS* MS * 64) level0_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  using Cfg = Level0Cfg<NS, MS, PG, CQB>;
  using G = typename Cfg::G;
  constexpr int S = 128, NPIX = S * S;
  THA4_DYN_LDS(smem);
  const WaveCtx w = wave_ctx<G>();
  char* ring = smem;
  f32x4* actv = reinterpret_cast<f32x4*>(smem + 2 * G::SLOT + w.ns * G::ACT_BYTES);
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos128, pix0, X0, Y, px, py);
  const char* gw = reinterpret_cast<const char*>(d.w_l0);
  const float* bias = d.b_l0;
  int slot = 0;
  fetch_pieces<Cfg::kP1, G::WAVES>(gw, ring, w.wave, w.lane);
  first_layer_pos<G, kNB0>(d.wx[1], d.wy[1], d.pbias + (size_t)n * kPbStride + kPbL0, px, py, actv, w);
  __syncthreads();
  sine_layer<G, kNB0, kKQ0, 1, Cfg::kP2>(gw, bias, ring, slot, actv, w);
  sine_layer<G, kNB1, kKQ0, 1, Cfg::kP3>(gw, bias, ring, slot, actv, w);
  z_layer<G, kNB1, kNB1, CQB>(gw, ring, slot, actv, d.z1 + (size_t)n * kNB1 * NPIX * 16, NPIX, pix0, w);
}

// ---------------------------------------------------------------------------------------------
// kernel 3: body level 1 (256x256): up2(z1)+pos+pose -> sin ->180 ->90, then z2 = W20[:, :90] h1
// ---------------------------------------------------------------------------------------------
template <int NS, int MS, int PG, int CQA, int CQB>
struct Level1Cfg {   // CQA: chunk of the 12-quad layers; CQB: chunk of the 6-quad z layer
  static constexpr int kP1 = CQA * kNB1, kP2 = CQA * kNB2, kP3 = CQB * kNB2;
  static constexpr int kSlotPieces = kP1 > kP3 ? kP1 : kP3;
  using G = Geo<NS, MS, PG, kNB1, kSlotPieces>;
};

template <int NS, int MS, int PG, int CQA, int CQB>
__global__ void __launch_bounds__(NS* MS * 64) level1_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  using Cfg = Level1Cfg<NS, MS, PG, CQA, CQB>;
  using G = typename Cfg::G;
  constexpr int S = 256, NPIX = S * S;
  THA4_DYN_LDS(smem);
  const WaveCtx w = wave_ctx<G>();
  char* ring = smem;
  f32x4* actv = reinterpret_cast<f32x4*>(smem + 2 * G::SLOT + w.ns * G::ACT_BYTES);
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos256, pix0, X0, Y, px, py);
  const char* gw = reinterpret_cast<const char*>(d.w_l1);
  const float* bias = d.b_l1;
  int slot = 0;
  fetch_pieces<Cfg::kP1, G::WAVES>(gw, ring, w.wave, w.lane);
  first_layer_up<G, kNB1>(d.z1 + (size_t)n * kNB1 * (128 * 128) * 16, 128, d.wx[2], d.wy[2],
                          d.pbias + (size_t)n * kPbStride + kPbL1, X0, Y, px, py, actv, w);
  __syncthreads();
  sine_layer<G, kNB1, kNB1, CQA, Cfg::kP2>(gw, bias, ring, slot, actv, w);
  sine_layer<G, kNB2, kNB1, CQA, Cfg::kP3>(gw, bias, ring, slot, actv, w);
  z_layer<G, kNB2, kNB2, CQB>(gw, ring, slot, actv, d.z2 + (size_t)n * kNB2 * NPIX * 16, NPIX, pix0, w);
}

// ---------------------------------------------------------------------------------------------
// kernel 4: body level 2 (512x512): up2(z2)+pos+pose -> sin ->90 ->90 -> head(7) -> warp -> blend
// ---------------------------------------------------------------------------------------------
// source of the warp = the image with the face patch pasted in (mode_14.py:72-78), read in place
THA4_DEV float body_source(const float* img, const float* face, int c, int y, int x) {
  const int fy = y - kFaceTop, fx = x - kFaceLeft;
  if ((unsigned)fy < (unsigned)kFaceSize && (unsigned)fx < (unsigned)kFaceSize)
    return face[((size_t)c * kFaceSize + fy) * kFaceSize + fx];
  return img[((size_t)c * kImg + y) * kImg + x];
}

template <int NS, int MS, int PG, int CQ>
struct Level2Cfg {
  static constexpr int kSlotPieces = CQ * kNB2;
  using G = Geo<NS, MS, PG, kNB2, kSlotPieces>;
};

template <int NS, int MS, int PG, int CQ>
__global__ void __launch_bounds__(NS* MS * 64) level2_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  using G = typename Level2Cfg<NS, MS, PG, CQ>::G;
  constexpr int S = kImg, NPIX = S * S;
  THA4_DYN_LDS(smem);
  const WaveCtx w = wave_ctx<G>();
  char* ring = smem;
  f32x4* actv = reinterpret_cast<f32x4*>(smem + 2 * G::SLOT + w.ns * G::ACT_BYTES);
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos512, pix0, X0, Y, px, py);
  const char* gw = reinterpret_cast<const char*>(d.w_l2);
  const float* bias = d.b_l2;
  int slot = 0;
  fetch_pieces<CQ * kNB2, G::WAVES>(gw, ring, w.wave, w.lane);
  first_layer_up<G, kNB2>(d.z2 + (size_t)n * kNB2 * (256 * 256) * 16, 256, d.wx[3], d.wy[3],
                          d.pbias + (size_t)n * kPbStride + kPbL2, X0, Y, px, py, actv, w);
  __syncthreads();
  sine_layer<G, kNB2, kNB2, CQ, CQ * kNB2>(gw, bias, ring, slot, actv, w);
  sine_layer<G, kNB2, kNB2, CQ, kNB2>(gw, bias, ring, slot, actv, w);
  // last_linear 90 -> 7: rows 0..3 (dx, dy, alpha, colour R) in lane group 0, rows 4..6 (G, B, A) in group 1
  f32x4 acc[1][PG];
  zero_acc<1, PG>(acc);
  gemm_stream<G, 1, 1, kNB2, kNB2, 0>(gw, ring, slot, actv, acc, w, 0, w.ms == 0);
  if (w.ms != 0) return;

  const int p = w.lane & 15, g = w.lane >> 4;
  const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + g * 4);
  const float* img = d.image + (size_t)n * d.image_stride;
  const float* face = d.face + (size_t)n * 4 * kFaceSize * kFaceSize;
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    const f32x4 v = acc[0][pg] + bb;
    const float dx = lane_read(v[0], p), dy = lane_read(v[1], p), al = lane_read(v[2], p);
    const float c0 = lane_read(v[3], p), c1 = lane_read(v[0], p + 16), c2 = lane_read(v[1], p + 16),
                c3 = lane_read(v[2], p + 16);
    const float col = g == 0 ? c0 : (g == 1 ? c1 : (g == 2 ? c2 : c3));   // lane group g handles image channel g
    // GridChangeApplier.apply: grid = affine_grid(identity) + change; grid_sample(bilinear, border, align_corners=False)
    const float gx = px[pg] + dx, gy = py[pg] + dy;
    float ix = ((gx + 1.0f) * (float)S - 1.0f) * 0.5f;
    float iy = ((gy + 1.0f) * (float)S - 1.0f) * 0.5f;
    ix = fminf((float)(S - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(S - 1), fmaxf(iy, 0.0f));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float tx = ix - fx0, ty = iy - fy0;
    const float wnw = (1.0f - tx) * (1.0f - ty), wne = tx * (1.0f - ty), wsw = (1.0f - tx) * ty, wse = tx * ty;
    const int x1 = min(x0 + 1, S - 1), y1 = min(y0 + 1, S - 1);   // out-of-range taps only occur with weight 0
    float wv = body_source(img, face, g, y0, x0) * wnw;
    wv += body_source(img, face, g, y0, x1) * wne;
    wv += body_source(img, face, g, y1, x0) * wsw;
    wv += body_source(img, face, g, y1, x1) * wse;
    const float blended = (1.0f - al) * wv + al * col;            // siren_morpher_03.py:131
    const size_t pix = (size_t)pix0[pg] + p;
    if (d.out_rgba8) store_display(d, n, pix, g, p, blended);      // (first: it LOADS the background colour, and a load behind a store waits for the store's acknowledgement)
    if (d.out_blended) d.out_blended[((size_t)n * 4 + g) * NPIX + pix] = blended;
    if (d.out_color) d.out_color[((size_t)n * 4 + g) * NPIX + pix] = col;
    if (d.out_warped) d.out_warped[((size_t)n * 4 + g) * NPIX + pix] = wv;
    if (d.out_alpha && g == 0) d.out_alpha[(size_t)n * NPIX + pix] = al;
    if (d.out_grid && g < 2) d.out_grid[((size_t)n * 2 + g) * NPIX + pix] = (g == 0 ? dx : dy);
  }
}

// ---------------------------------------------------------------------------------------------
// launch configuration shared by the C-ABI launcher and the CPU emulator tests.
// Every knob can be overridden with -D at build time (tools/sweep builds variant libraries).
// ---------------------------------------------------------------------------------------------
namespace cfg {
#ifndef THA4_FACE_CFG
#define THA4_FACE_CFG 4, 2, 1, 8        // NS, MS, PG, CQ   (whole layer per chunk: 1 barrier per layer)
#endif
#ifndef THA4_L0_CFG
#define THA4_L0_CFG 4, 2, 1, 2          // NS, MS, PG, CQB
#endif
#ifndef THA4_L1_CFG
#define THA4_L1_CFG 8, 1, 1, 2, 3       // NS, MS, PG, CQA, CQB
#endif
#ifndef THA4_L2_CFG
#define THA4_L2_CFG 4, 1, 1, 2          // NS, MS, PG, CQ   (48 KiB LDS: 3 workgroups per CU run out of phase)
#endif
using FaceG = FaceCfg<THA4_FACE_CFG>::G;
using L0G = Level0Cfg<THA4_L0_CFG>::G;
using L1G = Level1Cfg<THA4_L1_CFG>::G;
using L2G = Level2Cfg<THA4_L2_CFG>::G;
#define THA4_FACE_KERNEL face_kernel<THA4_FACE_CFG>
#define THA4_L0_KERNEL level0_kernel<THA4_L0_CFG>
#define THA4_L1_KERNEL level1_kernel<THA4_L1_CFG>
#define THA4_L2_KERNEL level2_kernel<THA4_L2_CFG>

template <class G>
constexpr int blocks_for(int batch, int side) { return batch * (side * side) / G::PX; }
constexpr int posebias_blocks() { return (kPbStride + kPoseBiasBlock - 1) / kPoseBiasBlock; }
}  // namespace cfg

}  // namespace tha4
