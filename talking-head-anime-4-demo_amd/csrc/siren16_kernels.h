// Second-generation student kernels: the SIREN contractions run on v_mfma_f32_16x16x32_f16 with BOTH
// operands split into fp16 hi + fp16 lo halves (v = hi + lo, 22 significant bits):
//     W x  ~=  W_hi x_hi  +  W_hi x_lo  +  W_lo x_hi            (3 MFMAs into ONE fp32 accumulator)
// fp16 products are exact in the fp32 accumulator; what is dropped is the lo*lo term (2^-22 relative) and the
// rounding of the lo halves (2^-22), i.e. about two bits less than an exact-fp32 product.  End to end the posed
// frame moves by < 6e-4 against the exact-fp32 kernels and sits at the SAME distance from the fp64 reference
// (1.5e-4 in a numpy simulation of the whole student; GPU parity gates on 1e-3 as before) - single-pass fp16/bf16
// weights would be off by 0.25 / 1.6.  Per 32-channel K group the matrix pipe spends 3 x 16 cycles
// instead of 8 x 32 (v_mfma_f32_16x16x4_f32): 5.3x fewer MFMA cycles, which moves the bound to the VALU - and VALU
// work does NOT hide under MFMAs on this chip (tools/microbench/valu_rate.hip) - so everything per-value is trimmed:
//   * weights are packed as W' = c S W with c = omega_0 / (2 pi) = 4.7746 for sine layers (and for the z hand-off layers,
//     whose consumer is a sine) - the sine takes its argument in TURNS (round 3, THA4_SIN_TURNS) - and S a per-layer power of
//     two that lifts the lo halves of small weights out of the fp16 subnormal range; the epilogue is ONE fma
//     t = acc / S + c b  and the sine needs no multiply;
//   * activations are in [-1, 1]: their lo halves are stored unscaled (abs error <= 2^-25, MFMA keeps fp16
//     subnormals), so the split of a PAIR of values is one v_cvt_pk_f16_f32 + one v_fma_mix per value (split_pair,
//     tha4_platform.h) and one accumulator serves all 3 MFMAs;
//   * sin(2 pi t) is ONE v_sin_f32 (sin_u, siren_kernels.h: the instruction reduces its argument itself, valid for
//     |t| <= 256 turns - checked for every layer when the weights are packed, siren_layout.h - 3.8e-7 max abs error).
//     (-DTHA4_SIN_TURNS=0 rebuilds round 2's pipeline: c = 30, radians, magic-add rint + 2-term Cody-Waite + degree-9
//     polynomial, 12 VALU ops per sine.)
// Everything else (fused level chains, streamed fragment-linear weights, z hand-off, warp epilogue) is the design of
// siren_kernels.h.
//
// Images.  K group Q = 32 channels = output blocks 2Q, 2Q+1 of the producing layer.  k-slot (g, j) of lane group
// g = lane>>4, j = 0..7 is channel 32Q + 16(j>>2) + 4g + (j&3), so the 4 rows a lane holds of block 2Q (j<4) and
// of block 2Q+1 (j>=4) ARE its B fragment for the next layer (C/D layout: row 4g+r, col lane&15).
//   weights     piece (Q, b) = 2 KiB: [hi: lane x 8 halves | lo: lane x 8 halves], W'[16b+(lane&15)][k-slot channel]
//   activations per pixel group and K group: 2 KiB: [hi: lane x 8 halves | lo: lane x 8 halves]
#pragma once
#include "siren_kernels.h"

namespace tha4 {
namespace v2 {

// In-kernel time stamps of the tuning builds (-DTHA4_STAMPS, tools/stamps_student.py): lane 0 of a chosen wave stores the shader clock into the (otherwise
// unused) pose-bias workspace, 64 stamps per slot.  The product build compiles them out.
#if defined(THA4_STAMPS) && !defined(THA4_EMU)
#define THA4_STAMP(d, on, slot, k)                                                                                       \
  do {                                                                                                                   \
    if ((on) && (threadIdx.x & 63) == 0) reinterpret_cast<unsigned long long*>((d).pbias)[(slot) * 64 + (k)] = clock64(); \
  } while (0)
// entry / exit of EVERY workgroup in the 100-MHz wall clock (comparable across XCDs): u64 [4 k .. 4 k + 3] of span slot k = min entry, max entry, min exit, max exit
#define THA4_SPAN(d, k, exit_)                                                                                                        \
  do {                                                                                                                               \
    if (threadIdx.x == 0) {                                                                                                          \
      unsigned long long* sp_ = reinterpret_cast<unsigned long long*>((d).pbias) + 320 + 4 * (k) + 2 * (exit_);                       \
      const unsigned long long t_ = wall_clock64();                                                                                  \
      atomicMin(sp_, t_);                                                                                                            \
      atomicMax(sp_ + 1, t_);                                                                                                        \
    }                                                                                                                                \
  } while (0)
#else
#define THA4_STAMP(d, on, slot, k)
#define THA4_SPAN(d, k, exit_)
#endif


// ---- host/device shared layout helpers ---------------------------------------------------------
// order of the pieces of one K group in the weight stream when a chunk is 1/HB of the group's blocks and MS waves
// split the rows: piece index of wave-local block bl of row split ms
constexpr int piece_index(int NB, int MS, int HB, int ms, int bl) {
  const int NBW = NB / MS, per = NBW / HB;
  return (bl / per) * (NB / HB) + ms * per + (bl % per);
}
constexpr int piece_block(int NB, int MS, int HB, int idx) {   // inverse: global block stored at piece idx
  const int NBW = NB / MS, per = NBW / HB;
  const int h = idx / (NB / HB), r = idx % (NB / HB);
  return (r / per) * NBW + h * per + (r % per);
}

// split 4 fp32 values (|v| <= 1: sines) into fp16 hi + fp16 lo, v = hi + lo
THA4_DEV void split4(const f32x4& v, f16x4& hi, f16x4& lo) {
  const float m1 = split_minus_one();
  _Float16 h[4], l[4];
  split_pair(v[0], v[1], m1, h[0], h[1], l[0], l[1]);
  split_pair(v[2], v[3], m1, h[2], h[3], l[2], l[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) { hi[j] = h[j]; lo[j] = l[j]; }
}

// act image of one pixel slot: [pg][Q][hi|lo][lane][8 halves]; block b -> (Q = b>>1, half = b&1)
template <class G>
THA4_DEV void store_block(char* act, int pg, int b, int lane, const f32x4& v) {
  f16x4 hi, lo;
  split4(v, hi, lo);
  char* base = act + ((size_t)(pg * G::ACTQ + (b >> 1)) * 2) * 1024 + lane * 16 + (b & 1) * 8;
  *reinterpret_cast<f16x4*>(base) = hi;
  *reinterpret_cast<f16x4*>(base + 1024) = lo;
}

// 16-byte load at (wave-uniform pointer) + (32-bit per-lane byte offset): the compiler emits the SGPR-base form of global_load and
// no 64-bit VALU address arithmetic (v_lshl_add_u64 / v_add_co + v_addc per load otherwise)
THA4_DEV f32x4 ldg4(const float* uniform_base, unsigned lane_bytes) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(uniform_base) + lane_bytes);
}

// Geometry: ACTQ counts K groups (2 KiB each) here; SLOT pieces are 2 KiB.
template <int NS_, int MS_, int PG_, int ACTG_, int SLOT_PIECES_>
struct Geo16 {
  static constexpr int NS = NS_, MS = MS_, PG = PG_, ACTQ = ACTG_;
  static constexpr int WAVES = NS * MS, THREADS = WAVES * 64;
  static constexpr int SLOT = SLOT_PIECES_ * 2048;
  static constexpr int ACT_BYTES = PG * ACTG_ * 2048;
  static constexpr int LDS = 2 * SLOT + NS * ACT_BYTES;
  static constexpr int PX = NS * PG * 16;
  static_assert(LDS <= 160 * 1024, "LDS budget exceeded");
};

template <int PIECES, int WAVES>
THA4_DEV void fetch2k(const char* g, char* l, int wave, int lane) {   // PIECES x 2 KiB = 2*PIECES x 1 KiB copies
  fetch_pieces<2 * PIECES, WAVES>(g, l, wave, lane);
}

// MFMAs of one resident chunk: CQ K groups x BPC of this wave's blocks (pieces [qq][ROW pieces per group] at wv,
// activations [pg][group] at av), accumulating into acc[bo0 + ...].  Software pipelined: the next block group's
// fragments are read from LDS under the last two thirds of the current group's MFMAs.
template <class G, int ROW, int BPC, int CQ, int NBW>
THA4_DEV void mma_chunk(const char* wv, const char* av, f32x4 (&acc)[NBW][G::PG], int bo0) {
  constexpr int PG = G::PG;
  constexpr int GB = group_blocks(BPC), NGB = BPC / GB, T = CQ * NGB;
  f16x8 ah[2][GB], al[2][GB], bh[2][PG], bl[2][PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    bh[0][pg] = *reinterpret_cast<const f16x8*>(av + (size_t)pg * G::ACTQ * 2048);
    bl[0][pg] = *reinterpret_cast<const f16x8*>(av + (size_t)pg * G::ACTQ * 2048 + 1024);
  }
#pragma unroll
  for (int b = 0; b < GB; ++b) {
    ah[0][b] = *reinterpret_cast<const f16x8*>(wv + (size_t)b * 2048);
    al[0][b] = *reinterpret_cast<const f16x8*>(wv + (size_t)b * 2048 + 1024);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int qq = t / NGB, g = t % NGB;
    const int bo = bo0 + g * GB;                            // first wave-local block of this group (compile-time after inlining)
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(ah[t & 1][b], bh[qq & 1][pg], acc[bo + b][pg]);
    THA4_SCHED_FENCE();
    if (t + 1 < T) {
      const int nq = (t + 1) / NGB, ng = (t + 1) % NGB;
      if (ng == 0) {
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
          bh[nq & 1][pg] = *reinterpret_cast<const f16x8*>(av + ((size_t)pg * G::ACTQ + nq) * 2048);
          bl[nq & 1][pg] = *reinterpret_cast<const f16x8*>(av + ((size_t)pg * G::ACTQ + nq) * 2048 + 1024);
        }
      }
#pragma unroll
      for (int b = 0; b < GB; ++b) {
        const char* pc = wv + ((size_t)nq * ROW + ng * GB + b) * 2048;
        ah[(t + 1) & 1][b] = *reinterpret_cast<const f16x8*>(pc);
        al[(t + 1) & 1][b] = *reinterpret_cast<const f16x8*>(pc + 1024);
      }
    }
    THA4_SCHED_FENCE();
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(ah[t & 1][b], bl[qq & 1][pg], acc[bo + b][pg]);
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(al[t & 1][b], bh[qq & 1][pg], acc[bo + b][pg]);
    THA4_SCHED_FENCE();
  }
}

// One linear layer, this wave's share (NBW of the NB output blocks) for its PG pixel groups.
//   KG K groups of 32 channels; chunk = CQ whole groups (HB == 1) or one 1/HB slice of a group's blocks (HB > 1, CQ == 1)
//   acc += W'hi Xhi + W'hi Xlo + W'lo Xhi;   W x = acc / S
template <class G, int NB, int NBW, int KG, int HB, int CQ, int NEXT_PIECES>
THA4_DEV void gemm16_stream(const char*& gw, char* ring, int& slot, const char* act, f32x4 (&acc)[NBW][G::PG], const WaveCtx& w,
                            bool active) {
  static_assert(HB == 1 || CQ == 1, "block slicing and multi-group chunks are exclusive");
  static_assert(NBW % HB == 0 && KG % CQ == 0, "bad chunking");
  constexpr int PG = G::PG;
  constexpr int CHUNK_PIECES = (NB / HB) * CQ;
  constexpr int NC = (KG / CQ) * HB;
  constexpr int CHUNK = CHUNK_PIECES * 2048;
  constexpr int BPC = NBW / HB;                               // this wave's blocks per chunk (per group)
  static_assert(CHUNK <= G::SLOT && NEXT_PIECES * 2048 <= G::SLOT, "chunk exceeds ring slot");
#pragma unroll 1
  for (int cg = 0; cg < KG / CQ; ++cg) {
#pragma unroll
  for (int h = 0; h < HB; ++h) {      // block slice: compile-time, so accumulator indices stay static
    const int c = cg * HB + h;
    const int nslot = slot ^ 1;
    if (c + 1 < NC) {
      fetch2k<CHUNK_PIECES, G::WAVES>(gw + (size_t)(c + 1) * CHUNK, ring + nslot * G::SLOT, w.wave, w.lane);
    } else if (NEXT_PIECES > 0) {
      fetch2k<NEXT_PIECES, G::WAVES>(gw + (size_t)NC * CHUNK, ring + nslot * G::SLOT, w.wave, w.lane);
    }
    if (active) {
      // this wave's pieces inside the slot: slice-local index = ms*BPC + i  (see piece_index)
      const char* wv = ring + slot * G::SLOT + (size_t)(w.ms * BPC) * 2048 + w.lane * 16;
      const char* av = act + (size_t)(cg * CQ) * 2048 + w.lane * 16;
      mma_chunk<G, NB / HB, BPC, CQ, NBW>(wv, av, acc, h * BPC);
    }
    THA4_HOOK_CHUNK_BARRIER();
    slot = nslot;
  }
  }
  gw += (size_t)NC * CHUNK;
}

#ifndef THA4_BIAS_AHEAD
#define THA4_BIAS_AHEAD 1
#endif
// sine hidden layer: act <- split(sin(acc / S + 30 b))          (30 and S are folded into the packed weights)
template <class G, int NB, int KG, int HB, int CQ, int NEXT_PIECES>
THA4_DEV void sine16_layer(const char*& gw, const float*& bias, const float*& scl, char* ring, int& slot, char* act, const WaveCtx& w) {
  static_assert(NB % G::MS == 0, "row split must divide the block count");
  constexpr int NBW = NB / G::MS, PG = G::PG;
  const int mbase = w.ms * NBW;
  f32x4 acc[NBW][PG];
  zero_acc<NBW, PG>(acc);
  // The layer's scale and biases are requested HERE, ahead of the GEMM (THA4_BIAS_AHEAD, round 4): requested in the epilogue they sat behind the
  // global_load_lds of the NEXT layer's first weight chunk in the in-order vmcnt queue - every wave's sine epilogue waited for that whole chunk
  // to land instead of running under it (tools/isa_waits.py: vmcnt(5) right behind three glds).  Older than every ring fetch, they do not disturb the
  // counted chunk waits.
  const int g4 = (w.lane >> 4) * 4;
  float inv_pre = 0.f;
  f32x4 bias_pre[NBW];
  if (THA4_BIAS_AHEAD) {
    inv_pre = *scl;
#pragma unroll
    for (int b = 0; b < NBW; ++b) bias_pre[b] = ldg4(bias + (mbase + b) * 16, g4 * 4u);
  }
  gemm16_stream<G, NB, NBW, KG, HB, CQ, NEXT_PIECES>(gw, ring, slot, act, acc, w, true);
  THA4_PRIO_VALU();
  const float inv = THA4_BIAS_AHEAD ? inv_pre : *scl;
  ++scl;
#pragma unroll
  for (int b = 0; b < NBW; ++b) {
    const f32x4 bb = THA4_BIAS_AHEAD ? bias_pre[b] : ldg4(bias + (mbase + b) * 16, g4 * 4u);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = sin_u(fmaf(acc[b][pg][j], inv, bb[j]));
      store_block<G>(act, pg, mbase + b, w.lane, v);
    }
  }
  bias += NB * 16;
  THA4_PRIO_MFMA();
  if (G::MS > 1) __syncthreads();
}

// z layer: z = 30 W act (fp32) -> global z[n][b][g][pix][4]; the consumer is the first (sine) layer of the next level
template <class G, int NB, int KG, int HB, int CQ>
THA4_DEV void z16_layer(const char*& gw, const float*& scl, char* ring, int& slot, const char* act, float* zframe, int npix,
                        const int (&pix0)[G::PG], const WaveCtx& w) {
  constexpr int NBW = NB / G::MS, PG = G::PG;
  const int mbase = w.ms * NBW;
  f32x4 acc[NBW][PG];
  zero_acc<NBW, PG>(acc);
  const float inv = *scl++;                                // (requested ahead of the GEMM: see sine16_layer)
  gemm16_stream<G, NB, NBW, KG, HB, CQ, 0>(gw, ring, slot, act, acc, w, true);
  const int p = w.lane & 15;
#pragma unroll
  for (int b = 0; b < NBW; ++b)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg)
      *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(zframe + z_offset(mbase + b, 0, 0, npix)) +
                                (unsigned)(z_offset(0, w.lane >> 4, pix0[pg] + p, npix) * sizeof(float))) = acc[b][pg] * inv;
}

// The pose-folded first-layer bias of one network, computed by the CONSUMER: pb[c] = (b[c] + sum_k Wpose[k][c] pose[n][k])
// * scale for the NB*16 channels of network `net` of frame n, into LDS.  Same arithmetic (and summation order) as
// posebias_kernel, which generation 2 no longer launches: all 45 loads of a channel are independent and go out with the
// first weight chunk, so the fold costs a barrier instead of a 7 us launch in front of every frame.
#ifndef THA4_PB_FOLD
#define THA4_PB_FOLD 1      // 0: A/B build - posebias_kernel is launched and the kernels read its result from HBM as in round 1
#endif
constexpr int pb_lds_bytes(int nb) { return nb * 16 * (int)sizeof(float); }
template <int NB, int THREADS>
THA4_DEV void pose_bias_to_lds(const StudentDev& d, int net, int n, float* pb) {
  constexpr int W = NB * 16;
  if (!THA4_PB_FOLD) return;
  for (int c = threadIdx.x; c < W; c += THREADS) {
    const float* wp = d.wpose[net] + c;
    const float* pose = d.pose + (size_t)n * kPose;
    float s[3] = {d.bias1[net][c], 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kPose; ++k) s[k % 3] = fmaf(wp[(size_t)k * W], pose[k], s[k % 3]);
    pb[c] = (s[0] + (s[1] + s[2])) * d.pb_scale;
  }
}

template <class G, int NB>
THA4_DEV void first16_pos(const float* wx, const float* wy, const float* pb, const float (&x)[G::PG], const float (&y)[G::PG],
                          char* act, const WaveCtx& w) {
  constexpr int NBW = NB / G::MS, PG = G::PG;
  const int g4 = (w.lane >> 4) * 4, mbase = w.ms * NBW;
#pragma unroll
  for (int bb = 0; bb < NBW; ++bb) {
    const int b = mbase + bb;
    const f32x4 vx = ldg4(wx + b * 16, g4 * 4u);
    const f32x4 vy = ldg4(wy + b * 16, g4 * 4u);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(pb + b * 16 + g4);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = sin_u(fmaf(vx[j], x[pg], fmaf(vy[j], y[pg], vb[j])));     // tables carry the 30x
      store_block<G>(act, pg, b, w.lane, v);
    }
  }
}

#ifndef THA4_TAP_PIPELINE
#define THA4_TAP_PIPELINE 1   // 1: explicit two-block software pipeline of the first layers' tap requests (round 4); 0: the loop of rounds 2-3
#endif
#ifndef THA4_TAP_BLOCKS
#define THA4_TAP_BLOCKS 4     // blocks whose 4 upsample taps are requested together in the first layers (power of two)
#endif
// sink(pg, block, v): where the layer's output rows go (the LDS activation image, or registers)
template <class G, int NB, class Sink>
THA4_DEV void first16_up_to(const float* zframe, int lowS, const float* wx, const float* wy, const float* pb,
                            const int (&X0)[G::PG], const int (&Y)[G::PG], const float (&x)[G::PG], const float (&y)[G::PG],
                            Sink&& sink, const WaveCtx& w) {
  constexpr int NBW = NB / G::MS, PG = G::PG;
  const int p = w.lane & 15, g4 = (w.lane >> 4) * 4, mbase = w.ms * NBW;
  const int npix = lowS * lowS;
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    int x0, x1, y0, y1;
    float lx0, lx1, ly0, ly1;
    up2_taps(X0[pg] + p, lowS, x0, x1, lx0, lx1);
    up2_taps(Y[pg], lowS, y0, y1, ly0, ly1);
    const float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
    // tap addresses as (wave-uniform block base) + (32-bit per-lane byte offset): global_load with an SGPR base, no 64-bit VALU address
    // arithmetic per load (48 v_lshl_add_u64 per pixel group in level 1 otherwise)
    const unsigned o00 = (unsigned)(z_offset(0, w.lane >> 4, y0 * lowS + x0, npix) * sizeof(float));
    const unsigned o01 = (unsigned)(z_offset(0, w.lane >> 4, y0 * lowS + x1, npix) * sizeof(float));
    const unsigned o10 = (unsigned)(z_offset(0, w.lane >> 4, y1 * lowS + x0, npix) * sizeof(float));
    const unsigned o11 = (unsigned)(z_offset(0, w.lane >> 4, y1 * lowS + x1, npix) * sizeof(float));
    // Two blocks' requests in flight (THA4_TAP_PIPELINE, round 4): block bb + 1 is requested BEFORE block bb is consumed, pinned by scheduling
    // fences.  Left to itself the compiler serialised the blocks of the weights-resident level 2 - six loads, s_waitcnt vmcnt(0), six loads ... :
    // five dependent memory round trips per strip (tools/isa_waits.py) where the streamed level 1 already overlapped two blocks.
    struct Taps { f32x4 vx, a, bq, c, d, vy; };
    auto request = [&](int bb) -> Taps {
      const int b = mbase + bb;
      const char* zb = reinterpret_cast<const char*>(zframe) + (size_t)b * npix * 16 * sizeof(float);      // wave-uniform
      Taps t;
      t.vx = ldg4(wx + b * 16, g4 * 4u);
      t.a = THA4_HOOK_ZLOAD(zb + o00, t.vx);
      t.bq = THA4_HOOK_ZLOAD(zb + o01, t.vx);
      t.c = THA4_HOOK_ZLOAD(zb + o10, t.vx);
      t.d = THA4_HOOK_ZLOAD(zb + o11, t.vx);
      t.vy = ldg4(wy + b * 16, g4 * 4u);
      return t;
    };
    auto consume = [&](int bb, const Taps& t) {
      const int b = mbase + bb;
      const f32x4 vb = *reinterpret_cast<const f32x4*>(pb + b * 16 + g4);
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // six FMAs per value: position + pose bias, then the four taps with their products of the axis weights (the separable
        // form ly0 (lx0 a + lx1 b) + ly1 (lx0 c + lx1 d) + position term took nine VALU instructions)
        const float u = fmaf(t.vx[j], x[pg], fmaf(t.vy[j], y[pg], vb[j]));                            // z and tables carry the 30x
        v[j] = sin_u(fmaf(w00, t.a[j], fmaf(w01, t.bq[j], fmaf(w10, t.c[j], fmaf(w11, t.d[j], u)))));
      }
      sink(pg, b, v);
    };
    if (THA4_TAP_PIPELINE) {
      Taps cur = request(0);
#pragma unroll
      for (int bb = 0; bb < NBW; ++bb) {
        Taps nxt = cur;
        if (bb + 1 < NBW) nxt = request(bb + 1);
        THA4_SCHED_FENCE();
        consume(bb, cur);
        THA4_SCHED_FENCE();
        cur = nxt;
      }
    } else {
#pragma unroll
      for (int bb = 0; bb < NBW; ++bb) {
        const Taps t = request(bb);
        consume(bb, t);
        if ((bb & (THA4_TAP_BLOCKS - 1)) == THA4_TAP_BLOCKS - 1) THA4_SCHED_FENCE();      // at most THA4_TAP_BLOCKS blocks (4 tap loads each) in flight: 12 blocks at once spill
      }
    }
  }
}

template <class G, int NB>
THA4_DEV void first16_up(const float* zframe, int lowS, const float* wx, const float* wy, const float* pb,
                         const int (&X0)[G::PG], const int (&Y)[G::PG], const float (&x)[G::PG], const float (&y)[G::PG],
                         char* act, const WaveCtx& w) {
  first16_up_to<G, NB>(zframe, lowS, wx, wy, pb, X0, Y, x, y,
                       [&](int pg, int b, const f32x4& v) { store_block<G>(act, pg, b, w.lane, v); }, w);
}

// K groups / block counts of the four networks in this generation (channels padded to 32 for K)
constexpr int kKGF = 4;    // face 128
constexpr int kKG0 = 12;   // 360 -> 384
constexpr int kKG1 = 6;    // 180 -> 192
constexpr int kKG2 = 3;    // 90 -> 96

// ---- face ------------------------------------------------------------------------------------------
template <int NS, int MS, int PG, int CQ>
struct Face16Cfg {
  static constexpr int kHB = 1;
  static constexpr int kSlotPieces = CQ * kNBF > kKGF ? CQ * kNBF : kKGF;
  using G = Geo16<NS, MS, PG, kKGF, kSlotPieces>;
};

template <int NS, int MS, int PG, int CQ>
THA4_DEV void face16_body(const StudentDev& d, char* smem, const WaveCtx& w) {
  using G = typename Face16Cfg<NS, MS, PG, CQ>::G;
  constexpr int S = kFaceSize, NPIX = S * S;
  char* ring = smem;
  char* act = smem + 2 * G::SLOT + w.ns * G::ACT_BYTES;
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos128, pix0, X0, Y, px, py);
  const char* gw = reinterpret_cast<const char*>(d.w_face);
  const float* bias = d.b_face;
  const float* scl = d.s_face;
  int slot = 0;
  fetch2k<CQ * kNBF, G::WAVES>(gw, ring, w.wave, w.lane);
  float* pb = reinterpret_cast<float*>(smem + G::SLOT);     // ring slot 1 is idle until the first streamed layer prefetches into it
  pose_bias_to_lds<kNBF, G::THREADS>(d, 0, n, pb);
  if (THA4_PB_FOLD) __syncthreads();
  first16_pos<G, kNBF>(d.wx[0], d.wy[0], THA4_PB_FOLD ? pb : d.pbias + (size_t)n * kPbStride + kPbFace, px, py, act, w);
  __syncthreads();
#pragma unroll 1
  for (int l = 0; l < 6; ++l) sine16_layer<G, kNBF, kKGF, 1, CQ, CQ * kNBF>(gw, bias, scl, ring, slot, act, w);
  sine16_layer<G, kNBF, kKGF, 1, CQ, kKGF>(gw, bias, scl, ring, slot, act, w);
  f32x4 a1[1][PG];
  zero_acc<1, PG>(a1);
  // head: ONE block; only row-split 0 computes, its piece is the first of every group (MS folded: NB = 1)
  gemm16_stream<Geo16<NS, 1, PG, kKGF, Face16Cfg<NS, MS, PG, CQ>::kSlotPieces>, 1, 1, kKGF, 1, kKGF, 0>(gw, ring, slot, act, a1,
                                                                                                   WaveCtx{w.lane, w.wave, w.ns, 0, w.blk, w.nblk}, w.ms == 0);
  if (w.ms == 0 && w.lane < 16) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias);
    const float inv = *scl;
    float* fo = d.face + (size_t)n * 4 * NPIX;
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      const f32x4 v = a1[0][pg] * inv + bb;
#pragma unroll
      for (int j = 0; j < 4; ++j) fo[(size_t)j * NPIX + pix0[pg] + w.lane] = v[j];
    }
  }
}

template <int NS, int MS, int PG, int CQ>
__global__ void __launch_bounds__(NS* MS * 64) face16_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  THA4_DYN_LDS(smem);
  face16_body<NS, MS, PG, CQ>(d, smem, wave_ctx<typename Face16Cfg<NS, MS, PG, CQ>::G>());
}

// ---- level 0 ----------------------------------------------------------------------------------------
template <int NS, int MS, int PG, int HBA, int CQB>
struct Level016Cfg {   // HBA: block slices per group of the 24/12-block layers; CQB: groups per chunk of the z layer
  static constexpr int kHBB = (kNB1 / MS) % HBA == 0 ? HBA : HBA / 2;     // slices of the 12-block layer (a wave's share must divide)
  static constexpr int kP1 = kNB0 / HBA, kP2 = kNB1 / kHBB, kP3 = CQB * kNB1;
  static constexpr int kSlotPieces = (kP1 > kP2 ? kP1 : kP2) > kP3 ? (kP1 > kP2 ? kP1 : kP2) : kP3;
  using G = Geo16<NS, MS, PG, kKG0, kSlotPieces>;
};

template <int NS, int MS, int PG, int HBA, int CQB>
THA4_DEV void level0_16_body(const StudentDev& d, char* smem, const WaveCtx& w) {
  using Cfg = Level016Cfg<NS, MS, PG, HBA, CQB>;
  using G = typename Cfg::G;
  constexpr int S = 128, NPIX = S * S;
  char* ring = smem;
  char* act = smem + 2 * G::SLOT + w.ns * G::ACT_BYTES;
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos128, pix0, X0, Y, px, py);
  const char* gw = reinterpret_cast<const char*>(d.w_l0);
  const float* bias = d.b_l0;
  const float* scl = d.s_l0;
  int slot = 0;
  fetch2k<Cfg::kP1, G::WAVES>(gw, ring, w.wave, w.lane);
  float* pb = reinterpret_cast<float*>(smem + G::SLOT);     // ring slot 1 is idle until the first streamed layer prefetches into it
  pose_bias_to_lds<kNB0, G::THREADS>(d, 1, n, pb);
  if (THA4_PB_FOLD) __syncthreads();
  first16_pos<G, kNB0>(d.wx[1], d.wy[1], THA4_PB_FOLD ? pb : d.pbias + (size_t)n * kPbStride + kPbL0, px, py, act, w);
  __syncthreads();
  sine16_layer<G, kNB0, kKG0, HBA, 1, Cfg::kP2>(gw, bias, scl, ring, slot, act, w);
  sine16_layer<G, kNB1, kKG0, Cfg::kHBB, 1, Cfg::kP3>(gw, bias, scl, ring, slot, act, w);
  z16_layer<G, kNB1, kKG1, 1, CQB>(gw, scl, ring, slot, act, d.z1 + (size_t)n * kNB1 * NPIX * 16, NPIX, pix0, w);
}

template <int NS, int MS, int PG, int HBA, int CQB>
__global__ void __launch_bounds__(NS* MS * 64) level0_16_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  THA4_DYN_LDS(smem);
  level0_16_body<NS, MS, PG, HBA, CQB>(d, smem, wave_ctx<typename Level016Cfg<NS, MS, PG, HBA, CQB>::G>());
}

// Face morpher and body level 0 in ONE launch: the face output is only needed by level 2's warp, so the two kernels are
// independent.  Level-0 workgroups come first (they are the longer ones); as each of them retires its CU picks up a
// face workgroup, instead of the whole chip draining at a kernel boundary first.  `d.front_l0_blocks` = level-0
// workgroups of this launch; both bodies use the same 512-thread geometry.
template <int FNS, int FMS, int FPG, int FCQ, int NS, int MS, int PG, int HBA, int CQB>
__global__ void __launch_bounds__(NS* MS * 64) front16_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  static_assert(FNS * FMS == NS * MS, "face and level 0 must use the same workgroup size");
  THA4_DYN_LDS(smem);
  const int nl0 = d.front_l0_blocks;
  if ((int)blockIdx.x < nl0) {
    WaveCtx w = wave_ctx<typename Level016Cfg<NS, MS, PG, HBA, CQB>::G>();
    w.nblk = nl0;
    level0_16_body<NS, MS, PG, HBA, CQB>(d, smem, w);
  } else {
    WaveCtx w = wave_ctx<typename Face16Cfg<FNS, FMS, FPG, FCQ>::G>();
    w.blk = blockIdx.x - nl0;
    w.nblk = gridDim.x - nl0;
    face16_body<FNS, FMS, FPG, FCQ>(d, smem, w);
  }
}

// ---- level 1 ----------------------------------------------------------------------------------------
template <int NS, int MS, int PG, int CQA, int CQB, int HBA = 1>   // HBA: block slices per group of the two sine layers (needs CQA == 1)
struct Level116Cfg {
  static constexpr int kHBA = HBA, kHBB = HBA > 1 ? HBA / 2 : 1;     // the 6-block layer is sliced half as often: equal chunk sizes
  static constexpr int kP1 = CQA * kNB1 / kHBA, kP2 = CQA * kNB2 / kHBB, kP3 = CQB * kNB2;
  static constexpr int kSlotPieces = (kP1 > kP2 ? kP1 : kP2) > kP3 ? (kP1 > kP2 ? kP1 : kP2) : kP3;
  using G = Geo16<NS, MS, PG, kKG1, kSlotPieces>;
};

template <int NS, int MS, int PG, int CQA, int CQB, int HBA = 1>
__global__ void __launch_bounds__(NS* MS * 64) level1_16_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  using Cfg = Level116Cfg<NS, MS, PG, CQA, CQB, HBA>;
  using G = typename Cfg::G;
  constexpr int S = 256, NPIX = S * S;
  THA4_DYN_LDS(smem);
  const WaveCtx w = wave_ctx<G>();
  char* ring = smem;
  char* act = smem + 2 * G::SLOT + w.ns * G::ACT_BYTES;
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos256, pix0, X0, Y, px, py);
  const char* gw = reinterpret_cast<const char*>(d.w_l1);
  const float* bias = d.b_l1;
  const float* scl = d.s_l1;
  int slot = 0;
  fetch2k<Cfg::kP1, G::WAVES>(gw, ring, w.wave, w.lane);
  float* pb = reinterpret_cast<float*>(smem + G::SLOT);     // ring slot 1 is idle until the first streamed layer prefetches into it
  pose_bias_to_lds<kNB1, G::THREADS>(d, 2, n, pb);
  if (THA4_PB_FOLD) __syncthreads();
  first16_up<G, kNB1>(d.z1 + (size_t)n * kNB1 * (128 * 128) * 16, 128, d.wx[2], d.wy[2], THA4_PB_FOLD ? pb : d.pbias + (size_t)n * kPbStride + kPbL1, X0, Y, px, py, act, w);
  __syncthreads();
  sine16_layer<G, kNB1, kKG1, HBA, CQA, Cfg::kP2>(gw, bias, scl, ring, slot, act, w);
  sine16_layer<G, kNB2, kKG1, Cfg::kHBB, CQA, Cfg::kP3>(gw, bias, scl, ring, slot, act, w);
  z16_layer<G, kNB2, kKG2, 1, CQB>(gw, scl, ring, slot, act, d.z2 + (size_t)n * kNB2 * NPIX * 16, NPIX, pix0, w);
}

// last_linear rows -> grid_sample warp of (image with the face patch) -> alpha blend -> NCHW outputs
// (siren_morpher_03.py:125-131, image_processing_util.py:33-54, mode_14.py:72-78).  Rows 0..3 (dx, dy, alpha,
// colour R) live in lane group 0, rows 4..6 (G, B, A) in group 1; lane group g then handles image channel g.
template <int PG>
THA4_DEV void warp_blend_store(const StudentDev& d, int n, const float* head_bias, float inv, const int (&pix0)[PG], const float (&px)[PG],
                               const float (&py)[PG], f32x4 (&a1)[1][PG], const WaveCtx& w) {
  constexpr int S = kImg, NPIX = S * S;
  const int p = w.lane & 15, g = w.lane >> 4;
  const f32x4 bb = *reinterpret_cast<const f32x4*>(head_bias + g * 4);
  const float* img = d.image + (size_t)n * d.image_stride;
  const float* face = d.face + (size_t)n * 4 * kFaceSize * kFaceSize;
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    const f32x4 v = a1[0][pg] * inv + bb;
    const float dx = lane_read(v[0], p), dy = lane_read(v[1], p), al = lane_read(v[2], p);
    const float c0 = lane_read(v[3], p), c1 = lane_read(v[0], p + 16), c2 = lane_read(v[1], p + 16),
                c3 = lane_read(v[2], p + 16);
    const float col = g == 0 ? c0 : (g == 1 ? c1 : (g == 2 ? c2 : c3));
    const float gx = px[pg] + dx, gy = py[pg] + dy;
    float ix = ((gx + 1.0f) * (float)S - 1.0f) * 0.5f;
    float iy = ((gy + 1.0f) * (float)S - 1.0f) * 0.5f;
    ix = fminf((float)(S - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(S - 1), fmaxf(iy, 0.0f));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float tx = ix - fx0, ty = iy - fy0;
    const float wnw = (1.0f - tx) * (1.0f - ty), wne = tx * (1.0f - ty), wsw = (1.0f - tx) * ty, wse = tx * ty;
    const int x1 = min(x0 + 1, S - 1), y1 = min(y0 + 1, S - 1);
    float wv = body_source(img, face, g, y0, x0) * wnw;
    wv += body_source(img, face, g, y0, x1) * wne;
    wv += body_source(img, face, g, y1, x0) * wsw;
    wv += body_source(img, face, g, y1, x1) * wse;
    const float blended = (1.0f - al) * wv + al * col;
    const size_t pix = (size_t)pix0[pg] + p;
    THA4_HOOK_BEFORE_STORES();
    if (d.out_rgba8) store_display(d, n, pix, g, p, blended);      // (first: it LOADS the background colour, and a load behind a store waits for the store's acknowledgement)
    if (d.out_blended) d.out_blended[((size_t)n * 4 + g) * NPIX + pix] = blended;      // (write-through stores of the posed frame: measured neutral, tools/runs_r06/gpu_r06_c23.sh)
    if (d.out_color) d.out_color[((size_t)n * 4 + g) * NPIX + pix] = col;
    if (d.out_warped) d.out_warped[((size_t)n * 4 + g) * NPIX + pix] = wv;
    if (d.out_alpha && g == 0) d.out_alpha[(size_t)n * NPIX + pix] = al;
    if (d.out_grid && g < 2) d.out_grid[((size_t)n * 2 + g) * NPIX + pix] = (g == 0 ? dx : dy);
  }
}

// ---- level 2 ----------------------------------------------------------------------------------------
template <int NS, int MS, int PG, int CQ>
struct Level216Cfg {
  static constexpr int kSlotPieces = CQ * kNB2 > kKG2 ? CQ * kNB2 : kKG2;
  using G = Geo16<NS, MS, PG, kKG2, kSlotPieces>;
};

template <int NS, int MS, int PG, int CQ>
__global__ void __launch_bounds__(NS* MS * 64) level2_16_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  using G = typename Level216Cfg<NS, MS, PG, CQ>::G;
  static_assert(MS == 1, "level 2 keeps whole rows per wave");
  constexpr int S = kImg, NPIX = S * S;
  THA4_DYN_LDS(smem);
  const WaveCtx w = wave_ctx<G>();
  char* ring = smem;
  char* act = smem + 2 * G::SLOT + w.ns * G::ACT_BYTES;
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos512, pix0, X0, Y, px, py);
  const char* gw = reinterpret_cast<const char*>(d.w_l2);
  const float* bias = d.b_l2;
  const float* scl = d.s_l2;
  int slot = 0;
  fetch2k<CQ * kNB2, G::WAVES>(gw, ring, w.wave, w.lane);
  float* pb = reinterpret_cast<float*>(smem + G::SLOT);     // ring slot 1 is idle until the first streamed layer prefetches into it
  pose_bias_to_lds<kNB2, G::THREADS>(d, 3, n, pb);
  if (THA4_PB_FOLD) __syncthreads();
  first16_up<G, kNB2>(d.z2 + (size_t)n * kNB2 * (256 * 256) * 16, 256, d.wx[3], d.wy[3], THA4_PB_FOLD ? pb : d.pbias + (size_t)n * kPbStride + kPbL2, X0, Y, px, py, act, w);
  __syncthreads();
  sine16_layer<G, kNB2, kKG2, 1, CQ, CQ * kNB2>(gw, bias, scl, ring, slot, act, w);
  sine16_layer<G, kNB2, kKG2, 1, CQ, kKG2>(gw, bias, scl, ring, slot, act, w);
  f32x4 a1[1][PG];
  zero_acc<1, PG>(a1);
  gemm16_stream<G, 1, 1, kKG2, 1, kKG2, 0>(gw, ring, slot, act, a1, w, true);

  warp_blend_store<PG>(d, n, bias, *scl, pix0, px, py, a1, w);
}

// ---- level 2, weights-resident variant -------------------------------------------------------------
// The three level-2 layers are only 78 KiB of fp16 hi/lo pieces: instead of re-streaming them through a ring for
// every 64-pixel workgroup (4096 workgroups, 320 MB of L2->LDS traffic and a barrier per chunk per frame), a
// workgroup loads them ONCE and its WAVES waves then walk PGW strips of PG pixel groups each with NO further
// barrier.  Because a wave owns whole rows here, the k-slot permutation (see "Images" above) makes the C/D rows a
// lane holds exactly its B fragment of the next layer: the activations never leave the wave's registers, LDS
// carries weights only, and one A fragment read feeds PG pixel groups.
template <int KG, int PG>
THA4_DEV void put_rows(f16x8 (&xh)[KG][PG], f16x8 (&xl)[KG][PG], int pg, int b, const f32x4& v) {
  f16x4 hi, lo;
  split4(v, hi, lo);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    xh[b >> 1][pg][(b & 1) * 4 + j] = hi[j];
    xl[b >> 1][pg][(b & 1) * 4 + j] = lo[j];
  }
}

// The optimiser SINKS pure per-value work (tap FMAs, sine, split) down to its first use: in the fully unrolled register-resident kernels that is
// a later chunk's first MFMA of that K group - every tap load's result (or every accumulator of the previous layer) then stays live across the
// chunks in between (level1_16r_kernel: 256 VGPRs + 158 AGPRs instead of ~180).  An empty volatile asm that "modifies" the finished rows pins the
// layer's epilogue where it is written.
template <int KG, int PG>
THA4_DEV void pin_rows(f16x8 (&xh)[KG][PG], f16x8 (&xl)[KG][PG]) {
#ifndef THA4_EMU
#pragma unroll
  for (int q = 0; q < KG; ++q)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) asm volatile("" : "+v"(xh[q][pg]), "+v"(xl[q][pg]));
#endif
}

// (Round 4, measured negative: the layer biases / scales of the weights-resident level 2 copied to LDS once per workgroup instead of three dependent
// global round trips per strip - the kernel alone 45.1 -> 44.6 us, the batch-1 STREAM 7640 -> 7535 frames/s on the same box, profiles/r04_raw/c31_student_bias_lds.txt.)
#ifndef THA4_L2_GROUP_BLOCKS
#define THA4_L2_GROUP_BLOCKS 3    // A fragments of this many blocks (hi + lo) are double-buffered in registers: 3 -> 48 VGPRs, 2 -> 32
#endif
// acc += W' x for one resident layer: pieces [Q][NB] at wv (lane offset applied), x in registers
template <int NB, int KG, int PG>
THA4_DEV void mma_resident(const char* wv, const f16x8 (&xh)[KG][PG], const f16x8 (&xl)[KG][PG], f32x4 (&acc)[NB][PG]) {
  constexpr int GB = (NB == kNB2) ? THA4_L2_GROUP_BLOCKS : group_blocks(NB), NGB = NB / GB, T = KG * NGB;
  static_assert(NB % GB == 0, "the block group must divide the layer");
  f16x8 ah[2][GB], al[2][GB];
#pragma unroll
  for (int b = 0; b < GB; ++b) {
    ah[0][b] = *reinterpret_cast<const f16x8*>(wv + (size_t)b * 2048);
    al[0][b] = *reinterpret_cast<const f16x8*>(wv + (size_t)b * 2048 + 1024);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int qq = t / NGB, bo = (t % NGB) * GB;
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(ah[t & 1][b], xh[qq][pg], acc[bo + b][pg]);
    THA4_SCHED_FENCE();
    if (t + 1 < T) {
#pragma unroll
      for (int b = 0; b < GB; ++b) {
        const char* pc = wv + ((size_t)((t + 1) / NGB) * NB + ((t + 1) % NGB) * GB + b) * 2048;
        ah[(t + 1) & 1][b] = *reinterpret_cast<const f16x8*>(pc);
        al[(t + 1) & 1][b] = *reinterpret_cast<const f16x8*>(pc + 1024);
      }
    }
    THA4_SCHED_FENCE();
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(ah[t & 1][b], xl[qq][pg], acc[bo + b][pg]);
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(al[t & 1][b], xh[qq][pg], acc[bo + b][pg]);
    THA4_SCHED_FENCE();
  }
}

constexpr bool defined_hw_sin() {
#ifdef THA4_HW_SIN
  return true;
#else
  return false;
#endif
}

// WAVES waves share SPW strips of PG pixel groups each; the strips are handed out by an LDS ticket, so WAVES need not divide SPW
// (12 waves = 3 per SIMD take 64 strips as well as 8 waves do)
template <int WAVES, int SPW, int PG>
struct Level2PCfg {
  static constexpr int kHidden = kNB2 * kKG2;                         // pieces of one 96->96 layer
  static constexpr int kPieces = 2 * kHidden + kKG2;                  // + head (1 block x 3 groups)
  static constexpr int kWeightBytes = kPieces * 2048;
  static constexpr int LDS = kWeightBytes + 16;                       // + the strip ticket counter (+ pb_lds_bytes(kNB2) at launch)
  static constexpr int THREADS = WAVES * 64;
  static constexpr int PX = SPW * PG * 16;
  using G = Geo16<WAVES, 1, PG, kKG2, 1>;
  static_assert((kImg * kImg) % PX == 0, "a frame must be a whole number of workgroups");
#if !defined(THA4_ALLOW_L216P_PG2) && !defined(THA4_EMU) && !defined(THA4_NO_PACKED_FP32)
  static_assert(PG == 1 || !(THA4_SIN_TURNS || defined_hw_sin()),
                "level2_16p_kernel with two pixel groups per strip is faulty on the device when the compiler may use packed-fp32 instructions "
                "and the sine is a v_sin_f32 (see THA4_L216P_CFG): build with tha4_amd._build.DEVICE_FLAGS (-packed-fp32-ops off, "
                "-DTHA4_NO_PACKED_FP32=1); -DTHA4_ALLOW_L216P_PG2 builds it anyway (fault-hunt builds only)");
#endif
  static_assert(LDS + 16 * kNB2 * 4 <= 80 * 1024, "two workgroups per CU must still fit");
};

template <int WAVES, int SPW, int PG>
__global__ void __launch_bounds__(WAVES * 64) level2_16p_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  using Cfg = Level2PCfg<WAVES, SPW, PG>;
  static_assert(SPW >= WAVES, "every wave takes its first strip statically");
  using G = typename Cfg::G;
  constexpr int S = kImg, STRIPS = S * S / (16 * PG);
  THA4_DYN_LDS(smem);
  const WaveCtx w = wave_ctx<G>();
  THA4_SPAN(d, 3, 0);
  fetch_pieces<2 * Cfg::kPieces, WAVES>(reinterpret_cast<const char*>(d.w_l2), smem, w.wave, w.lane);
  // Strips are handed out dynamically: a wave that hits slow image gathers in its warp epilogue does not hold the
  // workgroup back (static assignment: the slowest wave finished ~22 k cycles after the mean).  Which wave computes a
  // strip does not change its bytes.
  int* ticket = reinterpret_cast<int*>(smem + Cfg::kWeightBytes);
  if (threadIdx.x == 0) *ticket = WAVES;                               // strips 0..WAVES-1 are taken statically
  // a workgroup's strips all belong to one frame (STRIPS is a multiple of PGW * WAVES): its pose-folded bias goes to LDS
  float* pb = reinterpret_cast<float*>(smem + Cfg::LDS);
  pose_bias_to_lds<kNB2, WAVES * 64>(d, 3, (xcd_tile(blockIdx.x, gridDim.x) * SPW) / STRIPS, pb);
  // (Round 6, measured negative: publishing the 78 KiB of weight copies BEHIND the first strip's taps - an LDS-only barrier here, vmcnt(0) + s_barrier in
  //  front of the first strip's first matrix layer - so that the copies' round trip hides under the taps' (ablation: the launch is 4 us shorter without the
  //  copies).  With an LDS-DMA pending at the loop's entry the compiler's wait insertion puts s_waitcnt vmcnt(0) in front of every LDS read of the strip loop
  //  - the pose-bias reads between the tap requests among them - and the two-block tap pipeline collapses: 45 -> 109 us, tools/runs_r06/gpu_r06_c20.sh.)
  __syncthreads();
  const char* w1 = smem + w.lane * 16;
  const char* w2 = w1 + (size_t)Cfg::kHidden * 2048;
  const char* w3 = w2 + (size_t)Cfg::kHidden * 2048;
  const int g4 = (w.lane >> 4) * 4;
  const int strip0 = xcd_tile(blockIdx.x, gridDim.x) * SPW;
#pragma unroll 1
  for (int k = w.wave; k < SPW; k = wave_take_ticket(ticket, w.lane)) {
    THA4_HOOK_STRIP_TOP();
    const int strip = strip0 + k;
    const int n = strip / STRIPS;
    int pix0[PG], X0[PG], Y[PG];
    float px[PG], py[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      pix0[pg] = ((strip % STRIPS) * PG + pg) * 16;
      X0[pg] = pix0[pg] % S;
      Y[pg] = pix0[pg] / S;
      px[pg] = d.pos512[X0[pg] + (w.lane & 15)];
      py[pg] = d.pos512[Y[pg]];
    }
    f16x8 xh[kKG2][PG], xl[kKG2][PG];
    THA4_PRIO_VALU();
    first16_up_to<G, kNB2>(d.z2 + (size_t)n * kNB2 * (256 * 256) * 16, 256, d.wx[3], d.wy[3], THA4_PB_FOLD ? pb : d.pbias + (size_t)n * kPbStride + kPbL2, X0, Y, px, py,
                           [&](int pg, int b, const f32x4& v) { put_rows<kKG2, PG>(xh, xl, pg, b, v); }, w);
    const float* bias = d.b_l2;
#pragma unroll
    for (int layer = 0; layer < 2; ++layer) {
      f32x4 acc[kNB2][PG];
      zero_acc<kNB2, PG>(acc);
      THA4_PRIO_MFMA();
      mma_resident<kNB2, kKG2, PG>(layer == 0 ? w1 : w2, xh, xl, acc);
      THA4_PRIO_VALU();
      const float inv = d.s_l2[layer];
#pragma unroll
      for (int b = 0; b < kNB2; ++b) {
        const f32x4 bb = ldg4(bias + b * 16, g4 * 4u);
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = sin_u(fmaf(acc[b][pg][j], inv, bb[j]));
          put_rows<kKG2, PG>(xh, xl, pg, b, v);
        }
      }
      bias += kNB2 * 16;
    }
    f32x4 a1[1][PG];
    zero_acc<1, PG>(a1);
    THA4_PRIO_MFMA();
    mma_resident<1, kKG2, PG>(w3, xh, xl, a1);
    THA4_PRIO_VALU();
    warp_blend_store<PG>(d, n, bias, d.s_l2[2], pix0, px, py, a1, w);
  }
}


// ---- level 1, activations-in-registers variant (round 6) ------------------------------------------------
// level1_16_kernel re-streams the level's 252 KiB of weight pieces for every 64-pixel workgroup (1024 workgroups: 258 MB of L2 -> LDS
// traffic per frame) through a TWO-slot ring: one 12-KiB chunk in flight per workgroup, a barrier per chunk, and a round trip of every
// activation through LDS per layer (store_block + a B-fragment read per chunk).  Here a wave owns whole rows (MS = 1) of PG pixel
// groups, exactly like the weights-resident level 2: the k-slot permutation makes the C/D rows a lane holds its B fragment of the next
// layer, so the activations never leave the wave's registers (192-channel layers: 48 accumulator + 48 operand VGPRs per pixel group),
// LDS carries NOTHING but the weight ring - SLOTS x 24 KiB, SLOTS - 1 chunks in flight behind counted `s_waitcnt vmcnt` - and one
// pass over the stream serves WAVES x PG x 16 pixels (8 x 2 x 16 = 256: one image row per workgroup, 256 workgroups per frame, 65 MB
// of stream).  The stream is eleven 24-KiB chunks: six K groups x 12 blocks of the 180 -> 180 layer, three pairs of K groups x 6
// blocks of the 180 -> 90 layer, and the z layer's three K groups x 6 blocks as {0, 1}, {2, padding}.
// Biases and the pose-folded first-layer bias sit in LDS: no global load is issued between the first chunk's barrier and the z stores,
// so the counted waits see ring copies only.
// geometry of the kernels whose activations live in registers: a wave = a pixel slot that owns whole rows; no activation image in LDS
template <int WAVES_, int PG_>
struct GeoRegs {
  static constexpr int NS = WAVES_, MS = 1, PG = PG_, ACTQ = 0;
  static constexpr int WAVES = WAVES_, THREADS = WAVES_ * 64, PX = WAVES_ * PG_ * 16;
};

template <int WAVES, int PG, int SLOTS>
struct Level1RCfg {
  static constexpr int kChunkPieces = 12, kChunk = kChunkPieces * 2048;
  static constexpr int kChunksA = kKG1, kChunksB = kKG1 / 2, kChunksZ = (kKG2 + 1) / 2;
  static constexpr int kChunks = kChunksA + kChunksB + kChunksZ;                      // 11
  static constexpr int kStreamPieces = kChunks * kChunkPieces;                        // 132 (126 + 6 of padding)
  static constexpr int kCPW = 2 * kChunkPieces / WAVES;                               // 1-KiB copies per wave and chunk
  static constexpr int kDepth = SLOTS - 1;                                            // chunks in flight
  static constexpr int kPre = 2;                                                      // chunks requested before the first layer (the rest behind its tap loads)
  static constexpr int kRing = SLOTS * kChunk;
  static constexpr int kPbOff = kRing, kBiasOff = kPbOff + 3 * kNB1 * 16 * 4;         // floats: pb[192] | wx[192] | wy[192] | bias A[192] | bias B[96] | 1/S of the three layers (+ 1 pad)
  static constexpr int kBiasFloats = (kNB1 + kNB2) * 16;
  static constexpr int LDS = kBiasOff + (kBiasFloats + 4) * 4;
  static constexpr int THREADS = WAVES * 64, PX = WAVES * PG * 16;
  using G = GeoRegs<WAVES, PG>;
  static_assert((2 * kChunkPieces) % WAVES == 0, "every wave must issue the same number of copies per chunk (counted waits)");
  static_assert(kDepth >= kPre && kDepth * kCPW <= 16, "THA4_BARRIER_KEEP counts up to 16 copies");
  static_assert(LDS <= 160 * 1024, "LDS budget exceeded");
  static_assert((256 * 256) % PX == 0, "a frame must be a whole number of workgroups");
};

// acc[b][pg] += W' x for one landed chunk: pieces [qq < CQ][NBC blocks] at wv (lane offset applied), x K groups Q0 .. Q0 + CQ - 1 in registers
#ifndef THA4_REGS_GROUP_BLOCKS
#define THA4_REGS_GROUP_BLOCKS 0     // blocks whose A fragments (hi + lo) are double-buffered in registers; 0: 2 for two pixel groups / wide layers, 3 otherwise
#endif
// (Q0, B0: first K group of x / first accumulator block of the chunk - constants once the chunk sequence is unrolled)
// The A fragments (hi + lo of GB blocks per step) go through a ring of THA4_REGS_PREFETCH + 1 register buffers: the reads of step t + PREFETCH are
// issued at the top of step t.  With one step of look-ahead issued behind the step's first MFMAs (the form of mma_resident) a wave that is alone
// with its chain - one or two waves per SIMD instead of level 2's four - covers 4 MFMAs (~70 cycles) of a ~200-cycle LDS round trip under load and
// stalls at the top of EVERY step: front16r_kernel lost 2.4 us of 38 with its MFMAs removed (tools/runs_r06/gpu_r06_c3.sh).
#ifndef THA4_REGS_PREFETCH
#define THA4_REGS_PREFETCH 2
#endif
template <int NBC, int CQ, int KG, int PG, int NBT>
THA4_DEV void mma_chunk_regs(const char* wv, const f16x8 (&xh)[KG][PG], const f16x8 (&xl)[KG][PG], f32x4 (&acc)[NBT][PG], const int Q0, const int B0) {
  constexpr int GB = NBC == 1 ? 1 : THA4_REGS_GROUP_BLOCKS ? THA4_REGS_GROUP_BLOCKS : ((PG >= 2 || KG * PG > 8) ? 2 : (NBC % 3 == 0 ? 3 : 2)), NGB = NBC / GB, T = CQ * NGB;
  constexpr int D = THA4_REGS_PREFETCH, NBUF = D + 1;
  static_assert(NBC % GB == 0 && GB <= 3, "the block group must divide the chunk's blocks");
  static_assert(2 * GB * D <= 15, "the reads in flight must fit the 4-bit lgkmcnt");
  f16x8 ah[NBUF][GB], al[NBUF][GB];
  // The fragment reads are inline asm and their waits hand-counted (lds_read16 / lds_wait, tha4_platform.h): with the ring's LDS-DMA in flight the
  // compiler's own waits drain every outstanding read at each step.
  auto load = [&](auto tc) {
    constexpr int t = decltype(tc)::value;
    static_for<0, GB>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr int off = ((t / NGB) * NBC + (t % NGB) * GB + b) * 2048;
      lds_read16<off>(ah[t % NBUF][b], wv);
      lds_read16<off + 1024>(al[t % NBUF][b], wv);
    });
  };
  static_for<0, (D < T ? D : T)>(load);
  static_for<0, T>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr int qq = t / NGB, buf = t % NBUF;
    const int bo = B0 + (t % NGB) * GB;
    if constexpr (t + D < T) load(std::integral_constant<int, t + D>());
    constexpr int newer = ((t + D < T ? t + D : T - 1) - t) * 2 * GB;           // reads of the steps behind this one that may stay in flight
    if constexpr (GB == 1) lds_wait<newer>(ah[buf][0], al[buf][0]);
    else if constexpr (GB == 2) lds_wait<newer>(ah[buf][0], al[buf][0], ah[buf][1], al[buf][1]);
    else lds_wait<newer>(ah[buf][0], al[buf][0], ah[buf][1], al[buf][1], ah[buf][2], al[buf][2]);
    THA4_SCHED_FENCE();
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(ah[buf][b], xh[Q0 + qq][pg], acc[bo + b][pg]);
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(ah[buf][b], xl[Q0 + qq][pg], acc[bo + b][pg]);
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(al[buf][b], xh[Q0 + qq][pg], acc[bo + b][pg]);
    THA4_SCHED_FENCE();
  });
}

// First layer of a register-resident level (level1_16r_kernel): the x2-upsample taps of ALL pixel groups of the wave as one sequence of
// batches of TB blocks (4 tap loads each), TWO batches in flight.  first16_up_to keeps two BLOCKS in flight - right for 16 waves per CU, but a
// wave that owns whole rows of two pixel groups walks 24 dependent round trips to the z image that way (level 1: ~15 us of a 33-us launch).
// wx, wy, pb: LDS copies.
#ifndef THA4_TAP_BATCH
#define THA4_TAP_BATCH 4
#endif
// `between()` (round 6): called once BEHIND the first two batches' requests and in front of the first consume - the caller's prologue (table loads -> LDS,
// ring burst, barrier) then runs under the first taps' round trip instead of in front of it; wx / wy / pb need only be valid when it returns.
template <class G, int NB, int TB, class Sink, class Between>
THA4_DEV void first16_up_batched(const float* zframe, int lowS, const float* wx, const float* wy, const float* pb, const int (&X0)[G::PG], const int (&Y)[G::PG],
                                 const float (&x)[G::PG], const float (&y)[G::PG], Sink&& sink, const WaveCtx& w, Between&& between) {
  constexpr int PG = G::PG, KB = NB / TB, STEPS = PG * KB;
  static_assert(NB % TB == 0, "the tap batch must divide the layer's blocks");
  const int p = w.lane & 15, g4 = (w.lane >> 4) * 4;
  const int npix = lowS * lowS;
  unsigned o00[PG], o01[PG], o10[PG], o11[PG];
  float w00[PG], w01[PG], w10[PG], w11[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    int x0, x1, y0, y1;
    float lx0, lx1, ly0, ly1;
    up2_taps(X0[pg] + p, lowS, x0, x1, lx0, lx1);
    up2_taps(Y[pg], lowS, y0, y1, ly0, ly1);
    w00[pg] = ly0 * lx0; w01[pg] = ly0 * lx1; w10[pg] = ly1 * lx0; w11[pg] = ly1 * lx1;
    o00[pg] = (unsigned)(z_offset(0, w.lane >> 4, y0 * lowS + x0, npix) * sizeof(float));
    o01[pg] = (unsigned)(z_offset(0, w.lane >> 4, y0 * lowS + x1, npix) * sizeof(float));
    o10[pg] = (unsigned)(z_offset(0, w.lane >> 4, y1 * lowS + x0, npix) * sizeof(float));
    o11[pg] = (unsigned)(z_offset(0, w.lane >> 4, y1 * lowS + x1, npix) * sizeof(float));
  }
  struct Batch { f32x4 a[TB], bq[TB], c[TB], d[TB]; };
  auto request = [&](int s) -> Batch {
    const int pg = s / KB, b0 = (s % KB) * TB;
    Batch t;
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const char* zb = reinterpret_cast<const char*>(zframe) + (size_t)(b0 + i) * npix * 16 * sizeof(float);      // wave-uniform
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      t.a[i] = THA4_HOOK_ZLOAD(zb + o00[pg], zero);
      t.bq[i] = THA4_HOOK_ZLOAD(zb + o01[pg], zero);
      t.c[i] = THA4_HOOK_ZLOAD(zb + o10[pg], zero);
      t.d[i] = THA4_HOOK_ZLOAD(zb + o11[pg], zero);
    }
    return t;
  };
  auto consume = [&](int s, const Batch& t) {
    const int pg = s / KB, b0 = (s % KB) * TB;
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int b = b0 + i;
      const f32x4 vx = *reinterpret_cast<const f32x4*>(wx + b * 16 + g4);
      const f32x4 vy = *reinterpret_cast<const f32x4*>(wy + b * 16 + g4);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(pb + b * 16 + g4);
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = fmaf(vx[j], x[pg], fmaf(vy[j], y[pg], vb[j]));                            // z and tables carry the sine's scale
        v[j] = sin_u(fmaf(w00[pg], t.a[i][j], fmaf(w01[pg], t.bq[i][j], fmaf(w10[pg], t.c[i][j], fmaf(w11[pg], t.d[i][j], u)))));
      }
      sink(pg, b, v);
    }
  };
  static_assert(STEPS >= 2, "two batches are requested ahead");
  Batch cur = request(0), nxt = request(1);
  THA4_SCHED_FENCE();
  between();
  THA4_SCHED_FENCE();
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    THA4_SCHED_FENCE();
    consume(s, cur);
    THA4_SCHED_FENCE();
    cur = nxt;
    if (s + 2 < STEPS) nxt = request(s + 2);
  }
}

#ifndef THA4_L1_TAPS_FIRST
#define THA4_L1_TAPS_FIRST 0      // 1: the first two tap batches are requested in front of the prologue - measured neutral (35.7 vs 35.6 us, stream -0.4 %: tools/runs_r06/gpu_r06_c24.sh); 0: prologue, then taps
#endif
template <int NB, int THREADS, int NBIAS, int NSCL>
THA4_DEV void prologue_to_lds(const StudentDev& d, int net, int n, const float* bsrc, const float* ssrc, float* pb, float* bias_lds);      // (defined with the front kernel below)
template <int WAVES, int PG, int SLOTS>
__global__ void __launch_bounds__(WAVES * 64) level1_16r_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  using Cfg = Level1RCfg<WAVES, PG, SLOTS>;
  using G = typename Cfg::G;
  constexpr int S = 256, NPIX = S * S, NC = Cfg::kChunks, DEPTH = Cfg::kDepth, CPW = Cfg::kCPW;
  THA4_DYN_LDS(smem);
  const WaveCtx w = wave_ctx<G>();
  const char* gw = reinterpret_cast<const char*>(d.w_l1);
  auto fetch = [&](int c) { fetch_pieces<2 * Cfg::kChunkPieces, WAVES>(gw + (size_t)c * Cfg::kChunk, smem + (c % SLOTS) * Cfg::kChunk, w.wave, w.lane); };
  const bool son = blockIdx.x == 0 && w.wave == 0;
  THA4_SPAN(d, 2, 0);
  THA4_STAMP(d, son, 3, 0);
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos256, pix0, X0, Y, px, py);
  float* pb = reinterpret_cast<float*>(smem + Cfg::kPbOff);
  float* bias_lds = reinterpret_cast<float*>(smem + Cfg::kBiasOff);
  // The prologue - every table load requested before any is consumed (one memory round trip), IN FRONT of the ring's first 48 KiB (behind which the loads would
  // queue), LDS-only barrier (the ring copies stay in flight; chunk 0's own counted barrier publishes them) - runs UNDER the round trip of the first two tap
  // batches (THA4_L1_TAPS_FIRST; stamps of the first form: prologue 5.5 us, then 4.2 us of taps, before the first MFMA of a 27-us workgroup).
  auto prologue = [&]() {
    prologue_to_lds<kNB1, WAVES * 64, Cfg::kBiasFloats, 3>(d, 2, n, d.b_l1, d.s_l1, pb, bias_lds);
#pragma unroll
    for (int c = 0; c < Cfg::kPre; ++c) fetch(c);
    THA4_BARRIER_LDS();
    THA4_STAMP(d, son, 3, 2);
  };
  if (!THA4_L1_TAPS_FIRST) prologue();
  const int g4 = (w.lane >> 4) * 4;
  f16x8 xh[kKG1][PG], xl[kKG1][PG];
  THA4_PRIO_VALU();
  first16_up_batched<G, kNB1, THA4_TAP_BATCH>(d.z1 + (size_t)n * kNB1 * (128 * 128) * 16, 128, pb + kNB1 * 16, pb + 2 * kNB1 * 16, pb, X0, Y, px, py,
                                              [&](int pg, int b, const f32x4& v) { put_rows<kKG1, PG>(xh, xl, pg, b, v); }, w,
                                              [&]() { if (THA4_L1_TAPS_FIRST) prologue(); });
  pin_rows<kKG1, PG>(xh, xl);
  THA4_PRIO_MFMA();
#pragma unroll
  for (int c = Cfg::kPre; c < DEPTH; ++c) fetch(c);            // younger than every tap load: the taps' waits do not cover them
  const char* ring = smem + w.lane * 16;
  // chunk c: it has landed (for this wave: everything but the (chunks requested after it) x CPW newest copies; for the others: the barrier),
  // and every wave is done with chunk c - 1, whose slot chunk c + DEPTH goes into
#define THA4_L1R_CHUNK_TOP(c)                                                                  \
  do {                                                                                         \
    constexpr int newer_ = ((c) + DEPTH - 1 < NC - 1 ? (c) + DEPTH - 1 : NC - 1) - (c);        \
    THA4_BARRIER_KEEP(newer_ * CPW);                                                           \
    if ((c) + DEPTH < NC) fetch((c) + DEPTH);                                                  \
  } while (0)
  const float* scl = bias_lds + Cfg::kBiasFloats;              // (from LDS: a global load here would sit behind the ring copies in the in-order vmcnt queue)
  THA4_STAMP(d, son, 3, 3);
  {   // 180 -> 180, sine
    f32x4 acc[kNB1][PG];
    zero_acc<kNB1, PG>(acc);
    THA4_L1R_CHUNK_TOP(0); mma_chunk_regs<kNB1, 1>(ring + (0 % SLOTS) * Cfg::kChunk, xh, xl, acc, 0, 0);
    THA4_L1R_CHUNK_TOP(1); mma_chunk_regs<kNB1, 1>(ring + (1 % SLOTS) * Cfg::kChunk, xh, xl, acc, 1, 0);
    THA4_L1R_CHUNK_TOP(2); mma_chunk_regs<kNB1, 1>(ring + (2 % SLOTS) * Cfg::kChunk, xh, xl, acc, 2, 0);
    THA4_L1R_CHUNK_TOP(3); mma_chunk_regs<kNB1, 1>(ring + (3 % SLOTS) * Cfg::kChunk, xh, xl, acc, 3, 0);
    THA4_L1R_CHUNK_TOP(4); mma_chunk_regs<kNB1, 1>(ring + (4 % SLOTS) * Cfg::kChunk, xh, xl, acc, 4, 0);
    THA4_L1R_CHUNK_TOP(5); mma_chunk_regs<kNB1, 1>(ring + (5 % SLOTS) * Cfg::kChunk, xh, xl, acc, 5, 0);
    THA4_STAMP(d, son, 3, 4);
    THA4_PRIO_VALU();
    const float inv = scl[0];
#pragma unroll
    for (int b = 0; b < kNB1; ++b) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + b * 16 + g4);
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sin_u(fmaf(acc[b][pg][j], inv, bb[j]));
        put_rows<kKG1, PG>(xh, xl, pg, b, v);
      }
    }
    pin_rows<kKG1, PG>(xh, xl);
    THA4_PRIO_MFMA();
    THA4_STAMP(d, son, 3, 5);
  }
  f16x8 yh[kKG2][PG], yl[kKG2][PG];
  {   // 180 -> 90, sine
    f32x4 acc[kNB2][PG];
    zero_acc<kNB2, PG>(acc);
    THA4_L1R_CHUNK_TOP(6); mma_chunk_regs<kNB2, 2>(ring + (6 % SLOTS) * Cfg::kChunk, xh, xl, acc, 0, 0);
    THA4_L1R_CHUNK_TOP(7); mma_chunk_regs<kNB2, 2>(ring + (7 % SLOTS) * Cfg::kChunk, xh, xl, acc, 2, 0);
    THA4_L1R_CHUNK_TOP(8); mma_chunk_regs<kNB2, 2>(ring + (8 % SLOTS) * Cfg::kChunk, xh, xl, acc, 4, 0);
    THA4_STAMP(d, son, 3, 6);
    THA4_PRIO_VALU();
    const float inv = scl[1];
#pragma unroll
    for (int b = 0; b < kNB2; ++b) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + (kNB1 + b) * 16 + g4);
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sin_u(fmaf(acc[b][pg][j], inv, bb[j]));
        put_rows<kKG2, PG>(yh, yl, pg, b, v);
      }
    }
    pin_rows<kKG2, PG>(yh, yl);
    THA4_PRIO_MFMA();
    THA4_STAMP(d, son, 3, 7);
  }
  {   // z2 = c W20[:, :90] h1 (fp32) -> global z[n][b][g][pix][4]; the consumer is level 2's first (sine) layer
    f32x4 acc[kNB2][PG];
    zero_acc<kNB2, PG>(acc);
    THA4_L1R_CHUNK_TOP(9); mma_chunk_regs<kNB2, 2>(ring + (9 % SLOTS) * Cfg::kChunk, yh, yl, acc, 0, 0);
    THA4_L1R_CHUNK_TOP(10); mma_chunk_regs<kNB2, 1>(ring + (10 % SLOTS) * Cfg::kChunk, yh, yl, acc, 2, 0);
    THA4_STAMP(d, son, 3, 8);
    const float inv = scl[2];
    float* zframe = d.z2 + (size_t)n * kNB2 * NPIX * 16;
    const int p = w.lane & 15;
#pragma unroll
    for (int b = 0; b < kNB2; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg)
        store16_wt(reinterpret_cast<char*>(zframe + z_offset(b, 0, 0, NPIX)) + (unsigned)(z_offset(0, w.lane >> 4, pix0[pg] + p, NPIX) * sizeof(float)), acc[b][pg] * inv);
    THA4_STAMP(d, son, 3, 9);
  }
  THA4_SPAN(d, 2, 1);
#undef THA4_L1R_CHUNK_TOP
}


// ---- face + level 0 with the activations in registers (round 6) -------------------------------------------------------
// front16_kernel at batch 1: 16 waves per 64-pixel workgroup, rows split over four waves per pixel slot, activations through LDS, 42 (level 0) /
// 8 (face) chunk barriers with ONE chunk in flight - neither the matrix pipe (ablation: -1.6 us without MFMAs) nor the L2 -> LDS path
// (tools/microbench/lds_stream.hip: 115 GB/s per CU against the 30 it draws) bounds it; the serial chain barrier -> fragment reads -> 9 MFMAs
// -> barrier does.  Here a workgroup is FOUR waves, each owning whole rows of one pixel group (level 0: 96 operand + 96 accumulator VGPRs),
// LDS carries the weight ring only (<= 80 KiB), so TWO workgroups share a CU: the grid is [level-0 workgroups | face workgroups] and the
// dispatcher's breadth-first placement (tools/microbench/tg_id_probe.hip: workgroups b and b + 256 of a 512-workgroup grid share a CU) gives
// every SIMD one level-0 wave (1512 MFMAs) and one face wave (684) that never synchronise with each other.
template <int WAVES, int SLOTS, int CHUNK_PIECES>
struct RingRegs {
  static constexpr int kChunk = CHUNK_PIECES * 2048, kDepth = SLOTS - 1, kCPW = 2 * CHUNK_PIECES / WAVES, kBytes = SLOTS * kChunk;
  static_assert((2 * CHUNK_PIECES) % WAVES == 0, "every wave must issue the same number of copies per chunk (counted waits)");
  static_assert(kDepth * kCPW <= 16, "THA4_BARRIER_KEEP counts up to 16 copies");
  const char* gw;
  char* base;
  int wave, lane;
  THA4_DEV void fetch(int c) const { fetch_pieces<2 * CHUNK_PIECES, WAVES>(gw + (size_t)c * kChunk, base + (c % SLOTS) * kChunk, wave, lane); }
  // chunk c of nc has landed for every wave and every wave is done with chunk c - 1; then chunk c + kDepth is requested into that slot
  THA4_DEV void top(int c, int nc) const {
    const int last = c + kDepth - 1 < nc - 1 ? c + kDepth - 1 : nc - 1;
    THA4_BARRIER_KEEP((last - c) * kCPW);
    if (c + kDepth < nc) fetch(c + kDepth);
  }
  THA4_DEV const char* at(int c) const { return base + (c % SLOTS) * kChunk + lane * 16; }
};

struct StampCtx { const StudentDev* d; bool on; int slot; };

// ---- the ring with EARLY barriers: one continuous fragment pipeline per layer (round 6) ------------------------------------------------
// In-kernel stamps of the first register form (tools/stamps_student.py, profiles/r06_student_b1_reading.md): a 24-KiB chunk of level 0 took 0.47 us - 0.14 us
// at its top (counted wait + barrier + a burst of six LDS-DMA issues at ~60 cycles each), ~0.07 us refilling the fragment pipeline behind the barrier, 0.26 us
// of matrix work.  Here the barrier of chunk c + 1 is passed D steps BEFORE chunk c's last MFMAs - from then on the fragment reads run ahead across the
// chunk boundary and the pipeline never drains inside a layer - and the LDS-DMA copies it releases are issued one or two per step under the MFMAs instead
// of in a burst.  Price: the slot that is rewritten behind top(c) is the one of chunk c - 2 (chunk c - 1 is still being read), so of SLOTS slots one is in
// use, one has landed and SLOTS - 2 are in flight or being requested.
template <int WAVES, int SLOTS, int CHUNK_PIECES>
struct RingEarly {
  static constexpr int kChunk = CHUNK_PIECES * 2048, kCPW = 2 * CHUNK_PIECES / WAVES, kBytes = SLOTS * kChunk, kAhead = SLOTS - 2, kSlots = SLOTS;
  static_assert((2 * CHUNK_PIECES) % WAVES == 0, "every wave must issue the same number of copies per chunk (counted waits)");
  static_assert(SLOTS >= 3 && (SLOTS - 3) * kCPW <= 16, "THA4_BARRIER_KEEP_VM counts up to 16 copies");
  const char* gw;
  char* base;
  int wave, lane;
  // copies [i0, i1) of this wave's kCPW copies of chunk c
  THA4_DEV void fetch_part(int c, int i0, int i1) const {
    const char* g = gw + (size_t)c * kChunk;
    char* l = base + (c % SLOTS) * kChunk;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const int pc = i * WAVES + wave;
      THA4_HOOK_FETCH(glds16(g + pc * 1024 + (unsigned)(lane * 16), l + pc * 1024));
    }
  }
  THA4_DEV void fetch(int c) const { fetch_part(c, 0, kCPW); }
  // chunk c of nc has landed for every wave (everything but the copies of the chunks requested behind it), and every wave has consumed chunk c - 2
  THA4_DEV void wait_top(int c, int nc) const {
    const int last = c + SLOTS - 3 < nc - 1 ? c + SLOTS - 3 : nc - 1;
    THA4_BARRIER_KEEP_VM((last > c ? last - c : 0) * kCPW);
  }
  THA4_DEV const char* at(int c) const { return base + (c % SLOTS) * kChunk + lane * 16; }
};

#ifndef THA4_RING_SPREAD
#define THA4_RING_SPREAD 1     // 1: the LDS-DMA copies a barrier releases are issued under the following steps' MFMAs; 0: in a burst behind the barrier
#endif
// One linear layer with x in registers as ONE software pipeline over all its chunks.  Chunks: CQ K groups x NBC output blocks (CQ > 1 only with NBC == NB),
// global chunk index c0 + k.  acc: the layer's NB output blocks.
template <int NB, int KG, int NBC, int CQ, class Ring, int KGX, int PG>
THA4_DEV void layer_regs_e(const Ring& ring, const int c0, const int nc, const f16x8 (&xh)[KGX][PG], const f16x8 (&xl)[KGX][PG], f32x4 (&acc)[NB][PG],
                           const StampCtx st = StampCtx{nullptr, false, 0}) {
  static_assert(NB % NBC == 0 && KG <= KGX && (CQ == 1 || NBC == NB), "bad chunking");
  constexpr int NCHQ = NB / NBC;
  constexpr int GB = NBC == 1 ? 1 : THA4_REGS_GROUP_BLOCKS ? THA4_REGS_GROUP_BLOCKS : ((PG >= 2 || KGX * PG > 8) ? 2 : (NBC % 3 == 0 ? 3 : 2)), NGB = NBC / GB;
  constexpr int T = CQ * NGB, G = KG * NGB * NCHQ, NCH = (G + T - 1) / T;
  constexpr int D = THA4_REGS_PREFETCH < T ? THA4_REGS_PREFETCH : T, NBUF = D + 1;
  constexpr int CPW = Ring::kCPW, CPS = THA4_RING_SPREAD ? (CPW + T - 1) / T : CPW;          // copies issued per step behind a barrier
  static_assert(NBC % GB == 0 && GB <= 3 && 2 * GB * D <= 15, "fragment pipeline out of range");
  f16x8 ah[NBUF][GB], al[NBUF][GB];
  auto load = [&](auto gc) {
    constexpr int g = decltype(gc)::value, k = g / T, t = g % T;
    const char* wv = ring.at(c0 + k);
    static_for<0, GB>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr int off = ((t / NGB) * NBC + (t % NGB) * GB + b) * 2048;
      lds_read16<off>(ah[g % NBUF][b], wv);
      lds_read16<off + 1024>(al[g % NBUF][b], wv);
    });
  };
  ring.wait_top(c0, nc);
  if (c0 + Ring::kAhead < nc) ring.fetch(c0 + Ring::kAhead);
  static_for<0, (D < G ? D : G)>(load);
  static_for<0, G>([&](auto gc) {
    constexpr int g = decltype(gc)::value, k = g / T, t = g % T, buf = g % NBUF;
    constexpr int q = CQ == 1 ? k / NCHQ : k * CQ + t / NGB;
    constexpr int bo = (CQ == 1 ? (k % NCHQ) * NBC : 0) + (t % NGB) * GB;
    constexpr int ktop = (g + D) / T;                                  // the newest chunk of this layer whose barrier has been passed once this step's is
    if constexpr ((g + D) % T == 0 && ktop < NCH) {
      THA4_STAMP(*st.d, st.on && c0 + ktop >= 20 && c0 + ktop < 24, st.slot, 16 + 3 * (c0 + ktop - 20));
      ring.wait_top(c0 + ktop, nc);
      THA4_STAMP(*st.d, st.on && c0 + ktop >= 20 && c0 + ktop < 24, st.slot, 17 + 3 * (c0 + ktop - 20));
    }
    if constexpr (g + D < G) load(std::integral_constant<int, g + D>());
    constexpr int newer = ((g + D < G ? g + D : G - 1) - g) * 2 * GB;
    if constexpr (GB == 1) lds_wait<newer>(ah[buf][0], al[buf][0]);
    else if constexpr (GB == 2) lds_wait<newer>(ah[buf][0], al[buf][0], ah[buf][1], al[buf][1]);
    else lds_wait<newer>(ah[buf][0], al[buf][0], ah[buf][1], al[buf][1], ah[buf][2], al[buf][2]);
    THA4_SCHED_FENCE();
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(ah[buf][b], xh[q][pg], acc[bo + b][pg]);
    if constexpr (ktop >= 1 && ktop < NCH) {                            // this step's share of the copies the newest barrier released
      constexpr int s0 = g - (ktop * T - D);
      if (c0 + ktop + Ring::kAhead < nc) ring.fetch_part(c0 + ktop + Ring::kAhead, s0 * CPS < CPW ? s0 * CPS : CPW, (s0 + 1) * CPS < CPW ? (s0 + 1) * CPS : CPW);
    }
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(ah[buf][b], xl[q][pg], acc[bo + b][pg]);
#pragma unroll
    for (int b = 0; b < GB; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) acc[bo + b][pg] = mfma16h(al[buf][b], xh[q][pg], acc[bo + b][pg]);
    THA4_SCHED_FENCE();
  });
  // a short last chunk may leave copies of the newest barrier's chunk unissued: the next layer's counted wait expects them in the queue
  if constexpr (NCH >= 2) {
    constexpr int done = (G - ((NCH - 1) * T - D)) * CPS;
    if (done < CPW && c0 + NCH - 1 + Ring::kAhead < nc) ring.fetch_part(c0 + NCH - 1 + Ring::kAhead, done, CPW);
  }
}

// one linear layer, x in registers: chunks of ONE K group x NBC output blocks, chunk index c0 + q (NB / NBC) + h
template <int NB, int KG, int NBC, class Ring, int KGX, int PG>
THA4_DEV void layer_regs(const Ring& ring, int c0, int nc, const f16x8 (&xh)[KGX][PG], const f16x8 (&xl)[KGX][PG], f32x4 (&acc)[NB][PG], const StampCtx st = StampCtx{nullptr, false, 0}) {
  static_assert(NB % NBC == 0 && KG <= KGX, "bad chunking");
#pragma unroll
  for (int q = 0; q < KG; ++q)
#pragma unroll
    for (int h = 0; h < NB / NBC; ++h) {
      const int c = c0 + q * (NB / NBC) + h;
      if (c >= 10 && c < 14) THA4_STAMP(*st.d, st.on, st.slot, 16 + 3 * (c - 10));
      ring.top(c, nc);
      if (c >= 10 && c < 14) THA4_STAMP(*st.d, st.on, st.slot, 17 + 3 * (c - 10));
      mma_chunk_regs<NBC, 1>(ring.at(c), xh, xl, acc, q, h * NBC);
      if (c >= 10 && c < 14) THA4_STAMP(*st.d, st.on, st.slot, 18 + 3 * (c - 10));
    }
}

// sine epilogue of a register-resident layer: x <- split(sin(acc / S + c b)), biases and 1/S from LDS
// PB: priority base of the calling workgroup (front16r_kernel: THA4_FRONT_L0_PRIO lifts the level-0 waves - the launch's critical path - above the face
// waves they share their SIMDs with: matrix phase PB, VALU phase PB + 1)
#if !defined(THA4_EMU) && THA4_PHASE_PRIO
#define THA4_PRIO_SET(p) __builtin_amdgcn_s_setprio(p)
#else
#define THA4_PRIO_SET(p)
#endif
template <int NB, int KG, int PG, int PB = 0>
THA4_DEV void sine_regs(const f32x4 (&acc)[NB][PG], const float* bias_lds, float inv, int g4, f16x8 (&xh)[KG][PG], f16x8 (&xl)[KG][PG]) {
  THA4_PRIO_SET(PB + 1);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + b * 16 + g4);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = sin_u(fmaf(acc[b][pg][j], inv, bb[j]));
      put_rows<KG, PG>(xh, xl, pg, b, v);
    }
  }
  pin_rows<KG, PG>(xh, xl);
  THA4_PRIO_SET(PB);
}

// (wx, wy, pb: LDS copies - 2 x NB global loads per wave here would either all be in flight at once (4 VGPRs each) or serialise into NB round trips)
template <class G, int NB, class Sink>
THA4_DEV void first16_pos_to(const float* wx, const float* wy, const float* pb, const float (&x)[G::PG], const float (&y)[G::PG], Sink&& sink, const WaveCtx& w) {
  constexpr int PG = G::PG;
  const int g4 = (w.lane >> 4) * 4;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const f32x4 vx = *reinterpret_cast<const f32x4*>(wx + b * 16 + g4);
    const f32x4 vy = *reinterpret_cast<const f32x4*>(wy + b * 16 + g4);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(pb + b * 16 + g4);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = sin_u(fmaf(vx[j], x[pg], fmaf(vy[j], y[pg], vb[j])));     // tables carry the sine's scale
      sink(pg, b, v);
    }
  }
}

// Prologue of a register-resident workgroup: the pose-folded first-layer bias (same arithmetic and summation order as pose_bias_to_lds), the first
// layer's position columns, and the streamed layers' biases + 1/S - everything the kernel reads per value - into LDS.  ALL global loads are requested
// before the first result is needed (a thread folds up to RW channels at once: 45 loads each; the plain copies ride along): one memory round trip where the
// loop-after-loop form paid three to four (stamps: 3.6 us of a 28-us level-0 workgroup).
//   pb: [pose-folded bias W | wx W | wy W] floats, W = NB * 16;   bias_lds: [NBIAS biases | NSCL scales]
template <int NB, int THREADS, int NBIAS, int NSCL>
THA4_DEV void prologue_to_lds(const StudentDev& d, int net, int n, const float* bsrc, const float* ssrc, float* pb, float* bias_lds) {
  constexpr int W = NB * 16, RW = (W + THREADS - 1) / THREADS, RB = (NBIAS + NSCL + THREADS - 1) / THREADS, RT = (2 * W + THREADS - 1) / THREADS;
  const int tid = threadIdx.x;
  float bv[RB], tv[RT], s[RW][3];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int c = tid + r * THREADS;
    bv[r] = c < NBIAS ? bsrc[c] : (c < NBIAS + NSCL ? ssrc[c - NBIAS] : 0.0f);
  }
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int c = tid + r * THREADS;
    tv[r] = c < W ? d.wx[net][c] : (c < 2 * W ? d.wy[net][c - W] : 0.0f);
  }
  const float* pose = d.pose + (size_t)n * kPose;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int c = tid + r * THREADS, cc = c < W ? c : 0;
    const float* wp = d.wpose[net] + cc;
    s[r][0] = d.bias1[net][cc];
    s[r][1] = 0.f;
    s[r][2] = 0.f;
#pragma unroll
    for (int k = 0; k < kPose; ++k) s[r][k % 3] = fmaf(wp[(size_t)THA4_HOOK_POSE_ROW(k) * W], pose[k], s[r][k % 3]);
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int c = tid + r * THREADS;
    if (c < W) pb[c] = (s[r][0] + (s[r][1] + s[r][2])) * d.pb_scale;
  }
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int c = tid + r * THREADS;
    if (c < 2 * W) pb[W + c] = tv[r];
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int c = tid + r * THREADS;
    if (c < NBIAS + NSCL) bias_lds[c] = bv[r];
  }
}

#ifndef THA4_FRONT_L0_PRIO
#define THA4_FRONT_L0_PRIO 0     // priority base of the level-0 waves of front16r_kernel (the face waves stay at 0): 1 = level 0 wins every issue conflict
#endif
template <int NBC0, int SLOTS0, int NBCF, int SLOTSF>
struct FrontRCfg {
  static constexpr int WAVES = 4, PG = 1, THREADS = WAVES * 64, PX = WAVES * PG * 16;
  using G = GeoRegs<WAVES, PG>;
  // level 0: 360 -> 360 -> 180 (sine), then z1; chunks of one K group x NBC0 blocks
  using Ring0 = RingEarly<WAVES, SLOTS0, NBC0>;
  static constexpr int kChunksA = kKG0 * (kNB0 / NBC0), kChunksB = kKG0 * (kNB1 / NBC0), kChunksZ = kKG1 * (kNB1 / NBC0);
  static constexpr int kChunks0 = kChunksA + kChunksB + kChunksZ;
  static constexpr int kBias0 = (kNB0 + kNB1) * 16;                                            // floats: bias A | bias B, then 1/S x 3 (+ pad)
  static constexpr int kPb0Off = Ring0::kBytes, kBias0Off = kPb0Off + 3 * kNB0 * 16 * 4, kLds0 = kBias0Off + (kBias0 + 4) * 4;     // pb | wx | wy, then the biases
  // face: 7 x (128 -> 128, sine) + head 128 -> 4; chunks of one K group x NBCF blocks, the head's 4 pieces (1 block x 4 K groups) padded to one chunk
  using RingF = RingEarly<WAVES, SLOTSF, NBCF>;
  static constexpr int kChunksLayerF = kKGF * (kNBF / NBCF);
  static constexpr int kChunksF = 7 * kChunksLayerF + 1;
  static constexpr int kStreamPiecesF = kChunksF * NBCF;
  static constexpr int kBiasF = 7 * kNBF * 16 + 16;                                            // floats: 7 sine layers | head, then 1/S x 8
  static constexpr int kPbFOff = RingF::kBytes, kBiasFOff = kPbFOff + 3 * kNBF * 16 * 4, kLdsF = kBiasFOff + (kBiasF + 8) * 4;
  static constexpr int LDS = kLds0 > kLdsF ? kLds0 : kLdsF;
  static_assert(LDS <= 80 * 1024, "two workgroups must share a CU");
  static_assert(kNB0 % NBC0 == 0 && kNB1 % NBC0 == 0 && kNBF % NBCF == 0 && NBCF >= kKGF, "a chunk must be a whole number of blocks of every layer (and hold the head's 4 pieces)");
};

template <class Cfg>
THA4_DEV void level0_regs_body(const StudentDev& d, char* smem, const WaveCtx& w) {
  using G = typename Cfg::G;
  using Ring = typename Cfg::Ring0;
  constexpr int PG = Cfg::PG, S = 128, NPIX = S * S, NC = Cfg::kChunks0, NBC = Ring::kChunk / 2048;
  const Ring ring{reinterpret_cast<const char*>(d.w_l0), smem, w.wave, w.lane};
  THA4_PRIO_SET(THA4_FRONT_L0_PRIO);
  const bool son = (w.blk == 0 && w.wave == 0) || (w.blk == 131 && w.wave == 3);
  const StampCtx st{&d, son, w.blk == 0 ? 0 : 1};
  THA4_STAMP(d, son, st.slot, 0);
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos128, pix0, X0, Y, px, py);
  float* pb = reinterpret_cast<float*>(smem + Cfg::kPb0Off);
  float* bias_lds = reinterpret_cast<float*>(smem + Cfg::kBias0Off);
  // the prologue's latency-bound loads go out BEFORE the ring's first burst (48 KiB per workgroup, every workgroup of the chip at once): queued behind it
  // they took 2.3 us; the ring's first chunks have the whole first layer to land
  prologue_to_lds<kNB0, Cfg::THREADS, Cfg::kBias0, 3>(d, 1, n, d.b_l0, d.s_l0, pb, bias_lds);
  THA4_STAMP(d, son, st.slot, 1);
#pragma unroll
  for (int c = 0; c < Ring::kAhead; ++c) ring.fetch(c);        // (top(c) requests chunk c + kAhead)
  THA4_BARRIER_LDS();                                          // (the tables are in LDS for every wave; the ring copies stay in flight)
  THA4_SPAN(d, 4, 0);                                          // (stamps builds: earliest / latest "prologue done" over the level-0 workgroups)
  THA4_STAMP(d, son, st.slot, 2);
  const int g4 = (w.lane >> 4) * 4;
  const float* scl = bias_lds + Cfg::kBias0;
  f16x8 xh[kKG0][PG], xl[kKG0][PG];
  constexpr int PB = THA4_FRONT_L0_PRIO;
  THA4_PRIO_SET(PB + 1);
  first16_pos_to<G, kNB0>(pb + kNB0 * 16, pb + 2 * kNB0 * 16, pb, px, py, [&](int pg, int b, const f32x4& v) { put_rows<kKG0, PG>(xh, xl, pg, b, v); }, w);
  pin_rows<kKG0, PG>(xh, xl);
  THA4_PRIO_SET(PB);
  THA4_STAMP(d, son, st.slot, 3);
  {   // 360 -> 360, sine
    f32x4 acc[kNB0][PG];
    zero_acc<kNB0, PG>(acc);
    layer_regs_e<kNB0, kKG0, NBC, 1>(ring, 0, NC, xh, xl, acc, st);
    THA4_STAMP(d, son, st.slot, 4);
    sine_regs<kNB0, kKG0, PG, PB>(acc, bias_lds, scl[0], g4, xh, xl);
    THA4_STAMP(d, son, st.slot, 5);
  }
  f16x8 yh[kKG1][PG], yl[kKG1][PG];
  {   // 360 -> 180, sine
    f32x4 acc[kNB1][PG];
    zero_acc<kNB1, PG>(acc);
    layer_regs_e<kNB1, kKG0, NBC, 1>(ring, Cfg::kChunksA, NC, xh, xl, acc);
    THA4_STAMP(d, son, st.slot, 6);
    sine_regs<kNB1, kKG1, PG, PB>(acc, bias_lds + kNB0 * 16, scl[1], g4, yh, yl);
    THA4_STAMP(d, son, st.slot, 7);
  }
  {   // z1 = c W10[:, :180] h0 (fp32) -> global z[n][b][g][pix][4]; the consumer is level 1's first (sine) layer
    f32x4 acc[kNB1][PG];
    zero_acc<kNB1, PG>(acc);
    layer_regs_e<kNB1, kKG1, NBC, 1>(ring, Cfg::kChunksA + Cfg::kChunksB, NC, yh, yl, acc);
    THA4_STAMP(d, son, st.slot, 8);
    const float inv = scl[2];
    float* zframe = d.z1 + (size_t)n * kNB1 * NPIX * 16;
    const int p = w.lane & 15;
#pragma unroll
    for (int b = 0; b < kNB1; ++b)
#pragma unroll
      for (int pg = 0; pg < PG; ++pg)
        store16_wt(reinterpret_cast<char*>(zframe + z_offset(b, 0, 0, NPIX)) + (unsigned)(z_offset(0, w.lane >> 4, pix0[pg] + p, NPIX) * sizeof(float)), acc[b][pg] * inv);
    THA4_STAMP(d, son, st.slot, 9);
  }
}

template <class Cfg>
THA4_DEV void face_regs_body(const StudentDev& d, char* smem, const WaveCtx& w) {
  using G = typename Cfg::G;
  using Ring = typename Cfg::RingF;
  constexpr int PG = Cfg::PG, S = kFaceSize, NPIX = S * S, NC = Cfg::kChunksF, NBC = Ring::kChunk / 2048;
  const Ring ring{reinterpret_cast<const char*>(d.w_face), smem, w.wave, w.lane};
  const bool son = w.blk == 0 && w.wave == 0;
  THA4_STAMP(d, son, 2, 0);
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
  const int n = slot_pixels<G, S>(w, d.pos128, pix0, X0, Y, px, py);
  float* pb = reinterpret_cast<float*>(smem + Cfg::kPbFOff);
  float* bias_lds = reinterpret_cast<float*>(smem + Cfg::kBiasFOff);
  prologue_to_lds<kNBF, Cfg::THREADS, Cfg::kBiasF, 8>(d, 0, n, d.b_face, d.s_face, pb, bias_lds);
#pragma unroll
  for (int c = 0; c < Ring::kAhead; ++c) ring.fetch(c);
  THA4_BARRIER_LDS();
  THA4_SPAN(d, 4, 1);                                          // (... over the face workgroups)
  THA4_STAMP(d, son, 2, 2);
  const int g4 = (w.lane >> 4) * 4;
  const float* scl = bias_lds + Cfg::kBiasF;
  f16x8 xh[kKGF][PG], xl[kKGF][PG];
  THA4_PRIO_VALU();
  first16_pos_to<G, kNBF>(pb + kNBF * 16, pb + 2 * kNBF * 16, pb, px, py, [&](int pg, int b, const f32x4& v) { put_rows<kKGF, PG>(xh, xl, pg, b, v); }, w);
  pin_rows<kKGF, PG>(xh, xl);
  THA4_PRIO_MFMA();
  THA4_STAMP(d, son, 2, 3);
#pragma unroll
  for (int l = 0; l < 7; ++l) {
    f32x4 acc[kNBF][PG];
    zero_acc<kNBF, PG>(acc);
    layer_regs_e<kNBF, kKGF, NBC, 1>(ring, l * Cfg::kChunksLayerF, NC, xh, xl, acc);
    sine_regs<kNBF, kKGF, PG>(acc, bias_lds + l * kNBF * 16, scl[l], g4, xh, xl);
    THA4_STAMP(d, son, 2, 4 + l);
  }
  // head: ONE block x 4 K groups = the first 4 pieces of the last chunk; rows 0..3 (the image channels) live in lane group 0
  f32x4 a1[1][PG];
  zero_acc<1, PG>(a1);
  layer_regs_e<1, kKGF, 1, kKGF>(ring, NC - 1, NC, xh, xl, a1);
  if (w.lane < 16) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + 7 * kNBF * 16);
    const float inv = scl[7];
    float* fo = d.face + (size_t)n * 4 * NPIX;
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      const f32x4 v = a1[0][pg] * inv + bb;
#pragma unroll
      for (int j = 0; j < 4; ++j) fo[(size_t)j * NPIX + pix0[pg] + w.lane] = v[j];
    }
  }
  THA4_STAMP(d, son, 2, 12);
}

template <int NBC0, int SLOTS0, int NBCF, int SLOTSF>
__global__ void __launch_bounds__(256, 2) front16r_kernel(StudentDev d) {
  warm_kernarg<(int)sizeof(StudentDev)>();
  using Cfg = FrontRCfg<NBC0, SLOTS0, NBCF, SLOTSF>;
  THA4_DYN_LDS(smem);
  WaveCtx w = wave_ctx<typename Cfg::G>();
  const int nl0 = d.front_l0_blocks;
  THA4_SPAN(d, (int)blockIdx.x < nl0 ? 0 : 1, 0);
  if ((int)blockIdx.x < nl0) {
    w.nblk = nl0;
    level0_regs_body<Cfg>(d, smem, w);
  } else {
    w.blk = blockIdx.x - nl0;
    w.nblk = gridDim.x - nl0;
    face_regs_body<Cfg>(d, smem, w);
  }
  THA4_SPAN(d, (int)blockIdx.x < nl0 ? 0 : 1, 1);
}

// ---- launch configuration ------------------------------------------------------------------------------
namespace cfg {
#ifndef THA4_FACE16_CFG
#define THA4_FACE16_CFG 4, 4, 1, 4          // NS, MS, PG, CQ (groups per chunk; 4 = whole layer)
#endif
#ifndef THA4_L016_CFG
#define THA4_L016_CFG 4, 4, 1, 2, 1         // NS, MS, PG, HBA (block slices per group), CQB
#endif
#ifndef THA4_L116_CFG
#define THA4_L116_CFG 4, 2, 1, 1, 1, 2      // NS, MS, PG, CQA, CQB [, HBA]
#endif
#ifndef THA4_L216_CFG
#define THA4_L216_CFG 4, 1, 1, 1            // NS, MS, PG, CQ
#endif
#ifndef THA4_L216P_CFG
// WAVES, strips per workgroup, pixel groups per strip (weights-resident level 2).  SIXTEEN waves (four per SIMD) share the 64
// one-group strips of a workgroup: the strips are handed out by an LDS ticket, so the wave count need not divide them, and a strip
// spends half its time waiting (777 VALU + 117 MFMA instructions ~ 3.9 k issue cycles against 7.5 k measured at two waves per SIMD).
// Same box, library built without packed fp32 (below): <16,64,1> 45.4 us (126 VGPRs, no spill - with packed ops it was 128 + 16
// spilled), <12,64,1> 46.9, <12,32,2> 47.9, <8,32,2> 48.9.  More waves per SIMD is what every kernel of this path wants (round 3):
// the streamed kernels went from 8 to 16 waves per CU as well (rows of a pixel slot split over more waves: THA4_L016_CFG /
// THA4_FACE16_CFG MS = 4, level 1 as two 8-wave workgroups per CU), front 55.4 -> 50.0 us, level 1 43.7 -> 41.1; two INDEPENDENT
// 4-wave workgroups per CU (same waves per SIMD, no shared barrier) changed nothing - the stalls are each wave's own latency chain.
//
// The two-group geometry <8,.,2> that rounds 1-2 shipped has a history (profiles/r03_sin_cliff.md).  With the compiler's packed-fp32
// arithmetic (v_pk_mul / v_pk_fma / v_pk_add_f32: SLP vectorisation + float4 expressions) it is FAULTY on gfx950 as soon as the sine is
// a v_sin_f32 - round 2's "3e-2 cliff" of the hardware-sine A/B build, and errors of up to 1.5 with the turn-based sine: 30-100
// run-to-run varying pixels per frame in this one kernel, every hand-off image and every other kernel correct, the sine instruction
// accurate to 3.8e-7 on every argument of the frame.  Forced memory waits only lowered the rate, making every counted wait total changed
// nothing; eight wait states in FRONT of every v_pk_* (re-assembled ISA, tools/hunt/) lower the rate by orders of magnitude, a build
// without packed ops removes the fault in every geometry; no pairwise producer -> v_pk_* hazard exists beyond the documented ones
// (tools/microbench/gen_pk_hazard.py), so the mechanism stays open and the class is established by elimination.  The library is therefore built WITHOUT packed-fp32
// instructions (tha4_amd/_build.py DEVICE_FLAGS; a v_pk_fma_f32 costs two v_fma_f32: nothing is lost), every geometry is then sound -
// each equals its forced-wait twin bit for bit (tools/compare_libs.py) - and with packed ops the two-group form is refused at compile
// time.  The round-1 "256 VGPRs + scratch -> wrong, varying pixels" incident of that geometry was most likely the same hazard.
#define THA4_L216P_CFG 16, 64, 1
#endif
#ifndef THA4_L2_RESIDENT
#define THA4_L2_RESIDENT 1                  // 1: level2_16p_kernel, 0: streamed level2_16_kernel
#endif
#ifndef THA4_L1_REGS
#define THA4_L1_REGS 1                      // 1: level1_16r_kernel (activations in registers, LDS = weight ring only; round 6), 0: level1_16_kernel
#endif
#ifndef THA4_FRONT_REGS
#define THA4_FRONT_REGS 1                   // 1: front16r_kernel (face + level 0, activations in registers, two 4-wave workgroups per CU; round 6), 0: front16_kernel
#endif
#ifndef THA4_FRONT16R_CFG
#define THA4_FRONT16R_CFG 6, 6, 8, 4        // level 0: blocks per chunk (6: 12 KiB), ring slots; face: blocks per chunk (8: 16 KiB), ring slots
#endif
#ifndef THA4_L116R_CFG
#define THA4_L116R_CFG 8, 2, 6              // WAVES, pixel groups per wave, ring slots of 24 KiB
#endif
using L2P = Level2PCfg<THA4_L216P_CFG>;
#define THA4_L216P_KERNEL v2::level2_16p_kernel<THA4_L216P_CFG>
using FaceG = Face16Cfg<THA4_FACE16_CFG>::G;
using L0G = Level016Cfg<THA4_L016_CFG>::G;
using L1G = Level116Cfg<THA4_L116_CFG>::G;
using L2G = Level216Cfg<THA4_L216_CFG>::G;
using FrontR = FrontRCfg<THA4_FRONT16R_CFG>;
#define THA4_FRONT16R_KERNEL v2::front16r_kernel<THA4_FRONT16R_CFG>
constexpr int kL0HBA = THA4_FRONT_REGS ? 1 : (Level016Cfg<THA4_L016_CFG>::kP1 == kNB0 ? 1 : kNB0 / Level016Cfg<THA4_L016_CFG>::kP1);
constexpr int kL0HBB = THA4_FRONT_REGS ? 1 : Level016Cfg<THA4_L016_CFG>::kHBB;
using L1R = Level1RCfg<THA4_L116R_CFG>;
#define THA4_L116R_KERNEL v2::level1_16r_kernel<THA4_L116R_CFG>
// (the register form reads whole K groups in block order: no row split, no block slices; its stream ends with 6 pieces of padding)
constexpr int kL1HBA = THA4_L1_REGS ? 1 : Level116Cfg<THA4_L116_CFG>::kHBA, kL1HBB = THA4_L1_REGS ? 1 : Level116Cfg<THA4_L116_CFG>::kHBB;
constexpr int kFaceMS = THA4_FRONT_REGS ? 1 : FaceG::MS, kL0MS = THA4_FRONT_REGS ? 1 : L0G::MS, kL1MS = THA4_L1_REGS ? 1 : L1G::MS, kL2MS = L2G::MS;
constexpr int kL1Px = THA4_L1_REGS ? L1R::PX : L1G::PX, kL1Threads = THA4_L1_REGS ? L1R::THREADS : L1G::THREADS;
#define THA4_FACE16_KERNEL v2::face16_kernel<THA4_FACE16_CFG>
#define THA4_FRONT16_KERNEL v2::front16_kernel<THA4_FACE16_CFG, THA4_L016_CFG>
#ifndef THA4_FRONT_MERGE
#define THA4_FRONT_MERGE 1                  // 1: face + level 0 share one launch (front16_kernel); 0: two launches (A/B)
#endif
#define THA4_L016_KERNEL v2::level0_16_kernel<THA4_L016_CFG>
#define THA4_L116_KERNEL v2::level1_16_kernel<THA4_L116_CFG>
#define THA4_L216_KERNEL v2::level2_16_kernel<THA4_L216_CFG>
template <class G>
constexpr int blocks_for(int batch, int side) { return batch * (side * side) / G::PX; }
// dynamic LDS of each launch.  The streamed kernels park their pose-folded bias vector in ring slot 1 (idle until the first
// streamed layer prefetches into it); the weights-resident level 2 has no ring and appends it.
constexpr int kFrontLds = FaceG::LDS > L0G::LDS ? FaceG::LDS : L0G::LDS;
constexpr int kFrontRLds = FrontR::LDS;
constexpr int kFrontPx = THA4_FRONT_REGS ? FrontR::PX : L0G::PX;          // pixels per workgroup of both nets in the front launch
constexpr int kFaceLds = FaceG::LDS, kL0Lds = L0G::LDS, kL1Lds = L1G::LDS, kL1RLds = L1R::LDS, kL2Lds = L2G::LDS, kL2PLds = L2P::LDS + pb_lds_bytes(kNB2);
static_assert(pb_lds_bytes(kNB0) <= L0G::SLOT && pb_lds_bytes(kNBF) <= FaceG::SLOT && pb_lds_bytes(kNB1) <= L1G::SLOT && pb_lds_bytes(kNB2) <= L2G::SLOT,
              "the bias vector must fit a ring slot");

}  // namespace cfg

// ---- host packer -------------------------------------------------------------------------------------
// One layer [O x I] -> pieces [Q][piece idx][hi 1 KiB | lo 1 KiB] with the block order of (MS, HB).  The weights are
// multiplied by `pre` (30 for layers feeding a sine) and by a power of two S chosen so that max |W'| lies in
// [2^13, 2^14): hi = fp16(W'), lo = fp16(W' - hi) then keeps ~11 bits even for weights 2^-11 below the largest.
// Returns 1/S (exact), which the kernel multiplies the accumulator with.
inline float pack_layer16(const float* W, int ldw, int col0, int O, int I, int NB, int KG, int MS, int HB, float pre,
                          std::vector<char>& dst) {
  float mx = 0.f;
  for (int o = 0; o < O; ++o)
    for (int i = 0; i < I; ++i) mx = std::max(mx, std::fabs(W[(size_t)o * ldw + col0 + i] * pre));
  int e = 0;
  if (mx > 0.f) { std::frexp(16384.0f / mx, &e); e -= 1; }     // 2^e <= 16384 / mx < 2^(e+1)
  e = std::min(std::max(e, -24), 24);
  const float S = std::ldexp(1.0f, e);
  const size_t at = dst.size();
  dst.resize(at + (size_t)KG * NB * 2048);
  for (int Q = 0; Q < KG; ++Q)
    for (int idx = 0; idx < NB; ++idx) {
      const int b = piece_block(NB, MS, HB, idx);
      _Float16* hi = reinterpret_cast<_Float16*>(dst.data() + at + ((size_t)Q * NB + idx) * 2048);
      _Float16* lo = hi + 512;
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int o = 16 * b + (lane & 15);
          const int i = 32 * Q + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
          const float v = (o < O && i < I) ? (W[(size_t)o * ldw + col0 + i] * pre) * S : 0.0f;
          const _Float16 h = (_Float16)v;
          hi[lane * 8 + j] = h;
          lo[lane * 8 + j] = (_Float16)(v - (float)h);
        }
    }
  return std::ldexp(1.0f, -e);
}

struct StudentPacked16 {
  std::vector<char> w_face, w_l0, w_l1, w_l2;
  std::vector<float> s_face, s_l0, s_l1, s_l2;   // 1/S of every streamed layer, in execution order
  std::vector<float> b_face, b_l0, b_l1, b_l2;   // biases of the streamed layers; sine layers carry the 30x
  std::vector<float> wx[4], wy[4];               // first-layer position columns x 30 (0 face, 1..3 body levels)
};

// mirrors pack_student (siren_layout.h); `p1` is the generation-1 pack of the same weights (bias / first-layer tables)
inline void pack_student16(const StudentWeightsView& v, const StudentPacked& p1, StudentPacked16& p) {
  p = StudentPacked16();
  constexpr float W30 = kSineScale16;          // omega_0 in the unit the sine takes: turns (default) or radians
  for (int i = 1; i < 8; ++i) p.s_face.push_back(pack_layer16(v.face_sine[i].weight, kCF, 0, kCF, kCF, kNBF, kKGF, cfg::kFaceMS, 1, W30, p.w_face));
  p.s_face.push_back(pack_layer16(v.face_last.weight, kCF, 0, 4, kCF, 1, kKGF, 1, 1, 1.0f, p.w_face));
  if (THA4_FRONT_REGS) p.w_face.resize((size_t)cfg::FrontR::kStreamPiecesF * 2048, 0);      // front16r_kernel's last chunk is the head's 4 pieces + padding
  p.s_l0.push_back(pack_layer16(v.body_sine[0][1].weight, kC0, 0, kC0, kC0, kNB0, kKG0, cfg::kL0MS, cfg::kL0HBA, W30, p.w_l0));
  p.s_l0.push_back(pack_layer16(v.body_sine[0][2].weight, kC0, 0, kC1, kC0, kNB1, kKG0, cfg::kL0MS, cfg::kL0HBB, W30, p.w_l0));
  p.s_l0.push_back(pack_layer16(v.body_sine[1][0].weight, kC1 + 2 + kPose, 0, kC1, kC1, kNB1, kKG1, cfg::kL0MS, 1, W30, p.w_l0));
  p.s_l1.push_back(pack_layer16(v.body_sine[1][1].weight, kC1, 0, kC1, kC1, kNB1, kKG1, cfg::kL1MS, cfg::kL1HBA, W30, p.w_l1));
  p.s_l1.push_back(pack_layer16(v.body_sine[1][2].weight, kC1, 0, kC2, kC1, kNB2, kKG1, cfg::kL1MS, cfg::kL1HBB, W30, p.w_l1));
  p.s_l1.push_back(pack_layer16(v.body_sine[2][0].weight, kC2 + 2 + kPose, 0, kC2, kC2, kNB2, kKG2, cfg::kL1MS, 1, W30, p.w_l1));
  if (THA4_L1_REGS) p.w_l1.resize((size_t)cfg::L1R::kStreamPieces * 2048, 0);     // level1_16r_kernel's last chunk is K group 2 of the z layer + padding
  p.s_l2.push_back(pack_layer16(v.body_sine[2][1].weight, kC2, 0, kC2, kC2, kNB2, kKG2, 1, 1, W30, p.w_l2));
  p.s_l2.push_back(pack_layer16(v.body_sine[2][2].weight, kC2, 0, kC2, kC2, kNB2, kKG2, 1, 1, W30, p.w_l2));
  p.s_l2.push_back(pack_layer16(v.body_last.weight, kC2, 0, kHeadC, kC2, 1, kKG2, 1, 1, 1.0f, p.w_l2));
  // biases: every entry of the generation-1 arrays belongs to a sine layer except the trailing 16 of face / level 2
  auto scaled = [&](const std::vector<float>& b, size_t plain_tail) {
    std::vector<float> o(b);
    for (size_t i = 0; i + plain_tail < o.size(); ++i) o[i] *= W30;
    return o;
  };
  p.b_face = scaled(p1.b_face, 16);
  p.b_l0 = scaled(p1.b_l0, 0);
  p.b_l1 = scaled(p1.b_l1, 0);
  p.b_l2 = scaled(p1.b_l2, 16);
  const FirstLayerPack* fl[4] = {&p1.f_face, &p1.f_l0, &p1.f_l1, &p1.f_l2};
  for (int i = 0; i < 4; ++i) {
    p.wx[i] = fl[i]->wx;
    p.wy[i] = fl[i]->wy;
    for (auto& x : p.wx[i]) x *= W30;
    for (auto& x : p.wy[i]) x *= W30;
  }
}

}  // namespace v2
}  // namespace tha4
