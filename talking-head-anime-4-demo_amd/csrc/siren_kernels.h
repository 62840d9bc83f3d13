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
#pragma once
#include "tha4_platform.h"
#include "siren_layout.h"

namespace tha4 {

constexpr int kWaves = 4;            // waves per workgroup (one per SIMD)
constexpr int kBlock = kWaves * 64;

struct StudentDev {
  // packed parameters (device)
  const float *w_face, *w_l0, *w_l1, *w_l2;          // weight streams (siren_layout.h)
  const float *b_face, *b_l0, *b_l1, *b_l2;          // biases of the streamed layers
  const float *wx[4], *wy[4], *bias1[4], *wpose[4];  // first layers: 0 face, 1..3 body levels
  const float *pos128, *pos256, *pos512;             // affine_grid axes
  // workspace (device)
  float* pbias;   // [B][kPbStride]
  float* z1;      // [B][kNB1][128*128][16]
  float* z2;      // [B][kNB2][256*256][16]
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
  int batch;
};

// ---------------------------------------------------------------------------------------------
// sin(30 z) exactly as the reference rounds it: u = fl(30 z) (siren.py:39), then sin(u) to ~1.4e-7
// abs: k = rint(u/pi), r = u - k pi by 3-term Cody-Waite (k pi_A exact for |k| < 4096),
// sin(r) = r + r^3 q(r^2) on [-pi/2, pi/2] (degree-9 minimax), sign flipped for odd k.
// ---------------------------------------------------------------------------------------------
THA4_DEV float sin_omega(float z) {
  const float u = kOmega * z;
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

// ---------------------------------------------------------------------------------------------
// weight stream: all waves of the workgroup copy PIECES x 1 KiB from global to an LDS ring slot
// ---------------------------------------------------------------------------------------------
template <int PIECES>
THA4_DEV void fetch_pieces(const char* g, char* l, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < (PIECES + kWaves - 1) / kWaves; ++i) {
    const int pc = i * kWaves + wave;
    if (pc < PIECES) glds16(g + pc * 1024 + lane * 16, l + pc * 1024);
  }
}

// One linear layer for the wave's PG pixel groups (16 px each): acc[b][pg] += W-block b x act.
//   NB  output blocks (16 channels each)      KQ  input quads (16 channels each)
//   CQ  quads per streamed chunk (KQ % CQ == 0)   ACTQ  quads per pixel group in the act image
//   NEXT_PIECES  size (KiB) of the first chunk of whatever layer follows in the stream (0: none);
//                it is prefetched during this layer's last chunk so layer boundaries cost no bubble.
// On entry chunk 0 of this layer is resident in ring slot `slot` (the preceding barrier covered it).
template <int NB, int KQ, int CQ, int PG, int ACTQ, int SLOT_BYTES, int NEXT_PIECES>
THA4_DEV void gemm_stream(const char*& gw, char* ring, int& slot, const f32x4* actv,
                          f32x4 (&acc)[NB][PG], int wave, int lane) {
  static_assert(KQ % CQ == 0, "chunking must divide K");
  static_assert(CQ * NB * 1024 <= SLOT_BYTES, "chunk exceeds ring slot");
  static_assert(NEXT_PIECES * 1024 <= SLOT_BYTES, "next chunk exceeds ring slot");
  constexpr int NC = KQ / CQ;
  constexpr int CHUNK = CQ * NB * 1024;
#pragma unroll 1
  for (int c = 0; c < NC; ++c) {
    const int nslot = slot ^ 1;
    if (c + 1 < NC) {
      fetch_pieces<CQ * NB>(gw + (size_t)(c + 1) * CHUNK, ring + nslot * SLOT_BYTES, wave, lane);
    } else if (NEXT_PIECES > 0) {
      fetch_pieces<NEXT_PIECES>(gw + (size_t)NC * CHUNK, ring + nslot * SLOT_BYTES, wave, lane);
    }
    const f32x4* wv = reinterpret_cast<const f32x4*>(ring + slot * SLOT_BYTES);
#pragma unroll
    for (int qq = 0; qq < CQ; ++qq) {
      f32x4 bf[PG];
#pragma unroll
      for (int pg = 0; pg < PG; ++pg) bf[pg] = actv[(pg * ACTQ + c * CQ + qq) * 64 + lane];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const f32x4 a = wv[(qq * NB + b) * 64 + lane];
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[b][pg] = mfma16(a[j], bf[pg][j], acc[b][pg]);
        }
      }
    }
    __syncthreads();   // next slot landed (vmcnt(0)) and every wave is done reading this one
    slot = nslot;
  }
  gw += (size_t)NC * CHUNK;
}

template <int NB, int PG>
THA4_DEV void zero_acc(f32x4 (&acc)[NB][PG]) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) acc[b][pg] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc + bias -> (sin) -> act image (block b becomes input quad b of the next layer)
template <int NB, int PG, int ACTQ, bool SIN>
THA4_DEV void store_act(f32x4 (&acc)[NB][PG], const float* bias, f32x4* actv, int lane) {
  const int g4 = (lane >> 4) * 4;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + b * 16 + g4);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      f32x4 v = acc[b][pg] + bb;
      if (SIN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sin_omega(v[j]);
      }
      actv[(pg * ACTQ + b) * 64 + lane] = v;
    }
  }
}

// sine hidden layer: act <- sin(30 (W act + b))
template <int NB, int KQ, int CQ, int PG, int ACTQ, int SLOT_BYTES, int NEXT_PIECES>
THA4_DEV void sine_layer(const char*& gw, const float*& bias, char* ring, int& slot, f32x4* actv, int wave, int lane) {
  f32x4 acc[NB][PG];
  zero_acc<NB, PG>(acc);
  gemm_stream<NB, KQ, CQ, PG, ACTQ, SLOT_BYTES, NEXT_PIECES>(gw, ring, slot, actv, acc, wave, lane);
  store_act<NB, PG, ACTQ, true>(acc, bias, actv, lane);
  bias += NB * 16;
}

// z layer: z = W act, written to global as z[n][b][pix][16] (one 1 KiB run per (block, pixel group))
template <int NB, int KQ, int CQ, int PG, int ACTQ, int SLOT_BYTES>
THA4_DEV void z_layer(const char*& gw, char* ring, int& slot, const f32x4* actv, float* zframe, int npix,
                      const int (&pix0)[PG], int wave, int lane) {
  f32x4 acc[NB][PG];
  zero_acc<NB, PG>(acc);
  gemm_stream<NB, KQ, CQ, PG, ACTQ, SLOT_BYTES, 0>(gw, ring, slot, actv, acc, wave, lane);
  const int p = lane & 15, g4 = (lane >> 4) * 4;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int pg = 0; pg < PG; ++pg)
      *reinterpret_cast<f32x4*>(zframe + ((size_t)b * npix + pix0[pg] + p) * 16 + g4) = acc[b][pg];
}

// first layer from position only: act <- sin(30 (wx x + wy y + pb))      (pose folded into pb)
template <int NB, int PG, int ACTQ>
THA4_DEV void first_layer_pos(const float* wx, const float* wy, const float* pb, const float (&x)[PG],
                              const float (&y)[PG], f32x4* actv, int lane) {
  const int g4 = (lane >> 4) * 4;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const f32x4 vx = *reinterpret_cast<const f32x4*>(wx + b * 16 + g4);
    const f32x4 vy = *reinterpret_cast<const f32x4*>(wy + b * 16 + g4);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(pb + b * 16 + g4);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = sin_omega(fmaf(vx[j], x[pg], fmaf(vy[j], y[pg], vb[j])));
      actv[(pg * ACTQ + b) * 64 + lane] = v;
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
template <int NB, int PG, int ACTQ>
THA4_DEV void first_layer_up(const float* zframe, int lowS, const float* wx, const float* wy, const float* pb,
                             const int (&X0)[PG], const int (&Y)[PG], const float (&x)[PG], const float (&y)[PG],
                             f32x4* actv, int lane) {
  const int p = lane & 15, g4 = (lane >> 4) * 4;
  const int npix = lowS * lowS;
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    int x0, x1, y0, y1;
    float lx0, lx1, ly0, ly1;
    up2_taps(X0[pg] + p, lowS, x0, x1, lx0, lx1);
    up2_taps(Y[pg], lowS, y0, y1, ly0, ly1);
    const float* z00 = zframe + ((size_t)y0 * lowS + x0) * 16 + g4;
    const float* z01 = zframe + ((size_t)y0 * lowS + x1) * 16 + g4;
    const float* z10 = zframe + ((size_t)y1 * lowS + x0) * 16 + g4;
    const float* z11 = zframe + ((size_t)y1 * lowS + x1) * 16 + g4;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
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
      actv[(pg * ACTQ + b) * 64 + lane] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// kernel 0: pose-folded first-layer biases.  pb[n][c] = b[c] + sum_k Wpose[c][k] pose[n][k]
// grid (ceil(kPbStride/256), B)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) posebias_kernel(StudentDev d) {
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  const int n = blockIdx.y;
  if (idx >= kPbStride) return;
  int net, c;
  if (idx < kPbL0) { net = 0; c = idx - kPbFace; }
  else if (idx < kPbL1) { net = 1; c = idx - kPbL0; }
  else if (idx < kPbL2) { net = 2; c = idx - kPbL1; }
  else { net = 3; c = idx - kPbL2; }
  const int P = (net == 0) ? kFacePose : kPose;
  const float* w = d.wpose[net] + (size_t)c * P;
  const float* pose = d.pose + (size_t)n * kPose;
  float s = d.bias1[net][c];
  for (int k = 0; k < P; ++k) s = fmaf(w[k], pose[k], s);
  d.pbias[(size_t)n * kPbStride + idx] = s;
}

// ---------------------------------------------------------------------------------------------
// LDS carve: [ring slot 0][ring slot 1][act wave 0]..[act wave 3]
// ---------------------------------------------------------------------------------------------
template <int SLOT_BYTES, int PG, int ACTQ>
struct LdsPlan {
  static constexpr int kActBytesPerWave = PG * ACTQ * 1024;
  static constexpr int kBytes = 2 * SLOT_BYTES + kWaves * kActBytesPerWave;
  static_assert(kBytes <= 160 * 1024, "LDS budget exceeded");
};

// ---------------------------------------------------------------------------------------------
// kernel 1: face morpher.  Each wave owns PG strips of 16 px of the 128x128 face image.
// ---------------------------------------------------------------------------------------------
template <int PG, int CQ>
struct FaceCfg {
  static constexpr int kSlot = (CQ * kNBF > 8 ? CQ * kNBF : 8) * 1024;
  using Lds = LdsPlan<kSlot, PG, kNBF>;
};

template <int PG, int CQ>
__global__ void __launch_bounds__(kBlock) face_kernel(StudentDev d) {
  using Cfg = FaceCfg<PG, CQ>;
  constexpr int SLOT = Cfg::kSlot;
  constexpr int S = kFaceSize, NPIX = S * S, PGS = NPIX / 16;
  THA4_DYN_LDS(smem);
  const int lane = threadIdx.x & 63;
  const int wave = uniform_i32(threadIdx.x >> 6);
  char* ring = smem;
  f32x4* actv = reinterpret_cast<f32x4*>(smem + 2 * SLOT + wave * Cfg::Lds::kActBytesPerWave);

  const int pg_first = (blockIdx.x * kWaves + wave) * PG;     // global pixel-group id (over the batch)
  const int n = pg_first / PGS;                                  // PGS % (kWaves*PG) == 0: one frame per wave
  int pix0[PG];
  float px[PG], py[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    pix0[pg] = ((pg_first + pg) % PGS) * 16;
    px[pg] = d.pos128[(pix0[pg] % S) + (lane & 15)];
    py[pg] = d.pos128[pix0[pg] / S];
  }
  const char* gw = reinterpret_cast<const char*>(d.w_face);
  const float* bias = d.b_face;
  int slot = 0;
  fetch_pieces<CQ * kNBF>(gw, ring, wave, lane);
  first_layer_pos<kNBF, PG, kNBF>(d.wx[0], d.wy[0], d.pbias + (size_t)n * kPbStride + kPbFace, px, py, actv, lane);
  __syncthreads();
#pragma unroll 1
  for (int l = 0; l < 6; ++l)
    sine_layer<kNBF, kNBF, CQ, PG, kNBF, SLOT, CQ * kNBF>(gw, bias, ring, slot, actv, wave, lane);
  sine_layer<kNBF, kNBF, CQ, PG, kNBF, SLOT, 8>(gw, bias, ring, slot, actv, wave, lane);
  // last_linear 128 -> 4 (no nonlinearity, siren.py:87-91): one block, rows 0..3 live in lane group 0
  f32x4 acc[1][PG];
  zero_acc<1, PG>(acc);
  gemm_stream<1, kNBF, kNBF, PG, kNBF, SLOT, 0>(gw, ring, slot, actv, acc, wave, lane);
  if (lane < 16) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias);
    float* fo = d.face + (size_t)n * 4 * NPIX;
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
      const f32x4 v = acc[0][pg] + bb;
#pragma unroll
      for (int j = 0; j < 4; ++j) fo[(size_t)j * NPIX + pix0[pg] + lane] = v[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// kernel 2: body level 0 (128x128): 47->360 (VALU, pose folded) ->360 ->180, then z1 = W10[:, :180] h0
// ---------------------------------------------------------------------------------------------
template <int PG, int CQA, int CQB>
struct Level0Cfg {   // CQA: chunk of the 23-quad layers (23 is prime: 1 or 23); CQB: chunk of the 12-quad z layer
  static constexpr int kP1 = CQA * kNB0, kP2 = CQA * kNB1, kP3 = CQB * kNB1;
  static constexpr int kMax12 = kP1 > kP2 ? kP1 : kP2;
  static constexpr int kSlot = (kMax12 > kP3 ? kMax12 : kP3) * 1024;
  using Lds = LdsPlan<kSlot, PG, kNB0>;
};

template <int PG, int CQA, int CQB>
__global__ void __launch_bounds__(kBlock) level0_kernel(StudentDev d) {
  using Cfg = Level0Cfg<PG, CQA, CQB>;
  constexpr int SLOT = Cfg::kSlot;
  constexpr int S = 128, NPIX = S * S, PGS = NPIX / 16;
  THA4_DYN_LDS(smem);
  const int lane = threadIdx.x & 63;
  const int wave = uniform_i32(threadIdx.x >> 6);
  char* ring = smem;
  f32x4* actv = reinterpret_cast<f32x4*>(smem + 2 * SLOT + wave * Cfg::Lds::kActBytesPerWave);
  const int pg_first = (blockIdx.x * kWaves + wave) * PG;
  const int n = pg_first / PGS;
  int pix0[PG];
  float px[PG], py[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    pix0[pg] = ((pg_first + pg) % PGS) * 16;
    px[pg] = d.pos128[(pix0[pg] % S) + (lane & 15)];
    py[pg] = d.pos128[pix0[pg] / S];
  }
  const char* gw = reinterpret_cast<const char*>(d.w_l0);
  const float* bias = d.b_l0;
  int slot = 0;
  fetch_pieces<Cfg::kP1>(gw, ring, wave, lane);
  first_layer_pos<kNB0, PG, kNB0>(d.wx[1], d.wy[1], d.pbias + (size_t)n * kPbStride + kPbL0, px, py, actv, lane);
  __syncthreads();
  sine_layer<kNB0, kNB0, CQA, PG, kNB0, SLOT, Cfg::kP2>(gw, bias, ring, slot, actv, wave, lane);
  sine_layer<kNB1, kNB0, CQA, PG, kNB0, SLOT, Cfg::kP3>(gw, bias, ring, slot, actv, wave, lane);
  z_layer<kNB1, kNB1, CQB, PG, kNB0, SLOT>(gw, ring, slot, actv, d.z1 + (size_t)n * kNB1 * NPIX * 16, NPIX, pix0, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// kernel 3: body level 1 (256x256): up2(z1)+pos+pose -> sin ->180 ->90, then z2 = W20[:, :90] h1
// ---------------------------------------------------------------------------------------------
template <int PG, int CQA, int CQB>
struct Level1Cfg {   // CQA: chunk of the 12-quad layers; CQB: chunk of the 6-quad z layer
  static constexpr int kP1 = CQA * kNB1, kP2 = CQA * kNB2, kP3 = CQB * kNB2;
  static constexpr int kMax12 = kP1 > kP2 ? kP1 : kP2;
  static constexpr int kSlot = (kMax12 > kP3 ? kMax12 : kP3) * 1024;
  using Lds = LdsPlan<kSlot, PG, kNB1>;
};

template <int PG, int CQA, int CQB>
__global__ void __launch_bounds__(kBlock) level1_kernel(StudentDev d) {
  using Cfg = Level1Cfg<PG, CQA, CQB>;
  constexpr int SLOT = Cfg::kSlot;
  constexpr int S = 256, NPIX = S * S, PGS = NPIX / 16;
  THA4_DYN_LDS(smem);
  const int lane = threadIdx.x & 63;
  const int wave = uniform_i32(threadIdx.x >> 6);
  char* ring = smem;
  f32x4* actv = reinterpret_cast<f32x4*>(smem + 2 * SLOT + wave * Cfg::Lds::kActBytesPerWave);
  const int pg_first = (blockIdx.x * kWaves + wave) * PG;
  const int n = pg_first / PGS;
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    pix0[pg] = ((pg_first + pg) % PGS) * 16;
    X0[pg] = pix0[pg] % S;
    Y[pg] = pix0[pg] / S;
    px[pg] = d.pos256[X0[pg] + (lane & 15)];
    py[pg] = d.pos256[Y[pg]];
  }
  const char* gw = reinterpret_cast<const char*>(d.w_l1);
  const float* bias = d.b_l1;
  int slot = 0;
  fetch_pieces<Cfg::kP1>(gw, ring, wave, lane);
  first_layer_up<kNB1, PG, kNB1>(d.z1 + (size_t)n * kNB1 * (128 * 128) * 16, 128, d.wx[2], d.wy[2],
                                 d.pbias + (size_t)n * kPbStride + kPbL1, X0, Y, px, py, actv, lane);
  __syncthreads();
  sine_layer<kNB1, kNB1, CQA, PG, kNB1, SLOT, Cfg::kP2>(gw, bias, ring, slot, actv, wave, lane);
  sine_layer<kNB2, kNB1, CQA, PG, kNB1, SLOT, Cfg::kP3>(gw, bias, ring, slot, actv, wave, lane);
  z_layer<kNB2, kNB2, CQB, PG, kNB1, SLOT>(gw, ring, slot, actv, d.z2 + (size_t)n * kNB2 * NPIX * 16, NPIX, pix0, wave, lane);
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

template <int PG, int CQ>
struct Level2Cfg {
  static constexpr int kSlot = (CQ * kNB2 > kNB2 ? CQ * kNB2 : kNB2) * 1024;
  using Lds = LdsPlan<kSlot, PG, kNB2>;
};

template <int PG, int CQ>
__global__ void __launch_bounds__(kBlock) level2_kernel(StudentDev d) {
  using Cfg = Level2Cfg<PG, CQ>;
  constexpr int SLOT = Cfg::kSlot;
  constexpr int S = kImg, NPIX = S * S, PGS = NPIX / 16;
  THA4_DYN_LDS(smem);
  const int lane = threadIdx.x & 63;
  const int wave = uniform_i32(threadIdx.x >> 6);
  char* ring = smem;
  f32x4* actv = reinterpret_cast<f32x4*>(smem + 2 * SLOT + wave * Cfg::Lds::kActBytesPerWave);
  const int pg_first = (blockIdx.x * kWaves + wave) * PG;
  const int n = pg_first / PGS;
  int pix0[PG], X0[PG], Y[PG];
  float px[PG], py[PG];
#pragma unroll
  for (int pg = 0; pg < PG; ++pg) {
    pix0[pg] = ((pg_first + pg) % PGS) * 16;
    X0[pg] = pix0[pg] % S;
    Y[pg] = pix0[pg] / S;
    px[pg] = d.pos512[X0[pg] + (lane & 15)];
    py[pg] = d.pos512[Y[pg]];
  }
  const char* gw = reinterpret_cast<const char*>(d.w_l2);
  const float* bias = d.b_l2;
  int slot = 0;
  fetch_pieces<CQ * kNB2>(gw, ring, wave, lane);
  first_layer_up<kNB2, PG, kNB2>(d.z2 + (size_t)n * kNB2 * (256 * 256) * 16, 256, d.wx[3], d.wy[3],
                                 d.pbias + (size_t)n * kPbStride + kPbL2, X0, Y, px, py, actv, lane);
  __syncthreads();
  sine_layer<kNB2, kNB2, CQ, PG, kNB2, SLOT, CQ * kNB2>(gw, bias, ring, slot, actv, wave, lane);
  sine_layer<kNB2, kNB2, CQ, PG, kNB2, SLOT, kNB2>(gw, bias, ring, slot, actv, wave, lane);
  // last_linear 90 -> 7: rows 0..3 (dx, dy, alpha, colour R) in lane group 0, rows 4..6 (G, B, A) in group 1
  f32x4 acc[1][PG];
  zero_acc<1, PG>(acc);
  gemm_stream<1, kNB2, kNB2, PG, kNB2, SLOT, 0>(gw, ring, slot, actv, acc, wave, lane);

  const int p = lane & 15, g = lane >> 4;
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
    float w = body_source(img, face, g, y0, x0) * wnw;
    w += body_source(img, face, g, y0, x1) * wne;
    w += body_source(img, face, g, y1, x0) * wsw;
    w += body_source(img, face, g, y1, x1) * wse;
    const float blended = (1.0f - al) * w + al * col;            // siren_morpher_03.py:131
    const size_t pix = (size_t)pix0[pg] + p;
    d.out_blended[((size_t)n * 4 + g) * NPIX + pix] = blended;
    if (d.out_color) d.out_color[((size_t)n * 4 + g) * NPIX + pix] = col;
    if (d.out_warped) d.out_warped[((size_t)n * 4 + g) * NPIX + pix] = w;
    if (d.out_alpha && g == 0) d.out_alpha[(size_t)n * NPIX + pix] = al;
    if (d.out_grid && g < 2) d.out_grid[((size_t)n * 2 + g) * NPIX + pix] = (g == 0 ? dx : dy);
  }
}

}  // namespace tha4

// ---------------------------------------------------------------------------------------------
// launch configuration shared by the C-ABI launcher and the CPU emulator tests
// ---------------------------------------------------------------------------------------------
namespace tha4 {
namespace cfg {
// pixel groups (16 px) per wave / K quads per streamed chunk, per kernel
constexpr int kFacePG = 1, kFaceCQ = 2;
constexpr int kL0PG = 1, kL0CQA = 1, kL0CQB = 2;
constexpr int kL1PG = 2, kL1CQA = 2, kL1CQB = 3;
constexpr int kL2PG = 4, kL2CQ = 3;

constexpr int kFaceLds = FaceCfg<kFacePG, kFaceCQ>::Lds::kBytes;
constexpr int kL0Lds = Level0Cfg<kL0PG, kL0CQA, kL0CQB>::Lds::kBytes;
constexpr int kL1Lds = Level1Cfg<kL1PG, kL1CQA, kL1CQB>::Lds::kBytes;
constexpr int kL2Lds = Level2Cfg<kL2PG, kL2CQ>::Lds::kBytes;

constexpr int blocks_for(int batch, int side, int pg) { return batch * (side * side / 16) / (kWaves * pg); }
}  // namespace cfg
}  // namespace tha4
