// Data layout of the distilled-student (mode_14) parameters in HBM, and the host-side packer.
//
// Reference architecture being packed (citations relative to /root/reference/src/tha4):
//   poser/modes/mode_14.py:93-131   face SIREN 41->128 (x8 sine layers)->4 @128^2;
//                                   body SIREN levels 47->360->360->180 @128^2,
//                                   227->180->180->90 @256^2, 137->90->90->90 @512^2, head 90->7
//   nn/siren/vanilla/siren.py:23-39 every layer is Conv2d(k=1) + sin(30*x)
//   SURVEY.md Appendix B            state_dict key layout / input channel order
//
// Everything here is plain C++ (no HIP types) so the CPU unit tests can use it as well.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace tha4 {

// ---- problem constants ---------------------------------------------------------------------
constexpr int kPose = 45;
constexpr int kFacePose = 39;      // mode_14.py:66
constexpr int kImg = 512;
constexpr int kFaceSize = 128;
constexpr int kFaceTop = 80;       // mode_14.py:61-63: centre (256,144) +- 64
constexpr int kFaceLeft = 192;
constexpr float kOmega = 30.0f;    // siren.py:17
// Generation 2 measures sine arguments in TURNS: the packed weights, biases, first-layer tables and z hand-off of every layer
// that feeds a sine carry omega_0 / (2 pi) instead of omega_0, so that sin(30 (W x + b)) is ONE v_sin_f32 on the accumulator
// (the instruction takes revolutions and reduces the argument itself: fract() of an fp32 value is exact).  Round 2 carried 30 and
// spent 12 VALU slots per sine on k = rint(u / pi), a 2-term Cody-Waite reduction and a degree-9 polynomial; VALU time is
// additive to MFMA time on this chip (profiles/r03_mfma_valu_overlap2.txt) and the student evaluates 131.8 M sines per frame.
// fl32(30 / 2 pi) * 2 pi = 30 (1 - 2.2e-8): the posed frame moves by 3-4e-5 (fp64 simulation, 3 poses), a tenth of the distance
// between two correct fp32 evaluations of the reference.  -DTHA4_SIN_TURNS=0 restores the radian pipeline (A/B builds).
#ifndef THA4_SIN_TURNS
#define THA4_SIN_TURNS 1
#endif
constexpr float kSineTurns = 4.774648292756860f;     // 30 / (2 pi)
constexpr float kSineScale16 = THA4_SIN_TURNS ? kSineTurns : kOmega;

// z hand-off image between levels: z[n][block][g][pixel][4] - rows 4g..4g+3 of a 16-row block are what lane group g
// of an MFMA C/D fragment holds, so a quarter wave (fixed g, 16 consecutive pixels) writes one 256 B run and the x2
// upsample taps of the next level read a contiguous ~144 B span per quarter wave instead of sixteen 64 B-strided pieces.
constexpr size_t z_offset(int block, int g, int pixel, int npix) {
  return (((size_t)block * 4 + g) * npix + pixel) * 4;
}

// channel widths and their padding to 16-row MFMA blocks ("blocks") / 16-channel K groups ("quads")
constexpr int kCF = 128, kNBF = 8;    // face hidden width
constexpr int kC0 = 360, kNB0 = 24;   // body level 0 hidden width: 23 blocks, padded to 24 so two waves can split the rows
constexpr int kKQ0 = 23;              // ... but only 23 input quads (368 channels) are ever contracted
constexpr int kC1 = 180, kNB1 = 12;   // level 1 (192 padded)
constexpr int kC2 = 90, kNB2 = 6;     // level 2 (96 padded)
constexpr int kHeadC = 7;             // grid dx, dy, alpha, colour RGBA (siren_morpher_03.py:127-129)

// per-frame pose-folded first-layer bias vector: [face | L0 | L1 | L2], zero padded
constexpr int kPbFace = 0;
constexpr int kPbL0 = kPbFace + kNBF * 16;
constexpr int kPbL1 = kPbL0 + kNB0 * 16;
constexpr int kPbL2 = kPbL1 + kNB1 * 16;
constexpr int kPbStride = kPbL2 + kNB2 * 16;   // 800 floats per frame

// ---- MFMA-fragment-linear weight image --------------------------------------------------------
// One linear layer  y[o] = sum_i W[o][i] x[i]  with O outputs (NB = ceil(O/16) blocks) and I inputs
// (KQ = ceil(I/16) quads) is stored as  P[q][b][lane][j]  (q<KQ, b<NB, lane<64, j<4):
//     P = W[16*b + (lane & 15)][16*q + 4*(lane >> 4) + j]        (0 outside the matrix)
// i.e. exactly the A operand of four consecutive v_mfma_f32_16x16x4_f32 steps of lane `lane`
// (A[i = lane&15][k = lane>>4]), with the K index permuted so that k-step 4q+j of lane group
// g = lane>>4 contracts input channel 16q+4g+j.  Activations use the mirrored image
// X[q][lane][j] = x[channel 16q+4g+j][pixel lane&15], which is also the C/D fragment layout of
// the previous layer (row = 4g+j of block q, col = pixel) - so a layer's output block b IS the
// next layer's input quad b, and both operands are read with one conflict-free ds_read_b128.
// One (q,b) piece is 1 KiB; a global->LDS copy of any run of pieces is a linear memcpy.
inline size_t packed_floats(int NB, int KQ) { return (size_t)NB * KQ * 256; }

inline void pack_layer(const float* W, int ldw, int col0, int O, int I, int NB, int KQ, float* dst) {
  for (int q = 0; q < KQ; ++q)
    for (int b = 0; b < NB; ++b)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          int o = 16 * b + (lane & 15);
          int i = 16 * q + 4 * (lane >> 4) + j;
          float v = (o < O && i < I) ? W[(size_t)o * ldw + col0 + i] : 0.0f;
          dst[(((size_t)q * NB + b) * 64 + lane) * 4 + j] = v;
        }
}

// ---- host view of the reference weights (what the C-ABI receives) ---------------------------
struct LinearView {       // Conv2d(k=1): weight [out][in] row-major, bias [out]
  const float* weight;
  const float* bias;
  int out_ch;
  int in_ch;
};

struct StudentWeightsView {
  LinearView face_sine[8];     // siren.sine_layers.{0..7}.linear
  LinearView face_last;        // siren.last_linear
  LinearView body_sine[3][3];  // siren_layers.{L}.{j}.linear
  LinearView body_last;        // last_linear
};

// First layer of a SIREN stack: input = [features(F) | x | y | pose(P)]  (SURVEY.md App. B).
// Split into: feature part (packed as an MFMA layer executed by the PREVIOUS level at low
// resolution), position columns wx/wy, and pose columns + bias (folded per frame by the
// pose-bias kernel).
struct FirstLayerPack {
  std::vector<float> wx, wy, bias;   // [NB*16]
  std::vector<float> wpose;          // [kPose][NB*16]  (pose-major: coalesced across output channels; rows >= P are zero)
  int P = 0;
};

inline FirstLayerPack pack_first(const LinearView& l, int F, int P, int NB) {
  FirstLayerPack r;
  r.P = kPose;   // rows beyond P stay zero: the pose-bias kernel always contracts all 45 pose slots
  r.wx.assign(NB * 16, 0.f);
  r.wy.assign(NB * 16, 0.f);
  r.bias.assign(NB * 16, 0.f);
  r.wpose.assign((size_t)NB * 16 * kPose, 0.f);
  for (int o = 0; o < l.out_ch; ++o) {
    const float* row = l.weight + (size_t)o * l.in_ch;
    r.wx[o] = row[F];
    r.wy[o] = row[F + 1];
    r.bias[o] = l.bias[o];
    for (int k = 0; k < P; ++k) r.wpose[(size_t)k * NB * 16 + o] = row[F + 2 + k];
  }
  return r;
}

inline std::vector<float> pad_bias(const LinearView& l, int NB) {
  std::vector<float> b(NB * 16, 0.f);
  for (int o = 0; o < l.out_ch; ++o) b[o] = l.bias[o];
  return b;
}

// Everything the four SIREN kernels read, as host vectors ready for one hipMemcpy each.
struct StudentPacked {
  // weight streams: the layers of one kernel back to back, in execution order
  std::vector<float> w_face;   // 7 x [8x8] + [1x8]
  std::vector<float> w_l0;     // [24x23] [12x23] + z1 layer [12x12]   (blocks x quads)
  std::vector<float> w_l1;     // [12x12] [6x12]  + z2 layer [6x6]
  std::vector<float> w_l2;     // [6x6] [6x6] + head [1x6]
  // biases of the streamed layers, concatenated in the same order (padded to blocks)
  std::vector<float> b_face;   // 7*128 + 16
  std::vector<float> b_l0;     // 384 + 192          (z layers carry no bias)
  std::vector<float> b_l1;     // 192 + 96
  std::vector<float> b_l2;     // 96 + 96 + 16
  FirstLayerPack f_face, f_l0, f_l1, f_l2;
};

inline bool check_dims(const LinearView& l, int out, int in) { return l.out_ch == out && l.in_ch == in && l.weight && l.bias; }

// Returns "" on success, otherwise a description of the first mismatch with mode_14's architecture.
inline std::string pack_student(const StudentWeightsView& v, StudentPacked& p) {
  if (!check_dims(v.face_sine[0], kCF, 2 + kFacePose)) return "face sine layer 0 must be 41->128";
  for (int i = 1; i < 8; ++i)
    if (!check_dims(v.face_sine[i], kCF, kCF)) return "face sine layers 1..7 must be 128->128";
  if (!check_dims(v.face_last, 4, kCF)) return "face last_linear must be 128->4";
  const int dims[3][3][2] = {{{47, 360}, {360, 360}, {360, 180}},
                             {{227, 180}, {180, 180}, {180, 90}},
                             {{137, 90}, {90, 90}, {90, 90}}};
  for (int l = 0; l < 3; ++l)
    for (int j = 0; j < 3; ++j)
      if (!check_dims(v.body_sine[l][j], dims[l][j][1], dims[l][j][0]))
        return "body sine layer " + std::to_string(l) + "." + std::to_string(j) + " has unexpected shape";
  if (!check_dims(v.body_last, kHeadC, kC2)) return "body last_linear must be 90->7";

  auto append = [](std::vector<float>& dst, const LinearView& l, int col0, int I, int NB, int KQ) {
    size_t at = dst.size();
    dst.resize(at + packed_floats(NB, KQ));
    pack_layer(l.weight, l.in_ch, col0, l.out_ch, I, NB, KQ, dst.data() + at);
  };
  auto append_bias = [](std::vector<float>& dst, const LinearView& l, int NB) {
    auto b = pad_bias(l, NB);
    dst.insert(dst.end(), b.begin(), b.end());
  };

  p = StudentPacked();
  // face
  p.f_face = pack_first(v.face_sine[0], 0, kFacePose, kNBF);
  for (int i = 1; i < 8; ++i) {
    append(p.w_face, v.face_sine[i], 0, kCF, kNBF, kNBF);
    append_bias(p.b_face, v.face_sine[i], kNBF);
  }
  append(p.w_face, v.face_last, 0, kCF, 1, kNBF);
  append_bias(p.b_face, v.face_last, 1);
  // level 0
  p.f_l0 = pack_first(v.body_sine[0][0], 0, kPose, kNB0);
  append(p.w_l0, v.body_sine[0][1], 0, kC0, kNB0, kKQ0);
  append_bias(p.b_l0, v.body_sine[0][1], kNB0);
  append(p.w_l0, v.body_sine[0][2], 0, kC0, kNB1, kKQ0);
  append_bias(p.b_l0, v.body_sine[0][2], kNB1);
  append(p.w_l0, v.body_sine[1][0], 0, kC1, kNB1, kNB1);   // z1 = W_{1,0}[:, 0:180] h0  (no bias)
  // level 1
  p.f_l1 = pack_first(v.body_sine[1][0], kC1, kPose, kNB1);
  append(p.w_l1, v.body_sine[1][1], 0, kC1, kNB1, kNB1);
  append_bias(p.b_l1, v.body_sine[1][1], kNB1);
  append(p.w_l1, v.body_sine[1][2], 0, kC1, kNB2, kNB1);
  append_bias(p.b_l1, v.body_sine[1][2], kNB2);
  append(p.w_l1, v.body_sine[2][0], 0, kC2, kNB2, kNB2);   // z2 = W_{2,0}[:, 0:90] h1
  // level 2
  p.f_l2 = pack_first(v.body_sine[2][0], kC2, kPose, kNB2);
  append(p.w_l2, v.body_sine[2][1], 0, kC2, kNB2, kNB2);
  append_bias(p.b_l2, v.body_sine[2][1], kNB2);
  append(p.w_l2, v.body_sine[2][2], 0, kC2, kNB2, kNB2);
  append_bias(p.b_l2, v.body_sine[2][2], kNB2);
  append(p.w_l2, v.body_last, 0, kC2, 1, kNB2);
  append_bias(p.b_l2, v.body_last, 1);
  return "";
}

// Upper bound, in TURNS, of |omega_0 (W x + b)| / (2 pi) over every sine layer of the student for inputs inside their ranges -
// hidden activations are sine outputs (|x| <= 1), positions |x|, |y| < 1, pose parameters |p| <= 1 (pose_parameters.py:4-36), the
// upsampled features of levels 1 / 2 are convex combinations of sine outputs - i.e. max over rows of c (sum_j |W_ij| + |b_i|).
// The default kernels evaluate the sine with ONE v_sin_f32 on the argument in turns, which returns 0 beyond 256 turns where the
// reference's torch.sin accepts any argument: tha4_student_create / _set_weights refuse weights whose bound reaches the limit
// (the shipped students stay below 7 turns) and point at THA4_STUDENT_EXACT_FP32, whose radian pipeline reduces |u| < 12868 rad.
constexpr double kSineTurnsLimit = 256.0;
inline double sine_argument_bound_turns(const StudentWeightsView& v) {
  const double c = 30.0 / 6.283185307179586476925;
  double worst = 0.0;
  bool nan_seen = false;                                // sticky: a NaN row anywhere makes the bound NaN (reported as over the limit),
  auto layer = [&](const LinearView& l) {               // whatever rows follow it
    for (int o = 0; o < l.out_ch; ++o) {
      double s = std::fabs((double)l.bias[o]);
      for (int i = 0; i < l.in_ch; ++i) s += std::fabs((double)l.weight[(size_t)o * l.in_ch + i]);
      if (std::isnan(s)) nan_seen = true;
      else if (s * c > worst) worst = s * c;
    }
  };
  for (int i = 0; i < 8; ++i) layer(v.face_sine[i]);
  for (int l = 0; l < 3; ++l)
    for (int j = 0; j < 3; ++j) layer(v.body_sine[l][j]);
  return nan_seen ? std::nan("") : worst;
}

// exact affine_grid(identity, align_corners=False) axis: x_j = (2j+1)/S - 1 (dyadic, exact in fp32)
inline void exact_position_axis(int S, float* dst) {
  for (int j = 0; j < S; ++j) dst[j] = (float)((2.0 * j + 1.0) / S - 1.0);
}

}  // namespace tha4
