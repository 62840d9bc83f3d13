// Host-side layout helpers for the full-model kernels (plain C++, shared with the CPU unit tests).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace tha4 {

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
