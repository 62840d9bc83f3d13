// Image-domain kernels of the full THA4 pipeline (reference mode_07): crops / pastes, bilinear resizes,
// grid_sample warps and alpha blends between the five networks, plus the per-network "tails" that turn
// a head feature map (C16, one block) into the reference's NCHW outputs.  All pointwise, HBM-bound and
// tiny next to the convolutions: one thread per output pixel, coalesced along x.
//
// Reference (paths relative to /root/reference/src/tha4):
//   poser/modes/mode_07.py:72-118                      crops, pastes, resizes between the networks
//   nn/image_processing_util.py:6-24,33-58             apply_rgb_change / apply_grid_change / apply_color_change
//   nn/eyebrow_decomposer/eyebrow_decomposer_00.py:46-64, nn/eyebrow_morphing_combiner/..._00.py:47-72
//   nn/face_morpher/face_morpher_08.py:158-193, nn/morpher/morpher_00.py:42-66, nn/upscaler/upscaler_02.py:59-96
#pragma once
#include "tha4_platform.h"
#include "image_io_kernels.h"

namespace tha4 {

// grid_sample(bilinear, padding_mode='border', align_corners=False) of channel-planar image [C][H][W]
// at normalised coords (gx, gy); returns channel c.
THA4_DEV float sample_border(const float* img, int H, int W, int c, float gx, float gy) {
  float ix = ((gx + 1.0f) * (float)W - 1.0f) * 0.5f;
  float iy = ((gy + 1.0f) * (float)H - 1.0f) * 0.5f;
  ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float tx = ix - fx0, ty = iy - fy0;
  const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  const float* pc = img + (size_t)c * H * W;
  float v = pc[(size_t)y0 * W + x0] * ((1.0f - tx) * (1.0f - ty));
  v += pc[(size_t)y0 * W + x1] * (tx * (1.0f - ty));
  v += pc[(size_t)y1 * W + x0] * ((1.0f - tx) * ty);
  v += pc[(size_t)y1 * W + x1] * (tx * ty);
  return v;
}

THA4_DEV float axis_pos(int j, int S) { return (2.0f * (float)j + 1.0f) / (float)S - 1.0f; }   // affine_grid(identity), align_corners=False

// bilinear x2 upsample tap (F.interpolate align_corners=False), see siren_kernels.h up2_taps
THA4_DEV float up2_sample(const float* plane, int lowS, int X, int Y) {
  const float sx = fmaxf(0.0f, (X + 0.5f) * 0.5f - 0.5f), sy = fmaxf(0.0f, (Y + 0.5f) * 0.5f - 0.5f);
  const int x0 = (int)sx, y0 = (int)sy;
  const int x1 = min(x0 + 1, lowS - 1), y1 = min(y0 + 1, lowS - 1);
  const float lx1 = sx - (float)x0, ly1 = sy - (float)y0, lx0 = 1.0f - lx1, ly0 = 1.0f - ly1;
  return ly0 * (lx0 * plane[(size_t)y0 * lowS + x0] + lx1 * plane[(size_t)y0 * lowS + x1]) +
         ly1 * (lx0 * plane[(size_t)y1 * lowS + x0] + lx1 * plane[(size_t)y1 * lowS + x1]);
}

struct ImgArgs {
  const float* image;       // [B][4][512][512]
  long long image_stride;
  const float* pose;        // [B][45]
  // per-stage tensors (device); NCHW outputs are the reference's output list entries
  const float* head;        // C16 [B][1][S*S][16] head block of the current stage
  float* c16_out;           // C16 input of the next network
  float* out[8];            // NCHW outputs of this stage (see each kernel)
  const float* in0;         // stage-specific NCHW inputs
  const float* in1;
  const float* in2;
  int batch;
  int sel;                  // combiner: eyebrow_morphed_image_index (0 or 2, mode_07.py:275)
  int* fault;               // sticky numeric-fault flag of the handle (full_kernels.h report_fault_unless_finite), or null
  unsigned char* rgba8;     // unet_tail_kernel: fused display epilogue of out[0] (tha4_hip.h tha4_display), [B][S*S][4], or null
  int rgba8_has_bg;
  float rgba8_bg[3];
};

// NCHW outputs the caller did not ask for and no later stage reads are null (round 6; tha4_full_pose_ex: Poser.pose() wants ONE of the 33): their stores are
// skipped - a wave-uniform branch on a kernel-argument pointer.  (Rounds 1-5 wrote all 33 into scratch: 11 of the 15 output planes of the upscaler's tail,
// 15.7 MB per frame, for a caller that reads the posed frame only.)
THA4_DEV void put(float* plane, size_t off, float v) {
  if (plane) plane[off] = v;
}
// one pixel's 16 channels of a C16 tensor as four 16-byte stores (a lane's 64 bytes are contiguous: a wave writes 4 KiB runs)
THA4_DEV void put_c16(float* co, const f32x4& v0, const f32x4& v1, const f32x4& v2, const f32x4& v3) {
  reinterpret_cast<f32x4*>(co)[0] = v0;
  reinterpret_cast<f32x4*>(co)[1] = v1;
  reinterpret_cast<f32x4*>(co)[2] = v2;
  reinterpret_cast<f32x4*>(co)[3] = v3;
}

// the head block of a network (16 floats per pixel, padded channels are exact zeros) must be finite: a staged operand beyond
// the fp16 range, or a NaN from the weights, that no normalisation saw on its way here ends up in it
THA4_DEV void check_head_finite(const float* h, int* fault) {
  if (!fault) return;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += fabsf(h[i]);
  if (!(s < __builtin_inff())) *fault = 1;
}

// pose slices padded to C16 vectors: eyebrow pose[0:12] -> [B][16], face pose[12:39] -> [B][32]   (mode_07.py:80,91)
__global__ void __launch_bounds__(64) pose_pad_kernel(const float* pose, float* eyebrow16, float* face32, int batch) {
  const int n = blockIdx.x, i = threadIdx.x;
  if (i < 16) eyebrow16[n * 16 + i] = i < 12 ? pose[n * 45 + i] : 0.0f;
  if (i < 32) face32[n * 32 + i] = i < 27 ? pose[n * 45 + 12 + i] : 0.0f;
}

// stage 1 input: image[:, :, 64:192, 192:320] -> C16 (4 real channels)                                (mode_07.py:74)
__global__ void __launch_bounds__(256) crop_eyebrow_kernel(ImgArgs a) {
  warm_kernarg<(int)sizeof(ImgArgs)>();
  const int S = 128, idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (idx >= S * S) return;
  const int y = idx / S, x = idx % S;
  const float* img = a.image + (size_t)n * a.image_stride;
  float* o = a.c16_out + ((size_t)n * S * S + idx) * 16;
  f32x4 v;
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = img[((size_t)c * 512 + 64 + y) * 512 + 192 + x];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  reinterpret_cast<f32x4*>(o)[0] = v;
  reinterpret_cast<f32x4*>(o)[1] = z;
  reinterpret_cast<f32x4*>(o)[2] = z;
  reinterpret_cast<f32x4*>(o)[3] = z;
}

// stage 1 tail (eyebrow_decomposer_00.py:46-64).  head rows: 0 bg_alpha | 1-4 bg_colour | 5 eb_alpha | 6-9 eb_colour.
// out: 0 eyebrow_layer, 1 eb_alpha, 2 eb_colour, 3 background_layer, 4 bg_alpha, 5 bg_colour  (all [B][C][128][128])
// c16_out: combiner input = cat([background_layer, eyebrow_layer])                      (eyebrow_morphing_combiner_00.py:48)
__global__ void __launch_bounds__(256) decomposer_tail_kernel(ImgArgs a) {
  warm_kernarg<(int)sizeof(ImgArgs)>();
  const int S = 128, P = S * S, idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (idx >= P) return;
  const int y = idx / S, x = idx % S;
  const float* img = a.image + (size_t)n * a.image_stride;
  const float* h = a.head + ((size_t)n * P + idx) * 16;
  check_head_finite(h, a.fault);
  const float bga = h[0], eba = h[5];
  float* co = a.c16_out + ((size_t)n * P + idx) * 16;
  f32x4 vbg, veb;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float im = img[((size_t)c * 512 + 64 + y) * 512 + 192 + x];
    const float bgc = h[1 + c], ebc = h[6 + c];
    const float bg = bgc * bga + im * (1.0f - bga);
    const float eb = im * eba + ebc * (1.0f - eba);      // apply_color_change(alpha, image, colour): swapped (:55)
    vbg[c] = bg;
    veb[c] = eb;
    put(a.out[0], ((size_t)n * 4 + c) * P + idx, eb);
    put(a.out[2], ((size_t)n * 4 + c) * P + idx, ebc);
    put(a.out[3], ((size_t)n * 4 + c) * P + idx, bg);
    put(a.out[5], ((size_t)n * 4 + c) * P + idx, bgc);
  }
  put(a.out[1], (size_t)n * P + idx, eba);
  put(a.out[4], (size_t)n * P + idx, bga);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  put_c16(co, vbg, veb, z, z);
}

// stage 2 tail (eyebrow_morphing_combiner_00.py:47-72).  head rows: 0-1 grid | 2 alpha | 3-6 colour | 7 combine_alpha.
// in0 = eyebrow_layer (dec out 0), in1 = background_layer (dec out 3), both [B][4][128][128].
// out: 0 eyebrow_image, 1 combine_alpha, 2 eyebrow_image_no_combine_alpha, 3 morphed, 4 alpha, 5 colour, 6 warped, 7 grid
__global__ void __launch_bounds__(256) combiner_tail_kernel(ImgArgs a) {
  warm_kernarg<(int)sizeof(ImgArgs)>();
  const int S = 128, P = S * S, idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (idx >= P) return;
  const int y = idx / S, x = idx % S;
  const float* h = a.head + ((size_t)n * P + idx) * 16;
  check_head_finite(h, a.fault);
  const float* eb = a.in0 + (size_t)n * 4 * P;
  const float* bg = a.in1 + (size_t)n * 4 * P;
  const float gxc = h[0], gyc = h[1], al = h[2], ca = h[7];
  const float gx = axis_pos(x, S) + gxc, gy = axis_pos(y, S) + gyc;
  float morphed[4], warped[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    warped[c] = sample_border(eb, S, S, c, gx, gy);
    morphed[c] = h[3 + c] * al + warped[c] * (1.0f - al);
  }
  const float a2 = (morphed[3] + 1.0f) * 0.5f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float b = bg[(size_t)c * P + idx];
    const size_t o = ((size_t)n * 4 + c) * P + idx;
    put(a.out[0], o, c < 3 ? morphed[c] * ca + b * (1.0f - ca) : b);      // apply_rgb_change keeps the image's alpha
    put(a.out[2], o, c < 3 ? morphed[c] * a2 + b * (1.0f - a2) : b);
    put(a.out[3], o, morphed[c]);
    put(a.out[5], o, h[3 + c]);
    put(a.out[6], o, warped[c]);
  }
  put(a.out[1], (size_t)n * P + idx, ca);
  put(a.out[4], (size_t)n * P + idx, al);
  put(a.out[7], ((size_t)n * 2 + 0) * P + idx, gxc);
  put(a.out[7], ((size_t)n * 2 + 1) * P + idx, gyc);
}

// stage 3 input (mode_07.py:85-90): image[:, :, 32:224, 160:352] with [32:160, 32:160] <- combiner output `sel`.
// in0 = that combiner output [B][4][128][128].  out[0] = face input NCHW [B][4][192][192]; c16_out = same in C16.
__global__ void __launch_bounds__(256) face_input_kernel(ImgArgs a) {
  warm_kernarg<(int)sizeof(ImgArgs)>();
  const int S = 192, P = S * S, idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (idx >= P) return;
  const int y = idx / S, x = idx % S;
  const float* img = a.image + (size_t)n * a.image_stride;
  const bool in_eb = (unsigned)(y - 32) < 128u && (unsigned)(x - 32) < 128u;
  float* co = a.c16_out + ((size_t)n * P + idx) * 16;
  f32x4 vv;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float v = in_eb ? a.in0[((size_t)n * 4 + c) * 128 * 128 + (size_t)(y - 32) * 128 + (x - 32)]
                          : img[((size_t)c * 512 + 32 + y) * 512 + 160 + x];
    put(a.out[0], ((size_t)n * 4 + c) * P + idx, v);
    vv[c] = v;
  }
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  put_c16(co, vv, z, z, z);
}

// stage 3 tail (face_morpher_08.py:158-193).  head rows: 0-1 grid | 2-5 iris colour | 6 iris alpha | 7-10 eye colour | 11 eye alpha.
// in0 = face input NCHW [B][4][192][192].
// out: 0 output_image, 1 eye_alpha, 2 eye_colour, 3 iris_mouth_image_1, 4 iris_alpha, 5 iris_colour, 6 iris_mouth_image_0, 7 grid
__global__ void __launch_bounds__(256) face_tail_kernel(ImgArgs a) {
  warm_kernarg<(int)sizeof(ImgArgs)>();
  const int S = 192, P = S * S, idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (idx >= P) return;
  const int y = idx / S, x = idx % S;
  const float* h = a.head + ((size_t)n * P + idx) * 16;
  check_head_finite(h, a.fault);
  const float* fin = a.in0 + (size_t)n * 4 * P;
  const float gxc = h[0], gyc = h[1], ia = h[6], ea = h[11];
  const float gx = axis_pos(x, S) + gxc, gy = axis_pos(y, S) + gyc;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float im0 = sample_border(fin, S, S, c, gx, gy);
    const float icc = h[2 + c], ecc = h[7 + c];
    const float im1 = icc * ia + im0 * (1.0f - ia);
    const float o = ecc * ea + im1 * (1.0f - ea);
    const size_t off = ((size_t)n * 4 + c) * P + idx;
    put(a.out[0], off, o);
    put(a.out[2], off, ecc);
    put(a.out[3], off, im1);
    put(a.out[5], off, icc);
    put(a.out[6], off, im0);
  }
  put(a.out[1], (size_t)n * P + idx, ea);
  put(a.out[4], (size_t)n * P + idx, ia);
  put(a.out[7], ((size_t)n * 2 + 0) * P + idx, gxc);
  put(a.out[7], ((size_t)n * 2 + 1) * P + idx, gyc);
}

// mode_07.py:93-103: face_morphed_full = image with [32:224, 160:352] <- face output (in0, [B][4][192][192]);
// out[0] = face_morphed_full NCHW [B][4][512][512]
__global__ void __launch_bounds__(256) paste_face_kernel(ImgArgs a) {
  warm_kernarg<(int)sizeof(ImgArgs)>();
  const int S = 512, P = S * S, idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (idx >= P) return;
  const int y = idx / S, x = idx % S;
  const float* img = a.image + (size_t)n * a.image_stride;
  const bool in_face = (unsigned)(y - 32) < 192u && (unsigned)(x - 160) < 192u;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float v = in_face ? a.in0[((size_t)n * 4 + c) * 192 * 192 + (size_t)(y - 32) * 192 + (x - 160)]
                            : img[((size_t)c * S + y) * S + x];
    put(a.out[0], ((size_t)n * 4 + c) * P + idx, v);
  }
}

// face_morphed_half = bilinear 512 -> 256, align_corners=False (= mean of the 2x2 window).  in0 = full NCHW.
// out[0] = half NCHW [B][4][256][256], c16_out = half in C16 (body morpher input).
__global__ void __launch_bounds__(256) half_image_kernel(ImgArgs a) {
  warm_kernarg<(int)sizeof(ImgArgs)>();
  const int S = 256, P = S * S, idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (idx >= P) return;
  const int y = idx / S, x = idx % S;
  float* co = a.c16_out + ((size_t)n * P + idx) * 16;
  f32x4 vv;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float* pl = a.in0 + ((size_t)n * 4 + c) * 512 * 512;
    const float t0 = 0.5f * pl[(size_t)(2 * y) * 512 + 2 * x] + 0.5f * pl[(size_t)(2 * y) * 512 + 2 * x + 1];
    const float t1 = 0.5f * pl[(size_t)(2 * y + 1) * 512 + 2 * x] + 0.5f * pl[(size_t)(2 * y + 1) * 512 + 2 * x + 1];
    const float v = 0.5f * t0 + 0.5f * t1;
    put(a.out[0], ((size_t)n * 4 + c) * P + idx, v);
    vv[c] = v;
  }
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  put_c16(co, vv, z, z, z);
}

// U-Net tail of the body morpher (S=256) and the upscaler (S=512) (morpher_00.py:53-66, upscaler_02.py:85-96).
// head rows: 0-3 direct | 4-5 grid | 6 alpha (pre-sigmoid).  in0 = the image being warped, NCHW [B][4][S][S].
// out: 0 merged, 1 alpha, 2 warped, 3 grid, 4 direct
template <int S>
__global__ void __launch_bounds__(256) unet_tail_kernel(ImgArgs a) {
  warm_kernarg<(int)sizeof(ImgArgs)>();
  const int P = S * S, idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (idx >= P) return;
  const int y = idx / S, x = idx % S;
  const float* h = a.head + ((size_t)n * P + idx) * 16;
  check_head_finite(h, a.fault);
  const float* src = a.in0 + (size_t)n * 4 * P;
  const float gxc = h[4], gyc = h[5];
  const float al = 1.0f / (1.0f + expf(-h[6]));
  const float gx = axis_pos(x, S) + gxc, gy = axis_pos(y, S) + gyc;
  float merged[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float w = sample_border(src, S, S, c, gx, gy);
    const float d = h[c];
    const size_t off = ((size_t)n * 4 + c) * P + idx;
    merged[c] = d * al + w * (1.0f - al);
    put(a.out[0], off, merged[c]);
    put(a.out[2], off, w);
    put(a.out[4], off, d);
  }
  if (a.rgba8) {            // display epilogue on the values in registers: one packed 4-byte store per pixel
    const float a01 = fminf(fmaxf((merged[3] + 1.0f) * 0.5f, 0.0f), 1.0f);
    uchar4 o;
    o.x = display_channel(merged[0], 0, a01, a.rgba8_has_bg != 0, a.rgba8_bg[0]);
    o.y = display_channel(merged[1], 1, a01, a.rgba8_has_bg != 0, a.rgba8_bg[1]);
    o.z = display_channel(merged[2], 2, a01, a.rgba8_has_bg != 0, a.rgba8_bg[2]);
    o.w = display_channel(merged[3], 3, a01, a.rgba8_has_bg != 0, 0.0f);
    reinterpret_cast<uchar4*>(a.rgba8)[(size_t)n * P + idx] = o;
  }
  put(a.out[1], (size_t)n * P + idx, al);
  put(a.out[3], ((size_t)n * 2 + 0) * P + idx, gxc);
  put(a.out[3], ((size_t)n * 2 + 1) * P + idx, gyc);
}

// upscaler input (mode_07.py:108-118, upscaler_02.py:78-83): in0 = rest (face_morphed_full) NCHW 512^2,
// in1 = body merged [B][4][256][256], in2 = body grid [B][2][256][256].
// c16_out channels: 0-3 rest | 4-7 bilinear-up(merged) | 8-11 warp(rest, up(grid)) | 12-13 up(grid) | 14-15 zero
__global__ void __launch_bounds__(256) upscaler_input_kernel(ImgArgs a) {
  warm_kernarg<(int)sizeof(ImgArgs)>();
  const int S = 512, P = S * S, idx = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (idx >= P) return;
  const int y = idx / S, x = idx % S;
  const float* rest = a.in0 + (size_t)n * 4 * P;
  const float* merged = a.in1 + (size_t)n * 4 * 256 * 256;
  const float* grid = a.in2 + (size_t)n * 2 * 256 * 256;
  const float gxc = up2_sample(grid, 256, x, y), gyc = up2_sample(grid + 256 * 256, 256, x, y);
  const float gx = axis_pos(x, S) + gxc, gy = axis_pos(y, S) + gyc;
  float* co = a.c16_out + ((size_t)n * P + idx) * 16;
  f32x4 v0, v1, v2;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v0[c] = rest[(size_t)c * P + idx];
    v1[c] = up2_sample(merged + (size_t)c * 256 * 256, 256, x, y);
    v2[c] = sample_border(rest, S, S, c, gx, gy);
  }
  const f32x4 v3 = {gxc, gyc, 0.0f, 0.0f};
  put_c16(co, v0, v1, v2, v3);
}

}  // namespace tha4
