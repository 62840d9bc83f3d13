// §8(f) "next" rows either side of the poser path: the display epilogue every real-time caller runs
// right after pose(), and the image ingest that produces the path's input.  Pointwise and HBM-bound:
// 16-byte vector accesses, one thread per pixel, the RGBA8 side moves 4 bytes per pixel.
//
// Reference (paths relative to /root/reference/src/tha4):
//   app/character_model_ifacialmocap_puppeteer.py:325-349,377-381  clip((x+1)/2) -> linear->sRGB -> optional
//                                                                  background blend -> HWC -> *255 -> .byte()
//   image_util.py:56-58, shion/base/image_util.py:31-33            convert_linear_to_srgb / torch_linear_to_srgb
//   shion/base/image_util.py:10-12,127-149,194-198                 PIL RGBA8 -> sRGB->linear -> premultiply -> *2-1 -> CHW
#pragma once
#include "tha4_platform.h"

namespace tha4 {

// Every product and sum below is rounded on its own, like the reference's separate torch ops (`contract(off)`): the bytes must not depend
// on which multiply-add the compiler happens to fuse in the kernel this code is inlined into (fused epilogues vs display_rgba8_kernel)
THA4_DEV float linear_to_srgb(float x) {
#pragma clang fp contract(off)
  x = fminf(fmaxf(x, 0.0f), 1.0f);
  return x <= 0.003130804953560372f ? x * 12.92f : 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
}
// x^(1/2.4) on v_log_f32 / v_exp_f32 (~1 ulp each) for the display epilogue fused into compute kernels: 3 issue slots where
// libm's powf is ~60; the result differs from powf by a few ulp, i.e. by at most one step of the truncated 8-bit value at
// an exact boundary (tests gate on <= 1 LSB against the reference fixture, the same bound the standalone kernel gets)
THA4_DEV float linear_to_srgb_fast(float x) {
#ifdef THA4_EMU
  return linear_to_srgb(x);
#else
#pragma clang fp contract(off)
  x = fminf(fmaxf(x, 0.0f), 1.0f);
  const float pw = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.4f));
  return x <= 0.003130804953560372f ? x * 12.92f : 1.055f * pw - 0.055f;
#endif
}
// display value of ONE channel of one pixel (the arithmetic of display_rgba8_kernel): v = poser output in [-1,1],
// alpha01 = the pixel's clipped alpha in [0,1] (only read with a background)
THA4_DEV unsigned char display_channel(float v, int channel, float alpha01, bool has_background, float bg) {
#pragma clang fp contract(off)
  float c = fminf(fmaxf((v + 1.0f) * 0.5f, 0.0f), 1.0f);
  if (channel < 3) {
    c = linear_to_srgb_fast(c);
    if (has_background) c = c * alpha01 + (1.0f - alpha01) * bg;
  } else if (has_background) {
    c = 1.0f;
  }
  return (unsigned char)(255.0f * c);
}
THA4_DEV float srgb_to_linear(float x) {
  x = fminf(fmaxf(x, 0.0f), 1.0f);
  return x <= 0.04045f ? x / 12.92f : powf((x + 0.055f) / 1.055f, 2.4f);
}

struct DisplayArgs {
  const float* frames;     // [B][4][H*W] poser output, values in [-1,1]
  unsigned char* out;      // [B][H*W][4] RGBA8 (HWC, what wx.ImageFromBuffer / a video encoder takes)
  int pixels;
  int has_background;      // 0: keep alpha | 1: blend over an opaque background colour (alpha becomes 1)
  float bg[3];             // background colour in the SAME (sRGB-encoded, [0,1]) space the reference blends in
};

__global__ void __launch_bounds__(256) display_rgba8_kernel(DisplayArgs a) {
  warm_kernarg<(int)sizeof(DisplayArgs)>();
  const int i = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (i >= a.pixels) return;
  const float* f = a.frames + (size_t)n * 4 * a.pixels + i;
  // the per-channel arithmetic is display_channel(): the kernels that fuse this epilogue (the student's warp/blend tail, the
  // upscaler's tail) produce the same bytes
  const float a01 = fminf(fmaxf((f[(size_t)3 * a.pixels] + 1.0f) * 0.5f, 0.0f), 1.0f);
  uchar4 o;
  o.x = display_channel(f[0], 0, a01, a.has_background != 0, a.bg[0]);
  o.y = display_channel(f[(size_t)a.pixels], 1, a01, a.has_background != 0, a.bg[1]);
  o.z = display_channel(f[(size_t)2 * a.pixels], 2, a01, a.has_background != 0, a.bg[2]);
  o.w = display_channel(f[(size_t)3 * a.pixels], 3, a01, a.has_background != 0, 0.0f);
  reinterpret_cast<uchar4*>(a.out)[(size_t)n * a.pixels + i] = o;
}

struct IngestArgs {
  const unsigned char* rgba;   // [B][H*W][4]
  float* out;                  // [B][4][H*W]
  int pixels;
};

__global__ void __launch_bounds__(256) ingest_rgba8_kernel(IngestArgs a) {
  warm_kernarg<(int)sizeof(IngestArgs)>();
  const int i = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (i >= a.pixels) return;
  const uchar4 v = reinterpret_cast<const uchar4*>(a.rgba)[(size_t)n * a.pixels + i];
  const float al = (float)v.w / 255.0f;
  const float r = srgb_to_linear((float)v.x / 255.0f) * al;      // premultiplied, linear
  const float g = srgb_to_linear((float)v.y / 255.0f) * al;
  const float b = srgb_to_linear((float)v.z / 255.0f) * al;
  float* o = a.out + (size_t)n * 4 * a.pixels + i;
  o[0] = r * 2.0f - 1.0f;
  o[(size_t)a.pixels] = g * 2.0f - 1.0f;
  o[(size_t)2 * a.pixels] = b * 2.0f - 1.0f;
  o[(size_t)3 * a.pixels] = al * 2.0f - 1.0f;
}

}  // namespace tha4
