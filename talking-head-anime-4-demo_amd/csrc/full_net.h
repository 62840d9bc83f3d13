// Host-side executor of the full THA4 system (reference mode_07) on gfx950: turns the five networks'
// state_dicts into a static schedule of kernel launches (built once at create time) over one parameter
// blob and one workspace arena.  No allocation, no synchronisation and no host<->device traffic per frame.
//
// Schedule = the reference's cached-DAG evaluation (mode_07.py:54-134) flattened:
//   eyebrow_decomposer -> eyebrow_morphing_combiner -> face_morpher -> paste/half -> body_morpher -> upscaler
// Normalisation layers never run as standalone passes: producers emit per-tile moments, a tiny finalize
// kernel turns them into per-(frame, channel) scale/shift (FiLM folded in), and the CONSUMER convolution
// applies scale/shift + activation (+ nearest-up / avg-pool) while it loads its operand.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "full_image_kernels.h"
#include "full_conv16_kernels.h"
#include "full_conv_small_kernels.h"
#ifndef THA4_TILE16_DEFAULT
#define THA4_TILE16_DEFAULT 0      // conv_tile_kernel classes that run with sixteen waves (bit mask, see tile16_mask)
#endif
#ifndef THA4_TILE_NW4_DEFAULT
#define THA4_TILE_NW4_DEFAULT 50    // conv_tile_kernel classes that run as four-wave workgroups, two per CU (bit mask, see nw4_mask): <4,4> <4,2> <2,4>
#endif
#include "full_conv_point_kernels.h"
#include "full_kernels.h"
#include "full_layout.h"

namespace tha4 {

struct HostTensor {
  const float* data = nullptr;
  std::vector<int64_t> dims;
  int64_t numel() const { int64_t n = 1; for (auto d : dims) n *= d; return n; }
};
using WeightMap = std::map<std::string, HostTensor>;

struct FTensor {           // C16 feature map in the workspace
  size_t off = 0;          // float offset
  int cb = 0, h = 0, w = 0;
  size_t stats_off = 0;    // partial moments [n][tiles][cb*16][2]
  int stats_tiles = 0;
  size_t acc_off = (size_t)-1;   // moment accumulators [n][kMomentShards][cb*16] (MomentAcc) in the accumulator arena, or none
  int px() const { return h * w; }
};

struct Pending {           // transform a consumer applies while loading a tensor
  size_t scale_off = (size_t)-1, shift_off = (size_t)-1;   // workspace offsets of [n][cb*16] (written by norm_finalize_kernel)
  // ... or FUSED: no finalize launch; the consumer reduces the producer's per-tile moments itself (FusedNorm)
  bool fused = false;
  // ... from the producers' moment accumulators (round 5): `has_acc` - every source carries accumulators at acc_off[] (accumulator arena) besides its per-tile
  // moments, the consumer takes whichever it can fold faster; `acc_only` - too many tiles for the per-tile route: the consumer must be a kernel that reads them
  bool has_acc = false, acc_only = false;
  size_t acc_off[2] = {0, 0};
  int total_tiles = 0;
  size_t stats_off[2] = {0, 0};
  int tiles[2] = {0, 0};
  int channels = 0, groups = 0;
  float inv_count = 0.f;
  size_t gamma_off = 0, beta_off = 0, film0_off = (size_t)-1, film1_off = (size_t)-1;
  long long film1_stride = 0;
  bool has() const { return fused || scale_off != (size_t)-1; }
};

constexpr size_t kNone = (size_t)-1;

class FullModel {
 public:
  std::string error;
  int max_batch = 1;
  int sel_index = 2;       // eyebrow_morphed_image_index (mode_07.py:275)
  // THA4_FULL_EXACT_FP32 (round 4): every convolution on the exact-fp32 kernels (conv_mfma_kernel / conv_splitk_kernel:
  // v_mfma_f32_16x16x4_f32 on fp32 operands, no fp16 hi/lo staging and therefore no 65504 operand limit), every normalisation
  // through norm_finalize_kernel.  The plan a caller falls back to when the numeric-range guard of the default plan trips.
  bool exact_fp32 = false;
  // THA4_FULL_EXACT_DECOMPOSER (round 6): the mixed plan - exact_fp32 is raised while network 0 is planned and lowered again behind it.
  // exact_decomposer_outer (THA4_FULL_EXACT_DECOMPOSER_OUTER): the same for every convolution of network 0 EXCEPT the eleven 512 -> 512 convolutions of
  // its 16x16 bottleneck (they carry a fifth of the decomposer's share of the error and most of its launches)
  bool exact_decomposer = false, exact_decomposer_outer = false;
  bool bottleneck_on_split = false;      // planning state: inside network 0 of an "outer" mixed plan

  // ---- arenas -------------------------------------------------------------------------------
  std::vector<char> host_params;     // packed parameters, uploaded once
  char* dev_params = nullptr;
  char* dev_work = nullptr;
  size_t work_floats = 0;
  // split-K scratch shared by all convolutions (they run back to back on one stream): sized while the schedule is
  // built, placed in the workspace by finalize_scratch() before the arena is allocated (and zeroed) by the caller
  size_t partial_floats = 0, partial_off = 0;
  int conv_counter = 0;      // build-order index of every convolution (schedule dump / tuning aids)
  size_t dbg_off = 0;        // tuning aid (THA4_PHASE_TIMING): stamp buffer
  int* fault = nullptr;      // sticky numeric-fault flag (pinned host memory mapped into the device; set by the C ABI at create)
  void finalize_scratch() {
    if (std::getenv("THA4_DUMP_SCHEDULE"))
      std::fprintf(stderr, "plan: %zu + %zu ops (decomposer + rest); normalisations: %d foldable from per-tile moments, %d from moment accumulators only, %d finalize launches; "
                   "%d convolution launches fold accumulators; accumulator arena %zu KiB\n", ops_decomposer.size(), ops_rest.size(), n_norm_tiles, n_norm_acc, n_norm_finalize,
                   n_conv_acc, acc_floats * sizeof(float) >> 10);
    while (info_decomposer.size() < ops_decomposer.size()) info_decomposer.push_back(OpInfo{"(unlabelled)", 0.0});
    while (info_rest.size() < ops_rest.size()) info_rest.push_back(OpInfo{"(unlabelled)", 0.0});
    partial_off = alloc_work(partial_floats);
#ifdef THA4_PHASE_TIMING
    dbg_off = alloc_work((size_t)2 << 20);
#endif
  }

  size_t add_param(const void* p, size_t bytes) {
    size_t at = (host_params.size() + 255) / 256 * 256;
    host_params.resize(at + bytes);
    std::memcpy(host_params.data() + at, p, bytes);
    return at;
  }
  size_t add_param(const std::vector<float>& v) { return add_param(v.data(), v.size() * sizeof(float)); }
  size_t add_param_i(const std::vector<int>& v) { return add_param(v.data(), v.size() * sizeof(int)); }
  size_t alloc_work(size_t floats_per_frame) {      // per-frame size; arena holds max_batch frames, n-major per tensor
    size_t at = (work_floats + 63) / 64 * 64;
    work_floats = at + floats_per_frame * (size_t)max_batch;
    return at;
  }
  // moment accumulators (full_kernels.h MomentAcc): their own arena, zeroed by ONE memset at the top of every call
  char* dev_acc = nullptr;
  size_t acc_floats = 0;
  size_t alloc_acc(size_t floats_per_frame) {
    size_t at = (acc_floats + 63) / 64 * 64;
    acc_floats = at + floats_per_frame * (size_t)max_batch;
    return at;
  }
  MomentAcc* Acc(size_t off) const { return reinterpret_cast<MomentAcc*>(reinterpret_cast<float*>(dev_acc) + off); }
  // Producer-side accumulation + consumer-side folding instead of a norm_finalize_kernel launch (round 5).  MEASURED NEUTRAL TO NEGATIVE, off unless
  // THA4_TUNING + THA4_MOMENT_ACC are set (parity-clean: the GPU suite passes with it on): replacing the 54 finalize launches of a batch-1 frame (tensors of
  // more than 64 tiles) is +0.5 % / +-0.0 % on two boxes (profiles/r05_raw/c6_ab_moment_acc.txt, c7_ab_moment_acc.txt: 184.2 vs 183.3, 185.4 vs 185.4 frames/s) -
  // every consumer workgroup now folds its own table (8 shards x 4 int64 per channel, fp64 arithmetic, two barriers: ~2-3 us on a single-round grid's critical
  // path) where ONE 5.5 us launch did it for all of them, plus a memset of the arena per call; extending it to the 17-64-tile tensors (THA4_ACC_MIN_TILES=16:
  // one load round instead of 3-8) costs 2.8 % (180.2 vs 185.4): 128-256 more device-scope atomics in the epilogue of every producer and a 6.7 MB arena to zero.
  bool acc_planned() const { return !exact_fp32 && tune_env("THA4_MOMENT_ACC") && !tune_env("THA4_NO_TILE_CONV"); }
  template <class T = float> const T* P(size_t off) const { return reinterpret_cast<const T*>(dev_params + off); }
  float* Wk(size_t off) const { return reinterpret_cast<float*>(dev_work) + off; }

  // ---- schedule -------------------------------------------------------------------------------
  struct Frame {            // per-call bindings
    const float* image; long long image_stride; const float* pose; int batch; hipStream_t stream;
    float* out[33];         // NCHW outputs in the reference order; null = not requested and read by no later stage (the kernels skip its stores)
    unsigned char* rgba8; int rgba8_has_bg; float rgba8_bg[3];    // fused display epilogue of out[0] (tha4_display) or null
  };
  using Op = std::function<void(const Frame&)>;
  std::vector<Op> ops_decomposer, ops_rest;
  // one record per op of the schedule (ABI v5 tha4_full_op_info / tha4_full_last_op_ms: live per-launch-class timing for bench.py's roofline):
  // what it is and the as-written FLOPs (2 x MAC of the reference's layer, per frame) it stands for - 0 for everything that is not a convolution
  struct OpInfo { std::string label; double gflop = 0.0; };
  std::vector<OpInfo> info_decomposer, info_rest;
  void note(std::vector<Op>& ops, const std::string& label, double gflop = 0.0) {      // right behind every ops.push_back
    std::vector<OpInfo>& v = &ops == &ops_decomposer ? info_decomposer : info_rest;
    while (v.size() + 1 < ops.size()) v.push_back(OpInfo{"(unlabelled)", 0.0});
    if (v.size() < ops.size()) v.push_back(OpInfo{label, gflop});
  }

  // (Round 5 measured the frame's off-chain branches - the 27 skip convolutions, the FiLM gemvs - on a second stream of the handle between fork / join events:
  //  parity-clean, 185.2 -> 169.2 frames/s steady: a cross-queue dependency costs ~8 us of chain time, more than the launches it hides.  The code was removed in
  //  round 6; profiles/r05_boundaries_reading.md section 4 keeps the measurement.)
  size_t scratch_out[33];   // workspace offsets used for outputs the caller did not ask for
  // outputs of the 33-entry list that a later stage of the pipeline reads back (build() below): the combiner image the face morpher's input is pasted from
  // (19 + sel), the face morpher's output_image (11 -> paste_face), face_morphed_full (5 -> half image, upscaler input, the upscaler's warp source), the
  // body morpher's merged image and grid (6, 9 -> upscaler input).  Everything else is a leaf: unrequested, it is not written at all.
  bool read_by_later_stage(int i) const { return i == 5 || i == 6 || i == 9 || i == 11 || i == 19 + sel_index; }
  int out_ch[33], out_size[33];

  // ---- weights --------------------------------------------------------------------------------
  const HostTensor& get(const WeightMap& w, const std::string& k) {
    auto it = w.find(k);
    if (it == w.end()) { if (error.empty()) error = "missing tensor '" + k + "'"; static HostTensor empty; return empty; }
    return it->second;
  }
  bool expect(const HostTensor& t, std::initializer_list<int64_t> dims, const std::string& k) {
    if (!t.data || t.dims != std::vector<int64_t>(dims)) { if (error.empty()) error = "tensor '" + k + "' has an unexpected shape"; return false; }
    return true;
  }

  // ---- building blocks --------------------------------------------------------------------------
  FTensor new_tensor(int cb, int h, int w) {
    FTensor t; t.cb = cb; t.h = h; t.w = w; t.off = alloc_work((size_t)cb * h * w * 16); return t;
  }

  struct Src {              // one convolution operand
    FTensor t; Pending pend; bool vector = false; size_t vec_off = 0; int vec_cb = 0;
    int channels = 0;       // real channels contributed to the weight tensor
  };
  static Src src_tensor(const FTensor& t, int channels, Pending p = Pending()) { Src s; s.t = t; s.pend = p; s.channels = channels; return s; }
  static Src src_vector(size_t off, int cb, int channels) { Src s; s.vector = true; s.vec_off = off; s.vec_cb = cb; s.channels = channels; return s; }

  enum ConvKind { K_SAME3, K_SAME1, K_S2K4, K_CONVT };

  template <int TMB, int PG, int INMODE>
  static void launch_conv(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((conv_mfma_kernel<TMB, PG, INMODE>), grid, dim3(256), lds, s, a);
  }
  static void dispatch_conv(int tmb, int pg, int inmode, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
#define THA4_CASE(TM, PGV)                                                            \
  if (tmb == TM && pg == PGV) {                                                        \
    if (inmode == IN_DIRECT) return launch_conv<TM, PGV, IN_DIRECT>(a, grid, lds, s);  \
    if (inmode == IN_UP2) return launch_conv<TM, PGV, IN_UP2>(a, grid, lds, s);        \
    return launch_conv<TM, PGV, IN_POOL2>(a, grid, lds, s);                            \
  }
    THA4_CASE(4, 2) THA4_CASE(4, 1) THA4_CASE(2, 2) THA4_CASE(2, 1) THA4_CASE(1, 1)
#undef THA4_CASE
  }
  template <int TMB, int PG>
  static void launch_tile(int inmode, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    if (inmode == IN_DIRECT) hipLaunchKernelGGL((conv_tile_kernel<TMB, PG, IN_DIRECT>), grid, dim3(kTileThreads), lds, s, a);
    else if (inmode == IN_UP2) hipLaunchKernelGGL((conv_tile_kernel<TMB, PG, IN_UP2>), grid, dim3(kTileThreads), lds, s, a);
    else hipLaunchKernelGGL((conv_tile_kernel<TMB, PG, IN_POOL2>), grid, dim3(kTileThreads), lds, s, a);
  }
  // sixteen-wave form (MSW = 2: the tile's output blocks split over two halves of the workgroup).  MEASURED NEGATIVE for every class
  // (round 3, tools/runs_r03/gpu_r03_c37.sh: <2,4> +-0, <2,2> -0.4 %, <4,1> -0.6 %, <4,2> -2 %, <2,1> -2.4 % of a frame; all parity-clean):
  // unlike the student's kernels the convolution waves are not latency-starved at two per SIMD - each B fragment is now read by twice the
  // waves and the chunk barriers join sixteen.  Kept behind -DTHA4_TILE16_BUILD (the emulator and the per-op device harness always cover
  // the kernel's MSW = 2 path); the shipped library does not instantiate it.
#ifdef THA4_TILE16_BUILD
  template <int TMB, int PG>
  static void launch_tile16(int inmode, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    if (inmode == IN_DIRECT) hipLaunchKernelGGL((conv_tile_kernel<TMB, PG, IN_DIRECT, 2>), grid, dim3(kTileThreads * 2), lds, s, a);
    else if (inmode == IN_UP2) hipLaunchKernelGGL((conv_tile_kernel<TMB, PG, IN_UP2, 2>), grid, dim3(kTileThreads * 2), lds, s, a);
    else hipLaunchKernelGGL((conv_tile_kernel<TMB, PG, IN_POOL2, 2>), grid, dim3(kTileThreads * 2), lds, s, a);
  }
  // THA4_TILE16 (tuning aid): bit mask of the (TMB, PG) classes that run with sixteen waves: 1 <4,1>  2 <2,4>  4 <2,1>  8 <2,2>  16 <4,2>
  static int tile16_mask() {
    static const int m = tune_env("THA4_TILE16") ? std::atoi(tune_env("THA4_TILE16")) : THA4_TILE16_DEFAULT;
    return m;
  }
  static bool tile16(int tmb, int pg) {
    const int bit = tmb == 4 && pg == 1 ? 1 : tmb == 2 && pg == 4 ? 2 : tmb == 2 && pg == 1 ? 4 : tmb == 2 && pg == 2 ? 8 : tmb == 4 && pg == 2 ? 16 : 0;
    return (tile16_mask() & bit) != 0;
  }
#endif
  // four-wave form (NW = 4, round 4): half the pixel tile, two workgroups per CU
  template <int TMB, int PG>
  static void launch_tile4(int inmode, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    if (inmode == IN_DIRECT) hipLaunchKernelGGL((conv_tile_kernel<TMB, PG, IN_DIRECT, 1, 4>), grid, dim3(256), lds, s, a);
    else if (inmode == IN_UP2) hipLaunchKernelGGL((conv_tile_kernel<TMB, PG, IN_UP2, 1, 4>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_tile_kernel<TMB, PG, IN_POOL2, 1, 4>), grid, dim3(256), lds, s, a);
  }
  // classes that take the four-wave form: 1 <4,1>  2 <2,4>  4 <2,1>  8 <2,2>  16 <4,2>  32 <4,4>; THA4_TILE_NW4 is a tuning aid
  static int nw4_mask() {
    static const int m = tune_env("THA4_TILE_NW4") ? std::atoi(tune_env("THA4_TILE_NW4")) : THA4_TILE_NW4_DEFAULT;
    return m;
  }
  static bool nw4_class(int tmb, int pg) {
    const int bit = tmb == 4 && pg == 1 ? 1 : tmb == 2 && pg == 4 ? 2 : tmb == 2 && pg == 1 ? 4 : tmb == 2 && pg == 2 ? 8 : tmb == 4 && pg == 2 ? 16 :
                    tmb == 4 && pg == 4 ? 32 : 0;
    return (nw4_mask() & bit) != 0;
  }
  // XCD-aware workgroup order of conv_tile_kernel (ConvArgs::xcd_remap); THA4_NO_XCD_REMAP is a tuning aid (A/B)
  static bool xcd_remap_enabled() {
    static const bool on = !tune_env("THA4_NO_XCD_REMAP");
    return on;
  }
  static void dispatch_tile(int tmb, int pg, int inmode, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s, bool nw4 = false) {
    if (nw4) {
#define THA4_TCASE4(TM, PGV) if (tmb == TM && pg == PGV) return launch_tile4<TM, PGV>(inmode, a, grid, lds, s);
      THA4_TCASE4(8, 2) THA4_TCASE4(4, 4) THA4_TCASE4(4, 2) THA4_TCASE4(4, 1) THA4_TCASE4(2, 4) THA4_TCASE4(2, 2) THA4_TCASE4(2, 1)
#undef THA4_TCASE4
    }
#ifdef THA4_TILE16_BUILD
    if (a.phase != 2 && tile16(tmb, pg)) {
#define THA4_TCASE16(TM, PGV) if (tmb == TM && pg == PGV) return launch_tile16<TM, PGV>(inmode, a, grid, lds, s);
      THA4_TCASE16(4, 1) THA4_TCASE16(2, 4) THA4_TCASE16(2, 1) THA4_TCASE16(2, 2) THA4_TCASE16(4, 2)
#undef THA4_TCASE16
    }
#endif
#define THA4_TCASE(TM, PGV) if (tmb == TM && pg == PGV) return launch_tile<TM, PGV>(inmode, a, grid, lds, s);
    THA4_TCASE(8, 2) THA4_TCASE(4, 4) THA4_TCASE(4, 2) THA4_TCASE(4, 1) THA4_TCASE(2, 4) THA4_TCASE(2, 2) THA4_TCASE(2, 1)
    THA4_TCASE(1, 4) THA4_TCASE(1, 2) THA4_TCASE(1, 1)
#undef THA4_TCASE
  }
  template <int PGV>
  static void launch_small(int inmode, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    if (inmode == IN_DIRECT) hipLaunchKernelGGL((conv_small_kernel<PGV, IN_DIRECT>), grid, dim3(kSmallThreads), lds, s, a);
    else if (inmode == IN_UP2) hipLaunchKernelGGL((conv_small_kernel<PGV, IN_UP2>), grid, dim3(kSmallThreads), lds, s, a);
    else hipLaunchKernelGGL((conv_small_kernel<PGV, IN_POOL2>), grid, dim3(kSmallThreads), lds, s, a);
  }
  static void dispatch_small(int pg, int inmode, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    if (pg == 4) return launch_small<4>(inmode, a, grid, lds, s);
    if (pg == 2) return launch_small<2>(inmode, a, grid, lds, s);
    return launch_small<1>(inmode, a, grid, lds, s);
  }
  static void dispatch_point(int tmb, int pg, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
#define THA4_PCASE(TM, PGV) if (tmb == TM && pg == PGV) { hipLaunchKernelGGL((conv_point_kernel<TM, PGV>), grid, dim3(kPointThreads), lds, s, a); return; }
    THA4_PCASE(4, 2) THA4_PCASE(4, 1) THA4_PCASE(2, 2) THA4_PCASE(2, 1) THA4_PCASE(1, 2) THA4_PCASE(1, 1)
#undef THA4_PCASE
  }
  static hipError_t allow_all_conv_lds() {
    hipError_t e = hipSuccess;
    auto set = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
#define THA4_ALLOW(TM, PGV)                                                         \
  set(reinterpret_cast<const void*>(conv_mfma_kernel<TM, PGV, IN_DIRECT>));        \
  set(reinterpret_cast<const void*>(conv_mfma_kernel<TM, PGV, IN_UP2>));           \
  set(reinterpret_cast<const void*>(conv_mfma_kernel<TM, PGV, IN_POOL2>));
    THA4_ALLOW(4, 2) THA4_ALLOW(4, 1) THA4_ALLOW(2, 2) THA4_ALLOW(2, 1) THA4_ALLOW(1, 1)
#undef THA4_ALLOW
#define THA4_TALLOW(TM, PGV)                                                       \
  set(reinterpret_cast<const void*>(conv_tile_kernel<TM, PGV, IN_DIRECT>));        \
  set(reinterpret_cast<const void*>(conv_tile_kernel<TM, PGV, IN_UP2>));           \
  set(reinterpret_cast<const void*>(conv_tile_kernel<TM, PGV, IN_POOL2>));
    THA4_TALLOW(8, 2) THA4_TALLOW(4, 4) THA4_TALLOW(4, 2) THA4_TALLOW(4, 1) THA4_TALLOW(2, 4) THA4_TALLOW(2, 2) THA4_TALLOW(2, 1)
    THA4_TALLOW(1, 4) THA4_TALLOW(1, 2) THA4_TALLOW(1, 1)
#undef THA4_TALLOW
#define THA4_TALLOW4(TM, PGV)                                                            \
  set(reinterpret_cast<const void*>(conv_tile_kernel<TM, PGV, IN_DIRECT, 1, 4>));        \
  set(reinterpret_cast<const void*>(conv_tile_kernel<TM, PGV, IN_UP2, 1, 4>));           \
  set(reinterpret_cast<const void*>(conv_tile_kernel<TM, PGV, IN_POOL2, 1, 4>));
    THA4_TALLOW4(8, 2) THA4_TALLOW4(4, 4) THA4_TALLOW4(4, 2) THA4_TALLOW4(4, 1) THA4_TALLOW4(2, 4) THA4_TALLOW4(2, 2) THA4_TALLOW4(2, 1)
#undef THA4_TALLOW4
#ifdef THA4_TILE16_BUILD
#define THA4_TALLOW16(TM, PGV)                                                        \
  set(reinterpret_cast<const void*>(conv_tile_kernel<TM, PGV, IN_DIRECT, 2>));        \
  set(reinterpret_cast<const void*>(conv_tile_kernel<TM, PGV, IN_UP2, 2>));           \
  set(reinterpret_cast<const void*>(conv_tile_kernel<TM, PGV, IN_POOL2, 2>));
    THA4_TALLOW16(4, 1) THA4_TALLOW16(2, 4) THA4_TALLOW16(2, 1) THA4_TALLOW16(2, 2) THA4_TALLOW16(4, 2)
#undef THA4_TALLOW16
#endif
#define THA4_SALLOW(PGV)                                                           \
  set(reinterpret_cast<const void*>(conv_small_kernel<PGV, IN_DIRECT>));           \
  set(reinterpret_cast<const void*>(conv_small_kernel<PGV, IN_UP2>));              \
  set(reinterpret_cast<const void*>(conv_small_kernel<PGV, IN_POOL2>));
    THA4_SALLOW(4) THA4_SALLOW(2) THA4_SALLOW(1)
#undef THA4_SALLOW
#define THA4_PALLOW(TM, PGV) set(reinterpret_cast<const void*>(conv_point_kernel<TM, PGV>));
    THA4_PALLOW(4, 2) THA4_PALLOW(4, 1) THA4_PALLOW(2, 2) THA4_PALLOW(2, 1) THA4_PALLOW(1, 2) THA4_PALLOW(1, 1)
#undef THA4_PALLOW
    set(reinterpret_cast<const void*>(attention_kernel));
    return e;
  }

  // Convolution.  weight: Conv2d [cout][cin][k][k] or ConvTranspose2d [cin][cout][4][4]; the sources are
  // concatenated along the weight's input channels in order.  Returns the raw output tensor (+ moments).
  FTensor conv(std::vector<Op>& ops, ConvKind kind, const std::vector<Src>& srcs, int in_mode, int act_in,
               const HostTensor& weight, const float* bias_host, int cout, bool want_stats,
               const FTensor* residual = nullptr, int res_mode = IN_DIRECT, const std::vector<int>* act_out = nullptr,
               const std::vector<float>* bias_override = nullptr) {
    if (act_in != ACT_NONE && act_in != ACT_RELU && act_in != ACT_SILU) {      // apply_act4: sigmoid / tanh exist on the output side only
      if (error.empty()) error = "convolution input activation must be none, ReLU or SiLU";
      return FTensor();
    }
    const FTensor& t0 = srcs[0].t;
    const int ih = t0.h, iw = t0.w;
    const int vh = in_mode == IN_UP2 ? ih * 2 : (in_mode == IN_POOL2 ? ih / 2 : ih);
    const int vw = in_mode == IN_UP2 ? iw * 2 : (in_mode == IN_POOL2 ? iw / 2 : iw);
    int oh = vh, ow = vw, th = vh, tw = vw, nclass = 1, k = 3;
    if (kind == K_SAME1) k = 1;
    if (kind == K_S2K4) { oh = vh / 2; ow = vw / 2; th = oh; tw = ow; k = 4; }
    if (kind == K_CONVT) { oh = vh * 2; ow = vw * 2; nclass = 4; k = 4; }
    int cin = 0;
    std::vector<ChannelSegment> segs;
    for (auto& s : srcs) { segs.push_back({cin, s.channels}); cin += s.channels; }
    const int nb = (cout + 15) / 16;
    int tmb = nb % 4 == 0 ? 4 : (nb % 2 == 0 ? 2 : 1);
    int mtiles = nb / tmb;
    const int tile_px = th * tw;
    int cbtot = 0;
    for (auto& s : srcs) cbtot += (s.channels + 15) / 16;
    const int ntaps_k = kind == K_SAME3 ? 9 : kind == K_SAME1 ? 1 : kind == K_S2K4 ? 16 : 4;
    // every k > 1 convolution: LDS-staged window + fp16 hi/lo MFMA (conv_tile_kernel); small maps split K over blockIdx.z
    bool tiled = false;
    TilePlan plan;
    const int nq = (cbtot + 1) / 2;
    // normalisation folded into this convolution? (all tensor sources come from ONE norm() call)
    const Pending* fpend = nullptr;
    int ctab = 0;
    for (auto& sx : srcs)
      if (!sx.vector) { ctab += sx.t.cb * 16; if (sx.pend.fused) fpend = &sx.pend; }
    const size_t table_bytes = fpend ? (size_t)2 * ctab * sizeof(float) : 0;
    if ((kind != K_SAME1 || tune_env("THA4_TILE_1X1")) && !tune_env("THA4_NO_TILE_CONV") && !exact_fp32) {
      const ConvGeom g0 = kind == K_SAME3 ? geom_conv_same(3) : kind == K_SAME1 ? geom_conv_same(1) : kind == K_S2K4 ? geom_conv4_s2() : geom_convT4_s2(0, 0);
      plan = plan_tile_conv(g0, th, tw, tmb, mtiles, nq, 256, max_batch);
      if (plan.ok && table_bytes && !tile_geom(g0, th, tw, plan.pg, tmb, plan.geom.tw_log2, table_bytes).ok) plan.ok = false;
      // a half-filled chip without K split: halve the output tile instead (the window is staged twice as often, but
      // no partial-sum traffic and no second launch)
      if (plan.ok && plan.ksplit == 1 && tmb == 4 && plan.geom.tiles * mtiles * max_batch < 200 && !tune_env("THA4_NO_TMB_HALVE")) {
        const TilePlan p2 = plan_tile_conv(g0, th, tw, 2, mtiles * 2, nq, 256, max_batch);
        if (p2.ok && p2.ksplit == 1 && p2.pg >= plan.pg) { plan = p2; tmb = 2; mtiles *= 2; }
      }
      tiled = plan.ok;
      if (tune_env("THA4_NO_TILE_SPLITK") && plan.ksplit > 1) tiled = false;
    }
    // 1x1 convolutions on maps above 32x32: conv_point_kernel (operands straight from C16 global memory, fp16 hi/lo MFMA,
    // the K loop free of control flow around memory operations)
    const int max1x1 = tune_env("THA4_SMALL_1X1_MAX_PX") ? std::atoi(tune_env("THA4_SMALL_1X1_MAX_PX")) : 32 * 32;
    bool point = false;
    PointPlan pp;
    if (kind == K_SAME1 && in_mode == IN_DIRECT && tile_px > max1x1 && !tiled && !tune_env("THA4_NO_POINT_CONV") && !exact_fp32 &&
        (!residual || res_mode == IN_DIRECT) && point_act_supported(act_in)) {
      bool tensors = true;
      for (auto& sx : srcs) tensors = tensors && !sx.vector;
      pp = plan_point_conv(tile_px, nb, cbtot, max_batch, fpend != nullptr);
      point = tensors && pp.ok;
      if (point) { tmb = pp.tmb; mtiles = nb / tmb; }
    }
    // small maps (the tile plan would split K over two launches) and 1x1 convolutions: conv_small_kernel, K split across
    // the waves of one workgroup, ONE launch
    bool small = false;
    SmallPlan sp;
    if (!point && !exact_fp32) {
      const ConvGeom g0 = kind == K_SAME3 ? geom_conv_same(3) : kind == K_SAME1 ? geom_conv_same(1) : kind == K_S2K4 ? geom_conv4_s2() : geom_convT4_s2(0, 0);
      // where it wins (per-layer breakdown in profiles/r02_full_b1_reading.md): 3x3 / 1x1 / transposed-conv classes on maps
      // the tile plan would split over two launches, not the 16-tap stride-2 convolutions (their 108-pixel window per 16
      // outputs makes staging dominate) and not average-pooled inputs (four dependent samples per staged item)
      bool want = kind == K_SAME1 ? tile_px <= max1x1 : (!tiled || plan.ksplit > 1);
      if (fpend && fpend->acc_only) want = false;           // conv_small_kernel folds per-tile moments only (a tensor that NEEDS the accumulators has > 64 tiles: never a small map)
      if (!tune_env("THA4_SMALL_ALL_KINDS") && (kind == K_S2K4 || in_mode == IN_POOL2)) want = false;
      if (want && !tune_env("THA4_NO_SMALL_CONV")) {
        sp = plan_small_conv(g0, th, tw, nb, nq, 256, max_batch);
        small = sp.ok && sp.lds + table_bytes + 128 <= 160 * 1024;
      }
      if (!small && fpend && !fpend->acc_only && !tiled && !tune_env("THA4_NO_SMALL_CONV")) {
        // a folded normalisation needs one of the two kernels that can evaluate it: take conv_small_kernel even if its grid
        // runs in several rounds (1x1 projections behind a GroupNorm when the schedule is built for 2 frames)
        sp = plan_small_conv(g0, th, tw, nb, nq, 1 << 30, max_batch);
        small = sp.ok && sp.lds + table_bytes + 128 <= 160 * 1024;
      }
      if (small) { tiled = false; tmb = 1; mtiles = nb; }
    }
    bool tmb8_nw4 = false;
    int tmb8_twl4 = 4;
    // <8,2> output tile (round-5 review, task 2; tuning option THA4_TILE_TMB8): 128 output channels x 256 positions per eight-wave workgroup where the plan took <4,4> -
    // the same sixteen accumulators per wave, half the window staging per FLOP, 20 instead of 16 LDS fragment reads per tap (profiles/r06_full_b8_reading.md)
    if (tiled && !small && !point && tune_env("THA4_TILE_TMB8") && kind != K_SAME1 && nb % 8 == 0 && tmb == 4 && plan.pg == 4 && plan.ksplit == 1) {
      const ConvGeom g0 = kind == K_SAME3 ? geom_conv_same(3) : kind == K_S2K4 ? geom_conv4_s2() : geom_convT4_s2(0, 0);
      TileGeom best8;
      for (int twl : {4, 5, 3}) {
        const TileGeom t = tile_geom(g0, th, tw, 2, 8, twl, table_bytes);
        if (t.ok && t.efficiency > best8.efficiency * 1.1f) best8 = t;
      }
      if (best8.ok && (long)best8.tiles * (nb / 8) * max_batch >= 256) { tmb = 8; mtiles = nb / 8; plan.pg = 2; plan.geom = best8; }
      // THA4_TILE_TMB8=2: the same tile on FOUR-wave workgroups (128 positions, two workgroups per CU: one's staging under the other's MFMAs)
      if (tmb == 8 && std::atoi(tune_env("THA4_TILE_TMB8")) == 2) {
        TileGeom b4;
        for (int twl : {4, 3, 5}) {
          const TileGeom t = tile_geom(g0, th, tw, 2, 8, twl, table_bytes, 4);
          if (t.ok && t.efficiency > b4.efficiency * 1.1f) { b4 = t; tmb8_twl4 = twl; }
        }
        tmb8_nw4 = b4.ok && (long)b4.tiles * (nb / 8) * max_batch >= 512;
      }
    }
    // four-wave workgroups, two per CU (conv_tile_kernel<..., NW = 4>): same per-wave tile, half the workgroup tile.  Only without a
    // K split (phase 2 reads the partials of an eight-wave phase 1) and where every parity class has a geometry within 80 KiB
    bool nw4 = false;
    int twl4 = 4;
    if (tiled && plan.ksplit == 1 && nw4_class(tmb, plan.pg)) {
      const ConvGeom g0 = kind == K_SAME3 ? geom_conv_same(3) : kind == K_SAME1 ? geom_conv_same(1) : kind == K_S2K4 ? geom_conv4_s2() : geom_convT4_s2(0, 0);
      float best_eff = 0.f;
      int tiles4 = 0;
      for (int twl : {4, 3, 5}) {
        const TileGeom t = tile_geom(g0, th, tw, plan.pg, tmb, twl, table_bytes, 4);
        if (t.ok && t.efficiency > best_eff * 1.1f) { best_eff = t.efficiency; twl4 = twl; nw4 = true; tiles4 = t.tiles; }
      }
      // grids that fill the chip with co-resident pairs: at least one full round (512 four-wave workgroups) on a handle built for one
      // frame (+1.3 % steady on two of three same-box rounds, profiles/r04_raw/c12_ab.txt), two rounds otherwise (batch 8: +4-7 % with
      // two rounds, -0.5 % when single-round grids join in)
      const long min_wgs = tune_env("THA4_TILE_NW4_MIN_WGS") ? std::atol(tune_env("THA4_TILE_NW4_MIN_WGS")) : (max_batch == 1 ? 512 : 2 * 512);
      if (nw4 && (long)tiles4 * mtiles * max_batch < min_wgs) nw4 = false;
    }
    if (tmb8_nw4) { nw4 = true; twl4 = tmb8_twl4; }
    // 1x1 convolutions on SMALL maps of a batched plan (round 5): conv_small_kernel refuses grids of several rounds, and these 32 launches of a batch-8 step
    // (qkv / attention projections / skips at 16x16 and 32x32) used to fall back to the exact-fp32 conv_splitk_kernel (2 % of the step): with the frames of the
    // batch there are enough pixel tiles for conv_point_kernel
    if (!point && !small && !tiled && kind == K_SAME1 && in_mode == IN_DIRECT && !exact_fp32 && !tune_env("THA4_NO_POINT_CONV") && !tune_env("THA4_NO_POINT_SMALL_MAPS") &&
        (!residual || res_mode == IN_DIRECT) && point_act_supported(act_in) && tile_px % 64 == 0) {
      bool tensors = true;
      for (auto& sx : srcs) tensors = tensors && !sx.vector;
      pp = plan_point_conv(tile_px, nb, cbtot, max_batch, fpend != nullptr);
      point = tensors && pp.ok;
      if (point) { tmb = pp.tmb; mtiles = nb / tmb; }
    }
    // fallbacks (1x1 convolutions): small maps (<= 32x32) one pixel group per workgroup with K split over its 4 waves
    // (conv_splitk_kernel), otherwise the exact-fp32 pixel-tiled kernel (conv_mfma_kernel)
    const bool splitk = !small && !tiled && !point && tile_px <= 1024 && tmb == 4 && cbtot * ntaps_k >= 8;
    if (fpend && !small && !tiled && !point) { if (error.empty()) error = "internal: a fused normalisation needs the tile / small convolution kernels"; return FTensor(); }
    int pg = 1;
    if (small) pg = sp.pg;
    else if (point) pg = pp.pg;
    else if (tiled) pg = plan.pg;
    else if (!splitk && tmb >= 2 && tile_px % 128 == 0 && (tile_px / 128) * mtiles * nclass >= 512) pg = 2;
    if (!small && !tiled && !point && tile_px % (splitk ? 16 : 64 * pg) != 0) { if (error.empty()) error = "conv tile grid is not a multiple of the pixel tile"; return FTensor(); }
    int tiles = small ? sp.tiles : point ? pp.tiles : tiled ? plan.geom.tiles : splitk ? tile_px / 16 : tile_px / (64 * pg);
    if (nw4) {
      const ConvGeom g0 = kind == K_SAME3 ? geom_conv_same(3) : kind == K_SAME1 ? geom_conv_same(1) : kind == K_S2K4 ? geom_conv4_s2() : geom_convT4_s2(0, 0);
      tiles = tile_geom(g0, th, tw, pg, tmb, twl4, table_bytes, 4).tiles;
    }
    const int ksplit = tiled ? plan.ksplit : 1;
    if (tiled && ksplit > 1) {
      partial_floats = std::max(partial_floats, (size_t)ksplit * mtiles * tiles * tmb * 8 * pg * 64 * 4 * (nclass == 4 ? 4 : 1));     // merged classes: one image each
    }
    const int conv_index = conv_counter++;
    if (std::getenv("THA4_DUMP_SCHEDULE"))
      std::fprintf(stderr, "conv #%d kind=%d in=%dx%d mode=%d tile=%dx%d cin=%d(cb %d) cout=%d taps=%d splitk=%d tmb=%d pg=%d classes=%d wgs=%d tiled=%d ksplit=%d twl=%d nw=%d\n", conv_index, (int)kind, ih,
                   iw, in_mode, th, tw, cin, cbtot, cout, ntaps_k, (int)splitk, tmb, pg, nclass, tiles * mtiles * ksplit, small ? 2 : point ? 3 : (int)tiled, ksplit,
                   small ? sp.tw_log2 : nw4 ? twl4 : tiled ? plan.geom.tw_log2 : 0, nw4 ? 4 : 8);
    FTensor out = new_tensor(nb, oh, ow);
    if (want_stats) {
      out.stats_tiles = tiles * nclass;
      out.stats_off = alloc_work((size_t)out.stats_tiles * nb * 16 * 2);
      // tensors of more than a handful of tiles on handles built for one or two frames: the producer (conv_tile_kernel / conv_small_kernel) also feeds the moment
      // accumulators, so that norm() needs no finalize launch and the consumer one round of loads (below).  Few tiles: the consumer folds the per-tile moments, as before
      if ((tiled || small) && acc_planned() && out.stats_tiles > acc_min_tiles() / 2 && max_batch <= fused_norm_max_batch())
        out.acc_off = alloc_acc((size_t)kMomentShards * nb * 16 * (sizeof(MomentAcc) / sizeof(float)));
    }
    size_t bias_off = kNone;
    if (bias_host || bias_override) {
      std::vector<float> b(nb * 16, 0.f);
      for (int i = 0; i < cout; ++i) b[i] = bias_override ? (*bias_override)[i] : bias_host[i];
      bias_off = add_param(b);
    }
    size_t act_off = kNone;
    if (act_out) { std::vector<int> a(nb * 16, 0); for (int i = 0; i < cout; ++i) a[i] = (*act_out)[i]; act_off = add_param_i(a); }
    // The four output-parity classes of a transposed convolution run as ONE launch of conv_tile_kernel / conv_small_kernel (round 4: the
    // class is the slowest grid dimension, its geometry follows from the class id in the kernel, the four weight images are contiguous):
    // 4 launches -> 1, 8 -> 2 with a K split.  The exact-fp32 kernels keep one launch per class.
    const bool merge_classes = nclass == 4 && (small || tiled) && !tune_env("THA4_NO_CLASS_MERGE");
    const int launch_classes = merge_classes ? 1 : nclass, grid_classes = merge_classes ? 4 : 1;
    auto pack16_classes = [&](const ConvGeom& g0c, int tmb_pack, float* inv, size_t* class_bytes) {
      std::vector<char> all = pack_conv_weight16(weight.data, cout, cin, k, k, kind == K_CONVT, g0c, segs, tmb_pack, inv);
      *class_bytes = all.size();
      if (merge_classes)
        for (int c2 = 1; c2 < 4; ++c2) {
          float inv2 = 1.f;
          const std::vector<char> more = pack_conv_weight16(weight.data, cout, cin, k, k, true, geom_convT4_s2(c2 >> 1, c2 & 1), segs, tmb_pack, &inv2);
          if (more.size() != *class_bytes || inv2 != *inv) { if (error.empty()) error = "internal: parity classes of a transposed convolution pack differently"; }
          all.insert(all.end(), more.begin(), more.end());
        }
      return all;
    };
    for (int cls = 0; cls < launch_classes; ++cls) {
      ConvGeom g = kind == K_SAME3 ? geom_conv_same(3) : kind == K_SAME1 ? geom_conv_same(1)
                 : kind == K_S2K4 ? geom_conv4_s2() : geom_convT4_s2(cls >> 1, cls & 1);
      size_t w_off = 0, lds = 0;
      int cq = 1;
      ConvArgs a{};
      a.nclass = grid_classes;
      if (small) {
        const SmallPlan sg = small_geom(g, th, tw, sp.pg, sp.tw_log2);
        if (!sg.ok || sg.tiles != sp.tiles) { if (error.empty()) error = "small conv geometry differs between parity classes"; return FTensor(); }
        if (merge_classes)
          for (int c2 = 1; c2 < 4; ++c2) {
            const SmallPlan s2 = small_geom(geom_convT4_s2(c2 >> 1, c2 & 1), th, tw, sp.pg, sp.tw_log2);
            if (!s2.ok || s2.tiles != sg.tiles || s2.win_h != sg.win_h || s2.win_w != sg.win_w || s2.lds != sg.lds) {
              if (error.empty()) error = "small conv geometry differs between parity classes";
              return FTensor();
            }
          }
        float inv = 1.f;
        size_t class_bytes = 0;
        const std::vector<char> p16 = pack16_classes(g, 1, &inv, &class_bytes);
        a.w16_class_bytes = (long long)class_bytes;
        w_off = add_param(p16.data(), p16.size());
        lds = ((table_bytes + 127) & ~(size_t)127) + sg.lds;
        a.w16_inv_scale = inv; a.wg_tw_log2 = sg.tw_log2; a.win_h = sg.win_h; a.win_w = sg.win_w;
        a.win_dy0 = sg.dy0; a.win_dx0 = sg.dx0; a.units_per_q = sp.units_per_q;
      } else if (point) {
        float inv = 1.f;
        const std::vector<char> p16 = pack_conv_weight16(weight.data, cout, cin, k, k, false, g, segs, tmb, &inv);
        w_off = add_param(p16.data(), p16.size());
        lds = point_lds_bytes(tmb, cbtot);
        a.w16_inv_scale = inv;
      } else if (tiled) {
        const TileGeom tg = nw4 ? tile_geom(g, th, tw, pg, tmb, twl4, table_bytes, 4) : tile_geom(g, th, tw, pg, tmb, plan.geom.tw_log2, table_bytes);
        if (merge_classes)
          for (int c2 = 1; c2 < 4; ++c2) {
            const ConvGeom g2 = geom_convT4_s2(c2 >> 1, c2 & 1);
            const TileGeom t2 = nw4 ? tile_geom(g2, th, tw, pg, tmb, twl4, table_bytes, 4) : tile_geom(g2, th, tw, pg, tmb, plan.geom.tw_log2, table_bytes);
            if (!t2.ok || t2.tiles != tg.tiles || t2.win_h != tg.win_h || t2.win_w != tg.win_w || t2.taps_per_chunk != tg.taps_per_chunk ||
                t2.ring_slots != tg.ring_slots || t2.win_buffers != tg.win_buffers || t2.lds != tg.lds) {
              if (error.empty()) error = "conv tile geometry differs between parity classes";
              return FTensor();
            }
          }
        float inv = 1.f;
        size_t class_bytes = 0;
        const std::vector<char> p16 = pack16_classes(g, tmb, &inv, &class_bytes);
        a.w16_class_bytes = (long long)class_bytes;
        w_off = add_param(p16.data(), p16.size());
        lds = tg.lds;
        a.w16_inv_scale = inv; a.wg_tw_log2 = tg.tw_log2; a.win_h = tg.win_h; a.win_w = tg.win_w;
        a.win_dy0 = tg.dy0; a.win_dx0 = tg.dx0; a.taps_per_chunk = tg.taps_per_chunk; a.ring_slots = tg.ring_slots;
        a.win_buffers = tg.win_buffers;
        if (!tg.ok || tg.tiles != tiles) { if (error.empty()) error = "conv tile geometry differs between parity classes"; return FTensor(); }
      } else {
        w_off = add_param(pack_conv_weight(weight.data, cout, cin, k, k, kind == K_CONVT, g, segs, tmb));
        cq = std::max(1, 32 / (g.ntaps * tmb));
        cq = std::min(cq, cbtot);
        lds = 2 * (size_t)cq * g.ntaps * tmb * 1024 + 4 * tmb * 16 * 2 * sizeof(float);
      }
      a.nsrc = (int)srcs.size();
      a.in_h = ih; a.in_w = iw; a.in_mode = in_mode;
      a.ntaps = g.ntaps;
      for (int t = 0; t < g.ntaps; ++t) { a.tap_dy[t] = (signed char)g.dy[t]; a.tap_dx[t] = (signed char)g.dx[t]; }
      a.in_stride = g.in_stride;
      a.tile_h = th; a.tile_w = tw; a.out_h = oh; a.out_w = ow;
      a.out_sy = g.out_sy; a.out_sx = g.out_sx; a.out_oy = g.out_oy; a.out_ox = g.out_ox;
      a.res_mode = res_mode;
      a.stats_tiles = out.stats_tiles; a.stats_tile0 = cls * tiles;
      a.nb = nb; a.chunk_quads = cq;
      if (small || tiled) {                                 // launch constants of the prologues (FastDiv reciprocals, full_kernels.h)
        a.ksplit = ksplit;
        ConvArgs probe = a;
        probe.batch = max_batch;
        const long long gx = (long long)max_batch * tiles * (small ? nb : 1) * grid_classes;
        if (!finish_conv_args(a, 16 * pg * (small ? 1 : nw4 ? 4 : 8), (cbtot + 1) / 2) || a.tiles_per_frame != tiles ||
            (probe.tiles_per_frame = a.tiles_per_frame, !finish_conv_batch(probe, small, gx))) {
          if (error.empty()) error = "internal: convolution launch constants out of range";
          return FTensor();
        }
      }
      std::vector<Src> sv = srcs;
      const Pending fp = fpend ? *fpend : Pending();
      // which moments the consumer folds: the accumulators when it must (too many tiles for the per-tile route) or when it can and they are fewer loads
      const bool use_acc = fpend && (fpend->acc_only || (fpend->has_acc && !small && fpend->total_tiles > acc_min_tiles()));
      if (fpend && use_acc) ++n_conv_acc;
      const FTensor outc = out;
      const bool has_res = residual != nullptr;
      const FTensor resc = has_res ? *residual : FTensor();
      ops.push_back([=](const Frame& f) {
        ConvArgs c = a;
        for (size_t i = 0; i < sv.size(); ++i) {
          const Src& s = sv[i];
          c.src[i].kind = s.vector ? SRC_VECTOR : SRC_TENSOR;
          c.src[i].act = s.vector ? ACT_NONE : act_in;     // the pose is concatenated AFTER the activation (poser_encoder_decoder_00.py:108-113)
          c.src[i].cb = s.vector ? s.vec_cb : s.t.cb;
          c.src[i].data = Wk(s.vector ? s.vec_off : s.t.off);
          c.src[i].scale = s.pend.has() ? Wk(s.pend.scale_off) : nullptr;
          c.src[i].shift = s.pend.has() ? Wk(s.pend.shift_off) : nullptr;
        }
        if (fp.fused) {
          FusedNorm& fn = c.fnorm;
          fn.enabled = 1;
          for (int i = 0; i < 2; ++i) {
            fn.stats[i] = !fp.tiles[i] ? nullptr : use_acc ? reinterpret_cast<const float*>(Acc(fp.acc_off[i])) : Wk(fp.stats_off[i]);
            fn.tiles[i] = !fp.tiles[i] ? 0 : use_acc ? kMomentShards : fp.tiles[i];
          }
          fn.acc = use_acc ? 1 : 0;
          fn.channels = fp.channels; fn.groups = fp.groups; fn.inv_count = fp.inv_count; fn.eps = 1e-5f;
          fn.gamma = P(fp.gamma_off); fn.beta = P(fp.beta_off);
          fn.film0 = fp.film0_off == kNone ? nullptr : P(fp.film0_off);
          fn.film1 = fp.film1_off == kNone ? nullptr : Wk(fp.film1_off);
          fn.film1_stride = fp.film1_stride;
          fn.fault = fault;
        }
        c.w = P(w_off);
        c.w16 = P<char>(w_off);
        c.partial = ksplit > 1 ? Wk(partial_off) : nullptr;
#ifdef THA4_PHASE_TIMING
        c.dbg = (std::getenv("THA4_DBG_CONV") && std::atoi(std::getenv("THA4_DBG_CONV")) == conv_index) ? reinterpret_cast<long long*>(Wk(dbg_off)) : nullptr;
#endif
        c.ksplit = ksplit;
        c.bias = bias_off == kNone ? nullptr : P(bias_off);
        c.act_out = act_off == kNone ? nullptr : P<int>(act_off);
        c.residual = has_res ? Wk(resc.off) : nullptr;
        c.out = Wk(outc.off);
        c.stats = outc.stats_tiles ? Wk(outc.stats_off) : nullptr;
        c.stats_acc = outc.acc_off != kNone ? Acc(outc.acc_off) : nullptr;
        c.acc_fault = fault;
        c.batch = f.batch;
        if (small || tiled) finish_conv_batch(c, small, 0);
        if (small) {
          dispatch_small(pg, in_mode, c, dim3(f.batch * tiles * nb * grid_classes, 1, 1), lds, f.stream);
        } else if (point) {
          dispatch_point(tmb, pg, c, dim3(f.batch * tiles * mtiles, 1, 1), lds, f.stream);
        } else if (splitk) {
          const dim3 grid(f.batch * tiles, mtiles);
          if (in_mode == IN_DIRECT) hipLaunchKernelGGL((conv_splitk_kernel<4, IN_DIRECT>), grid, dim3(256), 16 * 1024, f.stream, c);
          else if (in_mode == IN_UP2) hipLaunchKernelGGL((conv_splitk_kernel<4, IN_UP2>), grid, dim3(256), 16 * 1024, f.stream, c);
          else hipLaunchKernelGGL((conv_splitk_kernel<4, IN_POOL2>), grid, dim3(256), 16 * 1024, f.stream, c);
        } else if (tiled) {
          if (ksplit > 1) {
            c.phase = 1;
            dispatch_tile(tmb, pg, in_mode, c, dim3(f.batch * tiles * grid_classes, mtiles, ksplit), lds, f.stream);
            c.phase = 2;                       // one output block per workgroup: 4x the workgroups, a quarter of the load rounds each
            dispatch_tile(1, pg, in_mode, c, dim3(f.batch * tiles * grid_classes, mtiles * tmb, 1), lds, f.stream);
          } else {
            const long long gxl = (long long)f.batch * tiles * grid_classes;
            if (xcd_remap_enabled() && finish_conv_remap(c, mtiles, gxl))
              dispatch_tile(tmb, pg, in_mode, c, dim3((unsigned)(gxl * mtiles), 1, 1), lds, f.stream, nw4);
            else
              dispatch_tile(tmb, pg, in_mode, c, dim3(f.batch * tiles * grid_classes, mtiles, 1), lds, f.stream, nw4);
          }
        } else {
          dispatch_conv(tmb, pg, in_mode, c, dim3(f.batch * tiles, mtiles), lds, f.stream);
        }
      });
      {
        static const char* kinds[] = {"conv3x3", "conv1x1", "conv4x4 stride 2", "convT4x4 stride 2"};
        static const char* modes[] = {"", " (nearest x2 on load)", " (2x2 mean on load)"};
        char buf[160];
        std::snprintf(buf, sizeof buf, "%s %dx%d%s cin=%d cout=%d [%s]", kinds[(int)kind], oh, ow, modes[in_mode], cin, cout,
                      small ? "conv_small_kernel" : point ? "conv_point_kernel" : tiled ? (ksplit > 1 ? "conv_tile_kernel, K split" : "conv_tile_kernel")
                            : splitk ? "conv_splitk_kernel" : "conv_mfma_kernel");
        // as written: Conv2d 2 * out_px * cout * cin * k * k; ConvTranspose2d(4, stride 2) 2 * in_px * cin * cout * 16 - shared evenly by the launches of its classes
        const double gf = kind == K_CONVT ? 2.0 * vh * vw * (double)cin * cout * 16 / launch_classes : 2.0 * oh * ow * (double)cout * cin * k * k;
        note(ops, buf, gf / 1e9);
      }
    }
    return out;
  }

  int n_norm_finalize = 0, n_norm_tiles = 0, n_norm_acc = 0, n_conv_acc = 0;      // normalisations by route (THA4_DUMP_SCHEDULE prints them)
  // a normalisation over more than this many tiles (all sources together) is folded from the moment accumulators when every source has them: the per-tile
  // route reads 8 tiles per dependent round (17-64 tiles: 3-8 rounds), the accumulators are one round whatever the tile count
  static int acc_min_tiles() { return tune_env("THA4_ACC_MIN_TILES") ? std::atoi(tune_env("THA4_ACC_MIN_TILES")) : 64; }
  static int fused_norm_max_tiles() { return tune_env("THA4_FUSED_NORM_MAX_TILES") ? std::atoi(tune_env("THA4_FUSED_NORM_MAX_TILES")) : 64; }
  static int fused_norm_max_batch() { return tune_env("THA4_FUSED_NORM_MAX_BATCH") ? std::atoi(tune_env("THA4_FUSED_NORM_MAX_BATCH")) : 2; }
  // Normalisation finalize over up to two concatenated tensors; returns the pending transform per source.
  std::vector<Pending> norm(std::vector<Op>& ops, const std::vector<FTensor>& srcs, int channels, int groups,
                            const HostTensor& gamma, const HostTensor& beta, size_t film0_off = kNone,
                            size_t film1_off = kNone, long long film1_stride = 0, bool consumers_read_acc = true) {
    // consumers_read_acc = false: a consumer of this normalisation (affine_add_kernel) folds per-tile moments only - with the moment-accumulator tuning option
    // on, a tensor of more than 64 tiles then takes the finalize launch instead of an accumulator-only Pending nobody could evaluate (round-5 advisor finding)
    std::vector<Pending> out(srcs.size());
    const size_t g_off = add_param(gamma.data, sizeof(float) * channels);
    const size_t b_off = add_param(beta.data, sizeof(float) * channels);
    // few tiles: no finalize launch - every consumer reduces the per-tile moments itself (FusedNorm / FusedInstanceNorm)
    int total_tiles = 0;
    for (auto& t : srcs) total_tiles += t.stats_tiles;
    const int fuse_max = fused_norm_max_tiles();
    // (a batched call multiplies the consumers' workgroups, each of which would redo the reduction, while one finalize launch
    // serves all frames: fusing pays for max_batch <= 2 only - measured, profiles/r02_full_b1_reading.md)
    const int fuse_batch = fused_norm_max_batch();
    bool all_acc = consumers_read_acc && srcs.size() <= 2 && max_batch <= fuse_batch && acc_planned() && !tune_env("THA4_NO_SMALL_CONV");
    for (auto& t : srcs) all_acc = all_acc && t.acc_off != kNone;
    const bool per_tile_ok = total_tiles <= fuse_max && max_batch <= fuse_batch && srcs.size() <= 2 && !tune_env("THA4_NO_SMALL_CONV") && !tune_env("THA4_NO_TILE_CONV") &&
                             !exact_fp32;
    // folded by the consumer: from the per-tile moments (few tiles), from the moment accumulators its producers fill with atomics (kMomentShards entries per
    // channel: one round of loads whatever the tile count, no finalize launch), or - 17 to 64 tiles with accumulators - from whichever the consuming kernel reads
    // faster (conv_small_kernel folds per-tile moments only; FullModel::conv decides)
    if (per_tile_ok || all_acc) {
      Pending p;
      p.fused = true;
      p.has_acc = all_acc;
      p.acc_only = all_acc && !per_tile_ok;
      p.total_tiles = total_tiles;
      if (p.acc_only) ++n_norm_acc; else ++n_norm_tiles;
      for (size_t i = 0; i < srcs.size(); ++i) {
        p.stats_off[i] = srcs[i].stats_off; p.tiles[i] = srcs[i].stats_tiles;
        p.acc_off[i] = all_acc ? srcs[i].acc_off : 0;
      }
      p.channels = channels; p.groups = groups; p.inv_count = 1.0f / (float)(srcs[0].h * srcs[0].w);
      p.gamma_off = g_off; p.beta_off = b_off; p.film0_off = film0_off; p.film1_off = film1_off; p.film1_stride = film1_stride;
      for (auto& o : out) o = p;
      return out;
    }
    ++n_norm_finalize;
    if (std::getenv("THA4_DUMP_SCHEDULE")) {
      std::fprintf(stderr, "norm finalize launch: channels=%d groups=%d sources:", channels, groups);
      for (auto& t : srcs) std::fprintf(stderr, " [%dx%d cb=%d tiles=%d acc=%d]", t.h, t.w, t.cb, t.stats_tiles, (int)(t.acc_off != kNone));
      std::fprintf(stderr, "\n");
    }
    for (size_t i = 0; i < srcs.size(); ++i) {
      out[i].scale_off = alloc_work((size_t)srcs[i].cb * 16);
      out[i].shift_off = alloc_work((size_t)srcs[i].cb * 16);
    }
    const std::vector<FTensor> sv = srcs;
    const std::vector<Pending> pv = out;
    int cbt = 0;
    for (auto& s : srcs) cbt += s.cb;
    ops.push_back([=](const Frame& f) {
      NormArgs a{};
      a.nsrc = (int)sv.size();
      for (size_t i = 0; i < sv.size(); ++i) {
        a.stats[i] = Wk(sv[i].stats_off); a.tiles[i] = sv[i].stats_tiles; a.cb[i] = sv[i].cb;
        a.scale[i] = Wk(pv[i].scale_off); a.shift[i] = Wk(pv[i].shift_off);
      }
      a.channels = channels; a.groups = groups;
      a.inv_count = 1.0f / (float)(sv[0].h * sv[0].w);
      a.eps = 1e-5f;
      a.gamma = P(g_off); a.beta = P(b_off);
      a.film0 = film0_off == kNone ? nullptr : P(film0_off); a.film0_stride = 0;
      a.film1 = film1_off == kNone ? nullptr : Wk(film1_off); a.film1_stride = film1_stride;
      a.fault = fault;
      const int ctot = cbt * 16;
      int max_tiles = 0;
      for (auto& t : sv) max_tiles = std::max(max_tiles, t.stats_tiles);
      a.cpb = tune_env("THA4_NORM_ONE_WG") ? ctot : norm_channels_per_block(ctot, channels, groups, tune_env("THA4_NORM_TILE_SPLIT") ? max_tiles : 0);
      const int S = std::max(1, kNormThreads / a.cpb);
      hipLaunchKernelGGL(norm_finalize_kernel, dim3(f.batch, (ctot + a.cpb - 1) / a.cpb), dim3(kNormThreads),
                         ((size_t)S * a.cpb * 2 + 2 * a.cpb) * sizeof(double), f.stream, a);
    });
    note(ops, groups ? "GroupNorm finalize [norm_finalize_kernel]" : "InstanceNorm finalize [norm_finalize_kernel]");
    return out;
  }

  FTensor affine_add(std::vector<Op>& ops, const FTensor& A, Pending pa, int act_a, const FTensor& B, Pending pb) {
    if (pa.acc_only || pb.acc_only) {
      if (error.empty()) error = "internal: affine_add does not read moment accumulators";
      return FTensor();
    }
    if (((pa.fused && pa.groups) || (pb.fused && pb.groups) || (A.px() * 4) % 256 != 0) && (pa.fused || pb.fused)) {
      if (error.empty()) error = "internal: affine_add supports fused InstanceNorm on maps of >= 64 pixels only";
      return FTensor();
    }
    FTensor out = new_tensor(A.cb, A.h, A.w);
    ops.push_back([=](const Frame& f) {
      AffineAddArgs k{};
      auto mat = [&](const Pending& p) { return p.has() && !p.fused; };
      auto fin = [&](const Pending& p, FusedInstanceNorm& fi) {       // InstanceNorm only (resnet_block.py:52-67)
        if (!p.fused) return;
        fi.stats = Wk(p.stats_off[0]); fi.tiles = p.tiles[0]; fi.channels = p.channels; fi.inv_count = p.inv_count; fi.eps = 1e-5f;
        fi.gamma = P(p.gamma_off); fi.beta = P(p.beta_off); fi.fault = fault;
      };
      k.a = Wk(A.off); k.sa = mat(pa) ? Wk(pa.scale_off) : nullptr; k.ha = mat(pa) ? Wk(pa.shift_off) : nullptr; k.act_a = act_a;
      k.b = Wk(B.off); k.sb = mat(pb) ? Wk(pb.scale_off) : nullptr; k.hb = mat(pb) ? Wk(pb.shift_off) : nullptr;
      fin(pa, k.fa); fin(pb, k.fb);
      k.out = Wk(out.off); k.cb = A.cb; k.px = A.px();
      const size_t quads = (size_t)A.cb * A.px() * 4;
      hipLaunchKernelGGL(affine_add_kernel, dim3((unsigned)((quads + 255) / 256), f.batch), dim3(256), 256, f.stream, k);
    });
    note(ops, "ResnetBlock add [affine_add_kernel]");
    return out;
  }

  // ---- encoder-decoder trunk (poser_encoder_decoder_00.py:99-121 / face_morpher_08.py:159-168) ----
  // returns the last feature map (64 ch, raw) and its pending InstanceNorm+ReLU
  bool encdec(std::vector<Op>& ops, const WeightMap& w, const std::string& pre, const FTensor& input, int in_ch,
              size_t pose_vec_off, int pose_count, FTensor& feat, Pending& feat_pend) {
    auto W = [&](const std::string& k) -> const HostTensor& { return get(w, pre + k); };
    FTensor x = conv(ops, K_SAME3, {src_tensor(input, in_ch)}, IN_DIRECT, ACT_NONE, W("downsample_blocks.0.0.weight"), nullptr, 64, true);
    Pending px = norm(ops, {x}, 64, 0, W("downsample_blocks.0.1.weight"), W("downsample_blocks.0.1.bias"))[0];
    int c = 64;
    for (int i = 1; i < 4; ++i) {
      const std::string b = "downsample_blocks." + std::to_string(i);
      x = conv(ops, K_S2K4, {src_tensor(x, c, px)}, IN_DIRECT, ACT_RELU, W(b + ".0.weight"), nullptr, 2 * c, true);
      c *= 2;
      px = norm(ops, {x}, c, 0, W(b + ".1.weight"), W(b + ".1.bias"))[0];
    }
    std::vector<Src> s0 = {src_tensor(x, 512, px)};
    if (pose_count > 0) s0.push_back(src_vector(pose_vec_off, (pose_count + 15) / 16, pose_count));
    // "outer" mixed plan: the bottleneck (convolutions AND the normalisations between them) on the default fp16 hi/lo plan.  The normalisation in front of it
    // was planned exact (a finalize launch: every kernel can consume its scale / shift vectors), the one tensor leaving it is a plain sum (affine_add)
    struct Lower { bool& f; bool saved; ~Lower() { f = saved; } } lower{exact_fp32, exact_fp32};
    if (bottleneck_on_split) exact_fp32 = false;
    x = conv(ops, K_SAME3, s0, IN_DIRECT, ACT_RELU, W("bottleneck_blocks.0.0.weight"), nullptr, 512, true);
    px = norm(ops, {x}, 512, 0, W("bottleneck_blocks.0.1.weight"), W("bottleneck_blocks.0.1.bias"), kNone, kNone, 0, false)[0];      // (also read by affine_add)
    int act_x = ACT_RELU;       // x is "raw + pending IN/ReLU" after block 0, a plain tensor after every ResnetBlock
    for (int i = 1; i < 6; ++i) {
      const std::string b = "bottleneck_blocks." + std::to_string(i) + ".resnet_path.";
      FTensor r1 = conv(ops, K_SAME3, {src_tensor(x, 512, px)}, IN_DIRECT, act_x, W(b + "0.weight"), nullptr, 512, true);
      Pending p1 = norm(ops, {r1}, 512, 0, W(b + "1.weight"), W(b + "1.bias"))[0];
      FTensor r2 = conv(ops, K_SAME3, {src_tensor(r1, 512, p1)}, IN_DIRECT, ACT_RELU, W(b + "3.weight"), nullptr, 512, true);
      Pending p2 = norm(ops, {r2}, 512, 0, W(b + "4.weight"), W(b + "4.bias"), kNone, kNone, 0, false)[0];                            // (read by affine_add)
      x = affine_add(ops, x, px, act_x, r2, p2);      // x + resnet_path(x)   (resnet_block.py:63-67)
      px = Pending();
      act_x = ACT_NONE;
    }
    if (!(bottleneck_on_split && tune_env("THA4_DEC_DOWN_ONLY"))) exact_fp32 = lower.saved;      // (tuning aid: the up-sampling convolutions on the split plan too)
    for (int i = 0; i < 3; ++i) {
      const std::string b = "upsample_blocks." + std::to_string(i);
      x = conv(ops, K_CONVT, {src_tensor(x, c, px)}, IN_DIRECT, act_x, W(b + ".0.weight"), nullptr, c / 2, true);
      c /= 2;
      px = norm(ops, {x}, c, 0, W(b + ".1.weight"), W(b + ".1.bias"))[0];
      act_x = ACT_RELU;
    }
    feat = x;
    feat_pend = px;
    return error.empty();
  }

  // heads: several conv3(64 -> k) layers fused into ONE output block (poser_args.py:31-68)
  struct HeadSpec { std::string name; int channels; int act; bool bias; };
  FTensor heads(std::vector<Op>& ops, const WeightMap& w, const std::vector<HeadSpec>& hs, const FTensor& feat, Pending fp) {
    int total = 0;
    for (auto& h : hs) total += h.channels;
    std::vector<float> wcat((size_t)total * 64 * 9, 0.f), bcat(total, 0.f);
    std::vector<int> acts(total, 0);
    int at = 0;
    for (auto& h : hs) {
      const HostTensor& wt = get(w, h.name + ".weight");
      if (!expect(wt, {h.channels, 64, 3, 3}, h.name + ".weight")) return FTensor();
      std::memcpy(wcat.data() + (size_t)at * 64 * 9, wt.data, sizeof(float) * (size_t)h.channels * 64 * 9);
      if (h.bias) { const HostTensor& bt = get(w, h.name + ".bias"); if (bt.data) for (int i = 0; i < h.channels; ++i) bcat[at + i] = bt.data[i]; }
      for (int i = 0; i < h.channels; ++i) acts[at + i] = h.act;
      at += h.channels;
    }
    head_storage.push_back(std::move(wcat));
    HostTensor hw; hw.data = head_storage.back().data(); hw.dims = {total, 64, 3, 3};
    return conv(ops, K_SAME3, {src_tensor(feat, 64, fp)}, IN_DIRECT, ACT_RELU, hw, nullptr, total, false, nullptr, IN_DIRECT, &acts, &bcat);
  }
  std::vector<std::vector<float>> head_storage;

  // ---- U-Net (unet.py) ---------------------------------------------------------------------------
  struct UnetCfg { int in_ch, model; std::vector<int> mults; std::vector<bool> attn; };
  struct Feat { FTensor t; int channels; };

  // small dense layers evaluated on the host at create time (constant t = 0 branch, unet.py:365-376, morpher_00.py:51)
  static std::vector<float> host_linear(const HostTensor& W, const HostTensor& b, const std::vector<float>& x, bool silu_in) {
    const int rows = (int)W.dims[0], k = (int)W.dims[1];
    std::vector<float> y(rows);
    for (int r = 0; r < rows; ++r) {
      float s = 0.f;
      for (int i = 0; i < k; ++i) {
        float v = x[i];
        if (silu_in) v = v / (1.0f + std::exp(-v));
        s = std::fma(W.data[(size_t)r * k + i], v, s);
      }
      y[r] = s + b.data[r];
    }
    return y;
  }

  void gemv(std::vector<Op>& ops, size_t w_off, size_t b_off, int rows, int k, std::function<const float*(const Frame&)> x,
            long long x_stride, size_t y_off, int act_in, int act_out) {
    ops.push_back([=](const Frame& f) {
      GemvArgs a{P(w_off), P(b_off), x(f), Wk(y_off), rows, k, x_stride, act_in, act_out};
      hipLaunchKernelGGL(gemv_kernel, dim3((rows + 3) / 4, f.batch), dim3(256), 0, f.stream, a);
    });
    note(ops, "cond / FiLM linear [gemv_kernel]", 2.0 * rows * k / 1e9);
  }

  struct ResPlan { std::string p; int cin, cout; int mode; size_t film0_off; size_t film1_row; };

  Feat attention(std::vector<Op>& ops, const WeightMap& w, const std::string& p, const Feat& x) {
    const int C = x.channels;
    Pending pn = norm(ops, {x.t}, C, 32, get(w, p + ".norm.weight"), get(w, p + ".norm.bias"))[0];
    FTensor qkv = conv(ops, K_SAME1, {src_tensor(x.t, C, pn)}, IN_DIRECT, ACT_NONE, get(w, p + ".qkv.weight"),
                       get(w, p + ".qkv.bias").data, 3 * C, false);
    FTensor att = new_tensor(C / 16, x.t.h, x.t.w);
    const int tokens = x.t.px();
    ops.push_back([=](const Frame& f) {
      AttnArgs a{Wk(qkv.off), Wk(att.off), C, 8, tokens};
      hipLaunchKernelGGL(attention_kernel, dim3(8, f.batch, tokens / kAttnQueries), dim3(256), (size_t)2 * tokens * kAttnRow * sizeof(f32x4), f.stream, a);
    });
    note(ops, "attention core [attention_kernel]", 4.0 * tokens * tokens * C / 1e9);      // q.k and p.v: 2 x 2 x L x L x C
    FTensor o = conv(ops, K_SAME1, {src_tensor(att, C)}, IN_DIRECT, ACT_NONE, get(w, p + ".conv.weight"), get(w, p + ".conv.bias").data,
                     C, true, &x.t, IN_DIRECT);
    return Feat{o, C};
  }

  // ResBlock (unet.py:154-165) on the concatenation of `ins`; film1 points at this block's rows of the per-frame FiLM vector
  Feat resblock(std::vector<Op>& ops, const WeightMap& w, const std::string& p, const std::vector<Feat>& ins, int cout, int mode,
                const std::vector<float>& t_emb, size_t film1_base, long long film1_stride, size_t& film1_row) {
    int cin = 0;
    std::vector<FTensor> ts;
    for (auto& f : ins) { cin += f.channels; ts.push_back(f.t); }
    // skip branch (unet.py:149-152,165) first: it reads the block's inputs only
    FTensor res;
    int res_mode = mode;
    const bool has_skip = cin != cout;
    if (has_skip) {
      std::vector<Src> ss;
      for (auto& f : ins) ss.push_back(src_tensor(f.t, f.channels));
      res = conv(ops, K_SAME1, ss, IN_DIRECT, ACT_NONE, get(w, p + ".skip.weight"), get(w, p + ".skip.bias").data, cout, false);
      res_mode = IN_DIRECT;
    } else {
      res = ins[0].t;     // resampling blocks and same-width blocks have a single input
    }
    std::vector<Pending> p0 = norm(ops, ts, cin, 32, get(w, p + ".norm0.weight"), get(w, p + ".norm0.bias"));
    std::vector<Src> s0;
    for (size_t i = 0; i < ins.size(); ++i) s0.push_back(src_tensor(ins[i].t, ins[i].channels, p0[i]));
    FTensor h = conv(ops, K_SAME3, s0, mode, ACT_SILU, get(w, p + ".conv0.weight"), get(w, p + ".conv0.bias").data, cout, true);
    // FiLM 0 comes from the constant time embedding: evaluate once on the host
    std::vector<float> f0 = host_linear(get(w, p + ".cond0_layers.1.weight"), get(w, p + ".cond0_layers.1.bias"), t_emb, true);
    const size_t f0_off = add_param(f0);
    const size_t my_row = film1_row;
    film1_row += 2 * (size_t)cout;
    Pending p1 = norm(ops, {h}, cout, 32, get(w, p + ".norm1.weight"), get(w, p + ".norm1.bias"), f0_off, film1_base + my_row, film1_stride)[0];
    FTensor o = conv(ops, K_SAME3, {src_tensor(h, cout, p1)}, IN_DIRECT, ACT_SILU, get(w, p + ".conv1.weight"),
                     get(w, p + ".conv1.bias").data, cout, true, &res, res_mode);
    return Feat{o, cout};
  }

  // film rows of all ResBlocks of a U-Net in execution order (must mirror unet() below)
  struct UnetWalk { std::vector<std::string> res_prefix; std::vector<int> res_cout; };

  FTensor unet(std::vector<Op>& ops, const WeightMap& w, const std::string& pre, const UnetCfg& cfg, const std::vector<Src>& first_srcs,
               const HostTensor& first_w, const std::vector<float>& first_bias) {
    const int L = (int)cfg.mults.size();
    auto key = [&](const std::string& k) { return pre + k; };
    // constant time embedding: t = 0 -> [cos 0 .. | sin 0 ..] = [1.. | 0..]
    std::vector<float> tin(cfg.model, 0.f);
    for (int i = 0; i < cfg.model / 2; ++i) tin[i] = 1.f;
    std::vector<float> t1 = host_linear(get(w, key("time_embed.1.weight")), get(w, key("time_embed.1.bias")), tin, false);
    std::vector<float> t_emb = host_linear(get(w, key("time_embed.3.weight")), get(w, key("time_embed.3.bias")), t1, true);
    if (!error.empty()) return FTensor();
    // enumerate the ResBlocks to lay out the per-frame FiLM-1 vector, then emit: cond MLP + one gemv for all blocks
    std::vector<std::pair<std::string, int>> blocks;    // (prefix, cout) in execution order
    {
      int cur = cfg.model;
      for (int i = 0; i < L; ++i) {
        const int out = cfg.model * cfg.mults[i];
        blocks.push_back({key("down_blocks." + std::to_string(i) + ".res_blocks.0"), out});
        if (i < L - 1) blocks.push_back({key("down_blocks." + std::to_string(i) + ".downsample"), out});
        cur = out;
      }
      for (int k = 0; k < 4; ++k) blocks.push_back({key("middle_blocks." + std::to_string(2 * k)), cur});
      for (int bi = 0; bi < L; ++bi) {
        const int i = L - 1 - bi, out = cfg.model * cfg.mults[i];
        for (int j = 0; j < 2; ++j) blocks.push_back({key("up_blocks." + std::to_string(bi) + ".resnet_blocks." + std::to_string(j)), out});
        if (i > 0) blocks.push_back({key("up_blocks." + std::to_string(bi) + ".upsample"), out});
      }
    }
    size_t rows = 0;
    for (auto& b : blocks) rows += 2 * (size_t)b.second;
    std::vector<float> wall(rows * 256), ball(rows);
    {
      size_t r = 0;
      for (auto& b : blocks) {
        const HostTensor& wt = get(w, b.first + ".cond1_layers.1.weight");
        const HostTensor& bt = get(w, b.first + ".cond1_layers.1.bias");
        if (!expect(wt, {2 * b.second, 256}, b.first + ".cond1_layers.1.weight")) return FTensor();
        std::memcpy(wall.data() + r * 256, wt.data, sizeof(float) * (size_t)2 * b.second * 256);
        std::memcpy(ball.data() + r, bt.data, sizeof(float) * (size_t)2 * b.second);
        r += 2 * (size_t)b.second;
      }
    }
    const size_t c0w = add_param(get(w, key("cond_embed.0.weight")).data, sizeof(float) * 256 * 6);
    const size_t c0b = add_param(get(w, key("cond_embed.0.bias")).data, sizeof(float) * 256);
    const size_t c2w = add_param(get(w, key("cond_embed.2.weight")).data, sizeof(float) * 256 * 256);
    const size_t c2b = add_param(get(w, key("cond_embed.2.bias")).data, sizeof(float) * 256);
    const size_t fw = add_param(wall), fb = add_param(ball);
    const size_t h1 = alloc_work(256), cemb = alloc_work(256), film1 = alloc_work(rows);
    // cond MLP + all FiLM-1 projections (they read the pose only)
    gemv(ops, c0w, c0b, 256, 6, [](const Frame& f) { return f.pose + 39; }, 45, h1, ACT_NONE, ACT_SILU);   // rotation pose = pose[:, 39:45]
    gemv(ops, c2w, c2b, 256, 256, [=](const Frame&) { return (const float*)Wk(h1); }, 256, cemb, ACT_NONE, ACT_NONE);
    gemv(ops, fw, fb, (int)rows, 256, [=](const Frame&) { return (const float*)Wk(cemb); }, 256, film1, ACT_SILU, ACT_NONE);

    size_t row = 0;
    const long long fstride = (long long)rows;
    FTensor h0 = conv(ops, K_SAME3, first_srcs, IN_DIRECT, ACT_NONE, first_w, nullptr, cfg.model, true, nullptr, IN_DIRECT, nullptr, &first_bias);
    std::vector<Feat> hs = {Feat{h0, cfg.model}};
    Feat h = hs[0];
    for (int i = 0; i < L; ++i) {
      const int out = cfg.model * cfg.mults[i];
      const std::string b = key("down_blocks." + std::to_string(i));
      h = resblock(ops, w, b + ".res_blocks.0", {h}, out, IN_DIRECT, t_emb, film1, fstride, row);
      if (cfg.attn[i]) h = attention(ops, w, b + ".attention_blocks.0", h);
      hs.push_back(h);
      if (i < L - 1) {
        h = resblock(ops, w, b + ".downsample", {h}, out, IN_POOL2, t_emb, film1, fstride, row);
        hs.push_back(h);
      }
    }
    for (int k = 0; k < 3; ++k) {
      h = resblock(ops, w, key("middle_blocks." + std::to_string(2 * k)), {h}, h.channels, IN_DIRECT, t_emb, film1, fstride, row);
      h = attention(ops, w, key("middle_blocks." + std::to_string(2 * k + 1) + ".module"), h);
    }
    h = resblock(ops, w, key("middle_blocks.6"), {h}, h.channels, IN_DIRECT, t_emb, film1, fstride, row);
    for (int bi = 0; bi < L; ++bi) {
      const int i = L - 1 - bi, out = cfg.model * cfg.mults[i];
      const std::string b = key("up_blocks." + std::to_string(bi));
      for (int j = 0; j < 2; ++j) {
        Feat skip = hs.back();
        hs.pop_back();
        h = resblock(ops, w, b + ".resnet_blocks." + std::to_string(j), {h, skip}, out, IN_DIRECT, t_emb, film1, fstride, row);
        if (cfg.attn[i]) h = attention(ops, w, b + ".attention_blocks." + std::to_string(j), h);
      }
      if (i > 0) h = resblock(ops, w, b + ".upsample", {h}, out, IN_UP2, t_emb, film1, fstride, row);
    }
    if (!hs.empty() || row != rows) { if (error.empty()) error = "internal: U-Net walk mismatch"; return FTensor(); }
    Pending pl = norm(ops, {h.t}, cfg.model, 32, get(w, key("last.0.weight")), get(w, key("last.0.bias")))[0];
    return conv(ops, K_SAME3, {src_tensor(h.t, cfg.model, pl)}, IN_DIRECT, ACT_SILU, get(w, key("last.2.weight")),
                get(w, key("last.2.bias")).data, 7, false);
  }

  // ---- image-domain launches -----------------------------------------------------------------------
  template <class K>
  void image_op(std::vector<Op>& ops, K kernel, int pixels, std::function<void(const Frame&, ImgArgs&)> bind) {
    ops.push_back([=](const Frame& f) {
      ImgArgs a{};
      a.image = f.image; a.image_stride = f.image_stride; a.pose = f.pose; a.batch = f.batch; a.sel = sel_index; a.fault = fault;
      bind(f, a);
      hipLaunchKernelGGL(kernel, dim3((pixels + 255) / 256, f.batch), dim3(256), 0, f.stream, a);
    });
    note(ops, "image-domain kernel (crop / paste / warp / blend)");
  }

  // ---- whole pipeline ------------------------------------------------------------------------------
  // Output order (mode_07.py:126-132): upscaler 0-4, face_morphed_full 5, body 6-10, face 11-18, combiner 19-26, decomposer 27-32
  // num_nets = 5: mode_07 (all 33 outputs); num_nets = 3: mode_12 (mode_12.py:42-97: eyebrow_decomposer ->
  // eyebrow_morphing_combiner -> face_morpher only; outputs 11..32 of the list below exist, 0..10 are never written)
  int num_networks = 5;
  bool build(const WeightMap nets[5], int max_batch_, int sel, int num_nets = 5, bool exact = false) {
    max_batch = max_batch_;
    exact_fp32 = exact;
    sel_index = sel;
    num_networks = num_nets;
    static const int och[33] = {4, 1, 4, 2, 4, 4, 4, 1, 4, 2, 4, 4, 1, 4, 4, 1, 4, 4, 2, 4, 1, 4, 4, 1, 4, 4, 2, 4, 1, 4, 4, 1, 4};
    static const int osz[33] = {512, 512, 512, 512, 512, 512, 256, 256, 256, 256, 256, 192, 192, 192, 192, 192, 192, 192, 192,
                                128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};
    for (int i = 0; i < 33; ++i) { out_ch[i] = och[i]; out_size[i] = osz[i]; scratch_out[i] = alloc_work((size_t)och[i] * osz[i] * osz[i]); }
    const size_t pose_eb = alloc_work(16), pose_face = alloc_work(32);

    // 1. eyebrow decomposer (its outputs are cached by the caller while the image is unchanged, mode_07.py:56-67)
    {
      struct Raise { bool& f; bool saved; ~Raise() { f = saved; } } raise{exact_fp32, exact_fp32};      // mixed plan: this network only on the exact-fp32 kernels
      struct Flag { bool& f; ~Flag() { f = false; } } lower{bottleneck_on_split};
      bottleneck_on_split = !exact_fp32 && !exact_decomposer && exact_decomposer_outer;
      exact_fp32 = exact_fp32 || exact_decomposer || exact_decomposer_outer;
      auto& ops = ops_decomposer;
      FTensor x = new_tensor(1, 128, 128);
      image_op(ops, crop_eyebrow_kernel, 128 * 128, [=](const Frame&, ImgArgs& a) { a.c16_out = Wk(x.off); });
      FTensor feat; Pending fp;
      if (!encdec(ops, nets[0], "body.", x, 4, 0, 0, feat, fp)) return false;
      if (bottleneck_on_split && tune_env("THA4_DEC_DOWN_ONLY")) exact_fp32 = false;                 // (... and the heads)
      FTensor hd = heads(ops, nets[0], {{"background_layer_alpha.0", 1, ACT_SIGMOID, true}, {"background_layer_color_change.0", 4, ACT_TANH, true},
                                        {"eyebrow_layer_alpha.0", 1, ACT_SIGMOID, true}, {"eyebrow_layer_color_change.0", 4, ACT_TANH, true}}, feat, fp);
      comb_in = new_tensor(1, 128, 128);
      const FTensor ci = comb_in;
      // the six decomposer outputs live in persistent workspace buffers: they are reused by later frames
      // while the image is unchanged (the reference caches them the same way, mode_07.py:56-67)
      for (int i = 0; i < 6; ++i) dec_persist[i] = alloc_work((size_t)och[27 + i] * 128 * 128);
      image_op(ops, decomposer_tail_kernel, 128 * 128, [=](const Frame&, ImgArgs& a) {
        a.head = Wk(hd.off); a.c16_out = Wk(ci.off);
        for (int i = 0; i < 6; ++i) a.out[i] = Wk(dec_persist[i]);
      });
    }
    auto& ops = ops_rest;
    ops.push_back([=](const Frame& f) {
      hipLaunchKernelGGL(pose_pad_kernel, dim3(f.batch), dim3(64), 0, f.stream, f.pose, Wk(pose_eb), Wk(pose_face), f.batch);
    });
    note(ops, "pose padding [pose_pad_kernel]");
    // 2. eyebrow morphing combiner
    {
      FTensor feat; Pending fp;
      if (!encdec(ops, nets[1], "body.", comb_in, 8, pose_eb, 12, feat, fp)) return false;
      FTensor hd = heads(ops, nets[1], {{"morphed_eyebrow_layer_grid_change", 2, ACT_NONE, false}, {"morphed_eyebrow_layer_alpha.0", 1, ACT_SIGMOID, true},
                                        {"morphed_eyebrow_layer_color_change.0", 4, ACT_TANH, true}, {"combine_alpha.0", 1, ACT_SIGMOID, true}}, feat, fp);
      image_op(ops, combiner_tail_kernel, 128 * 128, [=](const Frame& f, ImgArgs& a) {
        a.head = Wk(hd.off); a.in0 = Wk(dec_persist[0]); a.in1 = Wk(dec_persist[3]);
        for (int i = 0; i < 8; ++i) a.out[i] = f.out[19 + i];
      });
    }
    // 3. face morpher
    const size_t face_in_nchw = alloc_work((size_t)4 * 192 * 192);
    {
      FTensor x = new_tensor(1, 192, 192);
      image_op(ops, face_input_kernel, 192 * 192, [=](const Frame& f, ImgArgs& a) {
        a.in0 = f.out[19 + sel_index]; a.out[0] = Wk(face_in_nchw); a.c16_out = Wk(x.off);
      });
      FTensor feat; Pending fp;
      if (!encdec(ops, nets[2], "", x, 4, pose_face, 27, feat, fp)) return false;
      FTensor hd = heads(ops, nets[2], {{"iris_mouth_grid_change", 2, ACT_NONE, false}, {"iris_mouth_color_change.0", 4, ACT_TANH, true},
                                        {"iris_mouth_alpha.0", 1, ACT_SIGMOID, true}, {"eye_color_change.0", 4, ACT_TANH, true},
                                        {"eye_alpha.0", 1, ACT_SIGMOID, true}}, feat, fp);
      image_op(ops, face_tail_kernel, 192 * 192, [=](const Frame& f, ImgArgs& a) {
        a.head = Wk(hd.off); a.in0 = Wk(face_in_nchw);
        for (int i = 0; i < 8; ++i) a.out[i] = f.out[11 + i];
      });
    }
    if (num_nets == 3) {
      head_storage.clear();
      finalize_scratch();
      return error.empty();
    }
    // face_morphed_full / half (mode_07.py:93-103)
    image_op(ops, paste_face_kernel, 512 * 512, [=](const Frame& f, ImgArgs& a) { a.in0 = f.out[11]; a.out[0] = f.out[5]; });
    const size_t half_nchw = alloc_work((size_t)4 * 256 * 256);
    FTensor half = new_tensor(1, 256, 256);
    image_op(ops, half_image_kernel, 256 * 256, [=](const Frame& f, ImgArgs& a) { a.in0 = f.out[5]; a.out[0] = Wk(half_nchw); a.c16_out = Wk(half.off); });
    // 4. body morpher
    {
      UnetCfg cfg{4, 64, {1, 2, 4, 4, 4}, {false, false, false, false, true}};
      const HostTensor& fw = get(nets[3], "body.first_conv.weight");
      const HostTensor& fb = get(nets[3], "body.first_conv.bias");
      if (!expect(fw, {64, 4, 3, 3}, "body.first_conv.weight") || !fb.data) return false;
      std::vector<float> bias(fb.data, fb.data + 64);
      FTensor hd = unet(ops, nets[3], "body.", cfg, {src_tensor(half, 4)}, fw, bias);
      if (!error.empty()) return false;
      image_op(ops, unet_tail_kernel<256>, 256 * 256, [=](const Frame& f, ImgArgs& a) {
        a.head = Wk(hd.off); a.in0 = Wk(half_nchw);
        for (int i = 0; i < 5; ++i) a.out[i] = f.out[6 + i];
      });
    }
    // 5. upscaler: first_conv(rest) + coarse_image_conv(cat[coarse image, warped rest, coarse grid]) as ONE 14-channel conv
    {
      UnetCfg cfg{4, 32, {1, 2, 4, 8, 8, 8}, {false, false, false, false, false, true}};
      FTensor x = new_tensor(1, 512, 512);
      image_op(ops, upscaler_input_kernel, 512 * 512, [=](const Frame& f, ImgArgs& a) {
        a.in0 = f.out[5]; a.in1 = f.out[6]; a.in2 = f.out[9]; a.c16_out = Wk(x.off);
      });
      const HostTensor& fw = get(nets[4], "body.first_conv.weight");
      const HostTensor& fb = get(nets[4], "body.first_conv.bias");
      const HostTensor& cw = get(nets[4], "coarse_image_conv.weight");
      const HostTensor& cb = get(nets[4], "coarse_image_conv.bias");
      if (!expect(fw, {32, 4, 3, 3}, "body.first_conv.weight") || !expect(cw, {32, 10, 3, 3}, "coarse_image_conv.weight") || !fb.data || !cb.data) return false;
      head_storage.push_back(std::vector<float>((size_t)32 * 14 * 9));
      std::vector<float>& wc = head_storage.back();
      for (int o = 0; o < 32; ++o)
        for (int i = 0; i < 14; ++i)
          for (int t = 0; t < 9; ++t)
            wc[((size_t)o * 14 + i) * 9 + t] = i < 4 ? fw.data[((size_t)o * 4 + i) * 9 + t] : cw.data[((size_t)o * 10 + (i - 4)) * 9 + t];
      HostTensor hw; hw.data = wc.data(); hw.dims = {32, 14, 3, 3};
      std::vector<float> bias(32);
      for (int o = 0; o < 32; ++o) bias[o] = fb.data[o] + cb.data[o];
      FTensor hd = unet(ops, nets[4], "body.", cfg, {src_tensor(x, 14)}, hw, bias);
      if (!error.empty()) return false;
      image_op(ops, unet_tail_kernel<512>, 512 * 512, [=](const Frame& f, ImgArgs& a) {
        a.head = Wk(hd.off); a.in0 = f.out[5];
        for (int i = 0; i < 5; ++i) a.out[i] = f.out[i];
        a.rgba8 = f.rgba8; a.rgba8_has_bg = f.rgba8_has_bg;
        for (int k = 0; k < 3; ++k) a.rgba8_bg[k] = f.rgba8_bg[k];
      });
    }
    head_storage.clear();
    finalize_scratch();
    return error.empty();
  }

  FTensor comb_in;
  size_t dec_persist[6];

  // `want_dec[i]`: the caller asked for decomposer output i (copied out of the persistent buffers)
  // per-op timing (ABI v5): `timing_events` holds ops_decomposer.size() + ops_rest.size() + 2 events when enabled; event i is recorded on the launch stream in
  // FRONT of op i (decomposer ops first, then one separator, then the rest) and one more behind the last op
  std::vector<hipEvent_t> timing_events;
  bool timing_on = false, timing_recorded = false, timing_ran_decomposer = false;
  void run(const Frame& f, bool run_decomposer, const bool want_dec[6]) {
    if (acc_floats) (void)hipMemsetAsync(dev_acc, 0, acc_floats * sizeof(float), f.stream);      // the moment accumulators of this call's producers
    if (timing_on && timing_events.size() == ops_decomposer.size() + ops_rest.size() + 2) return run_timed(f, run_decomposer, want_dec);
    if (run_decomposer)
      for (auto& op : ops_decomposer) op(f);
    for (int i = 0; i < 6; ++i)
      if (want_dec[i])
        (void)hipMemcpyAsync(f.out[27 + i], Wk(dec_persist[i]), sizeof(float) * (size_t)f.batch * out_ch[27 + i] * 128 * 128,
                             hipMemcpyDeviceToDevice, f.stream);
    for (auto& op : ops_rest) op(f);
  }
  void run_timed(const Frame& f, bool run_decomposer, const bool want_dec[6]) {
    size_t e = 0;
    const size_t nd = ops_decomposer.size();
    for (size_t i = 0; i < nd; ++i) {
      (void)hipEventRecord(timing_events[e++], f.stream);
      if (run_decomposer) ops_decomposer[i](f);
    }
    (void)hipEventRecord(timing_events[e++], f.stream);      // separator: end of the decomposer ops (the copies below belong to nobody)
    for (int i = 0; i < 6; ++i)
      if (want_dec[i])
        (void)hipMemcpyAsync(f.out[27 + i], Wk(dec_persist[i]), sizeof(float) * (size_t)f.batch * out_ch[27 + i] * 128 * 128,
                             hipMemcpyDeviceToDevice, f.stream);
    for (size_t i = 0; i < ops_rest.size(); ++i) {
      (void)hipEventRecord(timing_events[e++], f.stream);
      ops_rest[i](f);
    }
    // (e == nd + 1 + ops_rest.size(): one event left for the end)
    (void)hipEventRecord(timing_events[e], f.stream);
    timing_recorded = true;
    timing_ran_decomposer = run_decomposer;
  }
};

}  // namespace tha4
