// C-ABI implementation (include/tha4_hip.h) of the student poser path for MI355X (gfx950).
// Host side only: weight packing/upload, workspace, and the 5-launch frame schedule.  The kernels
// live in siren_kernels.h.  No CPU fallback exists: without a gfx950 device create() fails.
#include "tha4_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "siren16_kernels.h"
#include "full_net.h"
#include "image_io_kernels.h"

using namespace tha4;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(THA4_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

constexpr int kNumKernels = 5;

const char* const kNumericFaultMessage =
    "numeric fault in an earlier tha4_full_pose on this handle: a normalisation scale/shift or a network's head block was not "
    "finite - a normalised + activated convolution operand left the fp16 hi/lo range (|v| >= 65520 is staged as inf) or the "
    "weights / inputs contain NaN or inf; the outputs of that call are invalid";

// One workspace per handle: two pose calls on DIFFERENT streams would race on it.  Calls on the same stream are ordered
// by the stream; when the stream changes, the event `done` (created with the handle: pose never allocates) is recorded on
// the previous stream - at that moment, so it covers the previous call and anything enqueued there since - and waited for
// by the new one.  Nothing is recorded on the common single-stream path (an event per frame would put a signal packet
// between the frames of a 160-us stream).  Contract (tha4_hip.h): a stream given to a pose call must stay valid until
// the handle's NEXT pose call has returned.
struct StreamOrder {
  hipStream_t last = nullptr;
  bool used = false;
  hipEvent_t done = nullptr;
  hipError_t create() { return hipEventCreateWithFlags(&done, hipEventDisableTiming); }
  hipError_t enter(hipStream_t s) {
    hipError_t e = hipSuccess;
    if (used && s != last && done) {
      e = hipEventRecord(done, last);
      if (e == hipSuccess) e = hipStreamWaitEvent(s, done, 0);
    }
    last = s;
    used = true;
    return e;
  }
  void destroy() {
    if (done) (void)hipEventDestroy(done);
    done = nullptr;
  }
};

// device that owns a device pointer (-1: unknown) - the stateless image entry points have no handle to ask
int device_of(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return a.device;
}

}  // namespace

struct tha4_student {
  int device = 0;
  int max_batch = 0;
  char* blob = nullptr;       // packed parameters, one allocation
  char* workspace = nullptr;  // pbias | face | z1 | z2
  StudentDev dev{};           // parameter + workspace pointers filled at create
  bool exact_fp32 = false;    // generation 1 kernels (v_mfma_f32_16x16x4_f32) instead of the fp16 hi/lo split
  bool timing = false;
  hipEvent_t ev[kNumKernels + 1] = {};
  bool ev_valid = false;
  bool ev_recorded = false;
  size_t blob_bytes = 0;
  std::vector<float> pos128, pos256, pos512;   // position axes given at create (reused by tha4_student_set_weights)
  StreamOrder order;                           // the handle's workspace is shared by consecutive calls
#ifdef THA4_L2_HOOK
  // fault-hunt aid (profiles/r03_sin_cliff.md, tools/hunt/): level 2 from an externally assembled code object instead of the built-in kernel
  // (env THA4_L2_CODE_OBJECT = path, THA4_L2_KERNEL = mangled name, THA4_L2_THREADS, THA4_L2_PX = pixels per workgroup).  Only in builds
  // with -DTHA4_L2_HOOK: the shipped library never loads code from outside itself
  hipModule_t l2_module = nullptr;
  hipFunction_t l2_function = nullptr;
  int l2_threads = 0, l2_px = 0;
#endif
};

namespace {

template <class T>
size_t align_up(T v, size_t a) { return ((size_t)v + a - 1) / a * a; }

struct BlobBuilder {
  std::vector<char> host;
  size_t add(const std::vector<float>& v) {
    size_t at = align_up(host.size(), 256);
    host.resize(at + v.size() * sizeof(float));
    std::memcpy(host.data() + at, v.data(), v.size() * sizeof(float));
    return at;
  }
  size_t add(const std::vector<char>& v) {
    size_t at = align_up(host.size(), 256);
    host.resize(at + v.size());
    std::memcpy(host.data() + at, v.data(), v.size());
    return at;
  }
};

StudentWeightsView to_view(const tha4_student_weights* w) {
  StudentWeightsView v{};
  auto cv = [](const tha4_linear& l) { return LinearView{l.weight, l.bias, l.out_ch, l.in_ch}; };
  for (int i = 0; i < 8; ++i) v.face_sine[i] = cv(w->face_sine[i]);
  v.face_last = cv(w->face_last);
  for (int l = 0; l < 3; ++l)
    for (int j = 0; j < 3; ++j) v.body_sine[l][j] = cv(w->body_sine[l][j]);
  v.body_last = cv(w->body_last);
  return v;
}

template <class K>
hipError_t allow_lds(K kernel, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace


namespace {
struct StudentBlobOffsets {
  size_t wf, w0, w1, w2, bf, b0, b1, b2, sf = 0, s0 = 0, s1 = 0, s2 = 0;
  size_t wx[4], wy[4], b[4], wp[4], p128, p256, p512;
};

// pack the two state_dicts into the parameter blob (layout depends on the architecture only, never on the values:
// tha4_student_set_weights overwrites the blob of a live handle in place)
std::string build_student_blob(const tha4_student_weights* weights, bool exact, const std::vector<float>& pos128,
                               const std::vector<float>& pos256, const std::vector<float>& pos512, BlobBuilder& bb,
                               StudentBlobOffsets& o) {
  StudentPacked p;
  std::string err = pack_student(to_view(weights), p);
  if (!err.empty()) return err;
  if (!exact && THA4_SIN_TURNS) {           // domain of the one-instruction sine (siren_layout.h sine_argument_bound_turns)
    const double bound = sine_argument_bound_turns(to_view(weights));
    if (!(bound < kSineTurnsLimit)) {
      char msg[256];
      std::snprintf(msg, sizeof msg, "a sine layer's argument can reach %.3g turns (limit %.0f: the default kernels' v_sin_f32 returns 0 beyond it); "
                    "create the poser with THA4_STUDENT_EXACT_FP32 (exact_fp32=True) for these weights", bound, kSineTurnsLimit);
      return msg;
    }
  }
  const FirstLayerPack* fl[4] = {&p.f_face, &p.f_l0, &p.f_l1, &p.f_l2};
  if (exact) {
    o.wf = bb.add(p.w_face); o.w0 = bb.add(p.w_l0); o.w1 = bb.add(p.w_l1); o.w2 = bb.add(p.w_l2);
    o.bf = bb.add(p.b_face); o.b0 = bb.add(p.b_l0); o.b1 = bb.add(p.b_l1); o.b2 = bb.add(p.b_l2);
    for (int i = 0; i < 4; ++i) { o.wx[i] = bb.add(fl[i]->wx); o.wy[i] = bb.add(fl[i]->wy); }
  } else {
    v2::StudentPacked16 p16;
    v2::pack_student16(to_view(weights), p, p16);
    o.wf = bb.add(p16.w_face); o.w0 = bb.add(p16.w_l0); o.w1 = bb.add(p16.w_l1); o.w2 = bb.add(p16.w_l2);
    o.bf = bb.add(p16.b_face); o.b0 = bb.add(p16.b_l0); o.b1 = bb.add(p16.b_l1); o.b2 = bb.add(p16.b_l2);
    o.sf = bb.add(p16.s_face); o.s0 = bb.add(p16.s_l0); o.s1 = bb.add(p16.s_l1); o.s2 = bb.add(p16.s_l2);
    for (int i = 0; i < 4; ++i) { o.wx[i] = bb.add(p16.wx[i]); o.wy[i] = bb.add(p16.wy[i]); }
  }
  for (int i = 0; i < 4; ++i) {
    o.b[i] = bb.add(fl[i]->bias);
    o.wp[i] = bb.add(fl[i]->wpose);
  }
  o.p128 = bb.add(pos128); o.p256 = bb.add(pos256); o.p512 = bb.add(pos512);
  return std::string();
}
}  // namespace

extern "C" {

int tha4_abi_version(void) { return THA4_ABI_VERSION; }

const char* tha4_last_error(void) { return g_last_error.c_str(); }

int tha4_student_create(const tha4_student_weights* weights, const tha4_position_axes* axes, int device,
                        int max_batch, tha4_student** out) {
  return tha4_student_create_ex(weights, axes, device, max_batch, 0, out);
}

int tha4_student_create_ex(const tha4_student_weights* weights, const tha4_position_axes* axes, int device,
                           int max_batch, int flags, tha4_student** out) {
  if (!weights || !out) return fail(THA4_ERR_INVALID_ARGUMENT, "weights/out must not be NULL");
  *out = nullptr;
  if (max_batch < 1 || max_batch > 4096) return fail(THA4_ERR_INVALID_ARGUMENT, "max_batch must be in [1, 4096]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(THA4_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(THA4_ERR_NO_DEVICE, "device index out of range");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(THA4_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library only contains gfx950 code");

  std::vector<float> pos128(128), pos256(256), pos512(512);
  exact_position_axis(128, pos128.data());
  exact_position_axis(256, pos256.data());
  exact_position_axis(512, pos512.data());
  if (axes) {
    if (axes->axis128) std::memcpy(pos128.data(), axes->axis128, 128 * sizeof(float));
    if (axes->axis256) std::memcpy(pos256.data(), axes->axis256, 256 * sizeof(float));
    if (axes->axis512) std::memcpy(pos512.data(), axes->axis512, 512 * sizeof(float));
  }
  const bool exact = (flags & THA4_STUDENT_EXACT_FP32) != 0;
  BlobBuilder bb;
  StudentBlobOffsets o;
  const std::string err = build_student_blob(weights, exact, pos128, pos256, pos512, bb, o);
  if (!err.empty()) return fail(THA4_ERR_INVALID_ARGUMENT, (err.rfind("a sine layer", 0) == 0 ? "student weights outside the default kernels' range: " : "not a mode_14 student: ") + err);

  DeviceGuard guard(device);
  auto* h = new tha4_student();
  h->device = device;
  h->max_batch = max_batch;
  h->exact_fp32 = exact;
  auto cleanup = [&]() {
    h->order.destroy();
    if (h->blob) (void)hipFree(h->blob);
    if (h->workspace) (void)hipFree(h->workspace);
    delete h;
  };
  h->blob_bytes = bb.host.size();
  h->pos128 = pos128; h->pos256 = pos256; h->pos512 = pos512;
  hipError_t e = hipMalloc((void**)&h->blob, bb.host.size());
  if (e == hipSuccess) e = hipMemcpy(h->blob, bb.host.data(), bb.host.size(), hipMemcpyHostToDevice);
  const size_t B = (size_t)max_batch;
  const size_t s_pb = align_up(B * kPbStride * sizeof(float), 256);
  const size_t s_face = align_up(B * 4 * kFaceSize * kFaceSize * sizeof(float), 256);
  const size_t s_z1 = align_up(B * kNB1 * 128 * 128 * 16 * sizeof(float), 256);
  const size_t s_z2 = align_up(B * kNB2 * 256 * 256 * 16 * sizeof(float), 256);
  if (e == hipSuccess) e = hipMalloc((void**)&h->workspace, s_pb + s_face + s_z1 + s_z2);
  if (e == hipSuccess) e = h->order.create();
  if (e == hipSuccess) e = allow_lds(THA4_FACE_KERNEL, cfg::FaceG::LDS);
  if (e == hipSuccess) e = allow_lds(THA4_L0_KERNEL, cfg::L0G::LDS);
  if (e == hipSuccess) e = allow_lds(THA4_L1_KERNEL, cfg::L1G::LDS);
  if (e == hipSuccess) e = allow_lds(THA4_L2_KERNEL, cfg::L2G::LDS);
  if (e == hipSuccess) e = allow_lds(THA4_FRONT16_KERNEL, v2::cfg::kFrontLds);
  if (e == hipSuccess) e = allow_lds(THA4_FRONT16R_KERNEL, v2::cfg::kFrontRLds);
  if (e == hipSuccess) e = allow_lds(THA4_FACE16_KERNEL, v2::cfg::kFaceLds);
  if (e == hipSuccess) e = allow_lds(THA4_L016_KERNEL, v2::cfg::kL0Lds);
  if (e == hipSuccess) e = allow_lds(THA4_L116_KERNEL, v2::cfg::kL1Lds);
  if (e == hipSuccess) e = allow_lds(THA4_L116R_KERNEL, v2::cfg::kL1RLds);
  if (e == hipSuccess) e = allow_lds(THA4_L216_KERNEL, v2::cfg::kL2Lds);
  if (e == hipSuccess) e = allow_lds(THA4_L216P_KERNEL, v2::cfg::kL2PLds);
#ifdef THA4_L2_HOOK
  if (e == hipSuccess && std::getenv("THA4_L2_CODE_OBJECT") && std::getenv("THA4_L2_KERNEL")) {
    e = hipModuleLoad(&h->l2_module, std::getenv("THA4_L2_CODE_OBJECT"));
    if (e == hipSuccess) e = hipModuleGetFunction(&h->l2_function, h->l2_module, std::getenv("THA4_L2_KERNEL"));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(h->l2_function), hipFuncAttributeMaxDynamicSharedMemorySize, v2::cfg::kL2PLds);
    h->l2_threads = std::getenv("THA4_L2_THREADS") ? std::atoi(std::getenv("THA4_L2_THREADS")) : 512;
    h->l2_px = std::getenv("THA4_L2_PX") ? std::atoi(std::getenv("THA4_L2_PX")) : 1024;
  }
#endif
  if (e != hipSuccess) {
    cleanup();
    return fail(THA4_ERR_HIP, std::string("tha4_student_create: ") + hipGetErrorString(e));
  }
  auto F = [&](size_t off) { return reinterpret_cast<const float*>(h->blob + off); };
  StudentDev& d = h->dev;
  d.w_face = F(o.wf); d.w_l0 = F(o.w0); d.w_l1 = F(o.w1); d.w_l2 = F(o.w2);
  d.b_face = F(o.bf); d.b_l0 = F(o.b0); d.b_l1 = F(o.b1); d.b_l2 = F(o.b2);
  for (int i = 0; i < 4; ++i) {
    d.wx[i] = F(o.wx[i]); d.wy[i] = F(o.wy[i]); d.bias1[i] = F(o.b[i]); d.wpose[i] = F(o.wp[i]);
  }
  d.pos128 = F(o.p128); d.pos256 = F(o.p256); d.pos512 = F(o.p512);
  d.s_face = F(o.sf); d.s_l0 = F(o.s0); d.s_l1 = F(o.s1); d.s_l2 = F(o.s2);
  d.pb_scale = exact ? 1.0f : kSineScale16;
  char* ws = h->workspace;
  d.pbias = reinterpret_cast<float*>(ws); ws += s_pb;
  d.face = reinterpret_cast<float*>(ws); ws += s_face;
  d.z1 = reinterpret_cast<float*>(ws); ws += s_z1;
  d.z2 = reinterpret_cast<float*>(ws);
  *out = h;
  return THA4_OK;
}

int tha4_student_pose(tha4_student* h, const float* image_dev, int64_t image_batch_stride, const float* pose_dev,
                      int batch, float* out_blended_dev, const tha4_student_aux* aux, void* stream) {
  if (!h || !image_dev || !pose_dev) return fail(THA4_ERR_INVALID_ARGUMENT, "handle/image/pose must not be NULL");
  unsigned char* rgba8 = aux ? aux->display.rgba8_dev : nullptr;
  if (!out_blended_dev && !rgba8)
    return fail(THA4_ERR_INVALID_ARGUMENT, "out_blended_dev must not be NULL unless aux->display.rgba8_dev is given");
  if (batch < 1) return fail(THA4_ERR_INVALID_ARGUMENT, "batch must be >= 1");
  if (batch > h->max_batch) return fail(THA4_ERR_BATCH_TOO_LARGE, "batch exceeds max_batch given at create");
  if (image_batch_stride != 0 && image_batch_stride < 4LL * kImg * kImg)
    return fail(THA4_ERR_INVALID_ARGUMENT, "image_batch_stride must be 0 (shared) or >= 4*512*512");
  if ((const void*)image_dev == (const void*)out_blended_dev)
    return fail(THA4_ERR_INVALID_ARGUMENT, "out_blended_dev must not alias image_dev");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(h->order.enter(s));
  StudentDev d = h->dev;
  d.image = image_dev;
  d.image_stride = image_batch_stride;
  d.pose = pose_dev;
  d.out_blended = out_blended_dev;
  d.out_alpha = aux ? aux->alpha_dev : nullptr;
  d.out_color = aux ? aux->color_change_dev : nullptr;
  d.out_warped = aux ? aux->warped_dev : nullptr;
  d.out_grid = aux ? aux->grid_change_dev : nullptr;
  d.out_rgba8 = rgba8;
  d.rgba8_has_bg = rgba8 && aux->display.background_rgb ? 1 : 0;
  for (int k = 0; k < 3; ++k) d.rgba8_bg[k] = d.rgba8_has_bg ? aux->display.background_rgb[k] : 0.0f;
  if (aux && aux->face_dev) d.face = aux->face_dev;   // face kernel writes output 5 directly; level2 reads it there
  d.batch = batch;

  const bool t = h->timing && h->ev_valid;
  if (t) HIP_TRY(hipEventRecord(h->ev[0], s));
  // generation 2 folds the pose bias into each kernel's prologue (pose_bias_to_lds); the exact-fp32 generation keeps the launch
  if (h->exact_fp32 || !THA4_PB_FOLD) hipLaunchKernelGGL(posebias_kernel, dim3(cfg::posebias_blocks(), batch), dim3(kPoseBiasBlock), 0, s, d);
  if (t) HIP_TRY(hipEventRecord(h->ev[1], s));
  if (h->exact_fp32) {
    hipLaunchKernelGGL((THA4_FACE_KERNEL), dim3(cfg::blocks_for<cfg::FaceG>(batch, 128)), dim3(cfg::FaceG::THREADS),
                       cfg::FaceG::LDS, s, d);
    if (t) HIP_TRY(hipEventRecord(h->ev[2], s));
    hipLaunchKernelGGL((THA4_L0_KERNEL), dim3(cfg::blocks_for<cfg::L0G>(batch, 128)), dim3(cfg::L0G::THREADS),
                       cfg::L0G::LDS, s, d);
    if (t) HIP_TRY(hipEventRecord(h->ev[3], s));
    hipLaunchKernelGGL((THA4_L1_KERNEL), dim3(cfg::blocks_for<cfg::L1G>(batch, 256)), dim3(cfg::L1G::THREADS),
                       cfg::L1G::LDS, s, d);
    if (t) HIP_TRY(hipEventRecord(h->ev[4], s));
    hipLaunchKernelGGL((THA4_L2_KERNEL), dim3(cfg::blocks_for<cfg::L2G>(batch, 512)), dim3(cfg::L2G::THREADS),
                       cfg::L2G::LDS, s, d);
  } else {
    if (THA4_FRONT_REGS) {
      // face + level 0 in one launch of 4-wave workgroups, two per CU (front16r_kernel): [level-0 workgroups | face workgroups]
      if (t) HIP_TRY(hipEventRecord(h->ev[2], s));
      d.front_l0_blocks = batch * (128 * 128) / v2::cfg::FrontR::PX;
      hipLaunchKernelGGL((THA4_FRONT16R_KERNEL), dim3(2 * d.front_l0_blocks), dim3(v2::cfg::FrontR::THREADS), v2::cfg::kFrontRLds, s, d);
    } else if (THA4_FRONT_MERGE) {
      // face + level 0 in one launch (front16_kernel): timing slot 1 (face) is empty, slot 2 holds the merged kernel
      if (t) HIP_TRY(hipEventRecord(h->ev[2], s));
      d.front_l0_blocks = v2::cfg::blocks_for<v2::cfg::L0G>(batch, 128);
      hipLaunchKernelGGL((THA4_FRONT16_KERNEL), dim3(d.front_l0_blocks + v2::cfg::blocks_for<v2::cfg::FaceG>(batch, 128)),
                         dim3(v2::cfg::L0G::THREADS), v2::cfg::kFrontLds, s, d);
    } else {
      hipLaunchKernelGGL((THA4_FACE16_KERNEL), dim3(v2::cfg::blocks_for<v2::cfg::FaceG>(batch, 128)), dim3(v2::cfg::FaceG::THREADS),
                         v2::cfg::kFaceLds, s, d);
      if (t) HIP_TRY(hipEventRecord(h->ev[2], s));
      hipLaunchKernelGGL((THA4_L016_KERNEL), dim3(v2::cfg::blocks_for<v2::cfg::L0G>(batch, 128)), dim3(v2::cfg::L0G::THREADS),
                         v2::cfg::kL0Lds, s, d);
    }
    if (t) HIP_TRY(hipEventRecord(h->ev[3], s));
    if (THA4_L1_REGS)
      hipLaunchKernelGGL((THA4_L116R_KERNEL), dim3(batch * (256 * 256) / v2::cfg::L1R::PX), dim3(v2::cfg::L1R::THREADS), v2::cfg::kL1RLds, s, d);
    else
      hipLaunchKernelGGL((THA4_L116_KERNEL), dim3(v2::cfg::blocks_for<v2::cfg::L1G>(batch, 256)), dim3(v2::cfg::L1G::THREADS),
                         v2::cfg::kL1Lds, s, d);
    if (t) HIP_TRY(hipEventRecord(h->ev[4], s));
#ifdef THA4_L2_HOOK
    if (h->l2_function) {
      void* params[] = {&d};
      HIP_TRY(hipModuleLaunchKernel(h->l2_function, batch * (512 * 512) / h->l2_px, 1, 1, h->l2_threads, 1, 1, v2::cfg::kL2PLds, s, params, nullptr));
    } else
#endif
    if (THA4_L2_RESIDENT)
      hipLaunchKernelGGL((THA4_L216P_KERNEL), dim3(batch * (512 * 512) / v2::cfg::L2P::PX), dim3(v2::cfg::L2P::THREADS),
                         v2::cfg::kL2PLds, s, d);
    else
      hipLaunchKernelGGL((THA4_L216_KERNEL), dim3(v2::cfg::blocks_for<v2::cfg::L2G>(batch, 512)), dim3(v2::cfg::L2G::THREADS),
                         v2::cfg::kL2Lds, s, d);
  }
  if (t) {
    HIP_TRY(hipEventRecord(h->ev[5], s));
    h->ev_recorded = true;
  }
  HIP_TRY(hipGetLastError());
  return THA4_OK;
}

void tha4_student_destroy(tha4_student* h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  if (h->ev_valid)
    for (auto& e : h->ev) (void)hipEventDestroy(e);
  h->order.destroy();
#ifdef THA4_L2_HOOK
  if (h->l2_module) (void)hipModuleUnload(h->l2_module);
#endif
  if (h->blob) (void)hipFree(h->blob);
  if (h->workspace) (void)hipFree(h->workspace);
  delete h;
}

int tha4_student_set_weights(tha4_student* h, const tha4_student_weights* weights) {
  if (!h || !weights) return fail(THA4_ERR_INVALID_ARGUMENT, "handle/weights must not be NULL");
  BlobBuilder bb;
  StudentBlobOffsets o;
  const std::string err = build_student_blob(weights, h->exact_fp32, h->pos128, h->pos256, h->pos512, bb, o);
  if (!err.empty()) return fail(THA4_ERR_INVALID_ARGUMENT, (err.rfind("a sine layer", 0) == 0 ? "student weights outside the default kernels' range: " : "not a mode_14 student: ") + err);
  if (bb.host.size() != h->blob_bytes) return fail(THA4_ERR_INVALID_ARGUMENT, "internal: packed size differs from the handle's blob");
  DeviceGuard guard(h->device);
  // rare operation: wait for every pose call in flight (whatever stream it is on), then overwrite the blob in place -
  // no allocation, workspace and handle stay valid
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->blob, bb.host.data(), bb.host.size(), hipMemcpyHostToDevice));
  return THA4_OK;
}

int tha4_student_debug_read(tha4_student* h, int which, int frame, float* host_out) {
  if (!h || !host_out) return fail(THA4_ERR_INVALID_ARGUMENT, "handle/host_out must not be NULL");
  if (frame < 0 || frame >= h->max_batch) return fail(THA4_ERR_INVALID_ARGUMENT, "frame out of range");
#ifdef THA4_STAMPS      // tuning builds: which == 2 reads the in-kernel time stamps (THA4_STAMP, siren16_kernels.h) out of the pose-bias workspace
  if (which == 2) {
    DeviceGuard guard(h->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, h->dev.pbias, (size_t)kPbStride * sizeof(float), hipMemcpyDeviceToHost));
    return THA4_OK;
  }
#endif
  if (which != 0 && which != 1) return fail(THA4_ERR_INVALID_ARGUMENT, "which must be 0 (z1) or 1 (z2)");
  const size_t per = which == 0 ? (size_t)kNB1 * 128 * 128 * 16 : (size_t)kNB2 * 256 * 256 * 16;
  const float* src = (which == 0 ? h->dev.z1 : h->dev.z2) + (size_t)frame * per;
  DeviceGuard guard(h->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host_out, src, per * sizeof(float), hipMemcpyDeviceToHost));
  return THA4_OK;
}

#ifdef THA4_STAMPS
extern "C" int tha4_student_debug_write(tha4_student* h, const void* host_in) {      // tuning builds: preset the stamp workspace (min / max fields)
  DeviceGuard guard(h->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->dev.pbias, host_in, (size_t)kPbStride * sizeof(float), hipMemcpyHostToDevice));
  return THA4_OK;
}
#endif

float tha4_student_hand_off_scale(const tha4_student* h) { return h ? h->dev.pb_scale : 0.0f; }

int tha4_student_max_batch(const tha4_student* h) { return h ? h->max_batch : THA4_ERR_INVALID_ARGUMENT; }
int tha4_student_device(const tha4_student* h) { return h ? h->device : THA4_ERR_INVALID_ARGUMENT; }

int tha4_student_set_timing(tha4_student* h, int enable) {
  if (!h) return fail(THA4_ERR_INVALID_ARGUMENT, "handle must not be NULL");
  DeviceGuard guard(h->device);
  if (enable && !h->ev_valid) {
    for (auto& e : h->ev) HIP_TRY(hipEventCreate(&e));
    h->ev_valid = true;
  }
  h->timing = enable != 0;
  h->ev_recorded = false;
  return THA4_OK;
}

int tha4_student_last_ms(tha4_student* h, int kernel, float* ms_out) {
  if (!h || !ms_out) return fail(THA4_ERR_INVALID_ARGUMENT, "handle/ms_out must not be NULL");
  if (!h->ev_valid || !h->ev_recorded) return fail(THA4_ERR_INVALID_ARGUMENT, "no timed pose call recorded");
  if (kernel < -1 || kernel >= kNumKernels) return fail(THA4_ERR_INVALID_ARGUMENT, "kernel index out of range");
  DeviceGuard guard(h->device);
  HIP_TRY(hipEventSynchronize(h->ev[kNumKernels]));
  if (kernel < 0) HIP_TRY(hipEventElapsedTime(ms_out, h->ev[0], h->ev[kNumKernels]));
  else HIP_TRY(hipEventElapsedTime(ms_out, h->ev[kernel], h->ev[kernel + 1]));
  return THA4_OK;
}

// ---------------------------------------------------------------------------------------------
// full model (mode_07)
// ---------------------------------------------------------------------------------------------
struct tha4_full {
  int device = 0;
  FullModel model;
  bool decomposer_valid = false;
  int last_batch = 0;
  int fault_policy = THA4_FAULT_REFUSE_NEXT;
  StreamOrder order;
  int* fault = nullptr;        // pinned host memory, device-visible: the kernels' sticky numeric-fault flag
};

int tha4_full_create(const tha4_full_weights* weights, int eyebrow_morphed_image_index, int device, int max_batch,
                     tha4_full** out) {
  return tha4_full_create_ex(weights, eyebrow_morphed_image_index, device, max_batch, 5, THA4_FULL_EXACT_DECOMPOSER_OUTER, out);      // the mixed default plan (ABI v6)
}

int tha4_full_create_ex(const tha4_full_weights* weights, int eyebrow_morphed_image_index, int device, int max_batch,
                        int num_networks, uint32_t flags, tha4_full** out) {
  if (!weights || !out) return fail(THA4_ERR_INVALID_ARGUMENT, "weights/out must not be NULL");
  if (flags & ~(uint32_t)(THA4_FULL_EXACT_FP32 | THA4_FULL_EXACT_DECOMPOSER | THA4_FULL_EXACT_DECOMPOSER_OUTER)) return fail(THA4_ERR_INVALID_ARGUMENT, "unknown flag bits");
  if (num_networks != 3 && num_networks != 5) return fail(THA4_ERR_INVALID_ARGUMENT, "num_networks must be 5 (mode_07) or 3 (mode_12)");
  *out = nullptr;
  if (max_batch < 1 || max_batch > 256) return fail(THA4_ERR_INVALID_ARGUMENT, "max_batch must be in [1, 256]");
  if (eyebrow_morphed_image_index != 0 && eyebrow_morphed_image_index != 2)
    return fail(THA4_ERR_INVALID_ARGUMENT, "eyebrow_morphed_image_index must be 0 or 2");
  int ndev = 0;
  const bool have_dev = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
  // diagnostics only: with THA4_DUMP_SCHEDULE set and no device (the build container) the launch plan is still built and printed
  // - planning is host code - before the call fails with THA4_ERR_NO_DEVICE like it always does without a GPU
  const bool plan_only = !have_dev && std::getenv("THA4_DUMP_SCHEDULE") != nullptr;
  if (!have_dev && !plan_only) return fail(THA4_ERR_NO_DEVICE, "no HIP device visible");
  if (have_dev) {
    if (device < 0 || device >= ndev) return fail(THA4_ERR_NO_DEVICE, "device index out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(THA4_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library only contains gfx950 code");
  }
  WeightMap nets[5];
  for (int n = 0; n < num_networks; ++n) {
    if (!weights->tensors[n] || weights->counts[n] <= 0) return fail(THA4_ERR_INVALID_ARGUMENT, "empty state_dict");
    for (int i = 0; i < weights->counts[n]; ++i) {
      const tha4_named_tensor& t = weights->tensors[n][i];
      if (!t.name || !t.data || t.ndim < 1 || t.ndim > 4) return fail(THA4_ERR_INVALID_ARGUMENT, "malformed named tensor");
      HostTensor h;
      h.data = t.data;
      h.dims.assign(t.dims, t.dims + t.ndim);
      nets[n][t.name] = h;
    }
  }
  auto* h = new tha4_full();
  h->device = device;
  h->model.exact_decomposer = (flags & THA4_FULL_EXACT_DECOMPOSER) != 0;
  h->model.exact_decomposer_outer = (flags & THA4_FULL_EXACT_DECOMPOSER_OUTER) != 0;
  if (!h->model.build(nets, max_batch, eyebrow_morphed_image_index, num_networks, (flags & THA4_FULL_EXACT_FP32) != 0)) {
    // a planner failure ("internal: ...") is ours, not a property of the caller's state_dicts: say so
    const bool internal = h->model.error.rfind("internal:", 0) == 0;
    std::string msg = (internal ? std::string("launch planning failed for max_batch = ") + std::to_string(max_batch) + ": "
                                : std::string(num_networks == 5 ? "not a mode_07 model: " : "not a mode_12 model: ")) + h->model.error;
    delete h;
    return fail(THA4_ERR_INVALID_ARGUMENT, msg);
  }
  if (plan_only) {
    delete h;
    return fail(THA4_ERR_NO_DEVICE, "no HIP device visible (the launch plan was printed: THA4_DUMP_SCHEDULE)");
  }
  DeviceGuard guard(device);
  FullModel& m = h->model;
  hipError_t e = hipMalloc((void**)&m.dev_params, m.host_params.size());
  if (e == hipSuccess) e = hipMemcpy(m.dev_params, m.host_params.data(), m.host_params.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&m.dev_work, m.work_floats * sizeof(float));
  if (e == hipSuccess) e = hipMemset(m.dev_work, 0, m.work_floats * sizeof(float));
  if (e == hipSuccess && m.acc_floats) e = hipMalloc((void**)&m.dev_acc, m.acc_floats * sizeof(float));
  if (e == hipSuccess && m.acc_floats) e = hipMemset(m.dev_acc, 0, m.acc_floats * sizeof(float));
  if (e == hipSuccess) e = FullModel::allow_all_conv_lds();
  if (e == hipSuccess) e = h->order.create();
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->fault, 64, hipHostMallocMapped);
  if (e == hipSuccess) { *h->fault = 0; m.fault = h->fault; }
  std::vector<char>().swap(m.host_params);
  if (e != hipSuccess) {
    h->order.destroy();
    if (h->fault) (void)hipHostFree(h->fault);
    if (m.dev_params) (void)hipFree(m.dev_params);
    if (m.dev_work) (void)hipFree(m.dev_work);
    if (m.dev_acc) (void)hipFree(m.dev_acc);
    delete h;
    return fail(THA4_ERR_HIP, std::string("tha4_full_create: ") + hipGetErrorString(e));
  }
  *out = h;
  return THA4_OK;
}

int tha4_full_pose(tha4_full* h, const float* image_dev, int64_t image_batch_stride, const float* pose_dev, int batch,
                   float* const* outputs_dev, int reuse_decomposer, void* stream) {
  return tha4_full_pose_ex(h, image_dev, image_batch_stride, pose_dev, batch, outputs_dev, reuse_decomposer, nullptr, stream);
}

int tha4_full_pose_ex(tha4_full* h, const float* image_dev, int64_t image_batch_stride, const float* pose_dev, int batch,
                      float* const* outputs_dev, int reuse_decomposer, const tha4_display* display, void* stream) {
  unsigned char* rgba8 = display ? display->rgba8_dev : nullptr;
  if (!h || !image_dev || !pose_dev || (!outputs_dev && !rgba8))
    return fail(THA4_ERR_INVALID_ARGUMENT, "handle/image/pose/outputs must not be NULL");
  if (rgba8 && h->model.num_networks != 5)
    return fail(THA4_ERR_INVALID_ARGUMENT, "the display epilogue belongs to output 0, which a 3-network (mode_12) handle does not produce");
  float* const no_outputs[33] = {};
  if (!outputs_dev) outputs_dev = no_outputs;
  {
    const int first = h->model.num_networks == 5 ? 0 : 11;     // a mode_12 handle produces outputs 11..32 only
    bool any = rgba8 != nullptr;
    for (int i = first; i < 33; ++i) any = any || outputs_dev[i] != nullptr;
    if (!any) return fail(THA4_ERR_INVALID_ARGUMENT, "no output requested");
    for (int i = 0; i < first; ++i)
      if (outputs_dev[i]) return fail(THA4_ERR_INVALID_ARGUMENT, "outputs 0..10 do not exist on a 3-network (mode_12) handle");
  }
  if (batch < 1) return fail(THA4_ERR_INVALID_ARGUMENT, "batch must be >= 1");
  if (batch > h->model.max_batch) return fail(THA4_ERR_BATCH_TOO_LARGE, "batch exceeds max_batch given at create");
  if (image_batch_stride != 0 && image_batch_stride < 4LL * 512 * 512)
    return fail(THA4_ERR_INVALID_ARGUMENT, "image_batch_stride must be 0 (shared) or >= 4*512*512");
  // sticky numeric-fault flag of EARLIER calls (no synchronisation: whatever has become visible by now; tha4_full_numeric_status
  // is the synchronous check).  Reported once, then cleared so that the caller can go on with sane inputs
  if (h->fault_policy == THA4_FAULT_REFUSE_NEXT && *reinterpret_cast<volatile int*>(h->fault)) {
    *reinterpret_cast<volatile int*>(h->fault) = 0;
    // the faulted call may be the one that filled the persistent eyebrow-decomposer outputs: they are not reused by anybody
    h->decomposer_valid = false;
    return fail(THA4_ERR_NUMERIC_RANGE, kNumericFaultMessage);
  }
  DeviceGuard guard(h->device);
  FullModel& m = h->model;
  FullModel::Frame f{};
  f.image = image_dev; f.image_stride = image_batch_stride; f.pose = pose_dev; f.batch = batch;
  f.stream = static_cast<hipStream_t>(stream);
  f.rgba8 = rgba8;
  f.rgba8_has_bg = rgba8 && display->background_rgb ? 1 : 0;
  for (int k = 0; k < 3; ++k) f.rgba8_bg[k] = f.rgba8_has_bg ? display->background_rgb[k] : 0.0f;
  HIP_TRY(h->order.enter(f.stream));
  bool want_dec[6];
  // an output the caller did not ask for is written to scratch only if a LATER STAGE reads it (FullModel::read_by_later_stage); otherwise its pointer stays
  // null and the image kernels skip its stores (round 6: Poser.pose() asks for one of the 33)
  static const bool write_all = tune_env("THA4_WRITE_ALL_OUTPUTS") != nullptr;      // tuning aid (A/B): rounds 1-5 wrote every output
  for (int i = 0; i < 33; ++i) f.out[i] = outputs_dev[i] ? outputs_dev[i] : ((write_all || m.read_by_later_stage(i)) ? m.Wk(m.scratch_out[i]) : nullptr);
  for (int i = 0; i < 6; ++i) want_dec[i] = outputs_dev[27 + i] != nullptr;
  // THA4_FAULT_STATUS_ONLY: the call is never refused, but while a fault is pending (raised and not yet polled through
  // tha4_full_numeric_status) the persistent decomposer outputs may be the faulted call's: they are recomputed, not reused.
  // The flag is only peeked at - reporting and clearing stay with tha4_full_numeric_status
  const bool fault_pending = *reinterpret_cast<volatile int*>(h->fault) != 0;
  const bool reuse = reuse_decomposer && h->decomposer_valid && h->last_batch == batch && !fault_pending;
  h->decomposer_valid = false;              // valid again only once every launch of this call has been accepted
  m.run(f, !reuse, want_dec);
  HIP_TRY(hipGetLastError());
  h->decomposer_valid = true;
  h->last_batch = batch;
  return THA4_OK;
}

int tha4_full_set_fault_policy(tha4_full* h, int policy) {
  if (!h) return fail(THA4_ERR_INVALID_ARGUMENT, "handle must not be NULL");
  if (policy != THA4_FAULT_REFUSE_NEXT && policy != THA4_FAULT_STATUS_ONLY) return fail(THA4_ERR_INVALID_ARGUMENT, "unknown fault policy");
  h->fault_policy = policy;
  return THA4_OK;
}

// ---- per-op timing (ABI v5) ---------------------------------------------------------------------------------------------------------------
int tha4_full_set_timing(tha4_full* h, int enable) {
  if (!h) return fail(THA4_ERR_INVALID_ARGUMENT, "handle must not be NULL");
  FullModel& m = h->model;
  DeviceGuard guard(h->device);
  const size_t want = m.ops_decomposer.size() + m.ops_rest.size() + 2;
  if (enable && m.timing_events.size() != want) {
    for (hipEvent_t e : m.timing_events) (void)hipEventDestroy(e);
    m.timing_events.clear();
    for (size_t i = 0; i < want; ++i) {
      hipEvent_t e = nullptr;
      const hipError_t err = hipEventCreate(&e);
      if (err != hipSuccess) {               // nothing half-built stays behind: timing is off and untouched by a failed switch
        for (hipEvent_t d : m.timing_events) (void)hipEventDestroy(d);
        m.timing_events.clear();
        m.timing_on = false;
        m.timing_recorded = false;
        return fail(THA4_ERR_HIP, std::string("tha4_full_set_timing: ") + hipGetErrorString(err));
      }
      m.timing_events.push_back(e);
    }
  }
  m.timing_on = enable != 0;
  m.timing_recorded = false;
  return THA4_OK;
}

int tha4_full_num_ops(const tha4_full* h) {
  return h ? (int)(h->model.ops_decomposer.size() + h->model.ops_rest.size()) : THA4_ERR_INVALID_ARGUMENT;
}

int tha4_full_op_info(const tha4_full* h, int index, const char** label, double* gflop) {
  if (!h) return fail(THA4_ERR_INVALID_ARGUMENT, "handle must not be NULL");
  const FullModel& m = h->model;
  const int nd = (int)m.info_decomposer.size(), n = nd + (int)m.info_rest.size();
  if (index < 0 || index >= n) return fail(THA4_ERR_INVALID_ARGUMENT, "op index out of range");
  const FullModel::OpInfo& o = index < nd ? m.info_decomposer[index] : m.info_rest[index - nd];
  if (label) *label = o.label.c_str();
  if (gflop) *gflop = o.gflop;
  return THA4_OK;
}

int tha4_full_last_op_ms(tha4_full* h, float* ms, int capacity) {
  if (!h || !ms) return fail(THA4_ERR_INVALID_ARGUMENT, "handle/ms must not be NULL");
  FullModel& m = h->model;
  const int nd = (int)m.ops_decomposer.size(), nr = (int)m.ops_rest.size();
  if (capacity < nd + nr) return fail(THA4_ERR_INVALID_ARGUMENT, "ms must hold tha4_full_num_ops() floats");
  if (!m.timing_on || !m.timing_recorded || (int)m.timing_events.size() != nd + nr + 2)
    return fail(THA4_ERR_INVALID_ARGUMENT, "no timed pose call recorded (tha4_full_set_timing)");
  DeviceGuard guard(h->device);
  HIP_TRY(hipEventSynchronize(m.timing_events[nd + nr + 1]));
  for (int i = 0; i < nd; ++i) {
    ms[i] = 0.0f;
    if (m.timing_ran_decomposer) HIP_TRY(hipEventElapsedTime(&ms[i], m.timing_events[i], m.timing_events[i + 1]));
  }
  for (int j = 0; j < nr; ++j) HIP_TRY(hipEventElapsedTime(&ms[nd + j], m.timing_events[nd + 1 + j], m.timing_events[nd + 2 + j]));
  return THA4_OK;
}

int tha4_full_numeric_status(tha4_full* h, int synchronize) {
  if (!h) return fail(THA4_ERR_INVALID_ARGUMENT, "handle must not be NULL");
  if (synchronize) {
    DeviceGuard guard(h->device);
    HIP_TRY(hipDeviceSynchronize());
  }
  if (*reinterpret_cast<volatile int*>(h->fault)) {
    *reinterpret_cast<volatile int*>(h->fault) = 0;
    h->decomposer_valid = false;
    return fail(THA4_ERR_NUMERIC_RANGE, kNumericFaultMessage);
  }
  return THA4_OK;
}

#ifdef THA4_PHASE_TIMING
// tuning aid (not part of the ABI header): copy the head of the split-K workspace (phase stamps) to the host
int tha4_full_debug_read(tha4_full* h, void* dst, size_t bytes) {
  DeviceGuard guard(h->device);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpy(dst, h->model.Wk(h->model.dbg_off), bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

void tha4_full_destroy(tha4_full* h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  h->order.destroy();
  for (hipEvent_t e : h->model.timing_events) (void)hipEventDestroy(e);
  h->model.timing_events.clear();
  if (h->fault) (void)hipHostFree(h->fault);
  if (h->model.dev_params) (void)hipFree(h->model.dev_params);
  if (h->model.dev_work) (void)hipFree(h->model.dev_work);
  if (h->model.dev_acc) (void)hipFree(h->model.dev_acc);
  delete h;
}

int tha4_full_max_batch(const tha4_full* h) { return h ? h->model.max_batch : THA4_ERR_INVALID_ARGUMENT; }
int tha4_full_num_networks(const tha4_full* h) { return h ? h->model.num_networks : THA4_ERR_INVALID_ARGUMENT; }
int tha4_full_flags(const tha4_full* h) {
  return h ? (int)((h->model.exact_fp32 ? THA4_FULL_EXACT_FP32 : 0u) | (h->model.exact_decomposer ? THA4_FULL_EXACT_DECOMPOSER : 0u) |
                  (h->model.exact_decomposer_outer ? THA4_FULL_EXACT_DECOMPOSER_OUTER : 0u)) : THA4_ERR_INVALID_ARGUMENT;
}

int tha4_display_rgba8(const float* frames_dev, int batch, int height, int width, const float* background_rgb,
                       uint8_t* out_dev, void* stream) {
  if (!frames_dev || !out_dev || batch < 1 || height < 1 || width < 1)
    return fail(THA4_ERR_INVALID_ARGUMENT, "frames/out must not be NULL and batch/size must be positive");
  const int dev = device_of(frames_dev);
  if (dev < 0 || dev != device_of(out_dev)) return fail(THA4_ERR_INVALID_ARGUMENT, "frames/out must be device pointers of one GPU");
  DeviceGuard guard(dev);
  DisplayArgs a{};
  a.frames = frames_dev; a.out = out_dev; a.pixels = height * width;
  a.has_background = background_rgb != nullptr;
  if (background_rgb) for (int k = 0; k < 3; ++k) a.bg[k] = background_rgb[k];
  hipLaunchKernelGGL(display_rgba8_kernel, dim3((a.pixels + 255) / 256, batch), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return THA4_OK;
}

int tha4_ingest_rgba8(const uint8_t* rgba_dev, int batch, int height, int width, float* out_dev, void* stream) {
  if (!rgba_dev || !out_dev || batch < 1 || height < 1 || width < 1)
    return fail(THA4_ERR_INVALID_ARGUMENT, "rgba/out must not be NULL and batch/size must be positive");
  const int dev = device_of(rgba_dev);
  if (dev < 0 || dev != device_of(out_dev)) return fail(THA4_ERR_INVALID_ARGUMENT, "rgba/out must be device pointers of one GPU");
  DeviceGuard guard(dev);
  IngestArgs a{rgba_dev, out_dev, height * width};
  hipLaunchKernelGGL(ingest_rgba8_kernel, dim3((a.pixels + 255) / 256, batch), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return THA4_OK;
}

}  // extern "C"
