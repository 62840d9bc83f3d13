// CPU emulation harness for the student kernels (TEST INFRASTRUCTURE ONLY).
// Builds the unmodified kernel source against emu_hip.h and exposes a tiny C API for pytest
// (tests/test_emu_kernels.py): host buffers stand in for HBM, selected workgroups are executed.
#define THA4_EMU 1
#include "siren16_kernels.h"
#include "tha4_hip.h"

#include <map>
#include <string>

using namespace tha4;

namespace {
struct EmuStudent {
  int gen = 1;
  v2::StudentPacked16 packed16;
  StudentPacked packed;
  std::vector<float> pos128, pos256, pos512;
  std::map<std::string, std::vector<float>> buf;
  StudentDev dev{};
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
}  // namespace

extern "C" {

void* emu_student_create_gen(const tha4_student_weights* w, const tha4_position_axes* axes, int gen);
void* emu_student_create(const tha4_student_weights* w, const tha4_position_axes* axes) { return emu_student_create_gen(w, axes, 1); }

void* emu_student_create_gen(const tha4_student_weights* w, const tha4_position_axes* axes, int gen) {
  auto* e = new EmuStudent();
  e->gen = gen;
  std::string err = pack_student(to_view(w), e->packed);
  if (!err.empty()) {
    std::fprintf(stderr, "emu_student_create: %s\n", err.c_str());
    delete e;
    return nullptr;
  }
  e->pos128.resize(128); e->pos256.resize(256); e->pos512.resize(512);
  exact_position_axis(128, e->pos128.data());
  exact_position_axis(256, e->pos256.data());
  exact_position_axis(512, e->pos512.data());
  if (axes) {
    if (axes->axis128) std::memcpy(e->pos128.data(), axes->axis128, 128 * 4);
    if (axes->axis256) std::memcpy(e->pos256.data(), axes->axis256, 256 * 4);
    if (axes->axis512) std::memcpy(e->pos512.data(), axes->axis512, 512 * 4);
  }
  auto& b = e->buf;
  b["pbias"].assign(kPbStride, 0.f);
  b["z1"].assign((size_t)kNB1 * 128 * 128 * 16, 0.f);
  b["z2"].assign((size_t)kNB2 * 256 * 256 * 16, 0.f);
  b["face"].assign(4 * 128 * 128, 0.f);
  b["image"].assign(4 * 512 * 512, 0.f);
  b["pose"].assign(kPose, 0.f);
  b["out_blended"].assign(4 * 512 * 512, 0.f);
  b["out_alpha"].assign(512 * 512, 0.f);
  b["out_color"].assign(4 * 512 * 512, 0.f);
  b["out_warped"].assign(4 * 512 * 512, 0.f);
  b["out_grid"].assign(2 * 512 * 512, 0.f);
  StudentDev& d = e->dev;
  const StudentPacked& p = e->packed;
  d.w_face = p.w_face.data(); d.w_l0 = p.w_l0.data(); d.w_l1 = p.w_l1.data(); d.w_l2 = p.w_l2.data();
  if (gen == 2) {
    v2::pack_student16(to_view(w), e->packed, e->packed16);
    d.w_face = reinterpret_cast<const float*>(e->packed16.w_face.data());
    d.w_l0 = reinterpret_cast<const float*>(e->packed16.w_l0.data());
    d.w_l1 = reinterpret_cast<const float*>(e->packed16.w_l1.data());
    d.w_l2 = reinterpret_cast<const float*>(e->packed16.w_l2.data());
  }
  d.b_face = p.b_face.data(); d.b_l0 = p.b_l0.data(); d.b_l1 = p.b_l1.data(); d.b_l2 = p.b_l2.data();
  const FirstLayerPack* f[4] = {&p.f_face, &p.f_l0, &p.f_l1, &p.f_l2};
  for (int i = 0; i < 4; ++i) {
    d.wx[i] = f[i]->wx.data(); d.wy[i] = f[i]->wy.data();
    d.bias1[i] = f[i]->bias.data(); d.wpose[i] = f[i]->wpose.data();
  }
  d.pb_scale = 1.0f;
  d.s_face = d.s_l0 = d.s_l1 = d.s_l2 = nullptr;
  if (gen == 2) {
    const v2::StudentPacked16& q = e->packed16;
    d.b_face = q.b_face.data(); d.b_l0 = q.b_l0.data(); d.b_l1 = q.b_l1.data(); d.b_l2 = q.b_l2.data();
    d.s_face = q.s_face.data(); d.s_l0 = q.s_l0.data(); d.s_l1 = q.s_l1.data(); d.s_l2 = q.s_l2.data();
    for (int i = 0; i < 4; ++i) { d.wx[i] = q.wx[i].data(); d.wy[i] = q.wy[i].data(); }
    d.pb_scale = kSineScale16;
  }
  d.pos128 = e->pos128.data(); d.pos256 = e->pos256.data(); d.pos512 = e->pos512.data();
  d.pbias = b["pbias"].data(); d.z1 = b["z1"].data(); d.z2 = b["z2"].data(); d.face = b["face"].data();
  d.image = b["image"].data(); d.image_stride = 4 * 512 * 512; d.pose = b["pose"].data();
  d.out_blended = b["out_blended"].data(); d.out_alpha = b["out_alpha"].data();
  d.out_color = b["out_color"].data(); d.out_warped = b["out_warped"].data(); d.out_grid = b["out_grid"].data();
  d.batch = 1;
  return e;
}

float* emu_student_buffer(void* h, const char* name, int64_t* nfloats) {
  auto* e = static_cast<EmuStudent*>(h);
  auto it = e->buf.find(name);
  if (it == e->buf.end()) return nullptr;
  if (nfloats) *nfloats = (int64_t)it->second.size();
  return it->second.data();
}

// number of workgroups of kernel k (0 posebias, 1 face, 2 level0, 3 level1, 4 level2) at batch 1
int emu_student_grid_gen(int kernel, int gen) {
  if (gen == 2) {
    switch (kernel) {
      case 0: return cfg::posebias_blocks();
      case 1: return THA4_FRONT_REGS ? (128 * 128) / v2::cfg::FrontR::PX : v2::cfg::blocks_for<v2::cfg::FaceG>(1, 128);
      case 2: return THA4_FRONT_REGS ? (128 * 128) / v2::cfg::FrontR::PX : v2::cfg::blocks_for<v2::cfg::L0G>(1, 128);
      case 3: return (256 * 256) / v2::cfg::kL1Px;
      case 4: return THA4_L2_RESIDENT ? (512 * 512) / v2::cfg::L2P::PX : v2::cfg::blocks_for<v2::cfg::L2G>(1, 512);
    }
    return -1;
  }
  switch (kernel) {
    case 0: return cfg::posebias_blocks();
    case 1: return cfg::blocks_for<cfg::FaceG>(1, 128);
    case 2: return cfg::blocks_for<cfg::L0G>(1, 128);
    case 3: return cfg::blocks_for<cfg::L1G>(1, 256);
    case 4: return cfg::blocks_for<cfg::L2G>(1, 512);
  }
  return -1;
}
int emu_student_grid(int kernel) { return emu_student_grid_gen(kernel, 1); }

int emu_student_run(void* h, int kernel, int first_block, int nblocks) {
  auto* e = static_cast<EmuStudent*>(h);
  const int grid = emu_student_grid_gen(kernel, e->gen);
  if (e->gen == 2) {
    if (grid < 0 || first_block < 0 || first_block + nblocks > grid) return -1;
    for (int b = first_block; b < first_block + nblocks; ++b) {
      switch (kernel) {
        case 0: emu::run_block(posebias_kernel, dim3(grid, 1), dim3(b, 0), kPoseBiasBlock, 0, e->dev); break;
        case 1:      // (front16r_kernel: the face workgroups are the second half of the merged grid)
          if (THA4_FRONT_REGS) { e->dev.front_l0_blocks = grid; emu::run_block(THA4_FRONT16R_KERNEL, dim3(2 * grid), dim3(grid + b), v2::cfg::FrontR::THREADS, v2::cfg::kFrontRLds, e->dev); }
          else emu::run_block(THA4_FACE16_KERNEL, dim3(grid), dim3(b), v2::cfg::FaceG::THREADS, v2::cfg::kFaceLds, e->dev);
          break;
        case 2:
          if (THA4_FRONT_REGS) { e->dev.front_l0_blocks = grid; emu::run_block(THA4_FRONT16R_KERNEL, dim3(2 * grid), dim3(b), v2::cfg::FrontR::THREADS, v2::cfg::kFrontRLds, e->dev); }
          else emu::run_block(THA4_L016_KERNEL, dim3(grid), dim3(b), v2::cfg::L0G::THREADS, v2::cfg::kL0Lds, e->dev);
          break;
        case 3:
          if (THA4_L1_REGS) emu::run_block(THA4_L116R_KERNEL, dim3(grid), dim3(b), v2::cfg::L1R::THREADS, v2::cfg::kL1RLds, e->dev);
          else emu::run_block(THA4_L116_KERNEL, dim3(grid), dim3(b), v2::cfg::L1G::THREADS, v2::cfg::kL1Lds, e->dev);
          break;
        case 4:
          if (THA4_L2_RESIDENT) emu::run_block(THA4_L216P_KERNEL, dim3(grid), dim3(b), v2::cfg::L2P::THREADS, v2::cfg::kL2PLds, e->dev);
          else emu::run_block(THA4_L216_KERNEL, dim3(grid), dim3(b), v2::cfg::L2G::THREADS, v2::cfg::kL2Lds, e->dev);
          break;
      }
    }
    return 0;
  }
  if (grid < 0 || first_block < 0 || first_block + nblocks > grid) return -1;
  for (int b = first_block; b < first_block + nblocks; ++b) {
    switch (kernel) {
      case 0: emu::run_block(posebias_kernel, dim3(grid, 1), dim3(b, 0), kPoseBiasBlock, 0, e->dev); break;
      case 1: emu::run_block(THA4_FACE_KERNEL, dim3(grid), dim3(b), cfg::FaceG::THREADS, cfg::FaceG::LDS, e->dev); break;
      case 2: emu::run_block(THA4_L0_KERNEL, dim3(grid), dim3(b), cfg::L0G::THREADS, cfg::L0G::LDS, e->dev); break;
      case 3: emu::run_block(THA4_L1_KERNEL, dim3(grid), dim3(b), cfg::L1G::THREADS, cfg::L1G::LDS, e->dev); break;
      case 4: emu::run_block(THA4_L2_KERNEL, dim3(grid), dim3(b), cfg::L2G::THREADS, cfg::L2G::LDS, e->dev); break;
    }
  }
  return 0;
}

// pixels [first, first+count) (row-major index at the kernel's resolution) covered by workgroup b
void emu_student_block_pixels_gen(int kernel, int block, int gen, int* first, int* count) {
  if (gen == 2)
    *count = kernel == 1 ? (THA4_FRONT_REGS ? v2::cfg::FrontR::PX : v2::cfg::FaceG::PX) : kernel == 2 ? (THA4_FRONT_REGS ? v2::cfg::FrontR::PX : v2::cfg::L0G::PX) : kernel == 3 ? v2::cfg::kL1Px
             : (THA4_L2_RESIDENT ? v2::cfg::L2P::PX : v2::cfg::L2G::PX);
  else
    *count = kernel == 1 ? cfg::FaceG::PX : kernel == 2 ? cfg::L0G::PX : kernel == 3 ? cfg::L1G::PX : cfg::L2G::PX;
  *first = xcd_tile(block, emu_student_grid_gen(kernel, gen)) * (*count);
}
void emu_student_block_pixels(int kernel, int block, int* first, int* count) { emu_student_block_pixels_gen(kernel, block, 1, first, count); }

// upper bound of the sine arguments in turns for a flat weight set (siren_layout.h): what tha4_student_create checks against 256
double emu_sine_argument_bound_turns(const tha4_student_weights* w) { return sine_argument_bound_turns(to_view(w)); }
double emu_sine_turns_limit() { return kSineTurnsLimit; }
float emu_sin_omega(float z) { return sin_omega(z); }
float emu_sin_u(float u) { return sin_u(u); }
float emu_sine_scale16() { return kSineScale16; }      // what generation 2 folds into biases / z hand-off: omega_0 / 2 pi (turns) or omega_0

void emu_student_destroy(void* h) { delete static_cast<EmuStudent*>(h); }

}  // extern "C"
