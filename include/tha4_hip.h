/* tha4_hip.h - C ABI of the MI355X-native THA4 poser hot path (libtha4_hip.so).
 *
 * The reference (pkhungurn/talking-head-anime-4-demo) has no FFI: its plugin boundary is the Python
 * ABC `Poser` (src/tha4/poser/poser.py:132-162) implemented by `GeneralPoser02`
 * (src/tha4/poser/general_poser_02.py:10-98) and built by `mode_14.create_poser`
 * (src/tha4/poser/modes/mode_14.py:134-162).  This header is the native boundary underneath that
 * surface: each entry point names the reference call it replaces.  The Python mirror of the
 * reference interface (tha4_amd.poser.modes.mode_14) binds these symbols with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only (no torch / HIP types in signatures; a stream is an
 * opaque `void*` = hipStream_t).  All `*_dev` pointers are device pointers on the handle's GPU.
 * Every function returns 0 on success or a negative tha4_status; nothing throws across the ABI.
 * `tha4_*_pose` never allocates, never synchronises and enqueues all work on the given stream
 * (callers bracket it with stream events exactly like full_manual_poser.py:388-398 does).
 * A handle owns ONE workspace: consecutive calls on the same stream are ordered by the stream; when the stream
 * differs from the previous call's, an event (created with the handle) is recorded on the PREVIOUS stream and the new
 * stream waits for it (no host sync).  A stream given to a pose call must therefore stay valid until the handle's next
 * pose call has returned.  A handle must not be used from two host threads at once (the reference is single-threaded,
 * SURVEY.md §8b); use one handle per thread.
 */
#ifndef THA4_HIP_H
#define THA4_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THA4_ABI_VERSION 6

typedef enum tha4_status {
  THA4_OK = 0,
  THA4_ERR_INVALID_ARGUMENT = -1,  /* null pointer, bad batch, wrong architecture dims            */
  THA4_ERR_HIP = -2,               /* a HIP runtime call failed (message in tha4_last_error())    */
  THA4_ERR_NO_DEVICE = -3,         /* no gfx950 device / device index out of range                */
  THA4_ERR_BATCH_TOO_LARGE = -4,   /* batch > max_batch given at create                           */
  THA4_ERR_NUMERIC_RANGE = -5      /* full model: an earlier pose call left the operand range (see tha4_full_numeric_status) */
} tha4_status;

/* One Conv2d(kernel_size=1) layer as stored in the reference state_dict
 * (src/tha4/nn/siren/vanilla/siren.py:23-29): weight [out_ch][in_ch] row-major fp32 (the [O,I,1,1]
 * tensor), bias [out_ch].  Host pointers, only read during tha4_student_create. */
typedef struct tha4_linear {
  const float* weight;
  const float* bias;
  int32_t out_ch;
  int32_t in_ch;
} tha4_linear;

/* The two student modules of mode_14 (src/tha4/poser/modes/mode_14.py:93-131), in state_dict order
 * (SURVEY.md Appendix B):
 *   face_sine[i]    <- face_morpher.pt  siren.sine_layers.{i}.linear     41->128, 7 x 128->128
 *   face_last       <- face_morpher.pt  siren.last_linear                128->4
 *   body_sine[l][j] <- body_morpher.pt  siren_layers.{l}.{j}.linear      47->360->360->180,
 *                                                                        227->180->180->90, 137->90->90->90
 *   body_last       <- body_morpher.pt  last_linear                      90->7 (grid dx,dy | alpha | colour RGBA) */
typedef struct tha4_student_weights {
  tha4_linear face_sine[8];
  tha4_linear face_last;
  tha4_linear body_sine[3][3];
  tha4_linear body_last;
} tha4_student_weights;

/* Optional fp32 affine_grid axes (src/tha4/nn/siren/morpher/siren_morpher_03.py:92-99).  The exact
 * values are the dyadic (2j+1)/S-1; ATen's fp32 linspace is off by one ulp in many entries and SIREN
 * amplifies that to ~1e-4 in the image, so a caller that wants to track a particular PyTorch build
 * bit-closely passes the axes that build produces.  Any pointer may be NULL (exact values used). */
typedef struct tha4_position_axes {
  const float* axis128; /* [128] */
  const float* axis256; /* [256] */
  const float* axis512; /* [512] */
} tha4_position_axes;

/* Display epilogue fused into the kernel that composes the posed frame: what every real-time caller
 * runs on the device right after pose() (src/tha4/app/character_model_ifacialmocap_puppeteer.py:325-349,377-381;
 * src/tha4/image_util.py:56-58): clip((x+1)/2,0,1) -> linear->sRGB on RGB -> optional blend over an opaque background
 * colour -> CHW->HWC -> *255 -> truncate to uint8.  The values are converted while still in registers: the fp32 frame
 * is neither re-read nor (when the fp32 output pointer is NULL) written. */
typedef struct tha4_display {
  uint8_t* rgba8_dev;          /* device, uint8 [B,512,512,4]; NULL = no display output                         */
  const float* background_rgb; /* HOST, 3 floats in [0,1] (sRGB-encoded, as the reference blends) or NULL = keep alpha */
} tha4_display;

/* Optional extra outputs of SirenMorpher03.forward / TwoStepPoserComputationProtocol
 * (src/tha4/nn/siren/morpher/siren_morpher_03.py:133-139, src/tha4/poser/modes/mode_14.py:85-88).
 * Device pointers, each may be NULL.  Output index in the reference list is given in brackets. */
typedef struct tha4_student_aux {
  float* alpha_dev;        /* [1] [B,1,512,512] raw alpha (no sigmoid)      */
  float* color_change_dev; /* [2] [B,4,512,512]                               */
  float* warped_dev;       /* [3] [B,4,512,512] grid_sample of the input     */
  float* grid_change_dev;  /* [4] [B,2,512,512] normalised offsets (x, y)    */
  float* face_dev;         /* [5] [B,4,128,128] face morpher output          */
  tha4_display display;    /* fused display epilogue of output 0 (ABI v3)    */
} tha4_student_aux;

typedef struct tha4_student tha4_student; /* opaque */

/* ABI version of the loaded library (== THA4_ABI_VERSION of the header it was built from). */
int tha4_abi_version(void);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char* tha4_last_error(void);

/* Replaces: mode_14.load_face_morpher / load_body_morpher + GeneralPoser02.get_modules
 * (mode_14.py:93-131, general_poser_02.py:41-49): validates the architecture, packs the weights into
 * the MFMA-fragment-linear HBM image, uploads them to `device` and sizes the workspace for
 * `max_batch` frames.  `axes` may be NULL. */
int tha4_student_create(const tha4_student_weights* weights, const tha4_position_axes* axes,
                        int device, int max_batch, tha4_student** out);

/* Same with options.  flags:
 *   THA4_STUDENT_EXACT_FP32  run the contractions on v_mfma_f32_16x16x4_f32 (exact fp32 products, csrc/siren_kernels.h)
 *                            instead of the default 3-pass fp16 hi/lo split on v_mfma_f32_16x16x32_f16
 *                            (csrc/siren16_kernels.h: 22-bit operands, fp32 accumulate; same distance to the fp64
 *                            reference, ~3x faster).  Both stay inside the 1e-3 gate; this flag is the A/B switch. */
#define THA4_STUDENT_EXACT_FP32 1
int tha4_student_create_ex(const tha4_student_weights* weights, const tha4_position_axes* axes,
                           int device, int max_batch, int flags, tha4_student** out);

/* Replaces: GeneralPoser02.get_posing_outputs -> TwoStepPoserComputationProtocol "all_outputs"
 * (general_poser_02.py:63-79, mode_14.py:58-90) for a batch of `batch` frames.
 *   image_dev          fp32 [B,4,512,512] (values in [-1,1], linear RGB premultiplied by alpha)
 *   image_batch_stride floats between consecutive frames' images; 0 = one image shared by the batch
 *                      (the reference needs B identical copies for that; 4*512*512 = dense batch)
 *   pose_dev           fp32 [B,45]
 *   out_blended_dev    fp32 [B,4,512,512]  output index 0 (the posed frame); must not alias image_dev; may be NULL
 *                      when aux->display.rgba8_dev is given (the caller only wants the displayable frame)
 *   aux                optional outputs 1..5, may be NULL
 *   stream             hipStream_t to enqueue on (NULL = the null stream) */
int tha4_student_pose(tha4_student* h, const float* image_dev, int64_t image_batch_stride,
                      const float* pose_dev, int batch, float* out_blended_dev,
                      const tha4_student_aux* aux, void* stream);

/* Hot-swap of the character (the puppeteers' "load model" action,
 * character_model_ifacialmocap_puppeteer.py:383-399 -> CharacterModel.get_poser, character_model.py:23-33): packs a new
 * pair of student state_dicts and overwrites the parameter blob of the live handle IN PLACE - no device allocation,
 * the workspace, max_batch, position axes and the handle itself stay valid.  Waits for pose calls in flight on the
 * handle's device before copying (a rare, host-synchronous operation; `tha4_student_pose` itself never synchronises). */
int tha4_student_set_weights(tha4_student* h, const tha4_student_weights* weights);

/* Replaces: GeneralPoser02.free (general_poser_02.py:84-85).  NULL is a no-op. */
void tha4_student_destroy(tha4_student* h);

/* Test / tuning hook (no reference counterpart): copies to host memory the inter-level hand-off image the most recent pose
 * call left in the handle's workspace for batch slot `frame`: which = 0: z1 = s W_{1,0}[:, :180] h_0
 * at 128^2 (12 x 16 x 128^2 floats), 1: z2 = s W_{2,0}[:, :90] h_1 at 256^2, s = tha4_student_hand_off_scale() (6 x 16 x 256^2 floats); fp32, layout
 * [block][4][pixels][4] with channel = 16 block + 4 g + j (csrc/siren_layout.h z_offset).  Synchronises the device. */
int tha4_student_debug_read(tha4_student* h, int which, int frame, float* host_out);
/* The factor the hand-off images (and every sine argument) carry: omega_0 / (2 pi) = 4.7746 - the sine takes turns - in the
 * default kernels, 1 with THA4_STUDENT_EXACT_FP32. */
float tha4_student_hand_off_scale(const tha4_student* h);

/* Introspection used by bench.py / tests (no reference counterpart). */
int tha4_student_max_batch(const tha4_student* h);
int tha4_student_device(const tha4_student* h);
/* Time (ms) of the most recent tha4_student_pose on this handle measured with HIP events recorded
 * on the SAME stream the kernels were launched on; enabled with tha4_student_set_timing(h, 1).
 * Reading it synchronises on the stop event.  kernel: 0 posebias, 1 face, 2 level0, 3 level1,
 * 4 level2(+warp), -1 whole call.  In the default (fp16-split) generation the pose bias is computed inside the kernels and
 * face + level 0 share one launch: slots 0 and 1 then measure an empty interval and slot 2 the merged kernel. */
int tha4_student_set_timing(tha4_student* h, int enable);
int tha4_student_last_ms(tha4_student* h, int kernel, float* ms_out);

/* ------------------------------------------------------------------------------------------------
 * Full THA4 system (reference mode_07: eyebrow_decomposer -> eyebrow_morphing_combiner -> face_morpher
 * -> body_morpher -> upscaler; src/tha4/poser/modes/mode_07.py:54-134, 272-315).
 * ------------------------------------------------------------------------------------------------ */

/* One entry of a state_dict: name exactly as in the reference module's state_dict(), fp32 host data. */
typedef struct tha4_named_tensor {
  const char* name;
  const float* data;
  int32_t ndim;
  int64_t dims[4];
} tha4_named_tensor;

/* The five state_dicts in the order of mode_07.Network (mode_07.py:24-29). */
typedef struct tha4_full_weights {
  const tha4_named_tensor* tensors[5];
  int32_t counts[5];
} tha4_full_weights;

#define THA4_FULL_NUM_OUTPUTS 33

typedef struct tha4_full tha4_full; /* opaque */

/* Replaces: the five mode_07.load_* loaders + GeneralPoser02.get_modules (mode_07.py:137-269,
 * general_poser_02.py:41-49).  `eyebrow_morphed_image_index` as in mode_07.create_poser (:275).
 * = tha4_full_create_ex(..., num_networks = 5, flags = THA4_FULL_EXACT_DECOMPOSER_OUTER): the recommended (mixed) plan, the one the
 * Python mirror creates by default (ABI v6; rounds 1-5: flags = 0, the pure fp16 hi/lo plan - still available through _ex). */
int tha4_full_create(const tha4_full_weights* weights, int eyebrow_morphed_image_index, int device, int max_batch,
                     tha4_full** out);

/* Same with the number of networks: 5 = mode_07; 3 = the reference's mode_12 (src/tha4/poser/modes/mode_12.py:169-202,
 * the teacher of the face-morpher distillation, siren_face_morpher_00_trainer.py:23-26): eyebrow_decomposer ->
 * eyebrow_morphing_combiner -> face_morpher only (weights->tensors[3], [4] are ignored).  Such a handle produces
 * outputs 11..32 of the list below (mode_12's list = face_morpher 8 + combiner 8 + decomposer 6, mode_12.py:92-97).
 *
 * flags (ABI v4):
 *   THA4_FULL_EXACT_FP32  plan EVERY convolution on the exact-fp32 kernels (v_mfma_f32_16x16x4_f32 on fp32 operands: fp32's own
 *                         range, no fp16 hi/lo staging) and every normalisation through the finalize kernel.  The default plan
 *                         (0) multiplies fp16 hi/lo operand halves (22 significant bits, |operand| <= 65504 after normalise +
 *                         activate; tha4_full_numeric_status reports a violation): ~2.5-3x faster, within 1e-3 of this plan on
 *                         every tested parameter set.  This is the plan to re-create the handle with when THA4_ERR_NUMERIC_RANGE
 *                         is reported for weights / inputs that are legitimate in fp32 (the reference computes in plain fp32,
 *                         mode_07.py:137-315 loads whatever the .pt files hold).
 *   THA4_FULL_EXACT_DECOMPOSER (ABI v6)  the MIXED plan: only the eyebrow decomposer (eyebrow_decomposer_00.py:46-64; network 0) on the exact-fp32
 *                         kernels, everything else on the default fp16 hi/lo plan.  Attribution of the split's share of the posed frame's
 *                         error (fp64 oracle, profiles/parity_r06/): the decomposer carries ~90 % of it (its outputs are thresholded layers the
 *                         four later networks all consume), the two U-Nets < 3 %.  The reference caches the decomposer's outputs while the
 *                         image is unchanged (mode_07.py:56-67) and so does this library: the flag costs nothing per steady frame and one
 *                         slower decomposer pass per new image.  Ignored (implied) with THA4_FULL_EXACT_FP32.
 *   THA4_FULL_EXACT_DECOMPOSER_OUTER (ABI v6)  the same for the decomposer's first, down-sampling, up-sampling and head convolutions only; the eleven
 *                         512 -> 512 convolutions of its 16x16 bottleneck (a fifth of its share of the error, most of its launches) stay on the
 *                         default plan.  Ignored (implied) with either flag above. */
#define THA4_FULL_EXACT_FP32 1u
#define THA4_FULL_EXACT_DECOMPOSER 2u
#define THA4_FULL_EXACT_DECOMPOSER_OUTER 4u
int tha4_full_create_ex(const tha4_full_weights* weights, int eyebrow_morphed_image_index, int device, int max_batch,
                        int num_networks, uint32_t flags, tha4_full** out);
/* The flags the handle was created with. */
int tha4_full_flags(const tha4_full* h);

/* Replaces: GeneralPoser02.get_posing_outputs -> FiveStepPoserComputationProtocol (mode_07.py:54-134).
 *   outputs_dev[i]  device pointer for output i of the reference's 33-entry list (order mode_07.py:126-132:
 *                   upscaler 0-4, face_morphed_full 5, body_morpher 6-10, face_morpher 11-18,
 *                   eyebrow_morphing_combiner 19-26, eyebrow_decomposer 27-32), fp32 NCHW [B,C,S,S];
 *                   NULL = not wanted: nothing is written for it at all unless a later stage of the pipeline reads it
 *                   (outputs 5, 6, 9, 11 and 19 + eyebrow_morphed_image_index then go to the handle's workspace); at least
 *                   one must be given (e.g. the distiller asks for 0,1,2,3,5 only, siren_morpher_protocols_03.py:56-72).
 *   reuse_decomposer  non-zero: the image (and batch) is unchanged since the previous call on this handle,
 *                   reuse the cached eyebrow-decomposer result (the reference detects this with a
 *                   max|delta| device->host sync, mode_07.py:56-61; here the caller states it). */
int tha4_full_pose(tha4_full* h, const float* image_dev, int64_t image_batch_stride, const float* pose_dev, int batch,
                   float* const* outputs_dev, int reuse_decomposer, void* stream);

/* Numeric-range guard of the full model (no reference counterpart: the reference computes in plain fp32).  The convolutions
 * stage their normalised + activated operands as unscaled fp16 hi + lo halves (22 significant bits, |v| <= 65504); beyond
 * that range - or with NaN / inf in weights or inputs - the affected outputs are not finite.  The kernels detect this where
 * every such fault ends up (the scale/shift of the next normalisation, a network's head block) and set a sticky flag in
 * host-visible memory.  tha4_full_pose returns THA4_ERR_NUMERIC_RANGE once, WITHOUT enqueueing work, when it finds the flag
 * set by an earlier call (it never synchronises, so the call that faulted itself returns THA4_OK); this function is the
 * synchronous check: synchronize != 0 waits for the handle's device first.  Both report-and-clear. */
int tha4_full_numeric_status(tha4_full* h, int synchronize);
/* How a numeric fault of an EARLIER call is delivered (ABI v4).  THA4_FAULT_REFUSE_NEXT (default): the next tha4_full_pose
 * returns THA4_ERR_NUMERIC_RANGE once without enqueueing that call (a caller that never polls still learns of the fault, at the
 * price of one refused frame).  THA4_FAULT_STATUS_ONLY: tha4_full_pose never refuses; the fault is reported through
 * tha4_full_numeric_status only - for real-time callers that poll it (synchronize = 0 costs nothing) and cannot lose a frame to a
 * deferred report; while a fault is pending (raised, not yet polled) reuse_decomposer is ignored - the persistent eyebrow-decomposer
 * outputs may be the faulted call's, so they are recomputed every call until the status is read.  The flag is peeked at ENQUEUE time
 * without synchronising: "pending" starts with the first call enqueued after the fault has become host-visible - calls enqueued while
 * the faulting call was still queued or running (a caller that submits several frames ahead) may still reuse the decomposer outputs
 * of the faulting call; a caller that needs the guarantee from the very next call polls with synchronize != 0 after each new image.
 * Either way the outputs of the faulting call itself are not finite / not trustworthy. */
#define THA4_FAULT_REFUSE_NEXT 0
#define THA4_FAULT_STATUS_ONLY 1
int tha4_full_set_fault_policy(tha4_full* h, int policy);

/* tha4_full_pose with the display epilogue of output 0 (the upscaler's merged frame) fused into the kernel that composes
 * it; `display` may be NULL (= tha4_full_pose).  With display->rgba8_dev set, outputs_dev may be all-NULL. */
int tha4_full_pose_ex(tha4_full* h, const float* image_dev, int64_t image_batch_stride, const float* pose_dev, int batch,
                      float* const* outputs_dev, int reuse_decomposer, const tha4_display* display, void* stream);

void tha4_full_destroy(tha4_full* h);
int tha4_full_max_batch(const tha4_full* h);

/* Per-op timing of the full model (ABI v5; measurement aid, no reference counterpart - the reference's callers bracket pose() with
 * torch.cuda.Event on the current stream, full_manual_poser.py:388-398).  A frame is a static schedule of tha4_full_num_ops() ops
 * (eyebrow-decomposer ops first, then the rest; an op is one launch, two for a convolution with a K split).  With timing enabled
 * every pose call records a HIP event on the launch stream in front of every op and behind the last one (they cost ~1-3 us of chain
 * time each: a timed call is slower than an untimed one); tha4_full_last_op_ms waits for the last timed call and returns the time
 * between consecutive events (0 for decomposer ops the call reused).  tha4_full_op_info: a label naming the reference layer + the kernel
 * that runs it, and the as-written GFLOP (2 x MAC per frame) of that layer - what bench.py prices `roofline.achieved` of the full
 * model's dominant launch class with.  The label pointer is owned by the handle. */
int tha4_full_set_timing(tha4_full* h, int enable);
int tha4_full_num_ops(const tha4_full* h);
int tha4_full_op_info(const tha4_full* h, int index, const char** label, double* gflop);
int tha4_full_last_op_ms(tha4_full* h, float* ms, int capacity);
int tha4_full_num_networks(const tha4_full* h);

/* ------------------------------------------------------------------------------------------------
 * Callers / data formats either side of the path.  Stateless, device pointers.
 * ------------------------------------------------------------------------------------------------ */

/* Replaces the on-device post-processing every puppeteer runs after pose()
 * (src/tha4/app/character_model_ifacialmocap_puppeteer.py:325-349,377-381; src/tha4/image_util.py:56-58):
 * clip((x+1)/2,0,1) -> linear->sRGB on RGB -> optional blend over an opaque background colour ->
 * CHW->HWC -> *255 -> truncate to uint8.   frames_dev fp32 [B,4,H,W] -> out_dev uint8 [B,H,W,4].
 * background_rgb: 3 host floats in [0,1] or NULL (keep the frame's alpha). */
int tha4_display_rgba8(const float* frames_dev, int batch, int height, int width, const float* background_rgb,
                       uint8_t* out_dev, void* stream);

/* Replaces extract_pytorch_image_from_PIL_image (src/tha4/shion/base/image_util.py:127-149,194-198) for an
 * RGBA8 image already on the device: /255 -> sRGB->linear on RGB -> premultiply alpha -> *2-1 -> HWC->CHW.
 * rgba_dev uint8 [B,H,W,4] -> out_dev fp32 [B,4,H,W]. */
int tha4_ingest_rgba8(const uint8_t* rgba_dev, int batch, int height, int width, float* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* THA4_HIP_H */
