#!/usr/bin/env python3
"""Golden fixtures for the BATCHED full model (mode_07) from the UNMODIFIED reference:

    python tests/golden/make_golden_full_batch.py      (build container only)

1. `full_batch_io.npz` - the teacher-in-the-loop call of the distiller
   (src/tha4/nn/siren/morpher/siren_morpher_protocols_03.py:102-108: `poser.get_posing_outputs(image[B], pose[B])`
   with B DISTINCT images, outputs 0,1,2,3,5 consumed, :56-72): B = 4 images of the SURVEY.md §8d config-5 recipe
   (oracle.student_oracle.synthetic_image, seeds 99..102), 4 poses (seed 777), standard synthetic weights
   (seed 20260925).  Stored: a stride-5 pixel subset of outputs 0,1,2,3,5 for all four frames (fp32 run) and of ALL 33
   outputs for frame 1 (fp32 and fp64 runs).
2. `full_adv_io.npz` - the ADVERSARIAL-RANGE parameter set `synth_full_weights(seed, small_gain=6, conv_gain=1000,
   film_gain=3)`: warps of +-0.3 (x6 the standard set), pre-normalisation activations of O(1e3), O(1) FiLM modulation;
   B = 2 distinct images (lambda_00, synthetic seed 99), poses seed 778; all 33 outputs, stride-5 subset, fp32 and fp64,
   plus the reference's own fp32-vs-fp64 distance per output (the yardstick of the test: the chained warps make this
   set ill conditioned for ANY fp32 implementation, the reference included).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from make_golden_full import build_reference_modules, m07, GeneralPoser02, get_pose_parameters  # noqa: E402  (reference imports)

from oracle import full_oracle as fo  # noqa: E402
from oracle.student_oracle import random_poses, synthetic_image  # noqa: E402

SUB5 = slice(2, None, 5)
SEED = 20260925
ADV = dict(small_gain=6.0, conv_gain=1000.0, film_gain=3.0)
TEACHER_OUTPUTS = (0, 1, 2, 3, 5)


def make_poser(mods):
    return GeneralPoser02(
        image_size=512, module_loaders={k: (lambda k=k: mods[k]) for k in mods},
        pose_parameters=get_pose_parameters().get_pose_parameter_groups(),
        output_list_func=m07.FiveStepPoserComputationProtocol(2).compute_func(),
        subrect=None, device=torch.device("cpu"), output_length=33, default_output_index=0)


def run_both(w, images, poses):
    """fp32 then fp64 run of the reference on one batch; returns (ref32, ref64): lists of 33 numpy arrays [B,...]."""
    mods = build_reference_modules()
    for k, mod in mods.items():
        mod.load_state_dict({kk: torch.from_numpy(v) for kk, v in w[k].items()}, strict=True)
        mod.train(False)
    torch.set_num_threads(8)
    with torch.no_grad():
        ref32 = [o.numpy().copy() for o in make_poser(mods).get_posing_outputs(torch.from_numpy(images), torch.from_numpy(poses))]
    torch.set_default_dtype(torch.float64)          # morpher_00.py:51 creates t with the default dtype
    for mod in mods.values():
        mod.double()
    with torch.no_grad():
        ref64 = [o.numpy().copy() for o in make_poser(mods).get_posing_outputs(torch.from_numpy(images).double(),
                                                                                  torch.from_numpy(poses).double())]
    torch.set_default_dtype(torch.float32)
    return ref32, ref64


def main():
    # ---- 1. dense batch, standard weights ---------------------------------------------------------------
    seeds = [99, 100, 101, 102]
    images = np.stack([synthetic_image(seed=s) for s in seeds])
    poses = random_poses(4, seed=777)
    ref32, ref64 = run_both(fo.synth_full_weights(SEED), images, poses)
    io = {"image_seeds": np.array(seeds), "poses": poses, "seed": np.int64(SEED)}
    for k in TEACHER_OUTPUTS:
        io[f"ref32_sub5_out{k}"] = ref32[k][:, :, SUB5, SUB5]
    for k in range(33):
        io[f"ref32_frame1_sub5_out{k}"] = ref32[k][1][:, SUB5, SUB5]
        io[f"ref64_frame1_sub5_out{k}"] = ref64[k][1][:, SUB5, SUB5].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "full_batch_io.npz"), **io)
    noise = {fo.OUTPUT_NAMES[k]: float(np.abs(ref32[k] - ref64[k]).max()) for k in range(33)}
    print("dense batch fp32-vs-fp64:", {k: f"{v:.2e}" for k, v in noise.items() if k.startswith("up_") or k == "face_morphed_full"})

    # ---- 2. adversarial-range weights ---------------------------------------------------------------------
    lam = np.load(os.path.join(HERE, "student_lambda_00_io.npz"))["image_f32"]
    images = np.stack([lam, synthetic_image(seed=99)])
    poses = random_poses(2, seed=778)
    ref32, ref64 = run_both(fo.synth_full_weights(SEED, **ADV), images, poses)
    io = {"poses": poses, "seed": np.int64(SEED), "gains": np.array([ADV["small_gain"], ADV["conv_gain"], ADV["film_gain"]])}
    for k in range(33):
        io[f"ref32_sub5_out{k}"] = ref32[k][:, :, SUB5, SUB5]
        io[f"ref64_sub5_out{k}"] = ref64[k][:, :, SUB5, SUB5].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "full_adv_io.npz"), **io)
    adv_noise = {fo.OUTPUT_NAMES[k]: float(np.abs(ref32[k] - ref64[k]).max()) for k in range(33)}
    rng = {fo.OUTPUT_NAMES[k]: [float(ref32[k].min()), float(ref32[k].max())] for k in range(33)}
    with open(os.path.join(HERE, "full_batch_noise.json"), "w") as f:
        json.dump({"dense_fp32_vs_fp64_maxabs": noise, "adv_fp32_vs_fp64_maxabs": adv_noise, "adv_range": rng, "adv_gains": ADV,
                   "seed": SEED}, f, indent=1)
    print("adversarial fp32-vs-fp64:", {k: f"{v:.2e}" for k, v in adv_noise.items()})
    print("adversarial grids:", {k: v for k, v in rng.items() if "grid" in k or k in ("face_0", "comb_0")})


if __name__ == "__main__":
    main()
