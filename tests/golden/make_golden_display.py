#!/usr/bin/env python3
"""Golden fixture for the display epilogue from the UNMODIFIED reference:

    python tests/golden/make_golden_display.py      (build container only)

Imports the reference's `convert_linear_to_srgb` (src/tha4/image_util.py:56-58 -> `torch_linear_to_srgb`,
src/tha4/shion/base/image_util.py:31-33) and applies it exactly where the puppeteers do
(src/tha4/app/character_model_ifacialmocap_puppeteer.py:325-349): clip((x+1)/2) -> convert_linear_to_srgb ->
optional blend_with_background (:377-381) -> CHW->HWC, *255, .byte().  The app module itself cannot be imported (wx is
not installed), so the three tensor statements around the imported function are restated here verbatim in structure.
Writes tests/golden/display_io.npz: the synthetic input frame, and for {lambda_00 posed frame, synthetic frame} x
{no background, green, blue, black, white} the uint8 [H,W,4] result.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
HERE = os.path.dirname(os.path.abspath(__file__))

from tha4.image_util import convert_linear_to_srgb  # noqa: E402  (reference, unmodified)

BACKGROUNDS = {"none": None, "green": (0.0, 1.0, 0.0), "blue": (0.0, 0.0, 1.0), "black": (0.0, 0.0, 0.0), "white": (1.0, 1.0, 1.0)}


def puppeteer_display(frame: torch.Tensor, bg):
    output_image = frame.float()
    output_image = torch.clip((output_image + 1.0) / 2.0, 0.0, 1.0)                 # puppeteer :326
    output_image = convert_linear_to_srgb(output_image)                             # :327 (reference function)
    if bg is not None:                                                               # :329-345
        background = torch.zeros(4, output_image.shape[1], output_image.shape[2])
        background[3, :, :] = 1.0
        for c in range(3):
            background[c, :, :] = bg[c]
        alpha = output_image[3:4, :, :]                                              # blend_with_background :377-381
        color = output_image[0:3, :, :]
        new_color = color * alpha + (1.0 - alpha) * background[0:3, :, :]
        output_image = torch.cat([new_color, background[3:4, :, :]], dim=0)
    c, h, w = output_image.shape
    output_image = 255.0 * torch.transpose(output_image.reshape(c, h * w), 0, 1).reshape(h, w, c)   # :347
    return output_image.byte().numpy()                                               # :348


def main():
    posed = torch.from_numpy(np.load(os.path.join(HERE, "student_lambda_00_io.npz"))["ref32_full_out0"][0])
    rng = np.random.default_rng(0)
    synth = rng.uniform(-1.2, 1.2, (4, 64, 48)).astype(np.float32)
    synth[:, 0, :8] = np.array([-1.0, -0.9937383901, -0.99373, -0.9937, 1.0, 0.0, -1.0000001, 0.9999999], np.float32)  # sRGB knee / clips
    out = {"synth_f32": synth}
    for name, bg in BACKGROUNDS.items():
        out[f"posed_{name}"] = puppeteer_display(posed, bg)
        out[f"synth_{name}"] = puppeteer_display(torch.from_numpy(synth), bg)
    np.savez_compressed(os.path.join(HERE, "display_io.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
