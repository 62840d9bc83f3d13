"""Architecture inventory and synthetic parameters of the full THA4 system (reference mode_07).

The reference checkout ships no full-model weights (``data/tha4/placeholder.txt``; download instructions
README.md:166-181), so benchmarks and parity fixtures use deterministic synthetic state_dicts with the
reference's exact key/shape layout (verified by ``load_state_dict(strict=True)`` into the reference modules in
tests/golden/make_golden_full.py).  This module is plain numpy: no reference code is imported.

Shapes follow SURVEY.md Appendix A; constructor arguments are those of mode_07.py:137-269.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

NETS = ["eyebrow_decomposer", "eyebrow_morphing_combiner", "face_morpher", "body_morpher", "upscaler"]  # mode_07.py:24-29
NUM_EYEBROW, NUM_FACE, NUM_ROT = 12, 27, 6                                                               # mode_07.py:42-44

UNET_BODY = dict(in_ch=4, out_ch=7, model=64, mults=[1, 2, 4, 4, 4], attn=[False, False, False, False, True],
                 cond_in=6, cond=256, heads=8)                       # mode_07.py:211-228
UNET_UP = dict(in_ch=4, out_ch=7, model=32, mults=[1, 2, 4, 8, 8, 8], attn=[False] * 5 + [True],
               cond_in=6, cond=256, heads=8)                         # mode_07.py:242-259


# --------------------------------------------------------------------------------------
# architecture enumeration -> state_dict keys and shapes
# --------------------------------------------------------------------------------------

def encdec_param_shapes(in_ch: int, pose: int, heads: Dict[str, tuple]) -> Dict[str, tuple]:
    """PoserEncoderDecoder00 (prefix 'body.') / FaceMorpher08 (no prefix) share the topology:
    start 64 ch, 3 stride-2 stages to 512 ch, 6 bottleneck blocks, 3 transposed-conv stages."""
    s: Dict[str, tuple] = {}

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    s["downsample_blocks.0.0.weight"] = (64, in_ch, 3, 3)
    norm("downsample_blocks.0.1", 64)
    c = 64
    for i in range(1, 4):
        s[f"downsample_blocks.{i}.0.weight"] = (2 * c, c, 4, 4)
        norm(f"downsample_blocks.{i}.1", 2 * c)
        c *= 2
    s["bottleneck_blocks.0.0.weight"] = (512, 512 + pose, 3, 3)
    norm("bottleneck_blocks.0.1", 512)
    for i in range(1, 6):
        s[f"bottleneck_blocks.{i}.resnet_path.0.weight"] = (512, 512, 3, 3)
        norm(f"bottleneck_blocks.{i}.resnet_path.1", 512)
        s[f"bottleneck_blocks.{i}.resnet_path.3.weight"] = (512, 512, 3, 3)
        norm(f"bottleneck_blocks.{i}.resnet_path.4", 512)
    for i in range(3):
        s[f"upsample_blocks.{i}.0.weight"] = (c, c // 2, 4, 4)       # ConvTranspose2d weight is [in, out, kh, kw]
        norm(f"upsample_blocks.{i}.1", c // 2)
        c //= 2
    return s


def unet_blocks(a: dict):
    """Yield (prefix, kind, cin, cout, mode) for every block of unet.py's Unet in execution order.
    kind: 'res' | 'attn'; mode: 'same' | 'down' | 'up'.  Also returns the skip bookkeeping."""
    chans = [a["model"]]
    cur = a["model"]
    down = []
    L = len(a["mults"])
    for i in range(L):
        out = a["model"] * a["mults"][i]
        down.append((f"down_blocks.{i}.res_blocks.0", "res", cur, out, "same"))
        if a["attn"][i]:
            down.append((f"down_blocks.{i}.attention_blocks.0", "attn", out, out, "same"))
        chans.append(out)
        if i < L - 1:
            down.append((f"down_blocks.{i}.downsample", "res", out, out, "down"))
            chans.append(out)
        cur = out
    middle = []
    for k in range(3):
        middle.append((f"middle_blocks.{2 * k}", "res", cur, cur, "same"))
        middle.append((f"middle_blocks.{2 * k + 1}.module", "attn", cur, cur, "same"))
    middle.append(("middle_blocks.6", "res", cur, cur, "same"))
    up = []
    for bi, i in enumerate(reversed(range(L))):
        out = a["model"] * a["mults"][i]
        for j in range(2):
            skip = chans.pop()
            up.append((f"up_blocks.{bi}.resnet_blocks.{j}", "res", (cur if j == 0 else out) + skip, out, "same"))
            if a["attn"][i]:
                up.append((f"up_blocks.{bi}.attention_blocks.{j}", "attn", out, out, "same"))
        if i > 0:
            up.append((f"up_blocks.{bi}.upsample", "res", out, out, "up"))
        cur = out
    assert not chans
    return down, middle, up


def unet_param_shapes(a: dict) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    C = a["cond"]
    s["time_embed.1.weight"] = (C, a["model"]); s["time_embed.1.bias"] = (C,)
    s["time_embed.3.weight"] = (C, C); s["time_embed.3.bias"] = (C,)
    s["cond_embed.0.weight"] = (C, a["cond_in"]); s["cond_embed.0.bias"] = (C,)
    s["cond_embed.2.weight"] = (C, C); s["cond_embed.2.bias"] = (C,)
    s["first_conv.weight"] = (a["model"], a["in_ch"], 3, 3); s["first_conv.bias"] = (a["model"],)
    down, middle, up = unet_blocks(a)
    for (p, kind, cin, cout, mode) in down + middle + up:
        if kind == "res":
            s[p + ".norm0.weight"] = (cin,); s[p + ".norm0.bias"] = (cin,)
            s[p + ".conv0.weight"] = (cout, cin, 3, 3); s[p + ".conv0.bias"] = (cout,)
            s[p + ".cond0_layers.1.weight"] = (2 * cout, C); s[p + ".cond0_layers.1.bias"] = (2 * cout,)
            s[p + ".norm1.weight"] = (cout,); s[p + ".norm1.bias"] = (cout,)
            s[p + ".conv1.weight"] = (cout, cout, 3, 3); s[p + ".conv1.bias"] = (cout,)
            s[p + ".cond1_layers.1.weight"] = (2 * cout, C); s[p + ".cond1_layers.1.bias"] = (2 * cout,)
            if cin != cout:
                s[p + ".skip.weight"] = (cout, cin, 1, 1); s[p + ".skip.bias"] = (cout,)
        else:
            s[p + ".norm.weight"] = (cin,); s[p + ".norm.bias"] = (cin,)
            s[p + ".qkv.weight"] = (3 * cin, cin, 1, 1); s[p + ".qkv.bias"] = (3 * cin,)
            s[p + ".conv.weight"] = (cin, cin, 1, 1); s[p + ".conv.bias"] = (cin,)
    s["last.0.weight"] = (a["model"],); s["last.0.bias"] = (a["model"],)
    s["last.2.weight"] = (a["out_ch"], a["model"], 3, 3); s["last.2.bias"] = (a["out_ch"],)
    return s


def full_param_shapes() -> Dict[str, Dict[str, tuple]]:
    def head(s, name, cout, bias=True):
        s[name + ".weight"] = (cout, 64, 3, 3)
        if bias:
            s[name + ".bias"] = (cout,)

    out: Dict[str, Dict[str, tuple]] = {}
    # eyebrow decomposer (eyebrow_decomposer_00.py:37-44): body + 2x(alpha, colour)
    s = {"body." + k: v for k, v in encdec_param_shapes(4, 0, {}).items()}
    for n, c in (("background_layer_alpha.0", 1), ("background_layer_color_change.0", 4),
                 ("eyebrow_layer_alpha.0", 1), ("eyebrow_layer_color_change.0", 4)):
        head(s, n, c)
    out["eyebrow_decomposer"] = s
    # eyebrow morphing combiner (eyebrow_morphing_combiner_00.py:38-45)
    s = {"body." + k: v for k, v in encdec_param_shapes(8, NUM_EYEBROW, {}).items()}
    s["morphed_eyebrow_layer_grid_change.weight"] = (2, 64, 3, 3)
    for n, c in (("morphed_eyebrow_layer_alpha.0", 1), ("morphed_eyebrow_layer_color_change.0", 4), ("combine_alpha.0", 1)):
        head(s, n, c)
    out["eyebrow_morphing_combiner"] = s
    # face morpher (face_morpher_08.py:44-99)
    s = dict(encdec_param_shapes(4, NUM_FACE, {}))
    s["iris_mouth_grid_change.weight"] = (2, 64, 3, 3)
    for n, c in (("iris_mouth_color_change.0", 4), ("iris_mouth_alpha.0", 1), ("eye_color_change.0", 4), ("eye_alpha.0", 1)):
        head(s, n, c)
    out["face_morpher"] = s
    out["body_morpher"] = {"body." + k: v for k, v in unet_param_shapes(UNET_BODY).items()}
    s = {"body." + k: v for k, v in unet_param_shapes(UNET_UP).items()}
    s["coarse_image_conv.weight"] = (32, 10, 3, 3)
    s["coarse_image_conv.bias"] = (32,)
    out["upscaler"] = s
    return out


def synth_full_weights(seed: int = 20260925, small_gain: float = 1.0, conv_gain: float = 1.0,
                       film_gain: float = 1.0, head_gains=(1.0, 1.0, 1.0)) -> Dict[str, Dict[str, np.ndarray]]:
    """Deterministic synthetic parameters (numpy PCG64) for all five networks, following SURVEY.md
    §8c: ordinary layers get He-normal conv weights / default-Linear-range weights / near-identity
    norm affines; every tensor the reference zero-initialises (ResBlock.conv1, attention.conv,
    U-Net last conv, coarse_image_conv, grid-change heads) gets N(0,(0.02/sqrt(fan_in))^2) weights
    and N(0,0.01^2) biases so that warps stay at the +-0.06 scale of real models.

    The three gains (all 1.0 by default: the parameter set every committed `full_synth_*` fixture uses) build the
    ADVERSARIAL-RANGE set of tests/golden/make_golden_full_batch.py: `small_gain` multiplies the zero-init family
    (larger warps, a residual stream that is no longer dominated by the skip path), `conv_gain` the He-normal
    convolutions that feed a normalisation layer (pre-normalisation activations and moments of O(conv_gain); the
    network function is unchanged up to the eps of the norm, so the reference stays well conditioned), `film_gain` the FiLM projections `cond*_layers` (O(1) scale/shift modulation as in a trained model).
    `head_gains` = (direct, grid, alpha): per-output-row gains of the two U-Nets' last convolution (`body.last.2`: rows 0-3 direct
    image, 4-5 grid change, 6 alpha logit - morpher_00.py:54-60, upscaler_02.py:84-90) - the MID-GAIN set of
    tests/golden/make_golden_full_midgain.py: U-Net outputs of O(0.2-0.3) and an alpha that spans most of (0, 1), so that the posed
    frame really carries the U-Net interior (with the standard set alpha ~ 0.5 +- 0.015 and direct ~ +-0.07: the frame is ~ half the
    warped input whatever the U-Net computes), while the reference stays well conditioned.
    The same random stream is drawn whatever the gains are."""
    rng = np.random.default_rng(seed)
    out: Dict[str, Dict[str, np.ndarray]] = {}
    for net, shapes in full_param_shapes().items():
        sd: Dict[str, np.ndarray] = {}
        for key, shp in shapes.items():
            small = (key.endswith("conv1.weight") or key.endswith("conv1.bias") or ".conv.weight" in key
                     or ".conv.bias" in key or "last.2." in key or "coarse_image_conv" in key
                     or "grid_change" in key)
            if len(shp) == 4:
                is_t = "upsample_blocks" in key
                fan_in = (shp[0] if is_t else shp[1]) * shp[2] * shp[3]
                normed = ("sample_blocks" in key or "bottleneck_blocks" in key or key.endswith(".conv0.weight"))   # followed by IN / GN
                std = small_gain * 0.02 / math.sqrt(fan_in) if small else (conv_gain if normed else 1.0) * math.sqrt(2.0 / fan_in)
                a = rng.standard_normal(shp) * std
            elif len(shp) == 2:
                a = rng.uniform(-1, 1, shp) / math.sqrt(shp[1])
                if "cond0_layers" in key or "cond1_layers" in key:
                    a = a * film_gain
            else:
                is_norm_w = key.endswith(".weight") and ("norm" in key or key.endswith(".1.weight") or key.endswith(".4.weight")
                                                         or key == "body.last.0.weight")
                if is_norm_w:
                    a = 1.0 + 0.1 * rng.standard_normal(shp)
                else:
                    a = (0.01 if small else 0.05) * rng.standard_normal(shp)
            if key in ("body.last.2.weight", "body.last.2.bias") and net in ("body_morpher", "upscaler"):
                rows = np.array([head_gains[0]] * 4 + [head_gains[1]] * 2 + [head_gains[2]], dtype=np.float64)
                a = a * rows.reshape((7,) + (1,) * (a.ndim - 1))
            sd[key] = a.astype(np.float32)
        out[net] = sd
    return out




def random_rgba_images(n: int, seed: int = 99, size: int = 512) -> np.ndarray:
    """`n` synthetic poser inputs [n,4,size,size] fp32 of the SURVEY.md §8d config-5 recipe: values U[-1,1) with the
    alpha channel U[0,1) and RGB premultiplied, i.e. linear rgb ~ U[0,1), a ~ U[0,1), image = cat(rgb * a, a) * 2 - 1
    (the convention of extract_pytorch_image_from_PIL_image, src/tha4/shion/base/image_util.py:127-149).  Drawn on a
    size/8 lattice and bilinearly enlarged so that warps sample a band-limited image.  Benchmark / test input only."""
    rng = np.random.default_rng(seed)
    lo = rng.uniform(0.0, 1.0, size=(n, 4, size // 8, size // 8))
    img = lo
    for _ in range(3):                              # x2 bilinear, align_corners=False, edge clamped (three times: x8)
        for axis in (2, 3):
            m = img.shape[axis]
            j = np.arange(2 * m)
            src = (j + 0.5) / 2.0 - 0.5
            i0 = np.clip(np.floor(src).astype(int), 0, m - 1)
            i1 = np.clip(i0 + 1, 0, m - 1)
            t = np.clip(src - np.floor(src), 0.0, 1.0)
            t = np.where(src < 0, 0.0, t)
            shape = [1, 1, 1, 1]
            shape[axis] = 2 * m
            img = np.take(img, i0, axis=axis) * (1.0 - t.reshape(shape)) + np.take(img, i1, axis=axis) * t.reshape(shape)
    img[:, 0:3] *= img[:, 3:4]
    return (img * 2.0 - 1.0).astype(np.float32)
