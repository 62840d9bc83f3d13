"""CPU oracle for the full THA4 system (reference mode_07: five networks, 33 outputs).

*** TEST INFRASTRUCTURE ONLY *** - same rules as oracle/student_oracle.py: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Parity status: PINNED BY LIVE REFERENCE WITH SYNTHETIC WEIGHTS.  The reference checkout ships no
full-model weights (data/tha4/placeholder.txt, README.md:166-181) and no tests.  This restatement
is pinned against the unmodified reference modules loaded with the deterministic synthetic
state_dicts of ``synth_full_weights`` (tests/golden/make_golden_full.py ->
tests/golden/full_synth_io.npz; tests/test_full_oracle_golden.py).  ``load_state_dict(strict=True)``
of those dicts into the reference modules also proves the key/shape layout below.

Restated reference code (paths relative to /root/reference/src/tha4):
  poser/modes/mode_07.py:54-134          five-step DAG, crops/pastes, resizes, output order
  nn/common/poser_encoder_decoder_00.py:99-121   conv encoder - bottleneck ResNet - convT decoder
  nn/conv.py:103-177, nn/resnet_block.py:52-67   conv3/conv4s2/convT4s2 + InstanceNorm(affine) + ReLU
  nn/common/poser_args.py:31-68          alpha (sigmoid) / colour (tanh) / grid heads
  nn/eyebrow_decomposer/eyebrow_decomposer_00.py:46-64
  nn/eyebrow_morphing_combiner/eyebrow_morphing_combiner_00.py:47-72
  nn/face_morpher/face_morpher_08.py:142-193
  nn/common/unet.py:90-97,154-165 (ResBlock/FiLM), :192-239 (attention), :365-376 (t-embedding),
                    :531-546 (Unet.forward), :642-658 (UnetWithFirstConvAddition.forward)
  nn/morpher/morpher_00.py:42-66, nn/upscaler/upscaler_02.py:59-96
  nn/image_processing_util.py:6-24,33-58  apply_rgb_change / apply_grid_change / apply_color_change
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

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


def synth_full_weights(seed: int = 20260925) -> Dict[str, Dict[str, np.ndarray]]:
    """Deterministic synthetic parameters (numpy PCG64) for all five networks, following SURVEY.md
    §8c: ordinary layers get He-normal conv weights / default-Linear-range weights / near-identity
    norm affines; every tensor the reference zero-initialises (ResBlock.conv1, attention.conv,
    U-Net last conv, coarse_image_conv, grid-change heads) gets N(0,(0.02/sqrt(fan_in))^2) weights
    and N(0,0.01^2) biases so that warps stay at the +-0.06 scale of real models."""
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
                std = 0.02 / math.sqrt(fan_in) if small else math.sqrt(2.0 / fan_in)
                a = rng.standard_normal(shp) * std
            elif len(shp) == 2:
                a = rng.uniform(-1, 1, shp) / math.sqrt(shp[1])
            else:
                is_norm_w = key.endswith(".weight") and ("norm" in key or key.endswith(".1.weight") or key.endswith(".4.weight")
                                                         or key == "body.last.0.weight")
                if is_norm_w:
                    a = 1.0 + 0.1 * rng.standard_normal(shp)
                else:
                    a = (0.01 if small else 0.05) * rng.standard_normal(shp)
            sd[key] = a.astype(np.float32)
        out[net] = sd
    return out


# --------------------------------------------------------------------------------------
# functional forward (torch ops, fp32 or fp64)
# --------------------------------------------------------------------------------------

def _T(sd, dt):
    import torch
    return {k: torch.from_numpy(v).to(dt) for k, v in sd.items()}


def _in_relu(x, P, p, relu=True):
    import torch.nn.functional as F
    y = F.instance_norm(x, weight=P[p + ".weight"], bias=P[p + ".bias"], eps=1e-5)
    return F.relu(y) if relu else y


def _encdec(P, pre, x, pose, bott):
    """PoserEncoderDecoder00.forward()[0] / FaceMorpher08 trunk."""
    import torch
    import torch.nn.functional as F
    f = _in_relu(F.conv2d(x, P[pre + "downsample_blocks.0.0.weight"], padding=1), P, pre + "downsample_blocks.0.1")
    for i in range(1, 4):
        f = _in_relu(F.conv2d(f, P[pre + f"downsample_blocks.{i}.0.weight"], stride=2, padding=1), P,
                     pre + f"downsample_blocks.{i}.1")
    if pose is not None:
        n, c = pose.shape
        f = torch.cat([f, pose.view(n, c, 1, 1).repeat(1, 1, bott, bott)], dim=1)
    f = _in_relu(F.conv2d(f, P[pre + "bottleneck_blocks.0.0.weight"], padding=1), P, pre + "bottleneck_blocks.0.1")
    for i in range(1, 6):
        q = pre + f"bottleneck_blocks.{i}.resnet_path."
        r = _in_relu(F.conv2d(f, P[q + "0.weight"], padding=1), P, q + "1")
        r = _in_relu(F.conv2d(r, P[q + "3.weight"], padding=1), P, q + "4", relu=False)
        f = f + r
    for i in range(3):
        f = _in_relu(F.conv_transpose2d(f, P[pre + f"upsample_blocks.{i}.0.weight"], stride=2, padding=1), P,
                     pre + f"upsample_blocks.{i}.1")
    return f


def _head(P, name, f, act):
    import torch
    import torch.nn.functional as F
    y = F.conv2d(f, P[name + ".weight"], P.get(name + ".bias"), padding=1)
    return torch.sigmoid(y) if act == "sigmoid" else (torch.tanh(y) if act == "tanh" else y)


def _warp(grid_change, image):
    """apply_grid_change / GridChangeApplier.apply (image_processing_util.py:13-24,33-54)."""
    import torch
    import torch.nn.functional as F
    n, c, h, w = image.shape
    gc = grid_change.reshape(n, 2, h * w).transpose(1, 2).reshape(n, h, w, 2)
    ident = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], dtype=image.dtype).unsqueeze(0).repeat(n, 1, 1)
    base = F.affine_grid(ident, [n, c, h, w], align_corners=False)
    return F.grid_sample(image, base + gc, mode="bilinear", padding_mode="border", align_corners=False)


def _color(alpha, cc, image):          # apply_color_change
    return cc * alpha + image * (1 - alpha)


def _rgb(alpha, cc, image):            # apply_rgb_change (image_processing_util.py:6-10)
    import torch
    return torch.cat([cc[:, 0:3] * alpha + image[:, 0:3] * (1 - alpha), image[:, 3:4]], dim=1)


def eyebrow_decomposer(P, image):
    f = _encdec(P, "body.", image, None, 16)
    bg_a = _head(P, "background_layer_alpha.0", f, "sigmoid")
    bg_c = _head(P, "background_layer_color_change.0", f, "tanh")
    bg = _color(bg_a, bg_c, image)
    eb_a = _head(P, "eyebrow_layer_alpha.0", f, "sigmoid")
    eb_c = _head(P, "eyebrow_layer_color_change.0", f, "tanh")
    eb = _color(eb_a, image, eb_c)          # note the swapped arguments (eyebrow_decomposer_00.py:55)
    return [eb, eb_a, eb_c, bg, bg_a, bg_c]


def eyebrow_morphing_combiner(P, bg, eb, pose):
    import torch
    f = _encdec(P, "body.", torch.cat([bg, eb], dim=1), pose, 16)
    gc = _head(P, "morphed_eyebrow_layer_grid_change", f, None)
    a = _head(P, "morphed_eyebrow_layer_alpha.0", f, "sigmoid")
    cc = _head(P, "morphed_eyebrow_layer_color_change.0", f, "tanh")
    warped = _warp(gc, eb)
    morphed = _color(a, cc, warped)
    ca = _head(P, "combine_alpha.0", f, "sigmoid")
    img = _rgb(ca, morphed, bg)
    img_nc = _rgb((morphed[:, 3:4] + 1.0) / 2.0, morphed, bg)
    return [img, ca, img_nc, morphed, a, cc, warped, gc]


def face_morpher(P, image, pose):
    f = _encdec(P, "", image, pose, 24)
    gc = _head(P, "iris_mouth_grid_change", f, None)
    im0 = _warp(gc, image)
    icc = _head(P, "iris_mouth_color_change.0", f, "tanh")
    ia = _head(P, "iris_mouth_alpha.0", f, "sigmoid")
    im1 = _color(ia, icc, im0)
    ecc = _head(P, "eye_color_change.0", f, "tanh")
    ea = _head(P, "eye_alpha.0", f, "sigmoid")
    out = _color(ea, ecc, im1)
    return [out, ea, ecc, im1, ia, icc, im0, gc]


def _gn(x, P, p):
    import torch.nn.functional as F
    return F.group_norm(x, min(32, x.shape[1]), P[p + ".weight"], P[p + ".bias"], eps=1e-5)


def _resblock(P, p, x, t_emb, c_emb, mode):
    import torch
    import torch.nn.functional as F
    h = F.silu(_gn(x, P, p + ".norm0"))
    xr = x
    if mode == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        xr = F.interpolate(x, scale_factor=2, mode="nearest")
    elif mode == "down":
        h = F.avg_pool2d(h, 2, 2)
        xr = F.avg_pool2d(x, 2, 2)
    h = F.conv2d(h, P[p + ".conv0.weight"], P[p + ".conv0.bias"], padding=1)
    h = _gn(h, P, p + ".norm1")
    for emb, name in ((t_emb, ".cond0_layers.1"), (c_emb, ".cond1_layers.1")):
        ss = F.linear(F.silu(emb), P[p + name + ".weight"], P[p + name + ".bias"])
        scale, shift = torch.chunk(ss.reshape(ss.shape[0], ss.shape[1], 1, 1), 2, dim=1)
        h = h * (1.0 + scale) + shift
    h = F.conv2d(F.silu(h), P[p + ".conv1.weight"], P[p + ".conv1.bias"], padding=1)
    if (p + ".skip.weight") in P:
        xr = F.conv2d(xr, P[p + ".skip.weight"], P[p + ".skip.bias"])
    return xr + h


def _attention(P, p, x, heads):
    import torch
    import torch.nn.functional as F
    B, C, H, W = x.shape
    qkv = F.conv2d(_gn(x, P, p + ".norm"), P[p + ".qkv.weight"], P[p + ".qkv.bias"]).reshape(B, 3 * C, H * W)
    ch = C // heads
    q, k, v = qkv.chunk(3, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", (q * scale).reshape(B * heads, ch, H * W), (k * scale).reshape(B * heads, ch, H * W))
    w = torch.softmax(w, dim=-1)
    h = torch.einsum("bts,bcs->bct", w, v.reshape(B * heads, ch, H * W)).reshape(B, C, H, W)
    return x + F.conv2d(h, P[p + ".conv.weight"], P[p + ".conv.bias"])


def unet(P, pre, a, x, cond, first_conv_addition=None):
    import torch
    import torch.nn.functional as F
    n = x.shape[0]
    half = a["model"] // 2
    # t = zeros (morpher_00.py:51): cos(0)=1, sin(0)=0  (unet.py:365-376)
    t_in = torch.cat([torch.ones(n, half, dtype=x.dtype), torch.zeros(n, half, dtype=x.dtype)], dim=1)
    t_emb = F.linear(F.silu(F.linear(t_in, P[pre + "time_embed.1.weight"], P[pre + "time_embed.1.bias"])),
                     P[pre + "time_embed.3.weight"], P[pre + "time_embed.3.bias"])
    c_emb = F.linear(F.silu(F.linear(cond, P[pre + "cond_embed.0.weight"], P[pre + "cond_embed.0.bias"])),
                     P[pre + "cond_embed.2.weight"], P[pre + "cond_embed.2.bias"])
    h = F.conv2d(x, P[pre + "first_conv.weight"], P[pre + "first_conv.bias"], padding=1)
    if first_conv_addition is not None:
        h = h + first_conv_addition
    down, middle, up = unet_blocks(a)
    hs = [h]
    for idx, (p, kind, cin, cout, mode) in enumerate(down):
        h = _resblock(P, pre + p, h, t_emb, c_emb, mode) if kind == "res" else _attention(P, pre + p, h, a["heads"])
        # DownsamplingBlock.forward pushes the level output AFTER its attention block (unet.py:314-323)
        followed_by_attn = kind == "res" and mode == "same" and idx + 1 < len(down) and down[idx + 1][1] == "attn"
        if not followed_by_attn:
            hs.append(h)
    for (p, kind, cin, cout, mode) in middle:
        h = _resblock(P, pre + p, h, t_emb, c_emb, mode) if kind == "res" else _attention(P, pre + p, h, a["heads"])
    for (p, kind, cin, cout, mode) in up:
        if kind == "res" and mode == "same":
            h = _resblock(P, pre + p, torch.cat([h, hs.pop()], dim=1), t_emb, c_emb, mode)
        elif kind == "res":
            h = _resblock(P, pre + p, h, t_emb, c_emb, mode)
        else:
            h = _attention(P, pre + p, h, a["heads"])
    assert not hs
    h = F.silu(_gn(h, P, pre + "last.0"))
    return F.conv2d(h, P[pre + "last.2.weight"], P[pre + "last.2.bias"], padding=1)


def body_morpher(P, image, pose):
    import torch
    o = unet(P, "body.", UNET_BODY, image, pose)
    direct, gc, alpha = o[:, 0:4], o[:, 4:6], torch.sigmoid(o[:, 6:7])
    warped = _warp(gc, image)
    return [_color(alpha, direct, warped), alpha, warped, gc, direct]


def upscaler(P, rest, coarse_img, coarse_gc, pose):
    import torch
    import torch.nn.functional as F
    warped_rest = _warp(coarse_gc, rest)
    feat = torch.cat([coarse_img, warped_rest, coarse_gc], dim=1)
    add = F.conv2d(feat, P["coarse_image_conv.weight"], P["coarse_image_conv.bias"], padding=1)
    o = unet(P, "body.", UNET_UP, rest, pose, add)
    direct, gc, alpha = o[:, 0:4], o[:, 4:6], torch.sigmoid(o[:, 6:7])
    warped = _warp(gc, rest)
    return [_color(alpha, direct, warped), alpha, warped, gc, direct]


def full_forward_torch(weights: Dict[str, Dict[str, np.ndarray]], image, pose, dtype: str = "float32",
                       eyebrow_morphed_image_index: int = 2):
    """mode_07 FiveStepPoserComputationProtocol "all_outputs" for a batch (33 tensors, mode_07.py:119-132):
    upscaler 5, face_morphed_full 1, body_morpher 5, face_morpher 8, combiner 8, decomposer 6."""
    import torch
    import torch.nn.functional as F
    dt = {"float32": torch.float32, "float64": torch.float64}[dtype]
    image = torch.as_tensor(np.asarray(image) if not torch.is_tensor(image) else image).to(dt)
    pose = torch.as_tensor(np.asarray(pose) if not torch.is_tensor(pose) else pose).to(dt)
    if image.dim() == 3:
        image = image.unsqueeze(0)
    if pose.dim() == 1:
        pose = pose.unsqueeze(0)
    if image.shape[0] != pose.shape[0]:
        image = image.expand(pose.shape[0], -1, -1, -1)
    with torch.no_grad():
        P = {n: _T(weights[n], dt) for n in NETS}
        dec = eyebrow_decomposer(P["eyebrow_decomposer"], image[:, :, 64:192, 192:320])
        comb = eyebrow_morphing_combiner(P["eyebrow_morphing_combiner"], dec[3], dec[0], pose[:, :NUM_EYEBROW])
        face_in = image[:, :, 32:224, 160:352].clone()
        face_in[:, :, 32:160, 32:160] = comb[eyebrow_morphed_image_index]
        face = face_morpher(P["face_morpher"], face_in, pose[:, NUM_EYEBROW:NUM_EYEBROW + NUM_FACE])
        full = image.clone()
        full[:, :, 32:224, 160:352] = face[0]
        half = F.interpolate(full, size=(256, 256), mode="bilinear", align_corners=False)
        rot = pose[:, NUM_EYEBROW + NUM_FACE:]
        body = body_morpher(P["body_morpher"], half, rot)
        coarse_img = F.interpolate(body[0], size=(512, 512), mode="bilinear")
        coarse_gc = F.interpolate(body[3], size=(512, 512), mode="bilinear")
        up = upscaler(P["upscaler"], full, coarse_img, coarse_gc, rot)
    return up + [full] + body + face + comb + dec


OUTPUT_NAMES = (["up_merged", "up_alpha", "up_warped", "up_grid", "up_direct", "face_morphed_full",
                 "body_merged", "body_alpha", "body_warped", "body_grid", "body_direct"]
                + [f"face_{i}" for i in range(8)] + [f"comb_{i}" for i in range(8)] + [f"dec_{i}" for i in range(6)])
