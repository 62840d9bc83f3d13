"""CPU oracle for the THA4 distilled-student poser path (mode_14).

*** TEST INFRASTRUCTURE ONLY ***
This module is the checker, never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``talking-head-anime-4-demo_amd/`` imports it, and the
product path raises if its HIP library is missing instead of falling back here.

Parity status: PINNED BY LIVE REFERENCE.  The reference ships no tests or golden
vectors; this restatement is pinned against outputs of the
unmodified reference modules run on CPU (``tests/golden/make_golden.py`` →
``tests/golden/student_lambda_00.npz``; ``tests/test_oracle_golden.py``).

Two independent restatements of the same algorithm live here:

* ``student_forward_torch``  – torch *functional* ops (conv2d / interpolate /
  affine_grid / grid_sample), i.e. the very ATen kernels the reference
  dispatches to, in fp32 or fp64.  This is the CPU baseline that ``bench.py``
  times (kind="port") and the bit-level twin of the reference on one machine.
* ``student_forward_numpy``  – explicit numpy arithmetic (matmul, hand-written
  bilinear resize and border-clamped bilinear warp) in fp64.  It states the
  closed forms the HIP kernels implement (analytic position grid, 2x bilinear
  weights, grid_sample index math) and so validates them independently of ATen.

It also exposes the *restructured* intermediates the HIP pipeline exchanges
between its kernels (``student_intermediates``): pose-folded first-layer biases
and the low-resolution pre-activation maps ``z1``/``z2`` obtained by commuting
the (linear) bilinear upsample with the (linear) feature part of the next
level's first layer.

Reference call sites restated (all paths relative to /root/reference/src/tha4):
  nn/siren/vanilla/siren.py:38-39        SineLinearLayer.forward  sin(30*(Wx+b))
  nn/siren/vanilla/siren.py:84-91        Siren.forward
  nn/siren/face_morpher/siren_face_morpher_00.py:34-51   position grid ++ pose → Siren
  nn/siren/morpher/siren_morpher_03.py:92-139            3-level body morpher, head, warp, blend
  nn/image_processing_util.py:33-54      GridChangeApplier.apply (affine_grid + grid_sample)
  poser/modes/mode_14.py:58-90           two-step DAG: face → paste at rows 80:208, cols 192:320 → body
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

OMEGA_0 = 30.0  # siren.py:17 (constructor constant, not stored in the state_dict)

FACE_SIZE = 128
IMAGE_SIZE = 512
LEVEL_SIZES = (128, 256, 512)          # mode_14.py:118-130
FACE_TOP, FACE_LEFT = 80, 192          # mode_14.py:61-63 (center_y=144, center_x=256, +-64)
NUM_FACE_POSE = 39                     # mode_14.py:66
NUM_POSE = 45


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------

def state_dicts_to_numpy(face_sd, body_sd) -> Dict[str, np.ndarray]:
    """Flatten the two reference ``state_dict``s into a dict of
    fp32 numpy arrays with 2-D weights ``[out, in]`` (the 1x1 conv kernels squeezed)."""
    out: Dict[str, np.ndarray] = {}
    for prefix, sd in (("face.", face_sd), ("body.", body_sd)):
        for k, v in sd.items():
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.ndim == 4:
                a = a.reshape(a.shape[0], a.shape[1])
            out[prefix + k] = a
    return out


def load_student_pt(face_pt: str, body_pt: str) -> Dict[str, np.ndarray]:
    """Read the two ``.pt`` files exactly like shion/core/load_save.py:12-14 does."""
    import torch
    with open(face_pt, "rb") as f:
        face_sd = torch.load(f, map_location="cpu")
    with open(body_pt, "rb") as f:
        body_sd = torch.load(f, map_location="cpu")
    return state_dicts_to_numpy(face_sd, body_sd)


def face_layers(w):
    return [(w[f"face.siren.sine_layers.{i}.linear.weight"], w[f"face.siren.sine_layers.{i}.linear.bias"])
            for i in range(8)], (w["face.siren.last_linear.weight"], w["face.siren.last_linear.bias"])


def body_layers(w):
    levels = [[(w[f"body.siren_layers.{l}.{j}.linear.weight"], w[f"body.siren_layers.{l}.{j}.linear.bias"])
               for j in range(3)] for l in range(3)]
    return levels, (w["body.last_linear.weight"], w["body.last_linear.bias"])


def random_student_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    """Random-init weights of the mode_14 architecture, following the init rules of
    siren.py:30-36 (uniform ranges) for the sine layers and a small normal for the
    two last_linear layers (the reference uses He init there, siren.py:52).  Used by
    bench.py (``data: synthetic``) and by size-independent property tests."""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}

    def sine(name, cin, cout, first):
        if first:
            lim = 1.0 / cin
        else:
            lim = np.sqrt(6.0 / cin) / OMEGA_0
        w[name + ".weight"] = rng.uniform(-lim, lim, size=(cout, cin)).astype(np.float32)
        b = 1.0 / np.sqrt(cin)  # torch Conv2d default bias init range
        w[name + ".bias"] = rng.uniform(-b, b, size=(cout,)).astype(np.float32)

    sine("face.siren.sine_layers.0.linear", 41, 128, True)
    for i in range(1, 8):
        sine(f"face.siren.sine_layers.{i}.linear", 128, 128, False)
    w["face.siren.last_linear.weight"] = (rng.standard_normal((4, 128)) * np.sqrt(2.0 / 128) * 0.3).astype(np.float32)
    w["face.siren.last_linear.bias"] = np.zeros(4, np.float32)
    dims = [[(47, 360), (360, 360), (360, 180)],
            [(227, 180), (180, 180), (180, 90)],
            [(137, 90), (90, 90), (90, 90)]]
    for l in range(3):
        for j in range(3):
            cin, cout = dims[l][j]
            sine(f"body.siren_layers.{l}.{j}.linear", cin, cout, l == 0 and j == 0)
    last = (rng.standard_normal((7, 90)) * np.sqrt(2.0 / 90)).astype(np.float32)
    last[0:2] *= 0.02   # grid_change rows: keep warps at the +-0.1 scale of trained students
    last[2:3] *= 0.3
    last[3:7] *= 0.3
    w["body.last_linear.weight"] = last
    w["body.last_linear.bias"] = np.zeros(7, np.float32)
    return w


# --------------------------------------------------------------------------------------
# closed forms shared by the numpy restatement and the HIP kernels
# --------------------------------------------------------------------------------------

_ATEN_POSITIONS = False   # module switch used by the *_numpy restatements (see aten_position_axis)


def aten_position_axis(size: int) -> np.ndarray:
    """The fp32 values ATen's affine_grid actually produces on THIS machine.  The reference
    builds its identity theta with the default dtype (float32) even when the modules run in
    fp64 (siren_face_morpher_00.py:41, siren_morpher_03.py:94), and ATen's vectorised fp32
    linspace is off by one ulp (6e-8) from the exact dyadic value in 25-75% of the entries.
    SIREN amplifies that to ~1e-4 in the final image, so the oracle can be asked to use the
    very same table (``use_aten_positions(True)``) when it is compared with the reference."""
    import torch
    import torch.nn.functional as F
    ident = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]).unsqueeze(0)
    g = F.affine_grid(ident, [1, 1, size, size], align_corners=False)
    return g[0, 0, :, 0].numpy().copy()


def use_aten_positions(flag: bool) -> None:
    global _ATEN_POSITIONS
    _ATEN_POSITIONS = bool(flag)


def position_axis(size: int, dtype=np.float64) -> np.ndarray:
    """affine_grid(identity, align_corners=False) along one axis:  x_j = (2j+1)/size - 1
    (siren_face_morpher_00.py:40-45, siren_morpher_03.py:92-99).  Exact dyadic values unless
    ``use_aten_positions(True)`` selected ATen's fp32 table."""
    if _ATEN_POSITIONS:
        return aten_position_axis(size).astype(dtype)
    j = np.arange(size, dtype=np.float64)
    return ((2.0 * j + 1.0) / size - 1.0).astype(dtype)


def upsample2x_axis_taps(dst_size: int):
    """F.interpolate(mode='bilinear', align_corners=False) with scale 2 along one axis
    (siren_morpher_03.py:121; ATen area_pixel_compute_source_index):
        src = max(0, (dst+0.5)/2 - 0.5);  i0 = floor(src);  i1 = min(i0+1, n-1);  l1 = src-i0
    Returns (i0, i1, l0, l1)."""
    n = dst_size // 2
    d = np.arange(dst_size, dtype=np.float64)
    src = np.maximum(0.0, (d + 0.5) * 0.5 - 0.5)
    i0 = np.floor(src).astype(np.int64)
    i1 = np.minimum(i0 + 1, n - 1)
    l1 = src - i0
    return i0, i1, 1.0 - l1, l1


def upsample2x_numpy(x: np.ndarray) -> np.ndarray:
    """x: [C, h, w] -> [C, 2h, 2w] bilinear, align_corners=False."""
    _, h, w = x.shape
    y0, y1, ly0, ly1 = upsample2x_axis_taps(2 * h)
    x0, x1, lx0, lx1 = upsample2x_axis_taps(2 * w)
    rows = x[:, y0, :] * ly0[None, :, None] + x[:, y1, :] * ly1[None, :, None]
    return rows[:, :, x0] * lx0[None, None, :] + rows[:, :, x1] * lx1[None, None, :]


def grid_sample_border_numpy(image: np.ndarray, gx: np.ndarray, gy: np.ndarray) -> np.ndarray:
    """grid_sample(mode='bilinear', padding_mode='border', align_corners=False)
    (image_processing_util.py:53; ATen grid_sampler_2d):
        ix = ((gx+1)*W - 1)/2, clamped to [0, W-1]; 4-tap bilinear, taps outside are skipped
        (only possible with weight 0 after the clamp).
    image: [C,H,W]; gx, gy: [H,W] normalised coords → [C,H,W]."""
    C, H, W = image.shape
    ix = np.clip(((gx + 1.0) * W - 1.0) / 2.0, 0.0, W - 1.0)
    iy = np.clip(((gy + 1.0) * H - 1.0) / 2.0, 0.0, H - 1.0)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    fx = ix - x0
    fy = iy - y0
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    x1 = np.minimum(x0 + 1, W - 1)   # weight is 0 whenever the clamp bites
    y1 = np.minimum(y0 + 1, H - 1)
    nw = (1 - fx) * (1 - fy)
    ne = fx * (1 - fy)
    sw = (1 - fx) * fy
    se = fx * fy
    return (image[:, y0, x0] * nw + image[:, y0, x1] * ne + image[:, y1, x0] * sw + image[:, y1, x1] * se)


def _sine_layer(x, W, b):
    """x: [Cin, P] → sin(30*(W x + b))   (siren.py:38-39)."""
    return np.sin(OMEGA_0 * (W @ x + b[:, None]))


# --------------------------------------------------------------------------------------
# numpy restatement (fp64 by default)
# --------------------------------------------------------------------------------------

def face_forward_numpy(w, pose39: np.ndarray, dtype=np.float64) -> np.ndarray:
    """SirenFaceMorpher00.forward for one frame → [4,128,128]."""
    sines, (Wl, bl) = face_layers(w)
    S = FACE_SIZE
    ax = position_axis(S, dtype)
    xs = np.broadcast_to(ax[None, :], (S, S)).reshape(-1)   # channel 0 = x (varies along width)
    ys = np.broadcast_to(ax[:, None], (S, S)).reshape(-1)   # channel 1 = y
    pose_img = np.broadcast_to(pose39.astype(dtype)[:, None], (NUM_FACE_POSE, S * S))
    x = np.concatenate([xs[None], ys[None], pose_img], axis=0)
    for (W, b) in sines:
        x = _sine_layer(x, W.astype(dtype), b.astype(dtype))
    out = Wl.astype(dtype) @ x + bl.astype(dtype)[:, None]
    return out.reshape(4, S, S)


def body_forward_numpy(w, image: np.ndarray, pose45: np.ndarray, dtype=np.float64) -> List[np.ndarray]:
    """SirenMorpher03.forward for one frame.  image: [4,512,512] (face already pasted)."""
    levels, (Wl, bl) = body_layers(w)
    x = None
    for li, S in enumerate(LEVEL_SIZES):
        ax = position_axis(S, dtype)
        xs = np.broadcast_to(ax[None, :], (S, S)).reshape(-1)
        ys = np.broadcast_to(ax[:, None], (S, S)).reshape(-1)
        pose_img = np.broadcast_to(pose45.astype(dtype)[:, None], (NUM_POSE, S * S))
        pp = np.concatenate([xs[None], ys[None], pose_img], axis=0)
        if li == 0:
            x = pp
        else:
            c = x.shape[0]
            up = upsample2x_numpy(x.reshape(c, S // 2, S // 2)).reshape(c, S * S)
            x = np.concatenate([up, pp], axis=0)     # features first, then x, y, pose (siren_morpher_03.py:122)
        for (W, b) in levels[li]:
            x = _sine_layer(x, W.astype(dtype), b.astype(dtype))
    S = IMAGE_SIZE
    so = (Wl.astype(dtype) @ x + bl.astype(dtype)[:, None]).reshape(7, S, S)
    grid_change, alpha, color_change = so[0:2], so[2:3], so[3:7]
    j = np.arange(S, dtype=np.float64)
    ax = ((2.0 * j + 1.0) / S - 1.0).astype(dtype)   # warp base grid is built in the working dtype (image_processing_util.py:41-50)
    if dtype == np.float32 and _ATEN_POSITIONS:
        ax = aten_position_axis(S)
    gx = ax[None, :] + grid_change[0]      # ch0 → x offset, ch1 → y offset (image_processing_util.py:36,51)
    gy = ax[:, None] + grid_change[1]
    warped = grid_sample_border_numpy(image.astype(dtype), gx, gy)
    blended = (1 - alpha) * warped + alpha * color_change     # siren_morpher_03.py:131
    return [blended, alpha, color_change, warped, grid_change]


def paste_face(image: np.ndarray, face: np.ndarray) -> np.ndarray:
    """mode_14.py:72-78: clone, then overwrite rows 80:208, cols 192:320."""
    out = np.array(image, copy=True)
    out[:, FACE_TOP:FACE_TOP + FACE_SIZE, FACE_LEFT:FACE_LEFT + FACE_SIZE] = face
    return out


def student_forward_numpy(w, image: np.ndarray, pose: np.ndarray, dtype=np.float64) -> List[np.ndarray]:
    """Full mode_14 pipeline for ONE frame (image [4,512,512], pose [45]).
    Returns the 6 outputs in the reference order (mode_14.py:85-88), without batch dim."""
    pose = np.asarray(pose, dtype=dtype)
    face = face_forward_numpy(w, pose[:NUM_FACE_POSE], dtype)
    body_in = paste_face(np.asarray(image, dtype=dtype), face)
    return body_forward_numpy(w, body_in, pose, dtype) + [face]


# --------------------------------------------------------------------------------------
# torch-functional restatement (same ATen kernels as the reference; fp32 or fp64)
# --------------------------------------------------------------------------------------

def student_forward_torch(w, image, pose, dtype="float32"):
    """Batched mode_14 pipeline with torch functional ops.
    image: [B,4,512,512] or [4,512,512]; pose: [B,45] or [45] (numpy or torch).
    Returns the 6 outputs as torch tensors [B,...] in the reference order."""
    import torch
    import torch.nn.functional as F
    tdt = {"float32": torch.float32, "float64": torch.float64}[dtype]
    image = torch.as_tensor(np.asarray(image) if not torch.is_tensor(image) else image).to(tdt)
    pose = torch.as_tensor(np.asarray(pose) if not torch.is_tensor(pose) else pose).to(tdt)
    if image.dim() == 3:
        image = image.unsqueeze(0)
    if pose.dim() == 1:
        pose = pose.unsqueeze(0)
    n = pose.shape[0]

    def t(a):
        return torch.from_numpy(a).to(tdt)

    def conv(x, W, b):
        return F.conv2d(x, t(W).view(W.shape[0], W.shape[1], 1, 1), t(b))

    def pos_grid(S):
        # default-dtype (fp32) identity, exactly as the reference builds it; promoted by the cat below
        ident = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]).unsqueeze(0)
        p = F.affine_grid(ident, [1, 1, S, S], align_corners=False).view(1, S * S, 2)
        return p.transpose(1, 2).reshape(1, 2, S, S).repeat(n, 1, 1, 1).to(tdt)

    def pose_img(p, S):
        return p.view(n, p.shape[1], 1, 1).repeat(1, 1, S, S)

    with torch.no_grad():
        # face morpher (siren_face_morpher_00.py:34-51)
        sines, (Wl, bl) = face_layers(w)
        x = torch.cat([pos_grid(FACE_SIZE), pose_img(pose[:, :NUM_FACE_POSE], FACE_SIZE)], dim=1)
        for (W, b) in sines:
            x = torch.sin(OMEGA_0 * conv(x, W, b))
        face = conv(x, Wl, bl)
        # paste (mode_14.py:72-78)
        body_in = image.clone()
        if body_in.shape[0] != n:
            body_in = body_in.expand(n, -1, -1, -1).clone()
        body_in[:, :, FACE_TOP:FACE_TOP + FACE_SIZE, FACE_LEFT:FACE_LEFT + FACE_SIZE] = face
        # body morpher (siren_morpher_03.py:107-139)
        levels, (Wl, bl) = body_layers(w)
        x = None
        for li, S in enumerate(LEVEL_SIZES):
            pp = torch.cat([pos_grid(S), pose_img(pose, S)], dim=1)
            if li == 0:
                x = pp
            else:
                x = F.interpolate(x, size=(S, S), mode="bilinear")
                x = torch.cat([x, pp], dim=1)
            for (W, b) in levels[li]:
                x = torch.sin(OMEGA_0 * conv(x, W, b))
        so = conv(x, Wl, bl)
        grid_change, alpha, color_change = so[:, 0:2], so[:, 2:3], so[:, 3:]
        S = IMAGE_SIZE
        gc = grid_change.reshape(n, 2, S * S).transpose(1, 2).reshape(n, S, S, 2)
        ident = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], dtype=tdt).unsqueeze(0).repeat(n, 1, 1)
        base = F.affine_grid(ident, [n, 4, S, S], align_corners=False)
        warped = F.grid_sample(body_in, base + gc, mode="bilinear", padding_mode="border", align_corners=False)
        blended = (1 - alpha) * warped + alpha * color_change
    return [blended, alpha, color_change, warped, grid_change, face]


# --------------------------------------------------------------------------------------
# restructured intermediates exchanged between the HIP kernels (fp64)
# --------------------------------------------------------------------------------------

def student_intermediates(w, pose: np.ndarray) -> Dict[str, np.ndarray]:
    """fp64 values of everything the HIP kernels hand to each other for ONE frame:

      pb_face[128], pb0[360], pb1[180], pb2[90] : first-layer bias with the pose columns folded in
      h0 [180,128,128]  level-0 output (after its 3 sine layers)
      z1 [180,128,128]  W_{1,0}[:, :180] @ h0    (level-1 first layer, feature part, at LOW res)
      h1 [90,256,256], z2 [90,256,256]           same for level 1 → 2
      siren_out [7,512,512]                      last_linear output
    The identity  W @ upsample(h) == upsample(W @ h)  (both linear) is what lets the
    HIP level-l kernel emit z_{l+1} at its own resolution; ``tests`` check it against
    ``body_forward_numpy`` which follows the reference's order of operations."""
    dtype = np.float64
    pose = np.asarray(pose, dtype)
    out: Dict[str, np.ndarray] = {}
    sines, _ = face_layers(w)
    W0, b0 = sines[0]
    out["pb_face"] = W0[:, 2:].astype(dtype) @ pose[:NUM_FACE_POSE] + b0
    levels, (Wl, bl) = body_layers(w)
    feat = [0, 180, 90]
    for l in range(3):
        W, b = levels[l][0]
        out[f"pb{l}"] = W[:, feat[l] + 2:].astype(dtype) @ pose + b
    h = None
    for l, S in enumerate(LEVEL_SIZES):
        ax = position_axis(S)
        xs = np.broadcast_to(ax[None, :], (S, S)).reshape(-1)
        ys = np.broadcast_to(ax[:, None], (S, S)).reshape(-1)
        W, b = levels[l][0]
        W = W.astype(dtype)
        pre = W[:, feat[l]:feat[l] + 1] * xs[None] + W[:, feat[l] + 1:feat[l] + 2] * ys[None] + out[f"pb{l}"][:, None]
        if l > 0:
            z = out[f"z{l}"]
            pre = pre + upsample2x_numpy(z).reshape(z.shape[0], S * S)
        x = np.sin(OMEGA_0 * pre)
        for (W, b) in levels[l][1:]:
            x = _sine_layer(x, W.astype(dtype), b.astype(dtype))
        out[f"h{l}"] = x.reshape(-1, S, S)
        if l < 2:
            Wn = levels[l + 1][0][0].astype(dtype)[:, :feat[l + 1]]
            out[f"z{l + 1}"] = (Wn @ x).reshape(-1, S, S)
    out["siren_out"] = (Wl.astype(dtype) @ out["h2"].reshape(90, -1) + bl[:, None]).reshape(7, IMAGE_SIZE, IMAGE_SIZE)
    return out


# --------------------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------------------

POSE_LO = np.array([0.0] * 37 + [-1.0] * 7 + [0.0], dtype=np.float32)   # pose_parameters.py:4-36
POSE_HI = np.ones(45, dtype=np.float32)


def random_poses(n: int, seed: int = 1234) -> np.ndarray:
    """lo + (hi-lo)*U[0,1)^45 with torch.Generator().manual_seed(seed)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, NUM_POSE, generator=g).numpy()
    return (POSE_LO + (POSE_HI - POSE_LO) * u).astype(np.float32)


def synthetic_image(seed: int = 99, size: int = IMAGE_SIZE) -> np.ndarray:
    """U[-1,1) fp32 image with alpha = U[0,1) mapped to [-1,1] and RGB premultiplied
    (SURVEY.md §8d config 5 recipe), smooth enough to have meaningful bilinear taps."""
    rng = np.random.default_rng(seed)
    lo = rng.uniform(0.0, 1.0, size=(4, size // 8, size // 8))
    img = upsample2x_numpy(upsample2x_numpy(upsample2x_numpy(lo)))     # [4,size,size] in [0,1]
    img[0:3] *= img[3:4]
    return (img * 2.0 - 1.0).astype(np.float32)
