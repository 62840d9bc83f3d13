"""CPU tests of the full-model kernels (csrc/full_kernels.h compiled against the SIMT emulator)
versus torch fp64 references of the same ops: implicit-GEMM conv (3x3 / 1x1 / 4x4 s2 / convT 4x4 s2,
fused input norm + activation + nearest-up / avg-pool, concatenated and broadcast-vector sources,
bias / residual / per-channel activation epilogue, per-tile statistics), norm finalize (InstanceNorm,
GroupNorm over a concatenation, FiLM folding), gemv and the attention core."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ACT = {"none": 0, "relu": 1, "silu": 2, "sigmoid": 3, "tanh": 4}
fp = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def lib(built):
    import __graft_entry__ as g
    g.build_emulator_full()
    L = C.CDLL(os.path.join(HERE, "emu", "libtha4_emu_full.so"))
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(fp)


def torch_act(x, name):
    return {"none": lambda v: v, "relu": F.relu, "silu": F.silu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[name](x)


def run_conv(lib, kind, k, tmb, pg, in_mode, act_in, x0, x1, vec1, scale, shift, weight, bias, residual, act_out, chunk_quads,
             tw_log2=4, ksplit=1):
    n, c0, h, w = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    cout = weight.shape[1] if kind == 2 else weight.shape[0]
    vh = h * 2 if in_mode == 1 else (h // 2 if in_mode == 2 else h)
    vw = w * 2 if in_mode == 1 else (w // 2 if in_mode == 2 else w)
    oh, ow = (vh, vw) if kind == 0 else ((vh // 2, vw // 2) if kind == 1 else (vh * 2, vw * 2))
    out = np.zeros((n, cout, oh, ow), np.float32)
    stats = np.zeros((n, cout, 2), np.float32)
    ao = None if act_out is None else np.ascontiguousarray(act_out, np.int32)
    rc = lib.emu_conv(kind, k, tmb, pg, in_mode, ACT[act_in], n, c0, c1, int(vec1), h, w, P(x0), P(x1), P(scale), P(shift),
                      P(weight), cout, P(bias), P(residual), None if ao is None else ao.ctypes.data_as(C.POINTER(C.c_int)),
                      chunk_quads, P(out), P(stats), tw_log2, ksplit)
    assert rc == 0, rc
    return out, stats


def ref_conv(kind, in_mode, act_in, x0, x1, vec1, scale, shift, weight, bias, residual, act_out):
    t = lambda a: None if a is None else torch.from_numpy(a).double()
    x = t(x0)
    if x1 is not None:
        xx1 = t(x1)
    if x1 is not None and vec1:
        # pose vector: concatenated raw AFTER the producer's norm + activation (poser_encoder_decoder_00.py:108-113)
        c0 = x.shape[1]
        if scale is not None:
            x = x * t(scale)[:, :c0, None, None] + t(shift)[:, :c0, None, None]
        x = torch.cat([torch_act(x, act_in), xx1[:, :, None, None].expand(-1, -1, x.shape[2], x.shape[3])], dim=1)
    else:
        if x1 is not None:
            x = torch.cat([x, xx1], dim=1)
        if scale is not None:
            x = x * t(scale)[:, :, None, None] + t(shift)[:, :, None, None]
        x = torch_act(x, act_in)
    if in_mode == 1:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif in_mode == 2:
        x = F.avg_pool2d(x, 2, 2)
    W = t(weight)
    if kind == 0:
        y = F.conv2d(x, W, t(bias), padding=W.shape[2] // 2)
    elif kind == 1:
        y = F.conv2d(x, W, t(bias), stride=2, padding=1)
    else:
        y = F.conv_transpose2d(x, W, t(bias), stride=2, padding=1)
    if residual is not None:
        y = y + t(residual)
    if act_out is not None:
        names = {v: k for k, v in ACT.items()}
        y = torch.stack([torch_act(y[:, c], names[int(act_out[c])]) for c in range(y.shape[1])], dim=1)
    return y.numpy()


CASES = [
    # kind k tmb pg in_mode act_in  c0  c1  vec  h   w  cout bias res  actout  scale chunk
    (0, 3, 2, 1, 0, "relu", 20, 0, False, 8, 16, 32, True, True, False, True, 2),     # conv3 + IN/ReLU input + residual
    (0, 3, 4, 2, 0, "silu", 32, 16, False, 16, 16, 64, True, False, False, True, 1),  # concat of two tensors (U-Net up path)
    (0, 3, 1, 1, 0, "relu", 16, 12, True, 8, 8, 16, False, False, False, True, 3),    # tensor ++ broadcast pose vector
    (0, 1, 2, 2, 0, "none", 32, 0, False, 8, 16, 24, True, True, False, True, 2),     # 1x1 skip / attention projection
    (1, 4, 2, 1, 0, "relu", 16, 0, False, 16, 16, 32, False, False, False, True, 1),  # conv 4x4 stride 2
    (2, 4, 2, 1, 0, "relu", 32, 0, False, 8, 8, 32, False, False, False, True, 2),    # convT 4x4 stride 2 (4 parity classes)
    (0, 3, 2, 1, 1, "silu", 16, 0, False, 4, 8, 32, True, False, False, True, 1),     # nearest-up x2 on load
    (0, 3, 2, 1, 2, "silu", 16, 0, False, 16, 16, 32, True, False, False, True, 1),   # avg-pool 2x2 on load
    (0, 3, 1, 1, 0, "relu", 64, 0, False, 8, 8, 10, True, False, True, False, 4),     # head block: mixed sigmoid/tanh/none rows
    (0, 3, 4, 0, 0, "relu", 48, 27, True, 12, 12, 64, False, False, False, True, 1),   # split-K kernel (pg=0): bottleneck entry with pose vector
    (0, 3, 4, 0, 0, "silu", 32, 32, False, 8, 8, 64, True, True, False, True, 1),     # split-K: concat + residual (U-Net middle)
    (0, 1, 4, 0, 0, "none", 64, 0, False, 8, 8, 192, True, False, False, True, 1),    # split-K: 1x1 qkv projection
    (2, 4, 4, 0, 0, "relu", 32, 0, False, 8, 8, 64, False, False, False, True, 1),    # split-K: convT 4x4 s2
    (0, 3, 4, 0, 2, "silu", 32, 0, False, 16, 16, 64, True, False, False, True, 1),   # split-K: avg-pool on load
    (0, 3, 2, 1, 0, "none", 4, 0, False, 24, 24, 32, True, False, False, False, 1),   # 4-channel image input, 24x24 (rows not a multiple of 16 px)
    # conv_tile_kernel (pg = 10 + PG): LDS-staged window, fp16 hi/lo split MFMA
    # last field for these: (log2 tile width, K split)
    (0, 3, 4, 12, 0, "relu", 40, 0, False, 16, 32, 64, True, True, False, True, (4, 1)),    # 3x3, odd quad count (phantom quad), residual
    (0, 3, 2, 14, 0, "silu", 32, 16, False, 32, 32, 32, True, False, False, True, (5, 1)),  # PG=4 (16x32 tiles), concat of two tensors
    (0, 3, 2, 11, 0, "relu", 16, 12, True, 16, 16, 32, False, False, False, True, (4, 1)),  # PG=1, tensor ++ broadcast pose vector
    (1, 4, 2, 11, 0, "relu", 16, 0, False, 32, 32, 32, False, False, False, True, (4, 1)),  # 4x4 stride 2 (8x16 tiles, 18x34 window)
    (2, 4, 2, 12, 0, "relu", 32, 0, False, 16, 16, 32, False, False, False, True, (4, 1)),  # convT 4x4 s2 (4 parity classes)
    (0, 3, 2, 12, 1, "silu", 16, 0, False, 8, 16, 32, True, False, False, True, (4, 1)),    # nearest-up x2 on load
    (0, 3, 2, 12, 2, "silu", 16, 0, False, 32, 32, 32, True, False, False, True, (4, 1)),   # avg-pool 2x2 on load
    (0, 3, 1, 12, 0, "relu", 64, 0, False, 16, 16, 10, True, False, True, False, (4, 1)),   # head block: mixed sigmoid/tanh/none rows
    (0, 3, 2, 12, 0, "none", 4, 0, False, 48, 48, 32, True, False, False, False, (4, 1)),   # 4-channel image input, 48x48 (3x3 tiles)
    (0, 3, 2, 12, 0, "relu", 96, 27, True, 16, 16, 64, True, True, False, True, (4, 4)),    # split-K x4 (8 K groups), pose vector, residual
    (0, 3, 4, 12, 0, "silu", 80, 0, False, 16, 16, 64, True, False, False, True, (4, 3)),   # split-K x3 over 3 K groups (5 quads)
    (0, 3, 2, 11, 0, "relu", 32, 0, False, 24, 24, 32, True, True, False, True, (3, 1)),    # ragged: 24x24 map in 16x8 tiles (rows 24..31 masked)
    (0, 3, 8, 12, 0, "silu", 48, 16, False, 16, 16, 128, True, True, False, True, (4, 1)),  # <8,2> tile (round 6, tuning option THA4_TILE_TMB8): 128 output channels per workgroup, concat + residual
    (0, 3, 2, 12, 0, "relu", 64, 0, False, 24, 24, 32, True, False, False, True, (5, 2)),   # ragged 8x32 tiles (cols 24..31 masked) + split-K
    (2, 4, 2, 11, 0, "relu", 64, 0, False, 12, 12, 32, False, False, False, True, (4, 2)),  # convT on a 12x12 map: ragged + split-K, 4 classes
    (1, 4, 2, 11, 0, "relu", 32, 0, False, 48, 48, 32, False, False, False, True, (3, 1)),  # 4x4 s2 -> 24x24 in 16x8 tiles (34x18 window)
    # conv_tile_kernel with SIXTEEN waves (pg = 40 + PG): the output blocks of the tile split over two halves of the workgroup
    (0, 3, 4, 42, 0, "relu", 40, 0, False, 16, 32, 64, True, True, False, True, (4, 1)),    # <4,2>: 3x3, phantom quad, residual
    (0, 3, 2, 44, 0, "silu", 32, 16, False, 32, 32, 32, True, False, False, True, (5, 1)),  # <2,4>: 16x32 tiles, concat of two tensors
    (0, 3, 2, 41, 0, "relu", 16, 12, True, 16, 16, 32, False, False, False, True, (4, 1)),  # <2,1>: tensor ++ broadcast pose vector
    (1, 4, 4, 41, 0, "relu", 16, 0, False, 32, 32, 64, False, False, False, True, (4, 1)),  # <4,1>: 4x4 stride 2 (18x34 window)
    (2, 4, 2, 42, 0, "relu", 32, 0, False, 16, 16, 32, False, False, False, True, (4, 1)),  # <2,2>: convT 4x4 s2 (4 parity classes)
    (0, 3, 2, 42, 1, "silu", 16, 0, False, 8, 16, 32, True, False, False, True, (4, 1)),    # nearest-up x2 on load
    (0, 3, 4, 41, 2, "silu", 16, 0, False, 32, 32, 64, True, False, False, True, (4, 1)),   # avg-pool 2x2 on load
    (0, 3, 2, 42, 0, "relu", 96, 27, True, 16, 16, 64, True, True, False, True, (4, 4)),    # split-K x4: phase 1 with 16 waves, phase 2 per block
    (0, 3, 2, 41, 0, "relu", 32, 0, False, 24, 24, 32, True, True, False, True, (3, 1)),    # ragged 24x24 map in 16x8 tiles
    # conv_tile_kernel as FOUR-wave workgroups (pg = 50 + PG): half the pixel tile, <= 80 KiB of LDS, two workgroups per CU
    (0, 3, 4, 54, 0, "silu", 40, 0, False, 32, 32, 64, True, True, False, True, (4, 1)),    # <4,4>: 16x16 tiles (18x18 window), phantom quad, residual
    (0, 3, 2, 54, 0, "silu", 32, 16, False, 32, 32, 32, True, False, False, True, (5, 1)),  # <2,4>: 8x32 tiles, concat of two tensors
    (0, 3, 4, 52, 0, "relu", 48, 0, False, 16, 32, 64, True, False, False, True, (4, 1)),   # <4,2>: 8x16 tiles
    (0, 3, 2, 51, 0, "relu", 16, 12, True, 16, 16, 32, False, False, False, True, (3, 1)),  # <2,1>: 8x8 tiles, tensor ++ broadcast pose vector
    (1, 4, 4, 51, 0, "relu", 16, 0, False, 32, 32, 64, False, False, False, True, (3, 1)),  # <4,1>: 4x4 stride 2 (18x18 window)
    (2, 4, 2, 52, 0, "relu", 32, 0, False, 16, 16, 32, False, False, False, True, (4, 1)),  # <2,2>: convT 4x4 s2 (4 parity classes)
    (0, 3, 2, 52, 1, "silu", 16, 0, False, 8, 16, 32, True, False, False, True, (4, 1)),    # nearest-up x2 on load
    (0, 3, 4, 51, 2, "silu", 16, 0, False, 32, 32, 64, True, False, False, True, (4, 1)),   # avg-pool 2x2 on load
    (0, 3, 2, 51, 0, "relu", 32, 0, False, 24, 24, 32, True, True, False, True, (3, 1)),    # ragged 24x24 map in 8x8 tiles
    (0, 3, 4, 54, 0, "none", 4, 0, False, 48, 48, 64, True, False, False, False, (4, 1)),   # 4-channel image input, 48x48 (3x3 tiles of 16x16)
    # conv_tile_kernel in the XCD-aware 1-D order (round 5, ConvArgs::xcd_remap): several output-channel tiles and gx = frames x tiles x classes a multiple of 8
    (0, 3, 2, 12, 0, "relu", 32, 0, False, 32, 32, 96, True, True, False, True, (4, 1)),    # <2,2>: 3 output tiles x (2 frames x 4 pixel tiles)
    (0, 3, 2, 52, 0, "silu", 32, 16, False, 32, 32, 64, True, False, False, True, (4, 1)),  # four-wave <2,2>: 2 output tiles x (2 x 8), concat
    (2, 4, 2, 12, 0, "relu", 32, 0, False, 16, 16, 64, False, False, False, True, (4, 1)),  # convT: 2 output tiles x (2 frames x 1 tile x 4 classes)
    # conv_small_kernel (pg = 20 + PG, tmb = 1): K split across the 8 waves of a workgroup, weights from L2 to registers
    # last field: (log2 tile width, units per K group; 0 = the planner's choice)
    (0, 3, 1, 21, 0, "relu", 96, 0, False, 16, 16, 32, True, True, False, True, (4, 0)),    # 3x3, 1x16 row tiles, 3 K groups -> tap split x4, residual
    (0, 3, 1, 22, 0, "silu", 64, 64, False, 16, 16, 48, True, False, False, True, (4, 1)),  # PG=2, concat of two tensors, 4 K groups on 4 waves
    (0, 3, 1, 24, 0, "relu", 160, 27, True, 24, 24, 32, False, False, False, True, (3, 0)),  # PG=4: 24x24 in 8x8 tiles, tensor ++ pose vector, 6 K groups
    (0, 3, 1, 21, 0, "relu", 288, 0, False, 16, 16, 16, True, False, False, True, (4, 0)),   # 9 K groups: wave 0 runs two units (prefetch path)
    (0, 1, 1, 22, 0, "none", 256, 0, False, 16, 16, 64, True, True, False, True, (4, 0)),    # 1x1 (attention projection + residual), 8 K groups
    (0, 1, 1, 24, 0, "none", 48, 0, False, 16, 32, 40, True, False, False, False, (4, 0)),   # 1x1 skip, odd quad count, 2 K groups on 2 waves
    (1, 4, 1, 21, 0, "relu", 64, 0, False, 32, 32, 32, False, False, False, True, (3, 0)),   # 4x4 stride 2 -> 16x16, 2x8 tiles (6x18 window), 16 taps
    (2, 4, 1, 22, 0, "relu", 64, 0, False, 16, 16, 32, False, False, False, True, (4, 0)),   # convT 4x4 s2: four parity classes
    (0, 3, 1, 21, 1, "silu", 64, 0, False, 8, 8, 32, True, False, False, True, (4, 0)),      # nearest-up x2 on load
    (0, 3, 1, 22, 2, "silu", 64, 0, False, 32, 32, 32, True, False, False, True, (4, 0)),    # avg-pool 2x2 on load
    (0, 3, 1, 24, 0, "relu", 32, 0, False, 20, 20, 16, True, False, True, False, (3, 0)),    # ragged 20x20 in 8x8 tiles, per-channel output activations
    (0, 1, 1, 21, 0, "none", 64, 0, False, 16, 16, 128, True, False, False, True, (4, 0)),   # 8 output blocks: XCD-aware (block -> XCD) workgroup order
    # conv_point_kernel (pg = 30 + PG): 1x1 convolutions on larger maps, operands straight from global memory
    (0, 1, 4, 32, 0, "none", 64, 32, False, 16, 32, 64, True, True, False, True, (4, 1)),    # U-Net skip: concat, residual, 3 K groups (< one chunk), 8 pixel tiles -> XCD order
    (0, 1, 2, 32, 0, "relu", 272, 0, False, 24, 24, 32, True, False, True, True, (4, 1)),    # 17 quads (phantom quad), 9 K groups = 3 chunks with a tail, ragged last tile, mixed output activations
    (0, 1, 1, 31, 0, "silu", 48, 16, False, 16, 16, 16, True, False, False, True, (4, 1)),   # SiLU input, first source with an odd quad count: a K group straddles the sources
    (0, 1, 4, 31, 0, "relu", 128, 0, False, 16, 16, 64, False, False, False, False, (4, 1)), # no scale/shift (identity table), exactly one chunk
    (0, 1, 2, 31, 0, "none", 160, 96, False, 8, 24, 96, True, True, False, True, (4, 1)),    # 3 output tiles, 8 K groups = 2 full chunks
    (0, 1, 1, 32, 0, "relu", 32, 0, False, 20, 20, 48, True, False, False, True, (4, 1)),    # one K group, ragged 400-pixel map
]


@pytest.mark.parametrize("case", CASES)
def test_conv_kernel_matches_torch(lib, case):
    kind, k, tmb, pg, in_mode, act_in, c0, c1, vec1, h, w, cout, has_bias, has_res, has_act, has_scale, chunk = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    n = 2
    x0 = rng.standard_normal((n, c0, h, w)).astype(np.float32)
    x1 = None
    if c1:
        x1 = rng.standard_normal((n, c1) if vec1 else (n, c1, h, w)).astype(np.float32)
    cin = c0 + c1
    scale = (1 + 0.3 * rng.standard_normal((n, cin))).astype(np.float32) if has_scale else None
    shift = (0.3 * rng.standard_normal((n, cin))).astype(np.float32) if has_scale else None
    kk = k if kind == 0 else 4
    wshape = (cin, cout, kk, kk) if kind == 2 else (cout, cin, kk, kk)
    weight = (rng.standard_normal(wshape) / np.sqrt(cin * kk * kk)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32) if has_bias else None
    act_out = np.array([3 if c % 3 == 0 else (4 if c % 3 == 1 else 0) for c in range(cout)], np.int32) if has_act else None
    ref_shape = ref_conv(kind, in_mode, act_in, x0, x1, vec1, scale, shift, weight, bias, None, act_out).shape
    residual = rng.standard_normal(ref_shape).astype(np.float32) if has_res else None
    ref = ref_conv(kind, in_mode, act_in, x0, x1, vec1, scale, shift, weight, bias, residual, act_out)
    twl, ksplit = chunk if isinstance(chunk, tuple) else (4, 1)
    out, stats = run_conv(lib, kind, k, tmb, pg, in_mode, act_in, x0, x1, vec1, scale, shift, weight, bias, residual, act_out,
                          1 if isinstance(chunk, tuple) else chunk, twl, ksplit)
    assert np.abs(out - ref).max() < (3e-5 if pg >= 10 else 2e-5)
    assert np.abs(stats[..., 0] - ref.sum(axis=(2, 3))).max() < 2e-3
    assert np.abs(stats[..., 1] - (ref ** 2).sum(axis=(2, 3))).max() < 5e-3


def test_xcd_aware_order_is_taken_where_the_product_takes_it(lib):
    """The three remap cases of CASES really run conv_tile_kernel on the 1-D grid (the emulator driver mirrors FullModel::conv's rule), a case with
    one output-channel tile does not."""
    lib.emu_remap_launches.restype = C.c_int
    remap = [c for c in CASES if (c[:4], c[9], c[11], c[16]) in (((0, 3, 2, 12), 32, 96, (4, 1)), ((0, 3, 2, 52), 32, 64, (4, 1)), ((2, 4, 2, 12), 16, 64, (4, 1)))]
    assert len(remap) == 3
    for case in remap:
        before = lib.emu_remap_launches()
        test_conv_kernel_matches_torch(lib, case)
        assert lib.emu_remap_launches() == before + 1, case
    before = lib.emu_remap_launches()
    test_conv_kernel_matches_torch(lib, CASES[[c[3] for c in CASES].index(14)])
    assert lib.emu_remap_launches() == before


def test_pixel_permutation_removes_the_window_read_conflicts(lib):
    """Round 5: which pixel of a 16-pixel group an MFMA column computes is chosen on the host so that the ds_read_b128 B-fragment reads of the LDS
    window are conflict-free (csrc/full_kernels.h).  For every 16-wide stride-1 geometry - all of the batch-8 plan's conv_tile launches - the
    identity costs one extra LDS cycle in each of the instruction's four lane groups and the choice none; elsewhere it is never worse."""
    ip = C.POINTER(C.c_int)
    for twl, tile_h, stride, halo in [(4, 16, 1, 2), (4, 8, 1, 2), (4, 4, 1, 2), (4, 1, 1, 2), (5, 8, 1, 2), (4, 16, 1, 1), (4, 8, 1, 0),   # 3x3 / convT / 1x1
                                      (3, 8, 1, 2), (3, 4, 1, 2), (2, 4, 1, 2), (3, 2, 1, 2), (4, 8, 2, 2), (3, 8, 2, 2), (3, 4, 1, 1), (2, 4, 1, 1)]:
        tw = 1 << twl
        win_w, win_h = tw * stride + halo, tile_h * stride + halo
        out = np.zeros(16, np.int32)
        ci, cb = C.c_int(), C.c_int()
        lib.emu_pixel_permutation(twl, win_w, win_h, stride, out.ctypes.data_as(ip), C.byref(ci), C.byref(cb))
        assert sorted(out.tolist()) == list(range(16)), (twl, tile_h, stride, out)
        assert 0 <= cb.value <= ci.value, (twl, tile_h, stride, ci.value, cb.value)
        if twl >= 4 and stride == 1:
            assert ci.value == 4 and cb.value == 0, (twl, tile_h, ci.value, cb.value)
        if cb.value == ci.value:
            assert out.tolist() == list(range(16))                     # identity is kept unless something is strictly better


def _partials(x, tiles):
    """[n][c][px] -> partial sums [n][tiles][cb*16][2] like the conv epilogue writes them."""
    n, c, px = x.shape
    cb = (c + 15) // 16
    st = np.zeros((n, tiles, cb * 16, 2), np.float32)
    chunk = px // tiles
    for t in range(tiles):
        seg = x[:, :, t * chunk:(t + 1) * chunk].astype(np.float64)
        st[:, t, :c, 0] = seg.sum(-1)
        st[:, t, :c, 1] = (seg ** 2).sum(-1)
    return st, cb


def test_norm_finalize_instance_and_group(lib):
    rng = np.random.default_rng(5)
    n, px = 2, 64
    # InstanceNorm over one 20-channel tensor
    x = (rng.standard_normal((n, 20, px)) * 2 + 1).astype(np.float32)
    st, cb = _partials(x, 4)
    gamma = (1 + 0.2 * rng.standard_normal(20)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(20)).astype(np.float32)
    sc = np.zeros((n, cb * 16), np.float32); sh = np.zeros_like(sc)
    lib.emu_norm(n, 1, P(st), 4, cb, None, 0, 0, 20, 0, C.c_float(1.0 / px), C.c_float(1e-5), P(gamma), P(beta), None, None,
                 P(sc), P(sh), None, None)
    ref = F.instance_norm(torch.from_numpy(x).double().reshape(n, 20, 8, 8), weight=torch.from_numpy(gamma).double(),
                          bias=torch.from_numpy(beta).double(), eps=1e-5).reshape(n, 20, px).numpy()
    got = x * sc[:, :20, None] + sh[:, :20, None]
    assert np.abs(got - ref).max() < 1e-5
    assert not sc[:, 20:].any() and not sh[:, 20:].any()
    # GroupNorm(32) over the concatenation of 64 + 32 channels (groups of 3 straddle the boundary) + two FiLM stages
    a = (rng.standard_normal((n, 64, px)) + 0.5).astype(np.float32)
    b = (rng.standard_normal((n, 32, px)) * 3).astype(np.float32)
    sa, cba = _partials(a, 2)
    sb, cbb = _partials(b, 4)
    C_ = 96
    gamma = (1 + 0.2 * rng.standard_normal(C_)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(C_)).astype(np.float32)
    f0 = (0.3 * rng.standard_normal((n, 2 * C_))).astype(np.float32)
    f1 = (0.3 * rng.standard_normal((n, 2 * C_))).astype(np.float32)
    sc0 = np.zeros((n, 64), np.float32); sh0 = np.zeros_like(sc0)
    sc1 = np.zeros((n, 32), np.float32); sh1 = np.zeros_like(sc1)
    lib.emu_norm(n, 2, P(sa), 2, cba, P(sb), 4, cbb, C_, 32, C.c_float(1.0 / px), C.c_float(1e-5), P(gamma), P(beta), P(f0), P(f1),
                 P(sc0), P(sh0), P(sc1), P(sh1))
    cat = torch.from_numpy(np.concatenate([a, b], 1)).double().reshape(n, C_, 8, 8)
    h = F.group_norm(cat, 32, torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), eps=1e-5)
    for f in (f0, f1):
        ft = torch.from_numpy(f).double()
        h = h * (1 + ft[:, :C_, None, None]) + ft[:, C_:, None, None]
    ref = h.reshape(n, C_, px).numpy()
    got = np.concatenate([a * sc0[:, :, None] + sh0[:, :, None], b * sc1[:, :, None] + sh1[:, :, None]], 1)
    assert np.abs(got - ref).max() < 2e-5


@pytest.mark.parametrize("split", [1, 0])
@pytest.mark.parametrize("tiles", [1024, 1031, 300, 72])
def test_norm_finalize_many_tiles_narrow_split(lib, tiles, split):
    """Round 5: with many tiles per channel (a 512x512 map: 1024+) the finalize launch takes few channels per workgroup (down to one GroupNorm group, at
    least 4) so that a thread walks about one round of eight tiles - main loop, conditional tail and the group reduction over a narrow range.
    `split` = 1: that tile-aware split (the product's THA4_NORM_TILE_SPLIT tuning option); 0: the SHIPPED split on the same many-tile tensors."""
    lib.emu_set_norm_tile_split(split)
    try:
        _norm_many_tiles(lib, tiles)
    finally:
        lib.emu_set_norm_tile_split(0)


def _norm_many_tiles(lib, tiles):
    rng = np.random.default_rng(tiles)
    n, C_ = 1, 32
    px = tiles * 4
    x = (rng.standard_normal((n, C_, px)) * 1.5 + 0.3).astype(np.float32)
    st, cb = _partials(x, tiles)
    gamma = (1 + 0.2 * rng.standard_normal(C_)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(C_)).astype(np.float32)
    f1 = (0.3 * rng.standard_normal((n, 2 * C_))).astype(np.float32)
    sc = np.zeros((n, cb * 16), np.float32); sh = np.zeros_like(sc)
    lib.emu_norm(n, 1, P(st), tiles, cb, None, 0, 0, C_, 32, C.c_float(1.0 / px), C.c_float(1e-5), P(gamma), P(beta), None, P(f1), P(sc), P(sh), None, None)
    h = F.group_norm(torch.from_numpy(x).double().reshape(n, C_, tiles, 4), 32, torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), eps=1e-5)
    ft = torch.from_numpy(f1).double()
    h = h * (1 + ft[:, :C_, None, None]) + ft[:, C_:, None, None]
    got = x * sc[:, :C_, None] + sh[:, :C_, None]
    assert np.abs(got - h.reshape(n, C_, px).numpy()).max() < 2e-5


def test_gemv_and_attention(lib):
    rng = np.random.default_rng(9)
    n, rows, k = 2, 37, 256
    w = rng.standard_normal((rows, k)).astype(np.float32) / 16
    b = rng.standard_normal(rows).astype(np.float32)
    x = rng.standard_normal((n, k)).astype(np.float32)
    y = np.zeros((n, rows), np.float32)
    lib.emu_gemv(n, rows, k, P(w), P(b), P(x), ACT["silu"], ACT["none"], P(y))
    ref = F.linear(F.silu(torch.from_numpy(x).double()), torch.from_numpy(w).double(), torch.from_numpy(b).double()).numpy()
    assert np.abs(y - ref).max() < 1e-5
    # attention core (unet.py:192-202) at reduced size: 64 channels, 2 heads (head dim 32), 64 tokens
    Cc, heads, L = 64, 2, 64
    qkv = rng.standard_normal((n, 3 * Cc, L)).astype(np.float32)
    out = np.zeros((n, Cc, L), np.float32)
    lib.emu_attention(n, Cc, heads, L, P(qkv), P(out))
    t = torch.from_numpy(qkv).double()
    q, kk, v = t.chunk(3, dim=1)
    ch = Cc // heads
    s = 1.0 / np.sqrt(np.sqrt(ch))
    wgt = torch.softmax(torch.einsum("bct,bcs->bts", (q * s).reshape(n * heads, ch, L), (kk * s).reshape(n * heads, ch, L)), -1)
    ref = torch.einsum("bts,bcs->bct", wgt, v.reshape(n * heads, ch, L)).reshape(n, Cc, L).numpy()
    assert np.abs(out - ref).max() < 1e-5


def test_tile_conv_launch_plans(lib):
    """plan_tile_conv (host logic) over every convolution shape of the full model: valid geometry, LDS and staging
    budgets respected, the tile grid covers the map, K splits cover the K groups with at least one group each, and
    maps >= 128x128 are never K-split (partial traffic would be ksplit x the output)."""
    shapes = []      # (kind, k, tile_h, tile_w, cin, cout)
    for hw in (128, 192):                                                    # encoder-decoders (192: face morpher)
        shapes += [(0, 3, hw, hw, 4, 64), (0, 3, hw, hw, 8, 64), (0, 3, hw, hw, 64, 12)]
        c = 64
        for lvl in range(3):
            shapes.append((1, 4, hw >> (lvl + 1), hw >> (lvl + 1), c, 2 * c))
            c *= 2
        shapes += [(0, 3, hw >> 3, hw >> 3, 512, 512), (0, 3, hw >> 3, hw >> 3, 539, 512)]
        for lvl in range(3):
            shapes.append((2, 4, hw >> (3 - lvl), hw >> (3 - lvl), c, c // 2))
            c //= 2
    for size, ch in ((256, 64), (512, 32)):                                   # U-Nets
        shapes += [(0, 3, size, size, 14, ch), (0, 3, size, size, ch, 7)]
        for lvl, mult in enumerate((1, 2, 4, 4, 4, 8)):
            s_ = size >> lvl
            if s_ < 16:
                break
            shapes += [(0, 3, s_, s_, ch * mult, ch * mult), (0, 3, s_, s_, 3 * ch * mult // 2, ch * mult), (0, 3, s_, s_, 2 * ch * mult, ch * mult)]
    out = (C.c_int * 11)()
    seen_split = seen_ragged = False
    for kind, k, th, tw, cin, cout in shapes:
        nb = (cout + 15) // 16
        tmb = 4 if nb % 4 == 0 else (2 if nb % 2 == 0 else 1)
        nq = ((cin + 15) // 16 + 1) // 2
        lib.emu_plan_tile_conv(kind, k, th, tw, tmb, nb // tmb, nq, out)
        ok, pg, ksplit, twl, tile_h, tiles, win_h, win_w, tpc, slots, lds = list(out)
        assert ok, (kind, k, th, tw, cin, cout)
        assert pg in (1, 2, 4) and 128 * pg == tile_h << twl
        assert tiles == -(-th // tile_h) * -(-tw // (1 << twl))
        assert win_h * win_w * 4 <= 5 * 512 and lds <= 160 * 1024 and 2 <= slots <= 4
        ntaps = {0: k * k, 1: 16, 2: 4}[kind]
        assert 1 <= tpc <= min(9, ntaps) and (tpc <= 4 or tpc * tmb * 2048 <= 72 * 1024)   # whole K groups per chunk where the ring fits, else 5 + 4
        assert 1 <= ksplit <= min(nq, 16)
        per = -(-nq // ksplit)
        assert (ksplit - 1) * per < nq                     # every split owns at least one K group
        if th * tw >= 128 * 128:
            assert ksplit == 1, (th, tw, cin, cout, ksplit)
        seen_split |= ksplit > 1
        seen_ragged |= tiles * 128 * pg != th * tw
    assert seen_split and seen_ragged


# ---- per-op edge cases; the same functions run on the device through tests/test_ops_device.py ----

def test_norm_finalize_near_zero_variance(lib):
    """InstanceNorm / GroupNorm(32) on channels whose variance (1e-8) is far below eps (1e-5) and below the fp32 noise of
    their own second moment: the fp64 finalize must neither produce NaN nor amplify (normalization.py:90-95, unet.py:65-66)."""
    rng = np.random.default_rng(11)
    n, px, c = 2, 256, 64
    x = (0.5 + 1e-4 * rng.standard_normal((n, c, px))).astype(np.float32)
    x[:, 5] = 0.25                                         # an exactly constant channel
    x[:, 6] = 0.0                                          # a dead channel
    st, cb = _partials(x, 4)
    gamma = (1 + 0.2 * rng.standard_normal(c)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(c)).astype(np.float32)
    for groups in (0, 32):
        sc = np.zeros((n, cb * 16), np.float32); sh = np.zeros_like(sc)
        lib.emu_norm(n, 1, P(st), 4, cb, None, 0, 0, c, groups, C.c_float(1.0 / px), C.c_float(1e-5), P(gamma), P(beta), None, None,
                     P(sc), P(sh), None, None)
        t = torch.from_numpy(x).double().reshape(n, c, 16, 16)
        g64, b64 = torch.from_numpy(gamma).double(), torch.from_numpy(beta).double()
        ref = (F.instance_norm(t, weight=g64, bias=b64, eps=1e-5) if groups == 0 else F.group_norm(t, 32, g64, b64, eps=1e-5)).reshape(n, c, px).numpy()
        got = x.astype(np.float64) * sc[:, :c, None] + sh[:, :c, None]
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() < 2e-4, groups     # (x - mean) * rstd with x ~ 0.5: one fp32 ulp of x*scale is 3e-5 * rstd/300


def test_attention_large_logits(lib):
    """softmax over logits of +-1e3 (q, k of magnitude 30): max-subtraction keeps exp() in range (unet.py:192-202)."""
    rng = np.random.default_rng(13)
    n, Cc, heads, L = 1, 64, 2, 64
    qkv = rng.standard_normal((n, 3 * Cc, L)).astype(np.float32)
    qkv[:, :2 * Cc] *= 30.0
    out = np.zeros((n, Cc, L), np.float32)
    assert lib.emu_attention(n, Cc, heads, L, P(qkv), P(out)) == 0
    t = torch.from_numpy(qkv).double()
    q, kk, v = t.chunk(3, dim=1)
    ch = Cc // heads
    s = 1.0 / np.sqrt(np.sqrt(ch))
    logits = torch.einsum("bct,bcs->bts", (q * s).reshape(n * heads, ch, L), (kk * s).reshape(n * heads, ch, L))
    assert float(logits.abs().max()) > 500
    ref = torch.einsum("bts,bcs->bct", torch.softmax(logits, -1), v.reshape(n * heads, ch, L)).reshape(n, Cc, L).numpy()
    assert np.isfinite(out).all()
    assert np.abs(out - ref).max() < 2e-3                  # near-one-hot rows: logit error 1e3 * 2^-23 * 32 terms -> weights move by ~1e-3


def run_unet_tail(lib, S, head, src):
    n = head.shape[0]
    outs = [np.zeros((n, c, S, S), np.float32) for c in (4, 1, 4, 2, 4)]
    assert lib.emu_unet_tail(S, n, P(head), P(src), *[P(o) for o in outs]) == 0
    return outs


def check_warp_blend_tail(lib, S):
    """GridChangeApplier.apply + blend (image_processing_util.py:33-54, morpher_00.py:53-66) with offsets up to +-1.5
    (three quarters of the image) so that every border clamps, against F.grid_sample in fp64."""
    rng = np.random.default_rng(17)
    n = 1
    head = rng.standard_normal((n, 7, S, S)).astype(np.float32)
    head[:, 4:6] = rng.uniform(-1.5, 1.5, (n, 2, S, S)).astype(np.float32)
    head[:, 4, :, :4] = -1.5; head[:, 4, :, -4:] = 1.5; head[:, 5, :4] = -1.5; head[:, 5, -4:] = 1.5     # push all four rims outside
    head[:, 4:6, S // 2, S // 2] = 0.0                                                               # and an exact identity tap
    src = rng.standard_normal((n, 4, S, S)).astype(np.float32)
    merged, alpha, warped, grid, direct = run_unet_tail(lib, S, head, src)
    h64, s64 = torch.from_numpy(head).double(), torch.from_numpy(src).double()
    ident = F.affine_grid(torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]], dtype=torch.float64), [n, 1, S, S], align_corners=False)
    g = ident + h64[:, 4:6].permute(0, 2, 3, 1)
    wref = F.grid_sample(s64, g, mode="bilinear", padding_mode="border", align_corners=False)
    aref = torch.sigmoid(h64[:, 6:7])
    assert np.abs(warped - wref.numpy()).max() < 3e-4     # tap weights from fp32 pixel coordinates (1 ulp of 512 = 6e-5) x gradient
    assert np.abs(alpha - aref.numpy()).max() < 1e-6
    assert np.array_equal(grid, head[:, 4:6]) and np.array_equal(direct, head[:, 0:4])
    mref = h64[:, 0:4] * aref + wref * (1 - aref)
    assert np.abs(merged - mref.numpy()).max() < 3e-4
    assert np.abs(warped[0, :, S // 2, S // 2] - src[0, :, S // 2, S // 2]).max() < 1e-6      # identity tap reproduces the pixel
    rim = np.abs(warped - wref.numpy())
    assert max(rim[..., :4].max(), rim[..., -4:].max(), rim[..., :4, :].max(), rim[..., -4:, :].max()) < 3e-4


def test_warp_blend_tail_borders(lib):
    check_warp_blend_tail(lib, 256)


def test_upscaler_input_bilinear_edges_and_warp(lib):
    """F.interpolate(bilinear, align_corners=False) x2 on the first / last rows and columns (mode_07.py:108-115) and the
    coarse warp of the upscaler input (upscaler_02.py:78-83)."""
    rng = np.random.default_rng(19)
    n = 1
    rest = rng.standard_normal((n, 4, 512, 512)).astype(np.float32)
    merged = rng.standard_normal((n, 4, 256, 256)).astype(np.float32)
    grid = rng.uniform(-0.6, 0.6, (n, 2, 256, 256)).astype(np.float32)
    out = np.zeros((n, 14, 512, 512), np.float32)
    assert lib.emu_upscaler_input(n, P(rest), P(merged), P(grid), P(out)) == 0
    r64 = torch.from_numpy(rest).double()
    up_m = F.interpolate(torch.from_numpy(merged).double(), size=(512, 512), mode="bilinear", align_corners=False)
    up_g = F.interpolate(torch.from_numpy(grid).double(), size=(512, 512), mode="bilinear", align_corners=False)
    ident = F.affine_grid(torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]], dtype=torch.float64), [n, 1, 512, 512], align_corners=False)
    w = F.grid_sample(r64, ident + up_g.permute(0, 2, 3, 1), mode="bilinear", padding_mode="border", align_corners=False)
    assert np.array_equal(out[:, 0:4], rest)
    e_m, e_g = np.abs(out[:, 4:8] - up_m.numpy()), np.abs(out[:, 12:14] - up_g.numpy())
    assert e_m.max() < 2e-6 and e_g.max() < 1e-6
    for band in (np.s_[..., 0:2, :], np.s_[..., 510:512, :], np.s_[..., :, 0:2], np.s_[..., :, 510:512]):      # edge-clamped taps
        assert e_m[band].max() < 2e-6 and e_g[band].max() < 1e-6
    assert np.abs(out[:, 8:12] - w.numpy()).max() < 5e-4


def _norm_ref(x, channels, groups, gamma, beta, f0, f1):
    """InstanceNorm / GroupNorm (+ two FiLM stages) of [n][c][h][w] in fp64 (unet.py:90-97,157-163)."""
    t = torch.from_numpy(x).double()
    g64, b64 = torch.from_numpy(gamma).double(), torch.from_numpy(beta).double()
    h = F.instance_norm(t, weight=g64, bias=b64, eps=1e-5) if groups == 0 else F.group_norm(t, groups, g64, b64, eps=1e-5)
    if f0 is not None:
        ft = torch.from_numpy(f0).double()
        h = h * (1 + ft[None, :channels, None, None]) + ft[None, channels:, None, None]
    if f1 is not None:
        ft = torch.from_numpy(f1).double()
        h = h * (1 + ft[:, :channels, None, None]) + ft[:, channels:, None, None]
    return h


FUSED_CASES = [
    # pg  k  act     c0   c1   h   w  cout groups film  tiles (twl, x)
    (21, 3, "relu", 64, 0, 16, 16, 32, 0, False, 4, (4, 0)),      # conv_small + InstanceNorm/ReLU (encoder-decoder bottleneck)
    (22, 3, "silu", 64, 32, 16, 16, 32, 32, True, 2, (4, 0)),     # conv_small + GroupNorm(32) over a concatenation (groups of 3 straddle it) + both FiLM stages
    (24, 1, "none", 128, 0, 16, 16, 48, 32, False, 8, (4, 0)),    # 1x1 qkv projection after GroupNorm
    (12, 3, "silu", 64, 32, 32, 32, 32, 32, True, 4, (4, 1)),     # conv_tile_kernel<2,2> with the table in its prologue
    (11, 3, "relu", 48, 0, 16, 16, 32, 0, False, 2, (4, 2)),      # conv_tile_kernel K split (phase 1 builds the table, phase 2 skips it)
    (42, 3, "silu", 64, 32, 32, 32, 32, 32, True, 4, (4, 1)),     # conv_tile_kernel<2,2,.,2>: sixteen waves build the table (1024 threads)
    (52, 3, "silu", 64, 32, 32, 32, 32, 32, True, 4, (4, 1)),     # conv_tile_kernel<2,2,.,1,4>: four waves (256 threads) build the table
    (32, 1, "silu", 96, 64, 16, 32, 32, 32, True, 4, (4, 0)),     # conv_point_kernel<2,2>: GroupNorm(32) over a concatenation + FiLM in its prologue
    (31, 1, "relu", 64, 0, 16, 16, 32, 0, False, 2, (4, 0)),      # conv_point_kernel<2,1>: InstanceNorm
]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_conv_with_fused_norm(lib, case):
    """The normalisation folded into the consumer convolution (FusedNorm): the kernel reduces the producer's per-tile moments
    itself.  Reference: torch norm (+FiLM) -> activation -> conv in fp64."""
    pg, k, act_in, c0, c1, h, w, cout, groups, film, tiles, (twl, extra) = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    n = 2
    x0 = (rng.standard_normal((n, c0, h, w)) * 1.7 + 0.4).astype(np.float32)
    x1 = (rng.standard_normal((n, c1, h, w)) * 0.6 - 0.2).astype(np.float32) if c1 else None
    cin = c0 + c1
    gamma = (1 + 0.2 * rng.standard_normal(cin)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(cin)).astype(np.float32)
    f0 = (0.3 * rng.standard_normal(2 * cin)).astype(np.float32) if film else None
    f1 = (0.3 * rng.standard_normal((n, 2 * cin))).astype(np.float32) if film else None
    st0, cb0 = _partials(x0.reshape(n, c0, h * w), tiles)
    st1 = _partials(x1.reshape(n, c1, h * w), tiles * 2)[0] if c1 else None
    weight = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    xin = np.concatenate([x0, x1], 1) if c1 else x0
    hn = torch_act(_norm_ref(xin, cin, groups, gamma, beta, f0, f1), act_in)
    ref = F.conv2d(hn, torch.from_numpy(weight).double(), torch.from_numpy(bias).double(), padding=k // 2).numpy()
    lib.emu_set_fused_norm(P(st0), tiles, P(st1), tiles * 2 if c1 else 0, cin, groups, C.c_float(1.0 / (h * w)), P(gamma), P(beta), P(f0), P(f1))
    tmb = 1 if 20 <= pg < 30 else 2
    out, stats = run_conv(lib, 0, k, tmb, pg, 0, act_in, x0, x1, False, None, None, weight, bias, None, None, 1, twl,
                          extra if pg < 20 or pg >= 50 else (1 if pg >= 30 else 0))
    assert np.abs(out - ref).max() < 5e-5, np.abs(out - ref).max()
    assert np.abs(stats[..., 0] - ref.sum(axis=(2, 3))).max() < 2e-3


def _acc_to_sums(acc):
    """MomentAcc accumulators [n][8][cw][4] int64 (hi, lo) pairs -> per-channel sums [n][cw][2] (float64): value = hi + lo 2^-32, shards added"""
    v = acc.astype(np.float64)
    s = (v[..., 0] + v[..., 1] / 4294967296.0).sum(1)
    q = (v[..., 2] + v[..., 3] / 4294967296.0).sum(1)
    return np.stack([s, q], -1)


@pytest.mark.parametrize("pg_prod,pg_cons,h,w,c0,c1,cmid", [(12, 12, 32, 32, 32, 0, 32), (54, 52, 32, 32, 16, 16, 64), (11, 14, 32, 32, 48, 0, 32), (22, 12, 16, 16, 64, 0, 32)])
def test_moment_accumulators_producer_to_consumer(lib, pg_prod, pg_cons, h, w, c0, c1, cmid):
    """Round 5: the producer convolution (conv_tile_kernel) adds its tiles' sums to per-channel integer accumulators with atomics and the CONSUMER folds
    the 8 shards into its GroupNorm + FiLM + SiLU table - no norm_finalize launch.  (a) the accumulators hold the tensor's moments to ~2^-32 per tile sum
    whatever the tiling; (b) the consumer (conv_tile_kernel, 8- and 4-wave forms; conv_small_kernel never gets one) reading them equals torch's norm -> act -> conv in fp64;
    (c) the same bits as folding per-tile moments is NOT promised (different summation), the parity gate is."""
    rng = np.random.default_rng(pg_prod * 100 + pg_cons)
    n = 2
    x0 = (rng.standard_normal((n, c0, h, w)) * 1.3).astype(np.float32)
    x1 = (rng.standard_normal((n, c1, h, w)) * 0.7).astype(np.float32) if c1 else None
    cin = c0 + c1
    w1 = (rng.standard_normal((cmid, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b1 = rng.standard_normal(cmid).astype(np.float32)
    nb = (cmid + 15) // 16
    acc = np.zeros((n, 8, nb * 16, 4), np.int64)
    lib.emu_set_acc_output(acc.ctypes.data_as(C.POINTER(C.c_longlong)))
    tmb = 1 if 20 <= pg_prod < 30 else 2                     # (pg 2x: conv_small_kernel as the producer)
    mid, stats = run_conv(lib, 0, 3, tmb, pg_prod, 0, "none", x0, x1, False, None, None, w1, b1, None, None, 1, 4, 0 if 20 <= pg_prod < 30 else 1)
    sums = _acc_to_sums(acc)
    mid64 = mid.astype(np.float64)
    assert np.abs(sums[:, :cmid, 0] - mid64.sum(axis=(2, 3))).max() < 1e-3
    assert np.abs(sums[:, :cmid, 1] - (mid64 ** 2).sum(axis=(2, 3))).max() < 3e-3
    assert np.abs(sums[:, :cmid, 0] - stats[..., 0]).max() < 1e-3          # the per-tile moments are still written beside them
    assert (acc[:, :, cmid:] == 0).all()                                    # padded channels: exact zeros
    assert (acc != 0).any(axis=(0, 2, 3)).sum() >= min(8, 2)                # several shards were used
    # consumer: GroupNorm(32 or 16) + FiLM + SiLU folded from the accumulators
    groups = 16
    gamma = (1 + 0.2 * rng.standard_normal(cmid)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(cmid)).astype(np.float32)
    f0 = (0.3 * rng.standard_normal(2 * cmid)).astype(np.float32)
    f1 = (0.3 * rng.standard_normal((n, 2 * cmid))).astype(np.float32)
    cout = 32
    w2 = (rng.standard_normal((cout, cmid, 3, 3)) / np.sqrt(cmid * 9)).astype(np.float32)
    b2 = rng.standard_normal(cout).astype(np.float32)
    hn = torch_act(_norm_ref(mid, cmid, groups, gamma, beta, f0, f1), "silu")
    ref = F.conv2d(hn, torch.from_numpy(w2).double(), torch.from_numpy(b2).double(), padding=1).numpy()
    lib.emu_set_fused_norm_acc(acc.ctypes.data_as(C.POINTER(C.c_longlong)), None, cmid, groups, C.c_float(1.0 / (h * w)), P(gamma), P(beta), P(f0), P(f1))
    out, _ = run_conv(lib, 0, 3, 1 if 20 <= pg_cons < 30 else 2, pg_cons, 0, "silu", mid, None, False, None, None, w2, b2, None, None, 1, 4,
                      1 if pg_cons < 20 or pg_cons >= 50 else 0)
    assert np.abs(out - ref).max() < 5e-5, np.abs(out - ref).max()


def test_small_conv_launch_plans(lib):
    """plan_small_conv over the small-map shapes of the full model: the chosen tile covers the map in ONE round of
    workgroups (<= 256: one workgroup per CU is all the kernel's registers allow), the window fits the staging budget,
    LDS <= 160 KiB, few K groups are spread over the waves by tap ranges; shapes that need more than one round are
    refused (they stay on conv_tile_kernel's K split)."""
    out = (C.c_int * 10)()
    one_round = [(0, 3, 16, 16, 256, 256), (0, 3, 16, 16, 512, 512), (0, 3, 16, 16, 524, 512), (0, 3, 32, 32, 256, 256), (0, 3, 32, 32, 512, 256),
                 (0, 1, 16, 16, 256, 768), (0, 1, 16, 16, 256, 256), (0, 1, 32, 32, 512, 256), (2, 4, 16, 16, 512, 256), (2, 4, 24, 24, 512, 256),
                 (2, 4, 32, 32, 256, 128), (2, 4, 64, 64, 128, 64)]
    refused = [(0, 3, 24, 24, 539, 512), (0, 3, 64, 64, 128, 128), (0, 3, 48, 48, 256, 256), (2, 4, 96, 96, 128, 64), (2, 4, 48, 48, 256, 128)]
    for (kind, k, th, tw, cin, cout) in one_round + refused:
        nb = (cout + 15) // 16
        nq = ((cin + 15) // 16 + 1) // 2
        lib.emu_plan_small_conv(kind, k, th, tw, nb, nq, out)
        ok, pg, twl, tile_h, tiles, win_h, win_w, upq, lds, wgs = list(out)
        if (kind, k, th, tw, cin, cout) in refused:
            assert not ok, (kind, k, th, tw, cin, cout)
            continue
        assert ok, (kind, k, th, tw, cin, cout)
        assert pg in (1, 2, 4) and 16 * pg == tile_h << twl
        assert tiles == -(-th // tile_h) * -(-tw // (1 << twl)) and wgs == tiles * nb
        assert win_h * win_w * 4 <= 7 * 64 and lds + 8192 <= 160 * 1024
        ntaps = {0: k * k, 1: 16, 2: 4}[kind]
        assert upq in (1, 2, 4, 8) and upq <= max(1, ntaps) and (nq * upq >= 8 or upq * 2 > ntaps)
        assert 128 <= wgs <= 256, (th, tw, cin, cout, wgs)


def test_fastdiv_is_exact_on_the_range_the_planner_allows(lib):
    """FastDiv (csrc/full_kernels.h): q = (x * ceil(2^42 / d)) >> 42 replaces every run-time integer division of the convolution prologues.  Exact for
    0 <= x < 2^22, 1 <= d < 2^20 - the bounds finish_conv_args / finish_conv_batch check (round 5: the range covers the batch-sized divisor of the
    documented handle limit, max_batch = 256 x 1024 tiles = 2^18, which round 4's 2^40 / d < 2^18 form refused); refused outside."""
    rng = np.random.default_rng(7)
    ds = np.concatenate([np.arange(1, 4097), rng.integers(1, 1 << 18, 20000), rng.integers(1 << 18, 1 << 20, 20000),
                         np.array([(1 << 18) - 1, 1 << 18, (1 << 18) + 1, (1 << 20) - 1, 65535, 65536, 65537, 3, 7, 9, 18, 24, 48, 96])]).astype(np.int32)
    xs = []
    for d in ds:
        k = rng.integers(0, ((1 << 22) - 1) // int(d) + 1, 6)
        xs.append(np.clip(np.array([0, d - 1, d, d + 1, (1 << 22) - 1, (1 << 22) - int(d), *(k * int(d)), *(k * int(d) + int(d) - 1)], np.int64), 0, (1 << 22) - 1))
    x = np.concatenate(xs).astype(np.int32)
    d = np.repeat(ds, len(xs[0])).astype(np.int32)
    out = np.empty_like(x)
    ip = C.POINTER(C.c_int)
    assert lib.emu_fast_div(x.ctypes.data_as(ip), d.ctypes.data_as(ip), int(x.size), out.ctypes.data_as(ip)) == 0
    np.testing.assert_array_equal(out, x // d)
    bad = np.array([0, -3, 1 << 20, 1 << 22], np.int32)
    o2 = np.empty_like(bad)
    lib.emu_fast_div(np.zeros(4, np.int32).ctypes.data_as(ip), bad.ctypes.data_as(ip), 4, o2.ctypes.data_as(ip))
    assert (o2 == -1).all()
