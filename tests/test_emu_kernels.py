"""Kernel-logic tests on CPU: the UNMODIFIED kernel source (csrc/siren_kernels.h) compiled against
the fiber SIMT emulator (tests/emu) and checked workgroup-by-workgroup against the fp64 oracle.
This validates the MFMA fragment packing, the LDS images, the weight streaming schedule, the
upsample taps and the warp/blend epilogue without a GPU.  (GPU parity proper: test_student_gpu.py.)"""
import numpy as np
import pytest

from oracle import student_oracle as so
from tests.emu_util import K_FACE, K_L0, K_L1, K_L2, K_POSEBIAS, EmuStudent, pack_z, unpack_z


@pytest.fixture(scope="module", params=[1, 2], ids=["fp32-mfma", "fp16x3-mfma"])
def ctx(request, built, golden_weights, golden_io):
    """Both kernel generations: 1 = exact-fp32 v_mfma_f32_16x16x4_f32, 2 = fp16 hi/lo split on v_mfma_f32_16x16x32_f16."""
    pose = golden_io["poses"][0]
    emu = EmuStudent(golden_weights, gen=request.param)
    emu.buf("pose")[:] = pose
    emu.buf("image")[:] = golden_io["image_f32"].reshape(-1)
    it = so.student_intermediates(golden_weights, pose)
    yield emu, it, pose
    emu.close()


def test_sin_omega_accuracy(ctx):
    emu = ctx[0]
    z = np.random.default_rng(0).uniform(-1.6, 1.6, 4000).astype(np.float32)
    err = max(abs(emu.sin_omega(v) - np.sin(np.float64(np.float32(30.0) * v))) for v in z)
    assert err < 2.5e-7
    assert emu.sin_omega(0.0) == 0.0          # padded channels must stay exactly zero


def test_sin_u_accuracy(ctx):
    """generation 2's sine takes TURNS (omega_0 / 2 pi is folded into the packed weights, siren_layout.h): on the device it is one
    v_sin_f32; the emulator restates it as sin(2 pi t) - exact to fp32 rounding - over the range the students reach (|t| < 7)"""
    emu = ctx[0]
    rng = np.random.default_rng(1)
    t = np.concatenate([rng.uniform(-10, 10, 4000), rng.uniform(-64, 64, 1000), [0.0, 0.25, -0.25, 0.5, 1e-8]]).astype(np.float32)
    err = max(abs(emu.sin_u(v) - np.sin(2.0 * np.pi * np.float64(v))) for v in t)
    assert err < 1.0e-7
    assert emu.sin_u(0.0) == 0.0


def test_posebias_kernel(ctx):
    emu, it, _ = ctx
    emu.run(K_POSEBIAS, 0, emu.grid(K_POSEBIAS))
    pb = emu.buf("pbias") / emu.handoff_scale
    tol = 1e-6 if emu.handoff_scale == 1.0 else 2e-6      # one more rounding (30x) in generation 2
    assert np.abs(pb[0:128] - it["pb_face"]).max() < tol
    assert np.abs(pb[128:488] - it["pb0"]).max() < tol
    assert np.abs(pb[512:692] - it["pb1"]).max() < tol
    assert np.abs(pb[704:794] - it["pb2"]).max() < tol
    assert pb.shape[0] == 800
    assert not pb[488:512].any() and not pb[692:704].any() and not pb[794:800].any()


def test_face_kernel_blocks(ctx, golden_weights):
    emu, it, pose = ctx
    emu.run(K_POSEBIAS, 0, emu.grid(K_POSEBIAS))
    ref = so.face_forward_numpy(golden_weights, pose[:39].astype(np.float64)).reshape(4, -1)
    for b in (0, 131, emu.grid(K_FACE) - 1):
        emu.run(K_FACE, b)
        px = emu.block_pixels(K_FACE, b)
        got = emu.buf("face").reshape(4, -1)[:, px]
        assert np.abs(got - ref[:, px]).max() < 5e-5


def test_level0_kernel_blocks(ctx):
    emu, it, _ = ctx
    emu.run(K_POSEBIAS, 0, emu.grid(K_POSEBIAS))
    ref = it["z1"].reshape(180, -1)
    for b in (emu.block_of_tile(K_L0, t) for t in (0, 77, emu.grid(K_L0) - 1)):
        emu.run(K_L0, b)
        px = emu.block_pixels(K_L0, b)
        got = unpack_z(emu.buf("z1"), 12, 128 * 128, 180)[:, px] / emu.handoff_scale
        assert np.abs(got - ref[:, px]).max() < 5e-4       # fp32 through 2x360-wide sine layers
        pad = emu.buf("z1").reshape(12, 4, 128 * 128, 4)[11, 1:, px, :]
        assert not pad.any()                                # channels 180..191 are exact zeros


def test_level1_kernel_blocks(ctx):
    emu, it, _ = ctx
    emu.run(K_POSEBIAS, 0, emu.grid(K_POSEBIAS))
    emu.buf("z1")[:] = pack_z(it["z1"].reshape(180, -1) * emu.handoff_scale, 12)
    ref = it["z2"].reshape(90, -1)
    for b in (emu.block_of_tile(K_L1, t) for t in (0, 1, 300, emu.grid(K_L1) - 1)):   # 0/1: image corners (clamped taps)
        emu.run(K_L1, b)
        px = emu.block_pixels(K_L1, b)
        got = unpack_z(emu.buf("z2"), 6, 256 * 256, 90)[:, px] / emu.handoff_scale
        assert np.abs(got - ref[:, px]).max() < 2e-5


def test_level2_kernel_blocks(ctx, golden_weights, golden_io):
    emu, it, pose = ctx
    emu.run(K_POSEBIAS, 0, emu.grid(K_POSEBIAS))
    emu.buf("z2")[:] = pack_z(it["z2"].reshape(90, -1) * emu.handoff_scale, 6)
    face = so.face_forward_numpy(golden_weights, pose[:39].astype(np.float64))
    emu.buf("face")[:] = face.astype(np.float32).reshape(-1)
    ref = so.student_forward_numpy(golden_weights, golden_io["image_f32"], pose)
    tol = {"out_blended": 5e-4, "out_alpha": 1e-5, "out_color": 2e-5, "out_warped": 5e-4, "out_grid": 5e-6}
    # 288/289: row 144 (inside the pasted face rows), 160..: row 80 = first face row
    g = emu.grid(K_L2)
    # rows 0, 80 (first face row), 144 (inside the pasted face), 200 and the last one
    for b in (emu.block_of_tile(K_L2, t) for t in sorted({0, 1, g * 80 // 512, g * 144 // 512, g * 144 // 512 + 1, g * 200 // 512, g - 1})):
        emu.run(K_L2, b)
        px = emu.block_pixels(K_L2, b)
        for name, k, c in (("out_blended", 0, 4), ("out_alpha", 1, 1), ("out_color", 2, 4), ("out_warped", 3, 4),
                           ("out_grid", 4, 2)):
            got = emu.buf(name).reshape(c, -1)[:, px]
            assert np.abs(got - ref[k].reshape(c, -1)[:, px]).max() < tol[name], (b, name)


def test_sine_argument_bound_and_domain(built, golden_weights, char_weights):
    """Round-3 advisor finding: the default kernels' sine is ONE v_sin_f32 on an argument in turns, which returns 0 beyond 256 turns
    where the reference's torch.sin accepts anything.  (a) The host-side bound tha4_student_create / _set_weights enforce
    (siren_layout.h sine_argument_bound_turns: max over rows of c (sum |W| + |b|)) is far below the limit for both shipped
    students and above it for weights scaled by 100; it really is an upper bound of what the oracle's layers see.  (b) The emulator
    models the instruction's domain (0 beyond 256 turns) instead of an ideal sine."""
    import ctypes as C
    import os
    from tha4_amd import _capi
    from tha4_amd.weights import split_flat_weights
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libtha4_emu.so"))
    lib.emu_sine_argument_bound_turns.restype = C.c_double
    lib.emu_sine_argument_bound_turns.argtypes = [C.c_void_p]
    lib.emu_sine_turns_limit.restype = C.c_double
    lib.emu_sin_u.restype = C.c_float
    lib.emu_sin_u.argtypes = [C.c_float]
    limit = lib.emu_sine_turns_limit()
    assert limit == 256.0

    def bound(flat):
        ws, keep = _capi.build_student_weights(*split_flat_weights(flat))
        return lib.emu_sine_argument_bound_turns(C.byref(ws))
    for name in ("lambda_00", "lambda_01"):
        b = bound(char_weights[name])
        assert 1.0 < b < 64.0, (name, b)                 # the shipped students: a wide margin to 256 turns
    big = {k: (v * 100.0 if k.endswith("sine_layers.3.linear.weight") else v) for k, v in golden_weights.items()}
    assert bound(big) > limit
    # a NaN weight in an EARLY layer makes the bound NaN whatever follows it (round-4 advisor finding: a later finite row used to
    # overwrite it), alone and behind an over-limit row
    def poisoned(with_big):
        d = {k: np.array(v, copy=True) for k, v in golden_weights.items()}
        if with_big:
            d["face.siren.sine_layers.0.linear.weight"] = d["face.siren.sine_layers.0.linear.weight"] * 1000.0
        d["face.siren.sine_layers.1.linear.weight"][3, 5] = np.nan
        return d
    for with_big in (False, True):
        b = bound(poisoned(with_big))
        assert np.isnan(b) and not (b < limit), (with_big, b)
    # it IS an upper bound: the largest |30 (W x + b)| / 2 pi the fp64 oracle meets in the face morpher's hidden layers for a pose
    pose = so.random_poses(1, seed=5)[0]
    x = None
    worst = 0.0
    c = 30.0 / (2.0 * np.pi)
    ax = so.position_axis(128)
    pos = np.stack([np.broadcast_to(ax[None, :], (128, 128)), np.broadcast_to(ax[:, None], (128, 128))])       # channel 0 = x, 1 = y
    inp = np.concatenate([pos, np.broadcast_to(pose[:39, None, None].astype(np.float64), (39, 128, 128))], 0).reshape(41, -1)
    x = inp
    for i in range(8):
        W = golden_weights[f"face.siren.sine_layers.{i}.linear.weight"].astype(np.float64)
        b = golden_weights[f"face.siren.sine_layers.{i}.linear.bias"].astype(np.float64)
        u = W @ x + b[:, None]
        worst = max(worst, float(np.abs(u).max()) * c)
        x = np.sin(30.0 * u)
    assert worst <= bound(golden_weights)
    # (b) the instruction's domain in the emulator
    assert lib.emu_sin_u(C.c_float(300.25)) == 0.0 and lib.emu_sin_u(C.c_float(-256.5)) == 0.0
    assert abs(lib.emu_sin_u(C.c_float(255.25)) - 1.0) < 1e-6
