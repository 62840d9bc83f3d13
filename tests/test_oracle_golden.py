"""Pin the CPU oracle (oracle/student_oracle.py) against outputs of the UNMODIFIED reference
(tests/golden/*.npz, produced by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import student_oracle as so

SUB = slice(1, None, 3)
NAMES = ["blended", "alpha", "color_change", "warped", "grid_change", "face"]
# cross-machine fp32 noise of the reference itself (tests/golden/student_lambda_00_noise.json):
# blended 3.6e-4, warped 1.3e-3 between 1 and 8 threads on ONE machine.
TOL32 = [6e-4, 1e-4, 2e-4, 2.5e-3, 2e-5, 6e-5]


@pytest.mark.parametrize("character", ["lambda_00", "lambda_01"])
def test_torch_restatement_matches_reference_fp32(character, char_weights, char_io):
    golden_weights, golden_io = char_weights[character], char_io[character]
    out = so.student_forward_torch(golden_weights, golden_io["image_f32"], golden_io["poses"][:3], "float32")
    full = out[0][0].numpy()
    assert np.abs(full - golden_io["ref32_full_out0"][0]).max() <= TOL32[0]
    for k in range(6):
        got = out[k].numpy()[:, :, SUB, SUB]
        ref = golden_io[f"ref32_sub_out{k}"]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= TOL32[k], NAMES[k]


@pytest.mark.parametrize("character", ["lambda_00", "lambda_01"])
def test_torch_restatement_matches_reference_fp64(character, char_weights, char_io):
    golden_weights, golden_io = char_weights[character], char_io[character]
    # the reference's fp64 run keeps fp32 position grids (default-dtype identity theta) and, through
    # GridChangeApplier's cache, an fp32 warp base grid; the oracle mirrors the former, hence 5e-5 on
    # the two warp-dependent outputs and storage rounding (fixtures hold ref64 rounded to fp32) elsewhere
    out = so.student_forward_torch(golden_weights, golden_io["image_f32"], golden_io["poses"][:2], "float64")
    tol = [5e-5, 2e-7, 2e-7, 5e-5, 2e-7, 2e-7]
    for k in range(6):
        got = out[k].numpy()[:, :, SUB, SUB]
        ref = golden_io[f"ref64_sub_out{k}"][:2]
        assert np.abs(got - ref).max() <= tol[k], NAMES[k]


def test_numpy_restatement_matches_reference_fp64(golden_weights, golden_io):
    so.use_aten_positions(True)
    try:
        out = so.student_forward_numpy(golden_weights, golden_io["image_f32"], golden_io["poses"][0])
    finally:
        so.use_aten_positions(False)
    tol = [5e-5, 2e-7, 2e-7, 5e-5, 2e-7, 2e-7]
    for k in range(6):
        assert np.abs(out[k][:, SUB, SUB] - golden_io[f"ref64_sub_out{k}"][0]).max() <= tol[k], NAMES[k]


def test_exact_positions_shift_output_by_1e4(golden_weights, golden_io):
    """Documents WHY the C ABI accepts position axes: exact dyadic positions vs ATen's fp32 table
    move the final image by ~1e-4 (SIREN gain), still well inside the 1e-3 budget."""
    out = so.student_forward_numpy(golden_weights, golden_io["image_f32"], golden_io["poses"][0])
    d = np.abs(out[0][:, SUB, SUB] - golden_io["ref64_sub_out0"][0]).max()
    assert 1e-6 < d < 5e-4


def test_upsample_and_warp_closed_forms_match_aten():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 8, 8))
    up = F.interpolate(torch.from_numpy(x)[None], size=(16, 16), mode="bilinear")[0].numpy()
    assert np.abs(so.upsample2x_numpy(x) - up).max() < 1e-14
    img = rng.standard_normal((4, 16, 16))
    gx = rng.uniform(-1.3, 1.3, (16, 16))
    gy = rng.uniform(-1.3, 1.3, (16, 16))
    grid = torch.from_numpy(np.stack([gx, gy], -1))[None]
    ref = F.grid_sample(torch.from_numpy(img)[None], grid, mode="bilinear", padding_mode="border",
                        align_corners=False)[0].numpy()
    assert np.abs(so.grid_sample_border_numpy(img, gx, gy) - ref).max() < 1e-13


def test_restructured_intermediates_equal_reference_order(golden_weights, golden_io):
    """Pose folding + commuting the x2 upsample with the next level's first layer (what the HIP
    kernels do) is algebraically the reference's computation: fp64 agreement to 1e-11."""
    pose = golden_io["poses"][1]
    it = so.student_intermediates(golden_weights, pose)
    face = so.face_forward_numpy(golden_weights, pose[:39].astype(np.float64))
    body = so.body_forward_numpy(golden_weights, so.paste_face(golden_io["image_f32"].astype(np.float64), face),
                                 pose.astype(np.float64))
    assert np.abs(it["siren_out"][0:2] - body[4]).max() < 1e-11
    assert np.abs(it["siren_out"][2:3] - body[1]).max() < 1e-11
    assert np.abs(it["siren_out"][3:7] - body[2]).max() < 1e-11


def test_random_pose_ranges():
    p = so.random_poses(64, seed=7)
    assert p.shape == (64, 45) and p.dtype == np.float32
    assert (p[:, :37] >= 0).all() and (p[:, :37] < 1).all()
    assert (p[:, 37:44] >= -1).all() and (p[:, 37:44] < 1).all() and (p[:, 37:44] < 0).any()
    assert (p[:, 44] >= 0).all()


@pytest.mark.parametrize("character", ["lambda_00", "lambda_01"])
def test_oracle_matches_the_pinned_sweep_and_edge_poses(character, char_weights, char_io):
    """The round-4 fixtures (tests/golden/make_golden_sweep.py): the four edge poses of the GPU suite's test_edge_poses and poses of
    the 64-pose sweep that round 3 had NOT pinned for lambda_01 (indices >= 16) - the oracle equals the unmodified reference on
    them too (stride-8 pixel subsets)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"student_{character}_sweep.npz"))
    assert z["ref32_sub8_out0"].shape[0] == 64 and z["ref32_sub8_edge_out0"].shape[0] == 4
    w, image = char_weights[character], char_io[character]["image_f32"]
    out = so.student_forward_torch(w, image, z["edge_poses"], "float32")[0].numpy()
    assert np.abs(out[:, :, 3::8, 3::8] - z["ref32_sub8_edge_out0"]).max() <= TOL32[0]
    idx = [17, 40, 63]
    out = so.student_forward_torch(w, image, z["poses"][idx], "float32")[0].numpy()
    assert np.abs(out[:, :, 3::8, 3::8] - z["ref32_sub8_out0"][idx]).max() <= TOL32[0]


def test_local_affine_grid_table_against_the_fixture_build():
    """The kernels take the LOCAL torch build's fp32 `affine_grid` axes (`match_aten_positions=True`: they track "the reference on
    this machine"); the committed reference frames were made with the torch build recorded in the sweep fixture.  A different
    local table moves up to ~1e-4 of the 1e-3 budget (test_exact_positions_shift_output_by_1e4): never silently - this test FAILS
    with both versions and the number of differing entries (round-4 review: a warning is not a gate), unless the environment says
    the mismatch is known and accepted (THA4_ACCEPT_AFFINE_GRID_MISMATCH=1: the reference frames then carry that much less margin);
    it also fails if an axis is outside what fp32 linspace rounding can produce (more than 1 ulp from the exact dyadic grid)."""
    import os
    import warnings
    import torch
    import torch.nn.functional as F
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "student_lambda_00_sweep.npz"))
    ident = torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]])
    differing = {}
    for sz in (128, 256, 512):
        local = F.affine_grid(ident, [1, 1, sz, sz], align_corners=False)[0, 0, :, 0].numpy()
        exact = ((2 * np.arange(sz) + 1) / sz - 1).astype(np.float64)
        assert np.abs(local.astype(np.float64) - exact).max() <= np.spacing(np.float32(1.0)), f"affine_grid axis {sz} is not an fp32 rounding of the dyadic grid"
        n = int((local != z[f"aten_axis{sz}"]).sum())
        if n:
            differing[sz] = n
    if differing:
        msg = (f"local torch {torch.__version__} produces a different fp32 affine_grid table than the fixtures' torch "
               f"{str(z['torch_version'])}: differing entries {differing} - the kernels follow the LOCAL table, the committed reference "
               f"frames the fixture's (up to ~1e-4 of the 1e-3 budget); regenerate the fixtures (tests/golden/make_golden*.py) or set "
               f"THA4_ACCEPT_AFFINE_GRID_MISMATCH=1")
        if os.environ.get("THA4_ACCEPT_AFFINE_GRID_MISMATCH") == "1":
            warnings.warn(msg, RuntimeWarning)
        else:
            pytest.fail(msg)
