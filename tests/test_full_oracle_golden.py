"""Pin oracle/full_oracle.py (mode_07 restatement) against the unmodified reference run with the
synthetic weights (tests/golden/full_synth_io.npz from tests/golden/make_golden_full.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import full_oracle as fo

SUB = slice(1, None, 3)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def full_io():
    z = np.load(os.path.join(GOLDEN, "full_synth_io.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def full_weights(full_io):
    return fo.synth_full_weights(int(full_io["seed"]))


def test_param_inventory_matches_survey():
    shapes = fo.full_param_shapes()
    count = {n: sum(int(np.prod(s)) for s in d.values()) for n, d in shapes.items()}
    # SURVEY.md Appendix A [measured on the reference modules]
    assert count == {"eyebrow_decomposer": 31479434, "eyebrow_morphing_combiner": 31535878, "face_morpher": 31605002,
                     "body_morpher": 34682119, "upscaler": 35015655}
    assert [len(d) for d in shapes.values()] == [62, 61, 63, 398, 466]


def test_full_forward_matches_reference_fp32(full_weights, full_io, golden_io):
    outs = fo.full_forward_torch(full_weights, golden_io["image_f32"], full_io["poses"][:2], "float32")
    assert len(outs) == 33
    assert np.abs(outs[0][0].numpy() - full_io["ref32_full_out0"][0]).max() < 5e-4
    for k in range(33):
        got = outs[k].numpy()[:, :, SUB, SUB]
        ref = full_io[f"ref32_sub_out{k}"]
        assert got.shape == ref.shape, fo.OUTPUT_NAMES[k]
        assert np.abs(got - ref).max() < 1e-3, fo.OUTPUT_NAMES[k]      # same ATen kernels: normally exact


def test_full_forward_matches_reference_fp64(full_weights, full_io, golden_io):
    outs = fo.full_forward_torch(full_weights, golden_io["image_f32"], full_io["poses"][:1], "float64")
    for k in range(33):
        got = outs[k].numpy()[:, :, SUB, SUB]
        ref = full_io[f"ref64_sub_out{k}"]
        # fixtures store ref64 rounded to fp32; the reference's GridChangeApplier caches the fp32 identity of
        # the earlier fp32 run (image_processing_util.py:38-49), so its fp64 warps carry an fp32 base grid
        tol = 5e-5 if fo.OUTPUT_NAMES[k] in ("up_merged", "up_warped", "body_merged", "body_warped") else 2e-7
        assert np.abs(got - ref).max() < tol, fo.OUTPUT_NAMES[k]
