"""GPU parity of the full THA4 system (mode_07) through the C ABI / Poser mirror against the CPU oracle
(pinned to the reference by test_full_oracle_golden.py) and the committed reference fixtures, using the
deterministic synthetic weights (the reference ships no full-model weights).  Gate: the posed frame
(output 0) within 1e-3 max-abs per channel of the reference fp32 CPU path."""
import os

import numpy as np
import pytest
import torch

import tha4_amd  # noqa: F401
from oracle import full_oracle as fo
from tha4_amd.poser.modes import mode_07

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUB = slice(1, None, 3)
TOL = 1e-3


@pytest.fixture(scope="module")
def full_io():
    z = np.load(os.path.join(GOLDEN, "full_synth_io.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def weights(full_io):
    return fo.synth_full_weights(int(full_io["seed"]))


@pytest.fixture(scope="module")
def poser(weights):
    p = mode_07.create_poser_from_state_dicts(torch.device("cuda:0"), weights, max_batch=2)
    p.get_modules()
    assert p._handle is not None
    return p


def test_all_33_outputs_vs_reference_fixture(poser, full_io, golden_io):
    dev = torch.device("cuda:0")
    image = torch.from_numpy(golden_io["image_f32"]).to(dev)
    report = []
    for i in range(2):
        outs = poser.get_posing_outputs(image, torch.from_numpy(full_io["poses"][i]).to(dev), image_changed=(i == 0))
        assert len(outs) == 33
        for k in range(33):
            got = outs[k][0].cpu().numpy()
            assert np.isfinite(got).all(), fo.OUTPUT_NAMES[k]
            err = float(np.abs(got[:, SUB, SUB] - full_io[f"ref32_sub_out{k}"][i]).max())
            report.append((i, fo.OUTPUT_NAMES[k], err))
    bad = [r for r in report if r[2] > TOL]
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/full_parity_report.txt", "w") as fh:
        fh.write("\n".join(f"pose {i} {n:20s} {e:.3e}" for i, n, e in report) + "\n")
    assert not bad, bad
    out0 = poser.pose(image, torch.from_numpy(full_io["poses"][0]).to(dev))[0].cpu().numpy()
    assert np.abs(out0 - full_io["ref32_full_out0"][0]).max() <= TOL


def test_full_vs_oracle_and_cache_and_batch(poser, weights, full_io, golden_io):
    dev = torch.device("cuda:0")
    image = torch.from_numpy(golden_io["image_f32"]).to(dev)
    poses = torch.from_numpy(full_io["poses"][2:4]).to(dev)
    ref = fo.full_forward_torch(weights, golden_io["image_f32"], full_io["poses"][2:4], "float32")
    cold = poser.pose(image, poses[0], image_changed=True)          # decomposer runs
    warm = poser.pose(image, poses[0])                               # decomposer output reused (mode_07.py:56-67)
    assert torch.equal(cold, warm)
    assert np.abs(cold[0].cpu().numpy() - ref[0][0].numpy()).max() <= TOL
    both = poser.pose(image, poses)                                  # batch of 2, one shared image
    assert both.shape == (2, 4, 512, 512)
    assert np.abs(both.cpu().numpy() - ref[0].numpy()).max() <= TOL
    assert torch.equal(both[0], cold[0])                             # a frame's bytes do not depend on the batch
    five = poser.pose(image, poses[1], 5)
    assert np.abs(five[0].cpu().numpy() - ref[5][1].numpy()).max() <= TOL
