"""`.pt` ingest: tha4_amd.weights.load_state_dict_file mirrors the reference's torch_load
(src/tha4/shion/core/load_save.py:12-14).  CPU only.

On the build box the shipped reference checkpoints are read through it and must equal the committed fixtures (which were
produced from the reference MODULES' state_dict(), tests/golden/make_golden.py) - that pins key names, kernel shapes and
values of the real files.  Everywhere (GPU box included) a round trip through torch.save covers the same code path."""
import collections
import os

import numpy as np
import pytest
import torch

import tha4_amd  # noqa: F401
from tha4_amd import _capi, weights
from tha4_amd.poser.modes import mode_14

REF_CM = "/root/reference/data/character_models"


@pytest.mark.parametrize("character", ["lambda_00", "lambda_01"])
@pytest.mark.skipif(not os.path.isdir(REF_CM), reason="reference checkout not present (GPU box)")
def test_reference_pt_files_equal_committed_fixtures(character, char_weights):
    face = weights.load_state_dict_file(os.path.join(REF_CM, character, "face_morpher.pt"))
    body = weights.load_state_dict_file(os.path.join(REF_CM, character, "body_morpher.pt"))
    gf, gb = weights.split_flat_weights(char_weights[character])
    assert list(face) == list(gf) and list(body) == list(gb)               # same keys, same order (Appendix B)
    assert face["siren.sine_layers.0.linear.weight"].shape == (128, 41, 1, 1)
    assert body["siren_layers.1.0.linear.weight"].shape == (180, 227, 1, 1) and body["last_linear.weight"].shape == (7, 90, 1, 1)
    for sd, g in ((face, gf), (body, gb)):
        for k, v in sd.items():
            assert v.dtype == np.float32
            assert np.array_equal(v.reshape(g[k].shape), g[k]), k
    # ... and the C struct built from the raw file is the one built from the fixture
    a, keep_a = _capi.build_student_weights(face, body)
    b, keep_b = _capi.build_student_weights(gf, gb)
    for la, lb in ((a.face_sine[0], b.face_sine[0]), (a.body_sine[2][1], b.body_sine[2][1]), (a.body_last, b.body_last)):
        assert (la.out_ch, la.in_ch) == (lb.out_ch, lb.in_ch)
        n = la.out_ch * la.in_ch
        assert np.array_equal(np.ctypeslib.as_array(la.weight, (n,)), np.ctypeslib.as_array(lb.weight, (n,)))


@pytest.mark.skipif(not os.path.isdir(REF_CM), reason="reference checkout not present (GPU box)")
def test_create_poser_lazy_loaders_read_the_reference_files():
    """mode_14.create_poser(device, module_file_names) builds loaders only (general_poser_02.py:41-49); the loaders
    return the file's state_dict.  (Device work needs a GPU: tests/test_student_gpu.py.)"""
    files = {"face_morpher": os.path.join(REF_CM, "lambda_01", "face_morpher.pt"),
             "body_morpher": os.path.join(REF_CM, "lambda_01", "body_morpher.pt")}
    p = mode_14.create_poser(torch.device("cpu"), module_file_names=files)
    assert p._state_dicts is None
    sd = {k: f() for k, f in p.state_dict_loaders.items()}
    assert set(sd) == {"face_morpher", "body_morpher"} and len(sd["face_morpher"]) == 18 and len(sd["body_morpher"]) == 20


def test_torch_save_roundtrip_any_box(golden_weights, tmp_path):
    face, body = weights.split_flat_weights(golden_weights)
    od = collections.OrderedDict((k, torch.from_numpy(v.reshape(v.shape + (1, 1)) if v.ndim == 2 else v)) for k, v in body.items())
    f = str(tmp_path / "body_morpher.pt")
    torch.save(od, f)
    back = mode_14.load_body_morpher(f)
    assert list(back) == list(body)
    for k in body:
        assert np.array_equal(back[k].reshape(body[k].shape), body[k])
    # half / double checkpoints are converted to fp32 like module.load_state_dict would
    torch.save(collections.OrderedDict((k, v.double()) for k, v in od.items()), f)
    assert all(v.dtype == np.float32 for v in weights.load_state_dict_file(f).values())
    with pytest.raises(FileNotFoundError):
        weights.load_state_dict_file(str(tmp_path / "missing.pt"))
