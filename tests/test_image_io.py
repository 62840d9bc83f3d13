"""§8f rows: image ingest, display epilogue, CharacterModel loader."""
import os

import numpy as np
import pytest
import torch

import tha4_amd  # noqa: F401
from oracle import image_oracle as io_oracle
from tha4_amd.charmodel.character_model import CharacterModel


def test_ingest_oracle_matches_reference_fixture(golden_io):
    got = io_oracle.ingest_rgba8_numpy(golden_io["image_rgba8"])
    assert got.shape == (4, 512, 512)
    assert np.abs(got - golden_io["image_f32"]).max() <= 1e-6       # same numpy ops as the reference


def test_character_model_yaml_roundtrip(tmp_path):
    d = tmp_path / "chars" / "x"
    cm = CharacterModel(str(d / "character.png"), str(d / "face_morpher.pt"), str(d / "body_morpher.pt"))
    y = str(d / "character_model.yaml")
    cm.save(y)
    text = open(y).read()
    # the format the reference ships (data/character_models/lambda_00/character_model.yaml)
    assert text.splitlines() == ["character_image_file_name: character.png", "face_morpher_file_name: face_morpher.pt",
                                 "body_morpher_file_name: body_morpher.pt"]
    back = CharacterModel.load(y)
    assert back.face_morpher_file_name == str(d / "face_morpher.pt")
    assert back.character_image_file_name == str(d / "character.png")
    poser = back.get_poser(torch.device("cpu"))          # lazy: nothing is read until the first pose()
    assert poser.get_num_parameters() == 45
    assert back.get_poser(torch.device("cpu")) is poser


@pytest.mark.skipif(not os.path.isdir("/root/reference/data/character_models/lambda_00"), reason="reference checkout absent")
def test_character_model_reads_reference_yaml():
    cm = CharacterModel.load("/root/reference/data/character_models/lambda_00/character_model.yaml")
    assert cm.body_morpher_file_name.endswith("lambda_00/body_morpher.pt") and os.path.exists(cm.body_morpher_file_name)
    assert os.path.exists(cm.character_image_file_name)


@pytest.mark.gpu
def test_ingest_and_display_on_gpu(golden_io):
    from tha4_amd import image_io
    dev = torch.device("cuda:0")
    rgba = torch.from_numpy(golden_io["image_rgba8"]).to(dev)
    img = image_io.image_from_rgba8(rgba)
    assert img.shape == (4, 512, 512)
    assert np.abs(img.cpu().numpy() - golden_io["image_f32"]).max() <= 2e-6       # reference tensor (fp32 pow ulp)
    batch = image_io.image_from_rgba8(torch.stack([rgba, rgba.flip(0)]))
    assert torch.equal(batch[0], img)
    # display epilogue on a real posed frame and on a synthetic one that hits both sRGB branches and the clips
    frames = torch.from_numpy(golden_io["ref32_full_out0"]).to(dev)
    rng = np.random.default_rng(0)
    synth = torch.from_numpy(rng.uniform(-1.2, 1.2, (1, 4, 64, 48)).astype(np.float32)).to(dev)
    for f in (frames, synth):
        for bg in (None, (0.0, 1.0, 0.0), (1.0, 1.0, 1.0)):
            got = image_io.to_display_rgba8(f, bg)[0].cpu().numpy().astype(np.int32)
            ref = io_oracle.display_rgba8_torch(f[0].cpu(), bg).numpy().astype(np.int32)
            diff = np.abs(got - ref)
            assert diff.max() <= 1                      # uint8 truncation of a pow() that may differ by an ulp
            assert (diff > 0).mean() < 2e-3
    assert image_io.to_display_rgba8(frames[0]).shape == (512, 512, 4)
