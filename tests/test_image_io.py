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


BACKGROUNDS = {"none": None, "green": (0.0, 1.0, 0.0), "blue": (0.0, 0.0, 1.0), "black": (0.0, 0.0, 0.0), "white": (1.0, 1.0, 1.0)}


@pytest.fixture(scope="module")
def display_io():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "display_io.npz"))
    return {k: z[k] for k in z.files}


def test_ingest_oracle_matches_reference_fixture_lambda_01(char_io):
    io = char_io["lambda_01"]
    assert np.abs(io_oracle.ingest_rgba8_numpy(io["image_rgba8"]) - io["image_f32"]).max() <= 1e-6


def test_display_oracle_pinned_to_reference(display_io, golden_io):
    """oracle.display_rgba8_torch == the reference's convert_linear_to_srgb inside the puppeteer sequence
    (tests/golden/make_golden_display.py).  Same torch ops: identical bytes on the machine that made the fixture; a
    different libm may move a pow() by an ulp across a uint8 truncation boundary, hence <= 1 LSB on < 0.1 % elsewhere."""
    frames = {"posed": torch.from_numpy(golden_io["ref32_full_out0"][0]), "synth": torch.from_numpy(display_io["synth_f32"])}
    for fname, f in frames.items():
        for bname, bg in BACKGROUNDS.items():
            got = io_oracle.display_rgba8_torch(f, bg).numpy().astype(np.int32)
            ref = display_io[f"{fname}_{bname}"].astype(np.int32)
            assert got.shape == ref.shape
            d = np.abs(got - ref)
            assert d.max() <= 1 and (d > 0).mean() < 1e-3, (fname, bname, d.max(), (d > 0).mean())


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout absent")
def test_resize_pil_image_equals_reference(golden_io):
    import sys
    import PIL.Image
    from tha4_amd import image_io
    sys.path.insert(0, "/root/reference/src")
    try:
        from tha4.image_util import resize_PIL_image as ref_resize
    finally:
        sys.path.remove("/root/reference/src")
    src = PIL.Image.fromarray(golden_io["image_rgba8"], "RGBA").crop((0, 40, 512, 400))     # 512 x 360: not square
    for size in ((512, 512), (256, 256)):
        assert np.array_equal(np.asarray(image_io.resize_PIL_image(src, size)), np.asarray(ref_resize(src, size)))


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


@pytest.mark.gpu
def test_display_on_gpu_vs_reference_fixture(display_io, golden_io):
    """tha4_display_rgba8 against bytes produced by the unmodified reference (five backgrounds, knee and clip pixels)."""
    from tha4_amd import image_io
    dev = torch.device("cuda:0")
    frames = {"posed": torch.from_numpy(golden_io["ref32_full_out0"][0]), "synth": torch.from_numpy(display_io["synth_f32"])}
    for fname, f in frames.items():
        for bname, bg in BACKGROUNDS.items():
            got = image_io.to_display_rgba8(f.to(dev), bg).cpu().numpy().astype(np.int32)
            ref = display_io[f"{fname}_{bname}"].astype(np.int32)
            d = np.abs(got - ref)
            assert d.max() <= 1 and (d > 0).mean() < 2e-3, (fname, bname, d.max(), (d > 0).mean())
