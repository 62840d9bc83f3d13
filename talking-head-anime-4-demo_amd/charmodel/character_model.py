"""Mirror of the reference ``tha4.charmodel.character_model.CharacterModel``
(src/tha4/charmodel/character_model.py:11-69): a YAML file with three paths relative to its own directory
(character image, face_morpher.pt, body_morpher.pt; written by distiller_config.py:273-299) -> a student
poser (mode_14) + the character image tensor.  The reference parses the YAML with OmegaConf; the file is
a flat mapping, so plain ``yaml`` reads/writes the same format.
"""
from __future__ import annotations

import os.path

import torch
import yaml

from .. import image_io
from ..poser.modes.mode_14 import KEY_BODY_MORPHER, KEY_FACE_MORPHER, create_poser


class CharacterModel:
    def __init__(self, character_image_file_name: str, face_morpher_file_name: str, body_morpher_file_name: str):
        self.body_morpher_file_name = body_morpher_file_name
        self.face_morpher_file_name = face_morpher_file_name
        self.character_image_file_name = character_image_file_name
        self.poser = None
        self.character_image = None

    def get_poser(self, device: torch.device):
        if self.poser is not None:
            self.poser.to(device)
        else:
            self.poser = create_poser(device, module_file_names={
                KEY_FACE_MORPHER: self.face_morpher_file_name,
                KEY_BODY_MORPHER: self.body_morpher_file_name})
        return self.poser

    def load_into(self, poser):
        """Extension: swap THIS character into an existing student poser without re-allocating anything on the device
        (the puppeteers' "load model" action re-creates the poser, character_model_ifacialmocap_puppeteer.py:383-399;
        with several characters served from one process the handle and workspace are reused instead)."""
        poser.load_character({KEY_FACE_MORPHER: self.face_morpher_file_name, KEY_BODY_MORPHER: self.body_morpher_file_name})
        self.poser = poser
        return poser

    def get_character_image(self, device: torch.device):
        if self.character_image is None:
            import PIL.Image
            pil_image = PIL.Image.open(self.character_image_file_name)
            self.character_image = image_io.image_from_pil(pil_image, device)
        self.character_image = self.character_image.to(device)
        return self.character_image

    def save(self, file_name: str):
        d = os.path.dirname(file_name)
        data = {
            "character_image_file_name": os.path.relpath(self.character_image_file_name, d),
            "face_morpher_file_name": os.path.relpath(self.face_morpher_file_name, d),
            "body_morpher_file_name": os.path.relpath(self.body_morpher_file_name, d),
        }
        os.makedirs(d, exist_ok=True)
        with open(file_name, "wt") as fout:
            yaml.safe_dump(data, fout, default_flow_style=False, sort_keys=False)

    @staticmethod
    def load(file_name: str) -> "CharacterModel":
        with open(file_name, "rt") as fin:
            conf = yaml.safe_load(fin)
        d = os.path.dirname(file_name)
        return CharacterModel(os.path.join(d, conf["character_image_file_name"]),
                              os.path.join(d, conf["face_morpher_file_name"]),
                              os.path.join(d, conf["body_morpher_file_name"]))
