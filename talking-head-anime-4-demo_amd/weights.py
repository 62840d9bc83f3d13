"""Weight ingest for the student path: the reference's ``torch.save``-d state_dicts.

Reference: src/tha4/shion/core/load_save.py:12-14 (``torch.load`` onto CPU storage) and the key
layout of SURVEY.md Appendix B.  Missing files raise ``FileNotFoundError`` exactly like the
reference's ``open()`` does.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def _to_numpy_sd(sd) -> Dict[str, np.ndarray]:
    out = {}
    for k, v in sd.items():
        out[k] = v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
    return out


def load_state_dict_file(file_name: str) -> Dict[str, np.ndarray]:
    """``torch_load`` equivalent returning fp32 numpy arrays keyed like the state_dict."""
    import torch
    with open(file_name, "rb") as f:
        sd = torch.load(f, map_location=lambda storage, loc: storage)
    return _to_numpy_sd(sd)


def split_flat_weights(flat: Dict[str, np.ndarray]):
    """Split a flat ``{'face.<key>': a, 'body.<key>': a}`` dict (tests/golden/*.npz, oracle
    ``random_student_weights``) into the two state_dicts."""
    face = {k[5:]: v for k, v in flat.items() if k.startswith("face.")}
    body = {k[5:]: v for k, v in flat.items() if k.startswith("body.")}
    return face, body
