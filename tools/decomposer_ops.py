#!/usr/bin/env python3
"""Per-op HIP-event times of the eyebrow decomposer (network 0) under the split / "outer" mixed / whole-decomposer-exact plans (GPU box; round 6).
What does the mixed default plan pay for, launch by launch?     python tools/decomposer_ops.py [--batch 8]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa
from tha4_amd.poser.modes import mode_07
from tha4_amd import synthetic as fo


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    w = fo.synth_full_weights()
    B = a.batch
    imgs = torch.from_numpy(fo.random_rgba_images(max(B, 1), seed=99)).to(dev)
    g = torch.Generator().manual_seed(77)
    lo = torch.tensor([0.0] * 37 + [-1.0] * 7 + [0.0])
    pose = (lo + (torch.ones(45) - lo) * torch.rand(B, 45, generator=g)).to(dev)
    table = {}
    for mode in (False, "outer", "all"):
        p = mode_07.create_poser_from_state_dicts(dev, w, max_batch=B, exact_decomposer=mode)
        x = imgs if B > 1 else imgs[0]
        q = pose if B > 1 else pose[0]
        for _ in range(3):
            p.pose(x, q, image_changed=True)
        p.set_timing(True)
        acc = None
        for _ in range(5):
            p.pose(x, q, image_changed=True)
            ms = np.array(p.last_op_ms())
            acc = ms if acc is None else acc + ms
        info = p.op_info()
        nd = next(i for i, (l, _) in enumerate(info) if "pose padding" in l)          # decomposer ops come first
        table[mode] = [(info[i][0], acc[i] / 5 * 1e3) for i in range(nd)]
        print(f"== exact_decomposer={mode}: decomposer {sum(t for _, t in table[mode]):.1f} us over {nd} ops, whole call {acc.sum() / 5:.3f} ms (event-timed)")
        p.free()
    for mode, rows in table.items():
        print(f"-- {mode}")
        for l, t in rows:
            print(f"   {t:8.1f} us  {l}")


if __name__ == "__main__":
    main()
