#!/usr/bin/env python3
"""Batch-8 full model as TWO half-batches on two streams (GPU box): does staggering the kernel boundaries of two independent
4-frame schedules (one's store tail / load head under the other's K loops) beat one 8-frame schedule?  Pure host-side experiment:
two handles (max_batch 4 each) on two streams against one handle (max_batch 8)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa
from tha4_amd.poser.modes import mode_07
from tha4_amd import synthetic as fo
dev = torch.device("cuda:0")
w = fo.synth_full_weights()
B = 8
imgs = [torch.from_numpy(fo.random_rgba_images(B, seed=99 + j)).to(dev) for j in range(2)]
g = torch.Generator().manual_seed(77)
lo = torch.tensor([0.0] * 37 + [-1.0] * 7 + [0.0]); hi = torch.ones(45)
poses = (lo + (hi - lo) * torch.rand(16, B, 45, generator=g)).to(dev)
N = 10


def timed(step):
    for i in range(3): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): step(i)
    torch.cuda.synchronize()
    return N * B / (time.perf_counter() - t0)


one = mode_07.create_poser_from_state_dicts(dev, w, max_batch=B)
print(f"one handle, max_batch 8, one stream: {timed(lambda i: one.pose(imgs[i % 2], poses[i % 16])):.2f} fps", flush=True)
ref = one.pose(imgs[0], poses[0]).clone()
one.free()
for parts in (2, 4):
    n = B // parts
    hs = [mode_07.create_poser_from_state_dicts(dev, w, max_batch=n) for _ in range(parts)]
    ss = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    outs = [None] * parts

    def step(i):
        cur = torch.cuda.current_stream(dev)
        for k in range(parts):
            ss[k].wait_stream(cur)
            with torch.cuda.stream(ss[k]):
                outs[k] = hs[k].pose(imgs[i % 2][k * n:(k + 1) * n], poses[i % 16][k * n:(k + 1) * n])
        for k in range(parts):
            cur.wait_stream(ss[k])
    print(f"{parts} handles, max_batch {n} each, {parts} streams: {timed(step):.2f} fps", flush=True)
    step(0)
    torch.cuda.synchronize()
    got = torch.cat(outs)
    print(f"   max |split - one handle| = {float((got - ref).abs().max()):.2e} (different launch plans)")
    # same handles, ONE stream (control: the plan of max_batch n without the overlap)
    def step1(i):
        for k in range(parts):
            outs[k] = hs[k].pose(imgs[i % 2][k * n:(k + 1) * n], poses[i % 16][k * n:(k + 1) * n])
    print(f"{parts} handles, max_batch {n} each, ONE stream (control): {timed(step1):.2f} fps", flush=True)
    for h in hs: h.free()
