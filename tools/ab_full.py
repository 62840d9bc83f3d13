#!/usr/bin/env python3
"""Same-box A/B of builds of libtha4_hip.so on the full model (GPU box): for every library given, ONE process measures
batch-1 steady / cold frames/s and batch-8 frames/s (tools/time_full.py three times would pay import + weight synthesis thrice).

  python tools/ab_full.py [--rounds R] [--no-b8] [--no-b1] name=path.so[@VAR=v,VAR2=v2] ...      ("default" = the shipped library;
                                                                 the optional @ list is added to the child's environment)
  python tools/ab_full.py --one                                         (child mode: THA4_HIP_LIB selects the library)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import tha4_amd  # noqa
    from tha4_amd.poser.modes import mode_07
    from tha4_amd import synthetic as fo
    dev = torch.device("cuda:0")
    w = fo.synth_full_weights()
    out = []
    if "--no-b1" not in sys.argv:
        p = mode_07.create_poser_from_state_dicts(dev, w)
        io = np.load(os.path.join(ROOT, "tests/golden/student_lambda_00_io.npz"))
        image = torch.from_numpy(io["image_f32"]).to(dev)
        poses = torch.from_numpy(io["poses"]).to(dev)
        for i in range(3):
            p.pose(image, poses[i % 8])
        torch.cuda.synchronize()
        for name, changed, n in (("steady", False, 60), ("cold", True, 30)):
            t0 = time.perf_counter()
            for i in range(n):
                p.pose(image, poses[i % 8], image_changed=changed)
            t1 = time.perf_counter()                       # host done submitting (queue depth permitting)
            torch.cuda.synchronize()
            out.append(f"{name} {n / (time.perf_counter() - t0):.2f}" + (f" (host submit {1e3 * (t1 - t0) / n:.2f} ms/frame)" if name == "steady" else ""))
        del p
    if "--no-b8" not in sys.argv:
        B = 8
        p = mode_07.create_poser_from_state_dicts(dev, w, max_batch=B)
        imgs = [torch.from_numpy(fo.random_rgba_images(B, seed=99 + j)).to(dev) for j in range(2)]
        g = torch.Generator().manual_seed(77)
        lo = torch.tensor([0.0] * 37 + [-1.0] * 7 + [0.0])
        hi = torch.ones(45)
        pb = (lo + (hi - lo) * torch.rand(16, B, 45, generator=g)).to(dev)
        for i in range(3):
            p.pose(imgs[i % 2], pb[i])
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for i in range(n):
            p.pose(imgs[i % 2], pb[i % 16])
        torch.cuda.synchronize()
        out.append(f"b8 {n * B / (time.perf_counter() - t0):.2f}")
    print("AB " + "  ".join(out), flush=True)


def main():
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 1
    libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
    extra = [f for f in ("--no-b8", "--no-b1") if f in sys.argv]
    for r in range(rounds):
        for name, path in libs:
            env = dict(os.environ)
            env.pop("THA4_HIP_LIB", None)
            if "@" in path:
                path, evs = path.split("@", 1)
                env.update(dict(kv.split("=", 1) for kv in evs.split(",") if kv))
            if path != "default":
                env["THA4_HIP_LIB"] = os.path.join(ROOT, path)
            try:
                res = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"] + extra, env=env, capture_output=True, text=True, timeout=240)
                line = [l for l in res.stdout.splitlines() if l.startswith("AB ")]
                print(f"{name:16s} {line[-1][3:] if line else 'FAILED ' + res.stderr[-300:]}", flush=True)
            except subprocess.TimeoutExpired:
                print(f"{name:16s} TIMEOUT", flush=True)


if __name__ == "__main__":
    one() if "--one" in sys.argv else main()
