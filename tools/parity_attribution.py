#!/usr/bin/env python3
"""Which convolutions carry the 22-bit operand split's share of the posed frame's error? (round-5 review, task 4; CPU only)

The default plan of the full model contracts fp16 hi/lo halves of both operands (22 significant bits, products exact, fp32 accumulate).  On the
mid-gain parameter set's dense batch of 8 random images the posed frame (`up_merged`) sits at 1.0e-3 from the reference's fp32 run where the
exact-fp32 plan sits at 7.0e-4 and the reference's own fp32-vs-fp64 distance is 6.6e-4.  This tool runs the fp64 oracle (test infrastructure)
on that batch and rounds the OPERANDS of one class of convolutions at a time exactly as the kernels stage them -

    weights      W' = S W, S the power of two with max |W'| in [2^13, 2^14); hi = fp16(W'), lo = fp16(W' - hi)      (full_layout.h pack_conv_weight16)
    activations  hi = fp16(v), lo = fp16(v - hi), unscaled                                                           (full_conv16_kernels.h staging)

- while every sum stays in fp64: the distance of the outputs from the unperturbed fp64 run is that class's contribution, free of the
accumulation-order noise both plans share.  Classes: the three encoder-decoder networks together, and per U-Net the small maps (<= 32x32: the
large-K 3x3 / 1x1 layers), the middle maps (64x64, 128x128), the large maps (>= 256x256), the `last` convolution (behind the final GroupNorm).

    python tools/parity_attribution.py [--threads 6] [--out profiles/parity_r06/split_attribution.txt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import full_oracle as fo                          # noqa: E402
from oracle.student_oracle import synthetic_image             # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
STATE = {"net": None, "active": set(), "seen": {}, "fine": False, "which": "both"}


def split22(t, scaled, pre=1.0):
    """v -> hi + lo as the kernels stage it (fp32 -> two fp16 halves), returned in the dtype of t.  pre: power of two applied before the split
    (activations: what-if for a pre-scaled staging that keeps the low halves of small activations out of the fp16 subnormal range)."""
    x = t.to(torch.float32)
    s = pre
    x = x * s
    if scaled:
        mx = float(x.abs().max())
        if mx > 0.0:
            s = 2.0 ** np.floor(np.log2(16384.0 / mx))
            x = x * s
    hi = x.to(torch.float16)
    lo = (x - hi.to(torch.float32)).to(torch.float16)
    return ((hi.to(torch.float64) + lo.to(torch.float64)) / s).to(t.dtype)


def conv_class(net, x, w, transposed):
    h = x.shape[-1] * (2 if transposed else 1)
    if net in ("dec", "comb", "face"):
        if not STATE["fine"]:
            return "encdec (decomposer + combiner + face morpher)"
        cin = w.shape[0] if transposed else w.shape[1]
        co = w.shape[1] if transposed else w.shape[0]
        role = ("first conv" if cin <= 8 else "heads" if co <= 4 else "up convs (transposed 4x4)" if transposed else
                "bottleneck 3x3" if w.shape[-1] == 3 else "down convs (4x4 stride 2)")
        return f"{net}: {role}"
    cout = w.shape[1] if transposed else w.shape[0]
    if cout == 7:
        return f"{net}: last convolution"
    if h <= 32:
        return f"{net}: maps <= 32x32"
    if h <= 128:
        return f"{net}: maps 64x64 .. 128x128"
    return f"{net}: maps >= 256x256"


def install():
    import torch.nn.functional as F
    real_conv, real_convt = F.conv2d, F.conv_transpose2d

    def conv2d(x, w, *a, **k):
        c = conv_class(STATE["net"], x, w, False)
        STATE["seen"][c] = STATE["seen"].get(c, 0) + 1
        if c in STATE["active"]:
            if STATE["which"] != "weights":
                x = split22(x, False, STATE.get("act_pre", 1.0))
            if STATE["which"] != "activations":
                w = split22(w, True)
        return real_conv(x, w, *a, **k)

    def conv_transpose2d(x, w, *a, **k):
        c = conv_class(STATE["net"], x, w, True)
        STATE["seen"][c] = STATE["seen"].get(c, 0) + 1
        if c in STATE["active"]:
            if STATE["which"] != "weights":
                x = split22(x, False, STATE.get("act_pre", 1.0))
            if STATE["which"] != "activations":
                w = split22(w, True)
        return real_convt(x, w, *a, **k)

    F.conv2d, F.conv_transpose2d = conv2d, conv_transpose2d
    for name, tag in (("eyebrow_decomposer", "dec"), ("eyebrow_morphing_combiner", "comb"), ("face_morpher", "face"), ("body_morpher", "body"), ("upscaler", "up")):
        real = getattr(fo, name)

        def wrapped(*a, _real=real, _tag=tag, **k):
            STATE["net"] = _tag
            return _real(*a, **k)
        setattr(fo, name, wrapped)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--frames", type=int, default=8, help="frames of the fixture's batch of 8 to evaluate")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "parity_r06", "split_attribution.txt"))
    ap.add_argument("--act-scale-log2", type=int, default=None, help="what-if: only the row 'ALL encoder-decoder classes, activations only' with the activations "
                                                                     "multiplied by 2^k before the split (and divided after)")
    ap.add_argument("--fine", action="store_true", help="split the encoder-decoder class by network and layer role, and by operand (weights / activations)")
    args = ap.parse_args()
    STATE["fine"] = args.fine
    torch.set_num_threads(args.threads)
    z = np.load(os.path.join(GOLDEN, "full_midgain_io.npz"))
    w = fo.synth_full_weights(int(z["seed"]), head_gains=tuple(float(x) for x in z["head_gains"]))
    images = np.stack([synthetic_image(seed=int(s)) for s in z["b8_image_seeds"]])[: args.frames]
    poses = z["b8_poses"][: args.frames]
    install()
    names = ("up_merged", "up_warped", "up_direct", "up_alpha", "up_grid", "body_merged", "body_grid", "face_0")
    idx = [fo.OUTPUT_NAMES.index(n) for n in names]

    def run(active, which="both"):
        STATE["active"] = set(active)
        STATE["which"] = which
        STATE["seen"] = {}
        t0 = time.time()
        outs = fo.full_forward_torch(w, images, poses, "float64")
        return [outs[i].numpy() for i in idx], time.time() - t0

    base, dt = run(())
    classes = sorted(STATE["seen"])
    lines = [f"# split-operand attribution on the mid-gain batch of {args.frames} (fp64 oracle, operands of ONE class rounded to fp16 hi + lo; {dt:.0f} s per run)",
             "# max |output - unperturbed fp64 output| per class;  convolutions of the class in one forward pass in brackets",
             f"{'class':52s} " + " ".join(f"{n:>11s}" for n in names)]
    print("\n".join(lines), flush=True)
    todo = [((c,), "both") for c in classes] + [(tuple(classes), "both")]
    if args.fine:
        enc = [c for c in classes if c.split(":")[0] in ("dec", "comb", "face")]
        todo = [((c,), "both") for c in enc] + [(tuple(enc), "weights"), (tuple(enc), "activations"), (tuple(enc), "both")]
    if args.act_scale_log2 is not None:
        STATE["act_pre"] = 2.0 ** args.act_scale_log2
        todo = [(tuple(c for c in classes if c.split(":")[0] in ("dec", "comb", "face")), "activations")]
    for active, which in todo:
        got, dt = run(active, which)
        label = (f"ALL {'encoder-decoder ' if args.fine else ''}classes" + ("" if which == "both" else f", {which} only")) if len(active) > 1 else f"{active[0]} [{STATE['seen'][active[0]]}]"
        row = f"{label:52s} " + " ".join(f"{float(np.abs(g - b).max()):11.3e}" for g, b in zip(got, base))
        print(row, flush=True)
        lines.append(row)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
