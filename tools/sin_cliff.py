#!/usr/bin/env python3
"""Why does a v_sin_f32 sine that is only 2.6e-7 less accurate move the posed frame by 3e-2?  (GPU box; tuning aid.)

Evaluates the student's sine variants ON THE DEVICE over exactly the arguments the network feeds them - every
pre-activation u = 30 (W x + b) of every sine layer of one pose (fp64 pipeline of the oracle's restructured intermediates,
rounded to fp32) - and reports, per layer: the argument range, max |sin_variant(u) - sin(u)|, the offending argument, the
signed mean error (bias) and the error correlated with sin / cos (a gain or phase error), for
  0: the shipped 12-op polynomial `sin_u`,   1: v_sin_f32 behind a 2-term Cody-Waite reduction by 2 pi (the -DTHA4_HW_SIN
  variant),   2: v_sin_f32(u / 2 pi) with no reduction.
Then runs the whole student through the library built with -DTHA4_HW_SIN (build_variants/libtha4_hwsin.so) and the shipped
one and prints all six outputs' distance to the oracle: which network (face / body) moves, and by how much.

  python tools/sin_cliff.py [pose_index]   ->  stdout (copy into profiles/r03_sin_cliff.md)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import student_oracle as so  # noqa: E402  (tools may use the oracle; the product never does)


def collect_arguments(w, pose):
    """(name, u[float32]) for every sine layer: u = 30 * pre-activation in fp64, rounded to fp32."""
    out = []
    pose = np.asarray(pose, np.float64)
    sines, _ = so.face_layers(w)
    S = so.FACE_SIZE
    ax = so.position_axis(S)
    xs = np.broadcast_to(ax[None, :], (S, S)).reshape(-1)
    ys = np.broadcast_to(ax[:, None], (S, S)).reshape(-1)
    x = np.concatenate([xs[None], ys[None], np.broadcast_to(pose[:so.NUM_FACE_POSE, None], (so.NUM_FACE_POSE, S * S))], 0)
    for i, (W, b) in enumerate(sines):
        u = so.OMEGA_0 * (W.astype(np.float64) @ x + b[:, None])
        out.append((f"face.{i}", u.astype(np.float32).ravel()))
        x = np.sin(u)
    levels, _ = so.body_layers(w)
    h = None
    for l, S in enumerate(so.LEVEL_SIZES):
        ax = so.position_axis(S)
        xs = np.broadcast_to(ax[None, :], (S, S)).reshape(-1)
        ys = np.broadcast_to(ax[:, None], (S, S)).reshape(-1)
        pp = np.concatenate([xs[None], ys[None], np.broadcast_to(pose[:, None], (so.NUM_POSE, S * S))], 0)
        if l == 0:
            x = pp
        else:
            c = h.shape[0]
            x = np.concatenate([so.upsample2x_numpy(h.reshape(c, S // 2, S // 2)).reshape(c, S * S), pp], 0)
        for j, (W, b) in enumerate(levels[l]):
            u = so.OMEGA_0 * (W.astype(np.float64) @ x + b[:, None])
            out.append((f"body.L{l}.{j}", u.astype(np.float32).ravel()))
            x = np.sin(u)
        h = x
    return out


def main():
    pose_index = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 0
    skip_eval = "--skip-eval" in sys.argv
    g = os.path.join(ROOT, "tests", "golden")
    w = dict(np.load(os.path.join(g, "student_lambda_00_weights.npz")))
    io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
    pose = io["poses"][pose_index]
    lib = C.CDLL(os.path.join(ROOT, "tools", "microbench", "libsin_eval.so"))
    lib.sin_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int]
    lib.sin_eval.restype = C.c_int
    names = ["polynomial (shipped)", "v_sin + Cody-Waite 2pi", "v_sin(u/2pi), no reduction"]
    print(f"# sine variants over the arguments of lambda_00, pose {pose_index} (device evaluation vs np.sin in fp64)\n")
    print("| layer | values | max abs(u) | variant | max abs err | at u = | got / want | mean err | corr with sin (gain) | corr with cos (phase) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    worst = {}
    for name, u in ([] if skip_eval else collect_arguments(w, pose)):
        ref = np.sin(u.astype(np.float64))
        cs = np.cos(u.astype(np.float64))
        for v in range(3):
            got = np.empty_like(u)
            rc = lib.sin_eval(u.ctypes.data, got.ctypes.data, u.size, v)
            assert rc == 0, rc
            err = got.astype(np.float64) - ref
            i = int(np.abs(err).argmax())
            gain = float((err * ref).sum() / (ref * ref).sum())
            phase = float((err * cs).sum() / (cs * cs).sum())
            print(f"| {name} | {u.size} | {np.abs(u).max():.1f} | {names[v]} | {np.abs(err).max():.3e} | {u[i]:.6f} | {got[i]:.8f} / {ref[i]:.8f} | "
                  f"{err.mean():+.2e} | {gain:+.2e} | {phase:+.2e} |")
            worst[v] = max(worst.get(v, 0.0), float(np.abs(err).max()))
    if worst:
        print("\nworst sine error over all layers: " + ", ".join(f"{names[v]} {worst[v]:.3e}" for v in range(3)))

    # ---- the whole student through both libraries ------------------------------------------------------------------
    hw = os.path.join(ROOT, "build_variants", "libtha4_hwsin.so")
    ref6 = [o.numpy() for o in so.student_forward_torch(w, io["image_f32"], pose, "float32")]
    inter = so.student_intermediates(w, pose)
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import tha4_amd
from tha4_amd.poser.modes import mode_14
from tha4_amd.weights import split_flat_weights
g = os.path.join(%r, "tests", "golden")
w = dict(np.load(os.path.join(g, "student_lambda_00_weights.npz"))); io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
f, b = split_flat_weights(w)
p = mode_14.create_poser_from_state_dicts(torch.device("cuda:0"), f, b)
img, pose = torch.from_numpy(io["image_f32"]).cuda(), torch.from_numpy(io["poses"][%d]).cuda()
outs = p.get_posing_outputs(img, pose)
z1, z2 = p.debug_hand_off(1).numpy(), p.debug_hand_off(2).numpy()
again = p.get_posing_outputs(img, pose)
np.savez(sys.argv[1], *[o.cpu().numpy() for o in outs], z1=z1, z2=z2, rerun_equal=np.array([bool(torch.equal(a, b)) for a, b in zip(outs, again)]))
""" % (ROOT, ROOT, pose_index)
    print("\n| library | blended | alpha | colour | warped | grid | face | z1 (level 0 -> 1) | z2 (level 1 -> 2) | rerun bitwise equal |   (max abs vs oracle, pose %d)" % pose_index)
    print("|---|---|---|---|---|---|---|---|---|---|")
    got = {}
    extra = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--libs=")]
    more = [(f"variant {n}", os.path.join(ROOT, "build_variants", f"libtha4_{n}.so")) for n in (extra[0].split(",") if extra else [])]
    for label, path in [("shipped polynomial", None), ("-DTHA4_HW_SIN", hw)] + more:
        if path and not os.path.exists(path):
            print(f"| {label} | (library {path} not built) |")
            continue
        env = dict(os.environ)
        if path:
            env["THA4_HIP_LIB"] = path
        out = f"/tmp/sin_cliff_{'hw' if path else 'poly'}.npz"
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env)
        z = np.load(out)
        got[label] = [z[f"arr_{k}"] for k in range(6)] + [z["z1"], z["z2"]]
        ez1 = np.abs(z["z1"][:180] - inter["z1"]).max()
        ez2 = np.abs(z["z2"][:90] - inter["z2"]).max()
        print(f"| {label} | " + " | ".join(f"{np.abs(got[label][k] - ref6[k]).max():.3e}" for k in range(6)) +
              f" | {ez1:.3e} (max abs z1 {np.abs(inter['z1']).max():.2f}) | {ez2:.3e} (max abs z2 {np.abs(inter['z2']).max():.2f}) | {z['rerun_equal'].all()} |")
    if "-DTHA4_HW_SIN" in got:
        a, b = got["shipped polynomial"], got["-DTHA4_HW_SIN"]
        d = np.abs(a[0] - b[0])[0].max(0)                  # [512, 512]
        print(f"\nposed frame, hardware-sine build vs shipped build: max {d.max():.3e}; pixels above 1e-3: {(d > 1e-3).sum()} of {512 * 512}; "
              f"grid change max delta {np.abs(a[4] - b[4]).max():.3e} (1.0 = 256 px); face max delta {np.abs(a[5] - b[5]).max():.3e}")
        ys, xs = np.nonzero(d > 1e-3)
        if ys.size:
            print(f"bad pixels: rows {ys.min()}..{ys.max()}, columns {xs.min()}..{xs.max()}; first 24 (y, x, delta): " +
                  " ".join(f"({y},{x},{d[y, x]:.1e})" for y, x in list(zip(ys, xs))[:24]))
        for name, k, lim in (("alpha", 1, 1e-4), ("colour", 2, 1e-3), ("grid", 4, 1e-5)):
            dk = np.abs(a[k] - b[k])[0].max(0)
            print(f"  {name}: pixels differing by more than {lim:g}: {(dk > lim).sum()}; max {dk.max():.3e}")
        for zi, lvl, side in ((6, 1, 128), (7, 2, 256)):
            dz = np.abs(a[zi] - b[zi]).max(0)
            ys, xs = np.nonzero(dz > 1e-4)
            print(f"  z{lvl} (level {lvl - 1} output side, {side}^2): max delta {dz.max():.3e}; positions above 1e-4: {ys.size}" +
                  (f" (rows {ys.min()}..{ys.max()}, columns {xs.min()}..{xs.max()})" if ys.size else ""))


if __name__ == "__main__":
    main()
