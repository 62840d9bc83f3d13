#!/usr/bin/env python3
"""rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one pass each - together they abort rocprofv3 on this image) ->
the machine-readable traffic summary bench.py reads from profiles/ (`roofline.traffic`).

  student:  python tools/traffic_json.py student <fetch dir> <write dir> -o profiles/r02_student_b1_traffic.json
  full:     python tools/traffic_json.py full <fetch dir> <write dir> --frames N [--cold-fetch D --cold-write D --cold-frames N] -o ...

Per kernel: average KiB per launch (rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB).  HBM bytes = 2 x FETCH_SIZE
(gfx950 wide-read correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE.
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

STUDENT = {"posebias_kernel": "posebias", "face16_kernel": "face", "level0_16_kernel": "level0", "level1_16_kernel": "level1",
           "level2_16p_kernel": "level2", "level2_16_kernel": "level2", "front16_kernel": "front", "front16r_kernel": "front", "level1_16r_kernel": "level1"}


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]].append(float(r["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", choices=["student", "full"])
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--cold-fetch")
    ap.add_argument("--cold-write")
    ap.add_argument("--cold-frames", type=int, default=0)
    ap.add_argument("--command", default="")
    ap.add_argument("--batch", type=int, default=1, help="frames per launch set (batched captures: per-frame bytes = per-step bytes / batch)")
    ap.add_argument("--steps", type=int, default=0, help="full model, batched: pose() calls in the capture incl. warm-up (all cold)")
    ap.add_argument("-o", required=True)
    a = ap.parse_args()
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per-launch averages; tools/traffic_json.py",
           "command": a.command, "unit": "KiB per launch", "fetch_correction": "x2 (gfx950 wide reads, MI355X_MICROARCH.md HBM section)"}
    fe, wr = per_kernel(a.fetch_dir, "FETCH_SIZE"), per_kernel(a.write_dir, "WRITE_SIZE")
    if a.model == "student":
        ks = {}
        for k, short in STUDENT.items():
            if k in fe:
                ks[short] = {"fetch_kib": round(sum(fe[k]) / len(fe[k]), 2), "write_kib": round(sum(wr[k]) / len(wr[k]), 2), "launches": len(fe[k])}
        out["kernels"] = ks
        out["batch"] = a.batch
        out["frame_bytes"] = int(round(sum((2.0 * k["fetch_kib"] + k["write_kib"]) * 1024 for k in ks.values()) / max(1, a.batch)))
    elif a.steps:
        # batched full model: every step is cold (distinct images each call)
        tot = sum(2.0 * sum(v) for v in fe.values()) * 1024 + sum(sum(v) for v in wr.values()) * 1024
        out["batch"] = a.batch
        out["captured"] = {"steps": a.steps, "bytes": int(tot)}
        out["step_bytes"] = int(round(tot / a.steps))
        out["cold_frame_bytes"] = int(round(tot / a.steps / a.batch))
        out["kernels"] = {k: {"fetch_kib": round(sum(v) / len(v), 2), "write_kib": round(sum(wr[k]) / max(1, len(wr[k])), 2), "launches_per_step": round(len(v) / a.steps, 2)}
                          for k, v in sorted(fe.items(), key=lambda kv: -sum(kv[1]))}
    else:
        def total_bytes(fe, wr):
            return sum(2.0 * sum(v) for v in fe.values()) * 1024 + sum(sum(v) for v in wr.values()) * 1024
        # tools/time_full.py runs 3 warm-up frames (1 cold + 2 steady) before the timed ones, so a "--mode steady --frames F"
        # capture holds F+2 steady + 1 cold frames and a "--mode cold" capture 2 steady + F+1 cold: two equations, two unknowns
        S = total_bytes(fe, wr)
        F = a.frames
        out["captured"] = {"steady_run": f"{F + 2} steady + 1 cold frames", "steady_run_bytes": int(S)}
        out["kernels"] = {k: {"fetch_kib": round(sum(v) / len(v), 2), "write_kib": round(sum(wr[k]) / max(1, len(wr[k])), 2), "launches_per_frame": round(len(v) / (F + 3), 2)}
                          for k, v in sorted(fe.items(), key=lambda kv: -sum(kv[1]))}
        if a.cold_fetch:
            Cb = total_bytes(per_kernel(a.cold_fetch, "FETCH_SIZE"), per_kernel(a.cold_write, "WRITE_SIZE"))
            Fc = a.cold_frames
            out["captured"]["cold_run"] = f"2 steady + {Fc + 1} cold frames"
            out["captured"]["cold_run_bytes"] = int(Cb)
            # (F+2) s + c = S ; 2 s + (Fc+1) c = Cb
            det = (F + 2) * (Fc + 1) - 2.0
            s_ = (S * (Fc + 1) - Cb) / det
            c_ = ((F + 2) * Cb - 2.0 * S) / det
            out["steady_frame_bytes"] = int(round(s_))
            out["cold_frame_bytes"] = int(round(c_))
        else:
            out["steady_frame_bytes"] = int(round(S / (F + 3)))
    with open(a.o, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main()
