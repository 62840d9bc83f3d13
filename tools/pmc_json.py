#!/usr/bin/env python3
"""rocprofv3 SQ PMC passes of the student stream -> the machine-readable per-kernel summary bench.py quotes (`roofline.mfma_busy`, `roofline.limiter`).

  python tools/pmc_json.py <dir with *_counter_collection.csv> [<dir> ...] --batch 1 -o profiles/r06_student_b1_pmc.json

Per kernel (launches of `--batch` frames only: the grid size selects them), averages per launch over the chip:
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles), launch cycles = GRBM_GUI_ACTIVE / 8 XCDs of the same pass when collected,
                   else duration x 2.4 GHz: the fraction of the launch during which a SIMD's matrix pipe is executing
  active / issue_stall / parked = SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES (each from its own pass): where a resident wave's time goes
  valu_per_mfma  = SQ_INSTS_VALU / SQ_INSTS_MFMA
SQ_* cycle counters are quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (MI355X_MICROARCH.md).  A PMC pass runs the kernels 1.2-1.4x slower than an unprofiled
stream: ratios inside one pass are what this file keeps."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

NAMES = {"front16r_kernel": "front", "front16_kernel": "front", "level1_16r_kernel": "level1", "level1_16_kernel": "level1", "level2_16p_kernel": "level2"}
GRID_B1 = {"front": (512,), "level1": (256, 1024), "level2": (256,)}         # workgroups of a batch-1 launch (register / LDS-activation forms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--mode", choices=("student", "full"), default="student",
                    help="full: every kernel of the capture under its full template name (conv_tile_kernel<4, 4, 0, 1, 4>, ...), no grid filter")
    ap.add_argument("-o", required=True)
    a = ap.parse_args()
    out = {"source": "rocprofv3 --kernel-trace --pmc <SQ counters> (one pass per counter group), per-launch averages; tools/pmc_json.py", "batch": a.batch, "kernels": {}}
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            dur = defaultdict(dict)
            for r in csv.DictReader(open(f)):
                if a.mode == "full":
                    short = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tha4::", "").strip()
                    if short.startswith("__amd") or "student" in short.lower():
                        continue
                else:
                    k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
                    if k not in NAMES:
                        continue
                    short = NAMES[k]
                    wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) if "Grid_Size" in r and r.get("Workgroup_Size") else None
                    if wgs is not None and wgs not in tuple(g * a.batch for g in GRID_B1[short]):
                        continue
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[short][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            for short, cs in acc.items():
                e = out["kernels"].setdefault(short, {"counters": {}, "passes": []})
                avg = {c: sum(v) / len(v) for c, v in cs.items()}
                us = sum(dur[short].values()) / len(dur[short]) / 1e3
                e["passes"].append({"counters": sorted(avg), "launches": len(dur[short]), "avg_us_in_pass": round(us, 2)})
                e["counters"].update({c: round(v, 1) for c, v in avg.items()})
                if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                    e["_mfma_pass_us"] = us
                if "GRBM_GUI_ACTIVE" in avg and us > 0:
                    # (diagnostic only: GRBM_GUI_ACTIVE / 8 XCDs / launch duration comes out at 2.5 - 2.9 GHz - the counter also runs while the launch is dispatched and
                    #  retired - so it is NOT used as the clock of the busy fraction)
                    e["grbm_gui_active_mhz"] = round(avg["GRBM_GUI_ACTIVE"] / 8.0 / us, 1)
                if "SQ_WAVE_CYCLES" in avg:
                    for key, c in (("active", "SQ_ACTIVE_INST_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("parked", "SQ_WAIT_ANY")):
                        if c in avg:
                            e[key] = round(avg[c] / avg["SQ_WAVE_CYCLES"], 4)
                if "SQ_INSTS_MFMA" in avg and avg["SQ_INSTS_MFMA"] > 0 and "SQ_INSTS_VALU" in avg:
                    e["valu_per_mfma"] = round(avg["SQ_INSTS_VALU"] / avg["SQ_INSTS_MFMA"], 2)
    for e in out["kernels"].values():          # matrix-pipe busy fraction: busy cycles / (1024 SIMDs x launch cycles), the cycles at the MEASURED clock where a GRBM pass exists
        us = e.pop("_mfma_pass_us", None)
        if us:
            e["mfma_busy"] = round(e["counters"]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * us * 2400.0), 4)
            e["mfma_busy_clock"] = "2.4 GHz (the part's peak shader clock) x the SQ pass's launch duration: a LOWER bound of the fraction when the chip clocks lower under load"
        e["launches_per_pass"] = max(p_["launches"] for p_ in e["passes"])
        e["avg_us"] = round(sum(p_["avg_us_in_pass"] for p_ in e["passes"]) / len(e["passes"]), 2)
    with open(a.o, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: {x: y for x, y in v.items() if x not in ("counters", "passes")} for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
