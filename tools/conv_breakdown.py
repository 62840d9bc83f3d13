#!/usr/bin/env python3
"""Join the conv schedule dump (THA4_DUMP_SCHEDULE=1, stderr of poser creation) with a rocprofv3 kernel trace:
per-layer time of the LAST cold frame.  usage: conv_breakdown.py <schedule.txt> <kernel_trace.csv> [<classes.json>]
(the JSON - profiles/rNN_full_b1_layers.json - is what bench.py's full-model roofline quotes its dominant launch class from)"""
import csv
import re
import sys
from collections import defaultdict

sched = []
for line in open(sys.argv[1]):
    if line.startswith("conv "):
        kv = dict(re.findall(r"(\w+)=([\w()x ]+?)(?= \w+=|$)", line.strip()[line.index("kind="):]))
        sched.append(kv)
def merged(kv):          # round 4: the four parity classes of a transposed convolution are ONE launch of conv_tile / conv_small (full_net.h merge_classes)
    return int(kv["classes"]) == 4 and kv.get("tiled") in ("1", "2")


launches = []
for kv in sched:
    for c in range((1 if merged(kv) else int(kv["classes"])) * (2 if int(kv.get("ksplit", "1")) > 1 else 1)):   # conv_small (tiled=2) never splits K over launches
        launches.append(kv)
rows = [r for r in csv.DictReader(open(sys.argv[2]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
convs = [r for r in rows if "conv_mfma_kernel" in r["Kernel_Name"] or "conv_splitk_kernel" in r["Kernel_Name"] or "conv_tile_kernel" in r["Kernel_Name"]
         or "conv_small_kernel" in r["Kernel_Name"] or "conv_point_kernel" in r["Kernel_Name"]]
n = len(launches)
last = convs[-n:]
print(f"{len(sched)} convs, {n} launches per cold frame, {len(convs)} conv launches in trace")
agg = defaultdict(lambda: [0, 0.0, 0.0])
tot = 0.0
for kv, r in zip(launches, last):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    taps = int(kv["taps"]) // (4 if int(kv["classes"]) == 4 else 1)
    th, tw = map(int, kv["tile"].split("x"))
    cin = int(kv["cin"].split("(")[0])
    gflop = 2.0 * th * tw * cin * int(kv["cout"]) * taps / 1e9 * (4 if merged(kv) else 1)
    key = (kv["kind"], kv["tile"], kv["mode"], cin, kv["cout"], kv["splitk"] + ("S" if kv.get("tiled") == "2" else "P" if kv.get("tiled") == "3" else ""), kv["tmb"], kv["pg"], kv["wgs"], kv.get("ksplit", "1"))
    agg[key][0] += 1
    agg[key][1] += us
    agg[key][2] += gflop / (2 if int(kv.get("ksplit", "1")) > 1 else 1)
    tot += us
print(f"total conv time {tot / 1e3:.2f} ms")
for key, (cnt, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"kind={key[0]} tile={key[1]:8s} mode={key[2]} cin={key[3]:4d} cout={key[4]:>4s} splitk={key[5]} tmb={key[6]} pg={key[7]} wgs={key[8]:>5s} ksplit={key[9]:>2s}  "
          f"launches={cnt:3d}  total={us:8.1f} us  avg={us / cnt:7.1f} us  {gf / (us * 1e-6) / 1e3:6.1f} TFLOP/s")

if len(sys.argv) > 3:
    import json
    kinds = {"0": "conv3x3", "1": "conv1x1", "2": "conv4x4 stride 2", "3": "convT4x4 stride 2"}
    classes = [{"class": f"{kinds.get(key[0], key[0])} {key[1]} cin={key[3]} cout={key[4]}", "in_mode": int(key[2]), "kernel": ("conv_small_kernel" if key[5].endswith("S") else
                "conv_point_kernel" if key[5].endswith("P") else "conv_tile_kernel"), "tmb": int(key[6]), "pg": int(key[7]), "workgroups": int(key[8]), "ksplit": int(key[9]),
                "launches": cnt, "total_us": round(us, 1), "avg_us": round(us / cnt, 2), "gflop_per_launch": round(gf / cnt, 4), "tflops": round(gf / (us * 1e-6) / 1e3, 1)}
               for key, (cnt, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    json.dump({"source": "tools/conv_breakdown.py: schedule dump (THA4_DUMP_SCHEDULE) joined with the rocprofv3 --kernel-trace of the last cold frame",
               "conv_launches_per_cold_frame": n, "total_conv_us": round(tot, 1), "classes": classes}, open(sys.argv[3], "w"), indent=1)
