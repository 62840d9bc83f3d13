import csv, glob, sys, os
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f)))
seg, segs = [], []
for s, e, n in ev:
    if "display_rgba8" in n:
        segs.append(seg); seg = []
    else:
        seg.append((s, e, n))
for name, sg in zip(["warm-up", "pose()", "pose(out=)", "raw C ABI"], segs):
    gaps_first, gaps_other = [], []
    for i in range(1, len(sg)):
        gap = (sg[i][0] - sg[i - 1][1]) / 1e3
        (gaps_first if "front16" in sg[i][2] or "posebias" in sg[i][2] else gaps_other).append(gap)
    span = (sg[-1][1] - sg[0][0]) / 1e3 if sg else 0
    frames = max(1, len(sg) // 3)
    print(f"{name:12s} kernels {len(sg):5d}  us/frame {span / frames:7.2f}  gap before first kernel of a frame {sum(gaps_first) / max(1, len(gaps_first)):.2f} us  other gaps {sum(gaps_other) / max(1, len(gaps_other)):.2f} us")
