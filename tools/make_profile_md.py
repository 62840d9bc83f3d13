#!/usr/bin/env python3
"""Turn the rocprofv3 summaries collected under gpurun_out/ (tools/profile_student.sh, profile_traffic.sh,
pmc_full.sh, breakdown_full.sh) into the tracked files under profiles/.  usage: make_profile_md.py <round tag>"""
import csv
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def stats_table(path, top=40):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        out.append(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} | {r['Percentage']} |")
    return "\n".join(out)


def read(path):
    return open(path).read() if os.path.exists(path) else "(not collected)\n"


def student():
    shutil.copy(os.path.join(G, "ps_kernel_stats.csv"), os.path.join(P, f"{tag}_student_b1_kernel_stats.csv"))
    if os.path.exists(os.path.join(G, "student_b1_traffic.json")):
        shutil.copy(os.path.join(G, "student_b1_traffic.json"), os.path.join(P, f"{tag}_student_b1_traffic.json"))
    for b in (1, 32):                                                    # SQ summary bench.py quotes (tools/pmc_json.py, round 6)
        if os.path.exists(os.path.join(G, f"student_b{b}_pmc.json")):
            shutil.copy(os.path.join(G, f"student_b{b}_pmc.json"), os.path.join(P, f"{tag}_student_b{b}_pmc.json"))
    md = [f"# {tag} - student path, batch 1 stream (`bench.py --steps 200 --warmup 50 --profile-frames 5 --full-frames 0 --d2h-frames 0 --exact-frames 0`), MI355X",
          f"Source: `tools/profile_{tag}.sh` (rocprofv3 --kernel-trace --stats, then separate --pmc passes; FETCH_SIZE and WRITE_SIZE each in",
          "its own pass: together they abort rocprofv3 on this image).  Kernels: generation 2 (fp16 hi/lo split MFMA), weights-resident",
          f"level 2, XCD-aware tile order, face + level 0 in one launch, pose bias folded into the prologues.  Machine-readable traffic: `{tag}_student_b1_traffic.json`.", "",
          f"## rocprofv3 --kernel-trace --stats (full CSV: {tag}_student_b1_kernel_stats.csv)", stats_table(os.path.join(G, "ps_kernel_stats.csv")), "",
          "## PMC passes (per-launch averages summed over the chip; SQ_* cycle counters count quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES)",
          "```", read(os.path.join(G, "ps_pmc_summary.txt")).strip(), "```", "",
          "## HBM-side traffic (FETCH_SIZE / WRITE_SIZE in KiB per launch; reads x2 for wide streams per MI355X_MICROARCH.md)",
          "```", read(os.path.join(G, "pt_summary.txt")).strip(), "```", "",
          read(os.path.join(P, f"{tag}_student_b1_reading.md"))]
    open(os.path.join(P, f"{tag}_student_b1_profile.md"), "w").write("\n".join(md))


def full():
    shutil.copy(os.path.join(G, "pf_kernel_stats.csv"), os.path.join(P, f"{tag}_full_b1_kernel_stats.csv"))
    if os.path.exists(os.path.join(G, "full_b1_traffic.json")):
        shutil.copy(os.path.join(G, "full_b1_traffic.json"), os.path.join(P, f"{tag}_full_b1_traffic.json"))
    for b in (1, 8):                                                     # SQ summary per kernel template instance (tools/pmc_json.py --mode full, round 6)
        if os.path.exists(os.path.join(G, f"full_b{b}_pmc.json")):
            shutil.copy(os.path.join(G, f"full_b{b}_pmc.json"), os.path.join(P, f"{tag}_full_b{b}_pmc.json"))
    if os.path.exists(os.path.join(G, "full_b1_layers.json")):          # per-launch-class table of the last cold frame (bench.py quotes its dominant class)
        shutil.copy(os.path.join(G, "full_b1_layers.json"), os.path.join(P, f"{tag}_full_b1_layers.json"))
    md = [f"# {tag} - full THA4 model (mode_07), batch 1, MI355X",
          f"Source: `tools/profile_{tag}.sh` (`rocprofv3 --kernel-trace --stats -- python tools/time_full.py`: 3 warm-up + 20 steady + 20 cold",
          "frames = 43 frames, 21 of them run the eyebrow decomposer; PMC passes over 9 frames; FETCH_SIZE / WRITE_SIZE passes over steady-only",
          f"and cold-only runs -> `{tag}_full_b1_traffic.json`; per-layer join of the schedule dump with the kernel trace).  Synthetic seeded weights.", "",
          "Un-profiled wall clock of the same script:", "```", read(os.path.join(G, "pf_time.log")).strip(), "```", "",
          f"## rocprofv3 --kernel-trace --stats (full CSV: {tag}_full_b1_kernel_stats.csv)", stats_table(os.path.join(G, "pf_kernel_stats.csv")), "",
          "## Per-layer breakdown of one cold frame (largest first; TFLOP/s = as-written FLOPs of the layer / its launches)",
          "```", "\n".join(read(os.path.join(G, "bd_report.txt")).splitlines()[:60]), "```", "",
          "## PMC passes (per-launch averages summed over the chip)", "```",
          "\n".join(l for l in read(os.path.join(G, "pmcfull_summary.txt")).splitlines() if l.startswith("==") or "conv_" in l or "norm_" in l or "attention" in l),
          "```", "", read(os.path.join(P, f"{tag}_full_b1_reading.md"))]
    open(os.path.join(P, f"{tag}_full_b1_profile.md"), "w").write("\n".join(md))


def batched():
    """configs[3] / configs[4], one GPU's share each (tools/profile_r03.sh): student batch 32, full model batch 8"""
    if os.path.exists(os.path.join(G, "pb32_kernel_stats.csv")):
        shutil.copy(os.path.join(G, "pb32_kernel_stats.csv"), os.path.join(P, f"{tag}_student_b32_kernel_stats.csv"))
        if os.path.exists(os.path.join(G, "student_b32_traffic.json")):
            shutil.copy(os.path.join(G, "student_b32_traffic.json"), os.path.join(P, f"{tag}_student_b32_traffic.json"))
        md = [f"# {tag} - student path, batch 32 per Poser.pose() call (BASELINE configs[3], one GPU's share), MI355X",
              f"Source: `tools/profile_{tag}.sh`: `bench.py --batch 32 --characters lambda_00 --steps 24 --warmup 4` under rocprofv3 --kernel-trace --stats, one SQ PMC",
              f"pass, FETCH_SIZE and WRITE_SIZE in separate passes (-> `{tag}_student_b32_traffic.json`).  Un-profiled `bench.py --batch 32 --steps 64`:", "```",
              read(os.path.join(G, "pb32_bench.json")).strip()[:700], "```", "",
              f"## rocprofv3 --kernel-trace --stats (full CSV: {tag}_student_b32_kernel_stats.csv)", stats_table(os.path.join(G, "pb32_kernel_stats.csv"), 8), "",
              "## PMC pass (per-launch averages summed over the chip; a launch = 32 frames)", "```", read(os.path.join(G, "pb32_pmc_summary.txt")).strip(), "```", "",
              "## HBM-side traffic (KiB per launch of 32 frames)", "```", read(os.path.join(G, "pb32_traffic_summary.txt")).strip(), "```", "",
              read(os.path.join(P, f"{tag}_student_b32_reading.md"))]
        open(os.path.join(P, f"{tag}_student_b32_profile.md"), "w").write("\n".join(md))
    if os.path.exists(os.path.join(G, "pfb8_kernel_stats.csv")):
        shutil.copy(os.path.join(G, "pfb8_kernel_stats.csv"), os.path.join(P, f"{tag}_full_b8_kernel_stats.csv"))
        if os.path.exists(os.path.join(G, "full_b8_traffic.json")):
            shutil.copy(os.path.join(G, "full_b8_traffic.json"), os.path.join(P, f"{tag}_full_b8_traffic.json"))
        md = [f"# {tag} - full THA4 model, batch 8 distinct images per call (BASELINE configs[4], one GPU's share), MI355X",
              f"Source: `tools/profile_{tag}.sh`: `tools/time_full.py --batch 8` (a `max_batch = 8` handle: no K split, no conv_small, no folded normalisations where 8",
              f"frames fill the chip; every step is cold) under rocprofv3 --kernel-trace --stats (3 warm-up + 10 steps), one SQ PMC pass and FETCH_SIZE / WRITE_SIZE passes",
              f"(3 + 5 steps -> `{tag}_full_b8_traffic.json`).  Un-profiled wall clock:", "```", read(os.path.join(G, "pfb8_time.log")).strip(), "```", "",
              f"## rocprofv3 --kernel-trace --stats (full CSV: {tag}_full_b8_kernel_stats.csv)", stats_table(os.path.join(G, "pfb8_kernel_stats.csv"), 30), "",
              "## PMC pass (per-launch averages summed over the chip; a launch covers the 8 frames of the step)", "```",
              "\n".join(l for l in read(os.path.join(G, "pfb8_pmc_summary.txt")).splitlines() if l.startswith("==") or "conv_" in l or "norm_" in l or "attention" in l),
              "```", "", read(os.path.join(P, f"{tag}_full_b8_reading.md"))]
        open(os.path.join(P, f"{tag}_full_b8_profile.md"), "w").write("\n".join(md))


if __name__ == "__main__":
    student()
    full()
    batched()
