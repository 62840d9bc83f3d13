#!/usr/bin/env python3
"""Fault hunt, second cut: keep the faulty kernel's schedule but make the compiler's COUNTED waits total inside the strip loop -
every `vmcnt(N)` -> `vmcnt(0)` (tests the in-order-return assumption of its load tracking), every `lgkmcnt(N)` -> `lgkmcnt(0)`, or both.
  python tools/hunt/asm_waits.py /tmp/hunt/full.s out_dir"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_regions import kernel_span, loop_span, CLANG, LLD

def assemble(lines, out):
    s = out + ".s"
    open(s, "w").writelines(lines)
    subprocess.run([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", out + ".o"], check=True)
    subprocess.run([LLD, "-shared", out + ".o", "-o", out + ".co"], check=True)
    os.remove(s); os.remove(out + ".o")

def main():
    src, outdir = sys.argv[1], sys.argv[2]
    os.makedirs(outdir, exist_ok=True)
    base = open(src).readlines()
    a, b = kernel_span(base)
    lo, hi = loop_span(base, a, b)
    for name, vm, lg in (("vm0", True, False), ("lgkm0", False, True), ("both0", True, True)):
        lines = list(base)
        n = 0
        for i in range(lo, hi):
            t = lines[i]
            if t.strip().startswith("s_waitcnt"):
                u = t
                if vm: u = re.sub(r"vmcnt\(\d+\)", "vmcnt(0)", u)
                if lg: u = re.sub(r"lgkmcnt\(\d+\)", "lgkmcnt(0)", u)
                n += u != t
                lines[i] = u
        assemble(lines, os.path.join(outdir, name))
        print(name, "waits changed:", n)
    # every load followed by a total wait (loads serialised, nothing else changed)
    lines = []
    for i, t in enumerate(base):
        lines.append(t)
        if lo <= i < hi and t.strip().startswith("global_load"):
            lines.append("\ts_waitcnt vmcnt(0)\n")
    assemble(lines, os.path.join(outdir, "load_sync"))
    lines = []
    for i, t in enumerate(base):
        lines.append(t)
        if lo <= i < hi and t.strip().startswith("v_sin_f32"):
            lines.append("\ts_nop 7\n\ts_nop 7\n")
    assemble(lines, os.path.join(outdir, "sin_nop16"))
    lines = []
    for i, t in enumerate(base):
        lines.append(t)
        if lo <= i < hi and t.strip().startswith("v_mfma"):
            lines.append("\ts_nop 7\n")
    assemble(lines, os.path.join(outdir, "mfma_nop8"))
    lines = []
    for i, t in enumerate(base):
        lines.append(t)
        if lo <= i < hi and t.strip().startswith("v_cvt_pk_f16_f32"):
            lines.append("\ts_nop 3\n")
    assemble(lines, os.path.join(outdir, "cvtpk_nop4"))

if __name__ == "__main__":
    main()
