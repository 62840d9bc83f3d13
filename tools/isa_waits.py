#!/usr/bin/env python3
"""Dependent memory round trips of a kernel, read off the ISA (build container, no GPU): lists every memory instruction, s_waitcnt, barrier and
s_endpgm of the named kernels with its instruction index.  A load followed by a wait that covers it, followed by another load ... is a chain of
round trips the source often does not show (indexed reads of a by-value argument struct, conditional loads, loads inside branchy helpers):
profiles/r04_full_conv_tile_reading.md sections 10-11 were found this way.

  python tools/isa_waits.py <libtha4_hip.so> <kernel-name-substring> [...] [--scalar] [--upto N]
     --scalar  also list s_load_* (argument-block reads)      --upto N  stop after instruction N"""
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin/"


def disassemble(lib):
    subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, "/tmp/_w.fat"], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--type=o", "--unbundle", "--input=/tmp/_w.fat", "--output=/tmp/_w.co",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    return subprocess.run([LLVM + "llvm-objdump", "-d", "--mcpu=gfx950", "/tmp/_w.co"], capture_output=True, text=True, check=True).stdout.split("\n")


def kernel(lines, name):
    on, ins = False, []
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            if on:
                break
            on = name in m.group(1)
            continue
        if on and l.strip():
            ins.append(l.split("//")[0].strip())
    return ins


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    scalar = "--scalar" in sys.argv
    upto = int(sys.argv[sys.argv.index("--upto") + 1]) if "--upto" in sys.argv else 1 << 30
    if "--upto" in sys.argv:
        args.remove(sys.argv[sys.argv.index("--upto") + 1])
    lines = disassemble(args[0])
    keep = ("s_waitcnt", "global_load", "global_store", "global_atomic", "buffer_", "s_barrier", "s_endpgm") + (("s_load",) if scalar else ())
    for name in args[1:]:
        ins = kernel(lines, name)
        print(f"== {name}: {len(ins)} instructions")
        for i, x in enumerate(ins[:upto]):
            if x.startswith(keep):
                print(f"  {i:5d} {x}")


if __name__ == "__main__":
    main()
