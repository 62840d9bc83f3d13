"""Build the gfx950 shared library in-tree with hipcc (no JIT cache: the .so must travel with the tree)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB = os.path.join(CSRC, "libtha4_hip.so")
RESOURCES = os.path.join(CSRC, "libtha4_hip.resources.txt")
# The same source compiled with every memory wait forced to zero (-mllvm -amdgpu-waitcnt-forcezero=1): identical arithmetic, so a byte
# that differs from the shipped library is a memory-ordering / hazard fault in one of them (how round 3 found the faulty level-2
# geometry).  TEST ARTEFACT: only tests/test_twin_gpu.py and tools/compare_libs.py load it, through THA4_HIP_LIB.
TWIN = os.path.join(CSRC, "libtha4_hip_wait0.so")
SOURCES = ["tha4_capi.hip"]
# No packed-fp32 VALU instructions (v_pk_mul / v_pk_fma / v_pk_add_f32) in any kernel of the library: the compiler's packed arithmetic is
# what made level2_16p_kernel<8,.,2> produce run-to-run varying pixels on gfx950 once the sine shrank to one instruction
# (profiles/r03_sin_cliff.md: wait states in front of every v_pk_* all but cure it, a build without them equals its forced-wait twin
# bit for bit), and it buys nothing here (a v_pk_fma_f32 costs two v_fma_f32).  THA4_NO_PACKED_FP32 tells the sources so.
DEVICE_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DTHA4_NO_PACKED_FP32=1"]


def _headers():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found (ROCm toolchain required to build libtha4_hip.so)")
    return p


def build_native(force: bool = False, verbose: bool = False) -> str:
    deps = [os.path.join(CSRC, f) for f in SOURCES + _headers()] + [os.path.join(INCLUDE, "tha4_hip.h"), os.path.abspath(__file__)]
    if not force and not _stale(LIB, deps) and os.path.exists(RESOURCES):
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Rpass-analysis=kernel-resource-usage"] + DEVICE_FLAGS + \
          ["-I", CSRC, "-I", INCLUDE] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    with open(RESOURCES, "w") as f:          # per-kernel VGPR / scratch / LDS report (tests/test_api_surface.py gates on it)
        f.write(parse_resource_remarks(r.stderr))
    return LIB


def build_forced_wait_twin(force: bool = False) -> str:
    deps = [os.path.join(CSRC, f) for f in SOURCES + _headers()] + [os.path.join(INCLUDE, "tha4_hip.h"), os.path.abspath(__file__)]
    if not force and not _stale(TWIN, deps):
        return TWIN
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-waitcnt-forcezero=1"] + DEVICE_FLAGS + \
          ["-I", CSRC, "-I", INCLUDE] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", TWIN]
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed (forced-wait twin):\n" + r.stderr[-4000:])
    return TWIN


def parse_resource_remarks(stderr: str) -> str:
    """`-Rpass-analysis=kernel-resource-usage` remarks -> one line per kernel: name vgprs vgpr_spill scratch_bytes occupancy."""
    out, cur = [], {}
    for line in stderr.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass", 1)[0].strip()
        if body.startswith("Function Name:"):
            if cur:
                out.append(cur)
            cur = {"name": body.split(":", 1)[1].strip()}
        elif ":" in body and cur:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    if cur:
        out.append(cur)
    return "".join(f"{c['name']} vgprs={c.get('VGPRs', '?')} vgpr_spill={c.get('VGPRs Spill', '?')} "
                   f"scratch={c.get('ScratchSize [bytes/lane]', '?')} occupancy={c.get('Occupancy [waves/SIMD]', '?')} "
                   f"sgpr_spill={c.get('SGPRs Spill', '?')}\n" for c in out)
