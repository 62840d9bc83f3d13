#!/usr/bin/env python3
"""Build / run launch-geometry variants of libtha4_hip.so (tuning aid, not part of the product).

  python tools/sweep.py build            # here (hipcc cross-compiles): variants -> build_variants/*.so
  python tools/sweep.py run [--steps N]  # on the GPU box: bench every variant, print per-kernel ms

Each variant overrides the -D knobs of csrc/siren_kernels.h (THA4_*_CFG, THA4_NO_PIPELINE).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "talking-head-anime-4-demo_amd", "csrc")
OUT = os.path.join(ROOT, "build_variants")

import importlib.util                                  # the product's own device flags (no packed fp32): ONE definition, tha4_amd/_build.py
_spec = importlib.util.spec_from_file_location("_tha4_build", os.path.join(ROOT, "talking-head-anime-4-demo_amd", "_build.py"))
_bld = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bld)
NOPK = list(_bld.DEVICE_FLAGS)
TUNE = ["-DTHA4_TUNING_BUILD"]                          # ablations / hazard-hunt switches live in csrc/tha4_tuning.h: tuning builds only
VARIANTS = {
    "default": [],
    "wait0": ["-mllvm", "-amdgpu-waitcnt-forcezero=1"],                              # the shipped source with every memory wait forced to zero (tools/compare_libs.py)
    "pk": ["-DTHA4_PACKED_FP32_BUILD", "-DTHA4_L216P_CFG=12,64,1"],                                   # with the compiler's packed fp32 ops (one pixel group per strip: sound)
    "pk_pg2": ["-DTHA4_PACKED_FP32_BUILD", "-DTHA4_L216P_CFG=8,32,2", "-DTHA4_ALLOW_L216P_PG2"],     # the FAULTY combination: packed fp32 + two pixel groups
    "pk_pg2_wait0": ["-DTHA4_PACKED_FP32_BUILD", "-DTHA4_L216P_CFG=8,32,2", "-DTHA4_ALLOW_L216P_PG2", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    "pg1w12": ["-DTHA4_L216P_CFG=12,64,1"], "pg1w8": ["-DTHA4_L216P_CFG=8,64,1"], "pg2w12": ["-DTHA4_L216P_CFG=12,32,2"],
    "pg2w12_wait0": ["-DTHA4_L216P_CFG=12,32,2", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    "poly": ["-DTHA4_SIN_TURNS=0", "-DTHA4_L216P_CFG=8,32,2"],                                                  # round 2: radians, 12-op polynomial
    "poly_wait0": ["-DTHA4_SIN_TURNS=0", "-DTHA4_L216P_CFG=8,32,2", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    "poly_pg1": ["-DTHA4_SIN_TURNS=0"],
    "hwsin": ["-DTHA4_SIN_TURNS=0", "-DTHA4_HW_SIN", "-DTHA4_L216P_CFG=8,32,2", "-DTHA4_ALLOW_L216P_PG2"],                                # radians, Cody-Waite + v_sin_f32: level2_16p<8,4,2> is faulty in this build
    "hwsin_wait0": ["-DTHA4_SIN_TURNS=0", "-DTHA4_HW_SIN", "-DTHA4_L216P_CFG=8,32,2", "-DTHA4_ALLOW_L216P_PG2", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    "hwsin_pg1": ["-DTHA4_SIN_TURNS=0", "-DTHA4_HW_SIN", "-DTHA4_L216P_CFG=8,64,1"],
    "hwsin_stream": ["-DTHA4_SIN_TURNS=0", "-DTHA4_HW_SIN", "-DTHA4_L2_RESIDENT=0"],
    # ---- weights-resident level 2 with three / four waves per SIMD (strips are handed out by ticket: any wave count shares the 64 strips) ----
    "l2w8": ["-DTHA4_L216P_CFG=8,64,1"], "l2w16": ["-DTHA4_L216P_CFG=16,64,1"],
    # ---- streamed kernels with two pixel groups per slot (one A fragment read feeds both; measured neutral in round 2 under the 12-op sine) ----
    "l1pg2": ["-DTHA4_L116_CFG=4,2,2,1,1"], "l0pg2": ["-DTHA4_L016_CFG=2,4,2,3,1"], "facepg2": ["-DTHA4_FACE16_CFG=2,4,2,2"],
    "l1pg2_wait0": ["-DTHA4_L116_CFG=4,2,2,1,1", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    "allpg2": ["-DTHA4_L116_CFG=4,2,2,1,1", "-DTHA4_L016_CFG=2,4,2,3,1", "-DTHA4_FACE16_CFG=2,4,2,2"],
    "allpg2_wait0": ["-DTHA4_L116_CFG=4,2,2,1,1", "-DTHA4_L016_CFG=2,4,2,3,1", "-DTHA4_FACE16_CFG=2,4,2,2", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    # ---- timing ablations (results are wrong; tools/runs_r03/gpu_r03_ablate.sh) ----
    "ab_mfma": ["-DTHA4_ABLATE_MFMA"], "ab_sin": ["-DTHA4_ABLATE_SIN"], "ab_fetch": ["-DTHA4_ABLATE_FETCH"], "ab_barrier": ["-DTHA4_ABLATE_BARRIER"],
    "ab_zload": ["-DTHA4_ABLATE_ZLOAD"], "ab_posefold": ["-DTHA4_ABLATE_POSEFOLD"], "ab_mfma_sin": ["-DTHA4_ABLATE_MFMA", "-DTHA4_ABLATE_SIN"],
    "ab_fetch_barrier": ["-DTHA4_ABLATE_FETCH", "-DTHA4_ABLATE_BARRIER"],
    "ab_all": ["-DTHA4_ABLATE_MFMA", "-DTHA4_ABLATE_SIN", "-DTHA4_ABLATE_FETCH", "-DTHA4_ABLATE_BARRIER", "-DTHA4_ABLATE_ZLOAD"],
    # ---- conv_tile_kernel ablations (round 4; results are wrong): what pre-staged operands / async window fills could buy at most ----
    "abt_valu": ["-DTHA4_ABLATE_TILE_STAGE_VALU"], "abt_window": ["-DTHA4_ABLATE_TILE_WINDOW"], "abt_epi": ["-DTHA4_ABLATE_TILE_EPILOGUE"],
    "abt_window_epi": ["-DTHA4_ABLATE_TILE_WINDOW", "-DTHA4_ABLATE_TILE_EPILOGUE"],
    "abt_all": ["-DTHA4_ABLATE_TILE_WINDOW", "-DTHA4_ABLATE_TILE_EPILOGUE", "-DTHA4_ABLATE_MFMA"],
    # ---- level 1 with independent workgroups sharing a CU (half chunks of 12 KiB: 72 KiB of LDS per 64-pixel workgroup -> two per CU) ----
    "l1w4": ["-DTHA4_L116_CFG=4,1,1,1,1,2"],       # 4 waves / workgroup: each SIMD hosts one wave of each of two workgroups
    "l1w8x2": ["-DTHA4_L116_CFG=4,2,1,1,1,2"],     # 8 waves / workgroup (rows split over two waves): four waves per SIMD
    "l1hb2": ["-DTHA4_L116_CFG=8,1,1,1,1,2"],      # control: the shipped 128-pixel workgroup with half chunks (twice the barriers)
    "l1m2": ["-DTHA4_L116_CFG=8,2,1,1,1,1"],       # 16 waves on the shipped 128-pixel workgroup (rows split over two waves): four per SIMD, same weight stream
    "l1m2h": ["-DTHA4_L116_CFG=8,2,1,1,1,2"],
    "l1m3": ["-DTHA4_L116_CFG=4,3,1,1,1,2"],       # 12 waves per 64-pixel workgroup, two workgroups per CU: six per SIMD
    "frontm4": ["-DTHA4_L016_CFG=4,4,1,2,1", "-DTHA4_FACE16_CFG=4,4,1,4"],     # front kernel with 16 waves (rows split over four)
    "frontm4_l1w8x2": ["-DTHA4_L016_CFG=4,4,1,2,1", "-DTHA4_FACE16_CFG=4,4,1,4", "-DTHA4_L116_CFG=4,2,1,1,1,2"],
    "frontm4_l1w8x2_wait0": ["-DTHA4_L016_CFG=4,4,1,2,1", "-DTHA4_FACE16_CFG=4,4,1,4", "-DTHA4_L116_CFG=4,2,1,1,1,2", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    "all16": ["-DTHA4_L016_CFG=4,4,1,2,1", "-DTHA4_FACE16_CFG=4,4,1,4", "-DTHA4_L116_CFG=4,2,1,1,1,2", "-DTHA4_L216P_CFG=16,64,1"],
    "all16_wait0": ["-DTHA4_L016_CFG=4,4,1,2,1", "-DTHA4_FACE16_CFG=4,4,1,4", "-DTHA4_L116_CFG=4,2,1,1,1,2", "-DTHA4_L216P_CFG=16,64,1", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    "l1w4_wait0": ["-DTHA4_L116_CFG=4,1,1,1,1,2", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    "l1w8x2_wait0": ["-DTHA4_L116_CFG=4,2,1,1,1,2", "-mllvm", "-amdgpu-waitcnt-forcezero=1"],
    "plainsplit": ["-DTHA4_PLAIN_SPLIT"],          # hi/lo split as `lo = fp16(v - float(hi))` (8 instructions per pair instead of 3-4 with v_fma_mix)
    "hook": ["-DTHA4_L2_HOOK"],                      # with the level-2 code-object hook of the fault hunt (tools/hunt/check_co.py)
    # round 4: the dependent-round-trip work, each switch back to the form it replaced (results identical either way; profiles/r04_full_conv_tile_reading.md sections 10-13)
    "no_kernarg_warm": ["-DTHA4_NO_KERNARG_WARM"],                 # without warm_kernarg(): the argument block read as a chain of dependent cold misses
    "no_tap_pipeline": ["-DTHA4_TAP_PIPELINE=0"],                  # student first layers: the tap requests as the compiler orders them
    "no_bias_ahead": ["-DTHA4_BIAS_AHEAD=0"],                      # student streamed layers: scale + biases requested in the epilogue (behind the next chunk's fetch)
    "small_epi_late": ["-DTHA4_SMALL_PREFETCH_EPI(PG,POOL)=0"],    # conv_small_kernel: bias + activation codes requested after the partial-sum exchange (still together)
    "prio": ["-DTHA4_PHASE_PRIO=1"],     # s_setprio 1 in the VALU phases (sine / staging epilogues), 0 in the MFMA phases
    # round 5 (profiles/r05_boundaries_reading.md, r05_full_b1_reading.md; the run-time switches of the round are environment variables read under THA4_TUNING=1,
    # A/B them with tools/ab_full.py name=default@THA4_TUNING=1,VAR=1: THA4_SIDE_STREAM, THA4_MOMENT_ACC [+ THA4_ACC_MIN_TILES=16], THA4_NO_XCD_REMAP, THA4_NORM_TILE_SPLIT)
    "identity_pixels": ["-DTHA4_IDENTITY_PIXELS=1"],               # pixel = MFMA column in the convolution windows (rounds 2-4: 22-30 % LDS bank-conflict cycles)
    "norm8": ["-DTHA4_NORM_LOADS_IN_FLIGHT=8"],                    # norm_finalize_kernel with eight moment loads in flight (measured -0.5 %)
    # round 6: level 1 with the activations in registers (level1_16r_kernel<WAVES, PG, SLOTS>) against the LDS-activation form it replaced
    "l1regs0": ["-DTHA4_L1_REGS=0"], "l1r816": ["-DTHA4_L116R_CFG=8,1,6"], "l1r413": ["-DTHA4_L116R_CFG=4,1,3"], "l1r824": ["-DTHA4_L116R_CFG=8,2,4"],
    "l1r823": ["-DTHA4_L116R_CFG=8,2,3"],
    # round 6: face + level 0 as two 4-wave register-resident workgroups per CU (front16r_kernel<level-0 blocks per chunk, level-0 slots, face slots>)
    "frontregs0": ["-DTHA4_FRONT_REGS=0"], "fr6649": ["-DTHA4_FRONT16R_CFG=6,6,4,9"], "fr12384": ["-DTHA4_FRONT16R_CFG=12,3,8,4"], "fr6584": ["-DTHA4_FRONT16R_CFG=6,5,8,4"],
    "zwt0": ["-DTHA4_Z_WRITE_THROUGH=0"],         # plain instead of write-through (sc1) stores of the z1 / z2 hand-off images
    "spread0": ["-DTHA4_RING_SPREAD=0"],          # the LDS-DMA copies a ring barrier releases in a burst behind it instead of one per step under the MFMAs
    "allregs0": ["-DTHA4_FRONT_REGS=0", "-DTHA4_L1_REGS=0"],
    "tilewt0": ["-DTHA4_TILE_OUT_WT=0"],           # full model: plain instead of write-through (sc1) output stores of conv_tile_kernel (A/B: tools/ab_full.py)
    "pointwt0": ["-DTHA4_POINT_OUT_WT=0"],         # ... and of conv_point_kernel
    "l1taps1": ["-DTHA4_L1_TAPS_FIRST=1"],         # level1_16r_kernel: the first two tap batches requested in front of the prologue
    "l0prio1": ["-DTHA4_FRONT_L0_PRIO=1"], "l0prio2": ["-DTHA4_FRONT_L0_PRIO=2"],   # front16r_kernel: level-0 waves above the face waves of their SIMDs
    "cwait": ["-DTHA4_NEVER_BUILT_HERE"],                                           # (a copy of an earlier build kept for a same-box A/B: never rebuilt by `build`)
    "stamps": ["-DTHA4_STAMPS"],                                                  # in-kernel time stamps (tools/stamps_student.py)
    "pf1": ["-DTHA4_REGS_PREFETCH=1"], "pf3": ["-DTHA4_REGS_PREFETCH=3"],        # A-fragment look-ahead of the register-resident kernels in steps (default 2)
    "l1r423": ["-DTHA4_L116R_CFG=4,2,3"], "tap3": ["-DTHA4_TAP_BATCH=3"], "tap6": ["-DTHA4_TAP_BATCH=6"],
}
if os.environ.get("THA4_SWEEP_VARIANTS"):      # comma-separated subset
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["THA4_SWEEP_VARIANTS"].split(",")}


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for name, flags in VARIANTS.items():            # all variants at once: one hipcc process each
        out = os.path.join(OUT, f"libtha4_{name}.so")
        base = [] if "-DTHA4_PACKED_FP32_BUILD" in flags else NOPK      # variants are relative to the shipped flags (no packed fp32)
        if any("THA4_ABLATE_" in f or "THA4_HUNT_" in f for f in flags):
            base = base + TUNE
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC, "-I",
               os.path.join(ROOT, "include")] + base + flags + [os.path.join(CSRC, "tha4_capi.hip"), "-o", out]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for name, p in procs:
        _, err = p.communicate()
        print(name, "OK" if p.returncode == 0 else "FAILED\n" + err[-2000:])


def run(steps):
    rows = []
    for name in VARIANTS:
        lib = os.path.join(OUT, f"libtha4_{name}.so")
        if not os.path.exists(lib):
            continue
        env = dict(os.environ, THA4_HIP_LIB=lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "100",
                            "--cpu-seconds", "0", "--profile-frames", "50"], capture_output=True, text=True, env=env,
                           timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", r.stderr[-1500:])
            continue
        j = json.loads(line[-1])
        km = j["roofline"]["kernel_ms"]
        rows.append((name, j["value"], km))
        print(f"{name:10s} fps {j['value']:8.1f}  " + "  ".join(f"{k} {v*1000:6.1f}us" for k, v in km.items()), flush=True)
        if "--parity" in sys.argv:
            r2 = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_student_gpu.py"), "-x", "-q", "-k",
                                 "output0_parity or all_six", "-s"], capture_output=True, text=True, env=env, timeout=600)
            print("   parity:", [l for l in r2.stdout.splitlines() if "passed" in l or "failed" in l or "PARITY" in l][-6:], flush=True)
    return rows


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 600
        run(steps)
