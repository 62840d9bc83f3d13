#!/usr/bin/env python3
"""Benchmark of the poser hot path through the drop-in Poser API (tha4_amd.poser.modes -> include/tha4_hip.h -> gfx950
kernels), for the configurations BASELINE.json names:

    python bench.py                                     configs[1]  lambda_00 student, batch-1 stream, 1 GPU (the headline)
    python bench.py --model full                        configs[2]  full THA4 model, batch 1, lambda_00 image
    torchrun ... bench.py --gpus 8 --batch 32           configs[3]  one student instance per GPU (lambda_00 / lambda_01
                                                                    alternating by rank), batches of 32 poses, gather
    torchrun ... bench.py --gpus 8 --model full --batch 8   configs[4]  full model, 8 random images + poses per GPU and step
                                                                    (64 frames per step over 8 GPUs), frame-parallel, gather

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [...]

A step = ONE Poser.pose() call on a batch of B frames (B = 1 by default), back-to-back on the rank's current stream,
inputs resident in HBM before the timed region.  With N > 1 every rank runs K steps of its own (weak scaling: frames
and character instances are independent, no collective on the compute path) and the finished frames - fp32, or RGBA8
after the display epilogue with --rgba8-gather - are gathered to rank 0 chunk by chunk over RCCL on a side stream
(the only exchange the path has; --no-gather skips it).  Timing: W untimed steps, then exactly K steps between
barrier + synchronize on both sides, maximum over ranks; rank 0 prints ONE JSON line.
"""
import argparse
import glob
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import tha4_amd  # noqa: E402,F401
from tha4_amd import synthetic  # noqa: E402
from tha4_amd.poser.modes import mode_07, mode_14  # noqa: E402
from tha4_amd.sharding import FrameShardedStream, tapered_schedule  # noqa: E402
from tha4_amd.weights import split_flat_weights  # noqa: E402

# Algorithmic work of the reference's forward passes as written (SURVEY.md §8d, 2*MAC), per 512x512 frame and per
# kernel of the student implementation.
GFLOP_FRAME = 37.885
# HBM bytes a launch of each student kernel cannot avoid in this design, per frame: fp32 hand-off images z1 [192 ch x 128^2] = 12.58 MB and
# z2 [96 ch x 256^2] = 25.17 MB, the 128^2 face patch, the frame in and the posed frame out (4.19 MB each), the kernel's packed weights once
ALGO_BYTES_KERNEL = {"front": 12.58e6 + 0.26e6 + 1.50e6, "level1": 12.58e6 + 25.17e6 + 0.26e6, "level2": 25.17e6 + 2 * 4.19e6 + 0.26e6 + 0.08e6}
GFLOP_KERNEL = {"face": 3.947, "level0": 6.924, "level1": 11.726, "level2": 15.288}
# What the student kernels actually execute after pose folding + commuting the x2 upsample with the next level's
# first layer: stated separately, never used for `roofline.achieved`.
GFLOP_EXECUTED_FRAME = 27.46
GFLOP_FULL_COLD = 645.90          # FlopCounterMode on the reference modules
GFLOP_FULL_STEADY = 625.90        # eyebrow decomposer cached (mode_07.py:56-67)
# The contractions issue v_mfma_f32_16x16x32_f16 on fp16 hi/lo halves of fp32 operands, three MFMAs per fp32-accurate
# product block (hi*hi + hi*lo + lo*hi, fp32 accumulate).  `roofline.peak` is the dense fp16 MFMA peak of the
# instruction actually issued (MI355X_MICROARCH.md: ~2.5 PFLOP/s); `roofline.achieved` stays ALGORITHMIC (every
# multiply-add of the reference counted once), so the ceiling of this arithmetic is peak/3 - reported next to it.
PEAK_F16_MFMA_TFLOPS = 2500.0
MFMA_PASSES = 3
PEAK_FP32_MFMA_TFLOPS = 157.3      # v_mfma_f32_16x16x4_f32 dense peak (what an exact-fp32 single pass could reach)
DTYPE = "f32 (I/O and accumulate; contractions: fp16 hi/lo 3-pass operands on v_mfma_f32_16x16x32_f16, 22-bit operands)"
KERNEL_NAMES = ["posebias", "face", "level0", "level1", "level2"]

POSE_LO = np.array([0.0] * 37 + [-1.0] * 7 + [0.0], dtype=np.float32)
POSE_HI = np.ones(45, dtype=np.float32)


def make_poses(n, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 45, generator=g).numpy()
    return torch.from_numpy((POSE_LO + (POSE_HI - POSE_LO) * u).astype(np.float32))


def load_character(name):
    """Student weights + image of one shipped character (tests/golden fixtures made from the reference's .pt / .png)."""
    g = os.path.join(ROOT, "tests", "golden")
    w = dict(np.load(os.path.join(g, f"student_{name}_weights.npz")))
    io = np.load(os.path.join(g, f"student_{name}_io.npz"))
    return w, io["image_f32"]


def newest_profile(pattern):
    """HBM-side bytes come from the rocprofv3 PMC passes committed under profiles/ (bench.py cannot run rocprofv3 on
    itself): the newest round's machine-readable summary, produced by tools/traffic_json.py from the PMC CSVs."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def rocprof_kernel_avgs(pattern, names):
    """Average launch durations (ms) from the newest committed `rocprofv3 --kernel-trace --stats` summary under profiles/, for the
    kernels whose name contains one of `names` ({label: substring}) - quoted NEXT to the in-bench HIP-event times, which carry the
    event overhead (their sum exceeds the step)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    import csv
    out = {}
    try:
        for row in csv.DictReader(open(files[-1])):
            for label, sub in names.items():
                if sub in row.get("Name", ""):
                    out[label] = round(float(row["AverageNs"]) / 1e6, 5)
    except (OSError, KeyError, ValueError):
        return None
    return {"file": os.path.relpath(files[-1], ROOT), "avg_ms": out} if out else None


def kernel_traffic_bytes(prof, name):
    k = (prof or {}).get("kernels", {}).get(name)
    if not k:
        return None
    return int(round((2.0 * k["fetch_kib"] + k["write_kib"]) * 1024))      # gfx950: FETCH_SIZE x2 (wide reads) + WRITE_SIZE


def roofline_evidence(roofline, dom, gflop_launch, event_ms, B):
    """What bounds the dominant kernel, from evidence instead of a constant label (round-5 review, task 5): `bound` stays the roofline the fraction is
    priced against - chosen by arithmetic intensity (algorithmic FLOP / counted HBM bytes against the ridge peak_flops / peak_bw) - and `limiter` names what
    the committed counters say the kernel actually waits for, with the one number that shows it.  `mfma_busy`, the wave-time split and VALU per MFMA come
    from the newest committed PMC capture (profiles/r*_student_b*_pmc.json, tools/pmc_json.py); `kernel_rocprof_frac` re-prices `frac` on the rocprofv3
    average of the committed capture instead of the in-bench HIP-event time."""
    ev = {}
    short = dom.split(" ")[0]
    traffic = roofline.get("traffic")
    # the roof the kernel sits under follows from its ALGORITHMIC intensity: as-written FLOP of the launch / the HBM bytes the launch cannot avoid in this
    # design (its hand-off images, the frame in and out, its weights once).  The counted traffic (2-4x that: re-fetched upsample rows across XCDs) is
    # reported beside it - it prices waste, it does not move the kernel to the other roof (at 1.3 TB/s of 8 the HBM roof is 6x away)
    algo = ALGO_BYTES_KERNEL.get(short)
    ev["ridge_flop_per_byte"] = round(PEAK_F16_MFMA_TFLOPS * 1e12 / 8.0e12, 1)
    if algo:
        ai = gflop_launch * 1e9 / (algo * B)
        ev["algorithmic_bytes"] = int(algo * B)
        ev["arithmetic_intensity_flop_per_byte"] = round(ai, 1)
        ev["bound"] = "mfma" if ai >= ev["ridge_flop_per_byte"] else "hbm"
    if traffic:
        ev["counted_intensity_flop_per_byte"] = round(gflop_launch * 1e9 / traffic, 1)
        ev["hbm_frac_counted"] = round(traffic / (event_ms * 1e-3) / 8.0e12, 4)
    rp = roofline.get("kernel_ms_rocprof") or {}
    ms = (rp.get("avg_ms") or {}).get(dom)
    if ms:
        ev["kernel_rocprof_frac"] = round(gflop_launch / ms / PEAK_F16_MFMA_TFLOPS, 4)
        ev["kernel_rocprof_what"] = f"as-written GFLOP of the launch / rocprofv3 average launch duration ({rp.get('file')}) / peak"
    pmc, pmc_file = newest_profile("r*_student_b1_pmc.json" if B == 1 else "r*_student_b32_pmc.json")
    k = (pmc or {}).get("kernels", {}).get(short)
    ev.update(pmc_evidence(k, pmc_file))
    return ev


def pmc_evidence(k, source):
    """`mfma_busy`, the wave-time split, VALU per MFMA and the `limiter` sentence for one kernel entry of a committed PMC summary (tools/pmc_json.py)."""
    if not k:
        return {"limiter": "unknown: no PMC capture committed for this kernel (tools/pmc_json.py)"}
    ev = {"mfma_busy": k.get("mfma_busy"), "wave_time": {x: k.get(x) for x in ("active", "issue_stall", "parked") if k.get(x) is not None},
          "valu_per_mfma": k.get("valu_per_mfma"), "pmc_source": source}
    busy = k.get("mfma_busy") or 0.0
    parked, stall = k.get("parked") or 0.0, k.get("issue_stall") or 0.0
    if busy >= 0.6:
        ev["limiter"] = f"mfma: matrix pipe busy {busy:.0%} of the launch"
    elif parked >= stall:
        ev["limiter"] = (f"memory / barrier waits: a resident wave is parked at s_waitcnt or s_barrier {parked:.0%} of its time "
                         f"(matrix pipe busy {busy:.0%})")
    else:
        ev["limiter"] = (f"instruction issue: MFMA and VALU share a SIMD's issue port - {k.get('valu_per_mfma')} VALU per MFMA, waves issue-stalled {stall:.0%} "
                         f"of their time (matrix pipe busy {busy:.0%})")
    return ev


def exchange_identity(backend):
    """What the multi-rank exchange ran on, for a reader who only has the JSON line: library version and every environment variable that steers it."""
    out = {"backend": backend, "torch": torch.__version__,
           "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "TORCH_NCCL_", "HSA_", "GLOO_", "HIP_VISIBLE", "ROCR_VISIBLE", "CUDA_VISIBLE"))}}
    if backend == "nccl":
        try:
            out["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:             # noqa: BLE001
            out["rccl_version"] = f"unavailable: {e}"
        out["devices"] = torch.cuda.device_count()
    return out


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline(run_frame, what, budget_s, max_frames, kind="port"):
    """The CPU path timed beside the GPU path, fp32, `torch.set_num_threads(k)` for k in {1, 8, 32, all
    physical cores}; the BEST k is the reported baseline (too many threads is slower for these small convolutions).
    `kind` = "reference": the UNMODIFIED reference (oracle/reference_runner.py, wherever /root/reference exists - the build
    container); "port": the oracle's torch-functional restatement (the same ATen ops the reference dispatches; what the GPU box,
    which has no /root/reference, can run).  Bounded sample: ~budget_s of CPU work in total."""
    phys = physical_cores()
    ks = sorted({k for k in (1, 8, 32, phys) if k <= max(phys, 1)})
    prev = torch.get_num_threads()
    by = {}
    per_k = budget_s / max(len(ks), 1)
    try:
        for k in ks:
            torch.set_num_threads(k)
            run_frame(0)                                           # warm-up (thread pool, allocator)
            n, t0 = 0, time.perf_counter()
            while True:
                run_frame(n)
                n += 1
                dt = time.perf_counter() - t0
                if dt >= per_k or n >= max_frames:
                    break
            by[k] = {"frames": n, "seconds": round(dt, 2), "fps": round(n / dt, 4), "ms_per_frame": round(1e3 * dt / n, 1)}
    finally:
        torch.set_num_threads(prev)
    best = max(by, key=lambda k: by[k]["fps"])
    return {"value": by[best]["fps"], "unit": "frames/s", "cores": best, "kind": kind,
            "sample": f"{what}; {by[best]['frames']} frames at the best thread count ({sum(v['seconds'] for v in by.values()):.0f} s of CPU work over thread counts {ks})",
            "ms_per_frame": by[best]["ms_per_frame"], "by_threads": {str(k): v for k, v in by.items()},
            "cpu_model": cpu_model_string(), "physical_cores": phys, "logical_cpus": os.cpu_count()}


# ------------------------------------------------------------------------------------------------------------------
# the two workloads: everything a step needs, resident in HBM
# ------------------------------------------------------------------------------------------------------------------
class StudentWork:
    kind = "student"

    def __init__(self, dev, rank, B, steps, characters):
        name = characters if characters != "alternate" else ("lambda_00", "lambda_01")[rank % 2]
        self.character = name
        self.w, self.image_np = load_character(name)
        face, body = split_flat_weights(self.w)
        self.poser = mode_14.create_poser_from_state_dicts(dev, face, body, max_batch=max(B, 4))
        self.image = torch.from_numpy(self.image_np).to(dev)
        self.poses_cpu = make_poses(steps * B, seed=1234 + rank)
        self.poses = self.poses_cpu.to(dev).reshape(steps, B, 45)
        self.B = B
        self.poser.get_modules()

    def step(self, i, out=None):
        return self.poser.pose(self.image, self.poses[i] if self.B > 1 else self.poses[i, 0], out=out)

    def step_rgba8(self, i, out=None):
        """pose + display epilogue fused in the composing kernel (tha4_display): uint8 [B,512,512,4], no fp32 frame written"""
        return self.poser.pose_display_rgba8(self.image, self.poses[i] if self.B > 1 else self.poses[i, 0], out=out)


class FullWork:
    kind = "full"

    def __init__(self, dev, rank, B, steps, steady, exact_fp32=False):
        self.poser = mode_07.create_poser_from_state_dicts(dev, synthetic.synth_full_weights(), max_batch=B, exact_fp32=exact_fp32)
        self.B, self.steady = B, steady
        _, self.image_np = load_character("lambda_00")
        if B == 1:
            # configs[2]: the lambda_00 image; steady = image unchanged (decomposer cached), cold = it changes every frame
            self.images = [torch.from_numpy(self.image_np).to(dev)]
        else:
            # configs[4]: B random images per step (SURVEY.md §8d config 5 recipe, seed 99 + rank); they change every
            # batch, so the eyebrow-decomposer cache never hits (cold cost, 645.9 GFLOP/frame)
            self.images = [torch.from_numpy(synthetic.random_rgba_images(B, seed=99 + 1000 * rank + j)).to(dev) for j in range(2)]
        self.poses = make_poses(steps * B, seed=77 + rank).to(dev).reshape(steps, B, 45)
        self.poser.get_modules()

    def step(self, i, out=None):
        if self.B == 1:
            r = self.poser.pose(self.images[0], self.poses[i, 0], image_changed=not self.steady)
        else:
            r = self.poser.pose(self.images[i % 2], self.poses[i])
        if out is not None:
            out.copy_(r)
            return out
        return r

    def step_rgba8(self, i, out=None):
        if self.B == 1:
            r = self.poser.pose_display_rgba8(self.images[0], self.poses[i, 0], image_changed=not self.steady)
        else:
            r = self.poser.pose_display_rgba8(self.images[i % 2], self.poses[i])
        if out is not None:
            out.copy_(r)
            return out
        return r


def timed_steps(work, lo, hi):
    for i in range(lo, hi):
        out = work.step(i)
    return out


def gather_plan(args, K, B):
    """Frames per gather round for a rank that poses K steps of B frames: rounds of `chunk` frames (default: a fifth of the rank's
    frames, at most 32) ending in a taper, so that the last round - the only exchange nothing overlaps - is at most 5 % of the
    rank's frames (sharding.tapered_schedule; round-4 review: `--steps 20` used to end on a 4-frame round, 20 % of the stream)."""
    want_chunk = args.gather_chunk if args.gather_chunk is not None else min(32, max(1, K * B // 5))
    chunk = max(B, want_chunk // B * B)
    return chunk, tapered_schedule(K * B, chunk, unit=B, tail_frac=0.05)


def measure(work, args, dev, rank, world, K, W, B, dist, return_frames=False, fresh=True, variant=None, rehearse=None):
    """Settle, W warm-up steps, then EXACTLY K timed steps between barriers (N > 1: with the gather of the finished frames to
    rank 0 inside the timed region, after an untimed rehearsal of the exchange).  Returns this rank's elapsed seconds (and, for
    the tests, what rank 0 gathered).  `fresh=False` (the `repeats` of the JSON line) times the same K steps again without
    settle / warm-up / rehearsal.  `variant` = "fp32" | "rgba8" | "none" overrides the exchange the command line asks for (the
    N > 1 line reports all three); `rehearse` forces / suppresses the untimed rehearsal of the exchange (default: with `fresh`).
    Device-agnostic: tests/test_bench_stream_gloo.py runs it on CPU tensors over gloo."""
    cuda = dev.type == "cuda"
    if variant is None:
        variant = "none" if args.no_gather else ("rgba8" if args.rgba8_gather else "fp32")
    if not hasattr(work, "exchange_log"):
        work.exchange_log = {}
    gather = world > 1 and variant != "none"
    rgba8 = variant == "rgba8"
    if rehearse is None:
        rehearse = fresh
    frames = None

    def barrier():
        if cuda:
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            if cuda:
                torch.cuda.synchronize(dev)

    settle_steps = 0
    with torch.no_grad():
        if fresh and args.settle_seconds > 0:             # same frames as the warm-up steps, not counted anywhere
            t_end = time.perf_counter() + args.settle_seconds
            while time.perf_counter() < t_end:
                timed_steps(work, 0, max(1, min(W, 8)))
                settle_steps += max(1, min(W, 8))
                if cuda:
                    torch.cuda.synchronize(dev)
        if fresh:
            work.settle_steps = settle_steps
            timed_steps(work, 0, W)
        if gather:
            chunk, schedule = gather_plan(args, K, B)
            work.gather_schedule = schedule
            shape, dtype = ((512, 512, 4), torch.uint8) if rgba8 else ((4, 512, 512), torch.float32)

            # frames leave the rank as fp32 [4,512,512] or - display epilogue FUSED into the composing kernel (tha4_display) -
            # as uint8 [512,512,4]: a quarter of the bytes, and no fp32 frame is ever written
            step = work.step_rgba8 if rgba8 else work.step

            def frame_fn(lo, hi):          # global frame ids of this rank start at rank*K*B; whole steps only
                base = rank * K * B
                blk = torch.empty((hi - lo,) + shape, dtype=dtype, device=dev)
                for f in range(lo, hi, B):                                   # straight into the gather block
                    step(W + (f - base) // B, out=blk[f - lo:f - lo + B])
                return blk

            # untimed rehearsal of the exchange - one full gather round and one ragged tail per rank - so that RCCL's
            # point-to-point connections (set up lazily on first use) exist before the clock starts
            def rehearsal_fn(lo, hi):
                blk = torch.empty((hi - lo,) + shape, dtype=dtype, device=dev)
                for f in range(0, hi - lo, B):
                    step(0, out=blk[f:f + B])
                return blk

            if rehearse:          # timed from outside the region: the first one of a run carries RCCL's connection set-up
                barrier()
                tr = time.perf_counter()
                FrameShardedStream(rehearsal_fn, total=(chunk + B) * world, frame_shape=shape, dtype=dtype, device=dev, chunk=chunk, gather=True).run()
                barrier()
                work.exchange_log.setdefault(variant, {}).setdefault("rehearsal_s", []).append(round(time.perf_counter() - tr, 4))
            if return_frames:              # tests: archive the whole (short) stream on rank 0
                stream = FrameShardedStream(frame_fn, total=K * B * world, frame_shape=shape, dtype=dtype, device=dev, chunk=chunk, gather=True,
                                            schedule=schedule)
                gathered = stream.allocate_result()       # allocated outside the timed region
            else:
                # the root is a STREAM: a ring of 3 gather rounds (world x chunk frames each; 8 x 32 x 4 MiB x 3 = 3 GiB) whose
                # consumer sees every frame once - the bench's consumer only counts them (a real one encodes / composites)
                work.delivered = 0

                def consume(lo, hi, frames_view):
                    work.delivered += hi - lo

                stream = FrameShardedStream(frame_fn, total=K * B * world, frame_shape=shape, dtype=dtype, device=dev, chunk=chunk, gather=True,
                                            on_chunk=consume, ring_slots=3, schedule=schedule, record_rounds=True)
                work.ring_bytes = stream.ring_bytes()
                gathered = None
            barrier()
            t0 = time.perf_counter()
            gathered = stream.run(gathered)
            barrier()
            t1 = time.perf_counter()
            if return_frames:
                frames = gathered
            elif rank == 0 and work.delivered != K * B * world:
                raise RuntimeError(f"gather delivered {work.delivered} of {K * B * world} frames")
            if not return_frames:      # what THIS form of the exchange held and did (the next variant's region overwrites work.ring_bytes / .delivered)
                log = work.exchange_log.setdefault(variant, {})
                log["ring_bytes"], log["delivered"] = work.ring_bytes, work.delivered
                log["rounds"] = stream.round_report()          # the last region of the variant (outside the clock: it synchronises)
            del gathered
        else:
            barrier()
            t0 = time.perf_counter()
            timed_steps(work, W, W + K)
            barrier()
            t1 = time.perf_counter()
    return (t1 - t0, frames) if return_frames else t1 - t0


def run_regions(work, args, dev, rank, world, K, W, B, dist):
    """The timed regions of one bench run: `value`'s region, its `--repeats`, and - N > 1 - the same K steps under each form of the
    exchange.  Returns (elapsed of the first region, frames/s of the repeats, per-variant dict or None, name of the primary form).
    Shared by the real run and `--stub-gloo` (tests)."""
    per_rank = {}                          # variant -> the ranks' own seconds of its last region (first-contact diagnostics, N > 1)

    def region(**kw):                      # one timed region: this rank's seconds -> maximum over ranks
        e = measure(work, args, dev, rank, world, K, W, B, dist, **kw)
        if world > 1:
            t = torch.zeros(world, dtype=torch.float64, device=dev)
            t[rank] = e
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            es = [float(x) for x in t.tolist()]
            per_rank[kw.get("variant") or primary] = es
            e = max(es)
        return e

    primary = "none" if args.no_gather else ("rgba8" if args.rgba8_gather else "fp32")
    elapsed = region()
    primary_rank_seconds = list(per_rank.get(primary, []))
    # `value` is the region above; the SAME K steps are timed `--repeats` more times so that a short region (20 steps of the student
    # stream = 2.5 ms) carries its own spread.  Same barriers, maximum over ranks per repeat.
    rep_fps = [K * B * world / region(fresh=False) for _ in range(max(0, args.repeats))]
    # N > 1: the same K steps under each form of the exchange - fp32 frames (what Poser.pose() returns: the default and `value`),
    # RGBA8 frames (display epilogue fused into the composing kernel: a quarter of the bytes) and no exchange at all - so that a
    # scaling figure below target is attributable to the gather (xGMI / root ingest) or to the compute path (round-4 review)
    variants = None
    if world > 1:
        variants = {}
        for v in ("fp32", "rgba8", "none"):
            if v == primary:
                es = [elapsed] + [K * B * world / f for f in rep_fps]
            else:
                region(fresh=False, rehearse=True, variant=v)              # untimed: first use of this form (buffers, connections)
                es = [region(fresh=False, variant=v) for _ in range(max(1, min(3, args.repeats)))]
            e = float(np.median(es))
            frame_bytes = {"fp32": 4 * 512 * 512 * 4, "rgba8": 512 * 512 * 4, "none": 0}[v]
            variants[v] = {"fps": round(K * B * world / e, 2), "per_gpu_fps": round(K * B / e, 2), "regions": len(es),
                           "root_ingest_GBps": round((world - 1) * K * B * frame_bytes / e / 1e9, 2),
                           "per_sender_GBps": round(K * B * frame_bytes / e / 1e9, 2)}
            rs = primary_rank_seconds if v == primary else per_rank.get(v, [])
            if rs:                         # every rank's OWN clock over its K steps (the line's time is their maximum): a slow rank / a slow link shows here
                rf = [K * B / x for x in rs]
                variants[v]["per_rank_fps"] = {"min": round(min(rf), 2), "max": round(max(rf), 2), "slowest_rank": int(np.argmin(rf)), "all": [round(x, 2) for x in rf],
                                               "what": "first region for the primary form, last region otherwise"}
            log = getattr(work, "exchange_log", {}).get(v, {})
            if v != "none":                # rank 0's view of the exchange, round by round (events on the gather's side stream)
                variants[v]["root_ring_bytes"] = log.get("ring_bytes")
                variants[v]["rehearsal_s"] = log.get("rehearsal_s")
                variants[v]["rounds"] = log.get("rounds")
    return elapsed, rep_fps, variants, primary


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["student", "full"], default="student",
                    help="student = configs[1]/[3] (default: the headline metric); full = configs[2]/[4]")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="frames per Poser.pose() call (1 = configs[1]/[2]; 32 = configs[3]; 8 = configs[4] per GPU)")
    ap.add_argument("--steps", type=int, default=None, help="timed pose() calls per rank (default: student 2000 at batch 1, 64 at batch > 1; full 100 / 20)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed pose() calls per rank (default 200 student / 5 full)")
    ap.add_argument("--characters", default=None, help="student: lambda_00 | lambda_01 | alternate (by rank; default for --batch > 1)")
    ap.add_argument("--cold", action="store_true", help="full model, batch 1: the image changes every frame (no decomposer cache)")
    ap.add_argument("--gather-chunk", type=int, default=None,
                    help="frames per gather round (N>1; rounded to a multiple of --batch; default: a fifth of the rank's frames, "
                         "at most 32 - a short run still overlaps its exchange with compute)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the gather of finished frames")
    ap.add_argument("--rgba8-gather", action="store_true", help="N>1: display epilogue (sRGB, uint8 HWC) before the gather: 4x fewer bytes")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline budget in seconds of CPU work (0 disables; N=1 only)")
    ap.add_argument("--profile-frames", type=int, default=100, help="student: steps for the per-kernel HIP-event pass")
    ap.add_argument("--d2h-frames", type=int, default=500, help="student N=1 B=1: steps for the RGBA8 + D2H inclusive measurement (0 = skip)")
    ap.add_argument("--exact-frames", type=int, default=500, help="student N=1 B=1: steps for the exact-fp32 generation A/B (0 = skip)")
    ap.add_argument("--full-frames", type=int, default=30, help="student N=1 B=1: frames of the appended full-model measurement (0 = skip)")
    ap.add_argument("--batched-steps", type=int, default=24,
                    help="student N=1 B=1: steps of the appended batched measurements - configs[3] (batch 32) and configs[4] (full model, batch 8), "
                         "one GPU's share of each (0 = skip)")
    ap.add_argument("--settle-seconds", type=float, default=0.3,
                    help="untimed frames posed for this long BEFORE the W warm-up steps, so that a short run (--steps 20 --warmup 5) "
                         "is timed at the GPU's steady clocks like the stream it samples (0 = none)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="time the same K steps this many more times after the region that gives `value`; the line reports median / min / max")
    ap.add_argument("--stub-gloo", action="store_true",
                    help="TEST HARNESS ONLY (tests/test_bench_launch_gloo.py): run the multi-rank driver logic - torchrun ranks, sharding, gather "
                         "to rank 0, barriers, max-over-ranks timing, the JSON line - over gloo on CPU tensors with a stub that POSES NOTHING "
                         "(constant frames).  Measures nothing: the line says \"data\": \"stub\" and carries no roofline / cpu_baseline.")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if args.stub_gloo:
        return stub_gloo_main(args, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    student = args.model == "student"
    B = max(1, args.batch)
    # defaults: configs[1] 2000-frame stream; configs[3] 64 batches of 32; configs[2] 100 frames;
    # configs[4] 20 batches of 8 per GPU
    K = args.steps if args.steps is not None else ((2000 if B == 1 else 64) if student else (100 if B == 1 else 20))
    W = args.warmup if args.warmup is not None else ((200 if B == 1 else 8) if student else (5 if B == 1 else 3))
    characters = args.characters or ("lambda_00" if B == 1 else "alternate")
    work = StudentWork(dev, rank, B, K + W, characters) if student else FullWork(dev, rank, B, K + W, steady=not args.cold)
    gather = world > 1 and not args.no_gather

    elapsed, rep_fps, variants, primary = run_regions(work, args, dev, rank, world, K, W, B, dist)

    if rank == 0:
        fps = K * B * world / elapsed
        result = {
            "metric": ("frames/sec (whole job) on 512x512 RGBA + 45-dim pose, " + ("distilled student" if student else "full THA4 model")),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(1e3 * elapsed / K, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE, "per_gpu_fps": round(fps / world, 2), "settle_seconds": args.settle_seconds,
            "settle_frames": getattr(work, "settle_steps", 0) * B,
            "repeats": ({"n": len(rep_fps), "what": f"the same {K} timed steps again, after `value` (frames/s of each region)",
                         "median": round(float(np.median(rep_fps)), 2), "min": round(min(rep_fps), 2), "max": round(max(rep_fps), 2),
                         "all": [round(v, 2) for v in rep_fps]} if rep_fps else None)}
        par = {"frames_per_gpu": K * B, "batch": B, "parallelism": f"frame-parallel x{world}",
               "gather": ("rgba8 (display epilogue fused into the composing kernel)" if args.rgba8_gather else "fp32") if gather else False}
        if gather:                        # the PRIMARY form's ring (each variant's own is in gather_variants[v].root_ring_bytes)
            par["gather_root_ring_bytes"] = getattr(work, "exchange_log", {}).get(primary, {}).get("ring_bytes", getattr(work, "ring_bytes", None))
        if variants is not None:
            sched = getattr(work, "gather_schedule", None)
            par["gather_schedule"] = ({"frames_per_round": sched, "rounds": len(sched),
                                       "unoverlapped_tail_fraction": round(sched[-1] / max(1, K * B), 4),
                                       "what": "frames of every rank per gather round; round c's exchange runs on a side stream under round c+1's compute, "
                                               "only the last round's is exposed (sharding.tapered_schedule)"} if sched else None)
            result["exchange"] = exchange_identity("nccl")
            result["gather_variants"] = dict(variants, value_is=primary,
                                             what="median of the timed regions of the same K steps under each form of the exchange to rank 0 (whole-job frames/s); "
                                                  "root_ingest_GBps = bytes arriving at rank 0 from the other ranks / region time; `value` is the "
                                                  f"'{primary}' form's first region")
        if student:
            result["data"] = ("synthetic pose stream (seed 1234+rank, pose_parameters ranges); shipped student weights + character image "
                              "(tests/golden fixtures made from the reference's lambda_00 / lambda_01 .pt and .png)")
            wl = ("configs[1]: lambda_00 distilled student, batch=1 real-time stream, 512x512 RGBA, one Poser.pose() per frame" if (B == 1 and characters == "lambda_00") else
                  f"configs[3]: one distilled-student character instance per GPU ({characters}), batch={B} pose stream per Poser.pose() call")
            result["config"] = dict(workload=wl, characters=characters, **par)
            result.update(student_extras(args, work, dev, world, fps, K, W, B))
        else:
            result["data"] = ("synthetic seeded weights (tha4_amd.synthetic; the reference checkout ships none); " +
                              ("lambda_00 image fixture" if B == 1 else "random 512x512 RGBA images (seed 99+, change every batch)") + "; random poses")
            wl = (f"configs[2]: THA4 full model (5 networks), batch=1, lambda_00 image, {'cold (image changes every frame)' if args.cold else 'steady state (eyebrow decomposer cached)'}"
                  if B == 1 else f"configs[4]: THA4 full model, batch={B} random images + poses per GPU and step ({B * world} frames per step), frame-parallel, decomposer never cached")
            result["config"] = dict(workload=wl, **par)
            result.update(full_extras(args, work, dev, world, fps, B))
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class _StubWork:
    """--stub-gloo: stands in for StudentWork / FullWork and poses nothing - step i of rank r fills its B frames with 1000 r + i."""

    def __init__(self, rank, B):
        self.rank, self.B = rank, B

    def step(self, i, out=None):
        if out is None:
            out = torch.empty(self.B, 4, 512, 512)
        out.fill_(1000.0 * self.rank + i)
        return out

    def step_rgba8(self, i, out=None):
        if out is None:
            out = torch.empty(self.B, 512, 512, 4, dtype=torch.uint8)
        out.fill_((7 * self.rank + i) % 251)
        return out


def stub_gloo_main(args, rank, world):
    """The launch / sharding / exchange / timing / reporting path of `bench.py --gpus N` with gloo instead of RCCL and a stub instead
    of a poser (test harness; no GPU in the build container, no multi-GPU node for the builder).  Same code from `measure` on."""
    dev = torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo")
    B = max(1, args.batch)
    K = args.steps if args.steps is not None else 4
    W = args.warmup if args.warmup is not None else 1
    work = _StubWork(rank, B)
    if args.repeats == 5:                  # (the real run's default; the stub's regions are CPU fills of 4 MiB frames)
        args.repeats = 1
    elapsed, rep_fps, variants, primary = run_regions(work, args, dev, rank, world, K, W, B, dist)
    if rank == 0:
        gather = world > 1 and not args.no_gather
        sched = getattr(work, "gather_schedule", None)
        print(json.dumps({"metric": "STUB: driver logic only, nothing was posed", "value": round(K * B * world / elapsed, 2), "unit": "stub frames/s",
                          "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 5), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "stub",
                          "gather_variants": dict(variants, value_is=primary) if variants else None,
                          "exchange": exchange_identity("gloo") if world > 1 else None,
                          "config": {"workload": "stub (tests/test_bench_launch_gloo.py)", "frames_per_gpu": K * B, "batch": B,
                                     "parallelism": f"frame-parallel x{world}",
                                     "gather": ("rgba8" if args.rgba8_gather else "fp32") if gather else False,
                                     "gather_schedule": sched if world > 1 else None,
                                     "gather_root_ring_bytes": getattr(work, "exchange_log", {}).get(primary, {}).get("ring_bytes") if gather else None,
                                     "delivered": getattr(work, "exchange_log", {}).get(primary, {}).get("delivered") if gather else None}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def student_extras(args, work, dev, world, fps, K, W, B):
    poser, image, poses = work.poser, work.image, work.poses
    # per-kernel durations from HIP events recorded on the launch stream inside the C ABI
    poser.set_timing(True)
    acc = np.zeros(len(KERNEL_NAMES))
    whole = 0.0
    nprof = max(1, args.profile_frames)
    with torch.no_grad():
        for i in range(nprof):
            work.step(W + (i % K))
            for k in range(len(KERNEL_NAMES)):
                acc[k] += poser.last_kernel_ms(k)
            whole += poser.last_kernel_ms(-1)
    poser.set_timing(False)
    kernel_ms = {n: float(acc[k] / nprof) for k, n in enumerate(KERNEL_NAMES)}
    gflop = dict(GFLOP_KERNEL)
    if kernel_ms["face"] < 0.012 * B:       # face + level 0 share one launch (v2::front16_kernel): event slot 1 is empty, slot 2 is the merged kernel
        kernel_ms["front (level0 + face, one launch)"] = kernel_ms.pop("level0")
        gflop["front (level0 + face, one launch)"] = gflop.pop("level0") + gflop.pop("face")
        kernel_ms["face"] = 0.0
    dom = max(gflop, key=lambda n: kernel_ms[n])
    achieved = gflop[dom] * B / kernel_ms[dom]        # GFLOP / ms = TFLOP/s
    prof, prof_file = newest_profile("r*_student_b1_traffic.json")
    roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": PEAK_F16_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4),
                "mfma": "v_mfma_f32_16x16x32_f16 on fp16 hi/lo operand halves, 3 per product block, fp32 accumulate",
                "frac_of_split_ceiling": round(achieved * MFMA_PASSES / PEAK_F16_MFMA_TFLOPS, 4),
                "vs_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                "traffic": kernel_traffic_bytes(prof, dom.split(" ")[0]) if B == 1 else None,
                "traffic_unit": f"bytes/launch = 2 x FETCH_SIZE + WRITE_SIZE of the rocprofv3 PMC passes in {prof_file}",
                "traffic_per_frame": (sum(kernel_traffic_bytes(prof, n) or 0 for n in prof.get("kernels", {})) if (B == 1 and prof) else None),
                "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
                "kernel_ms_what": "HIP events on the launch stream inside the C ABI (they add ~3-4 us per kernel: a conservative frac)",
                "kernel_ms_rocprof": rocprof_kernel_avgs("r*_student_b1_kernel_stats.csv" if B == 1 else "r*_student_b32_kernel_stats.csv",
                                                         {"front (level0 + face, one launch)": "front16", "level1": "level1_16",
                                                          "level2": "level2_16p_kernel"}),
                "frame_event_ms": round(whole / nprof, 4),
                "whole_frame_achieved_tflops": round(fps / world * GFLOP_FRAME / 1e3, 3),
                "whole_frame_frac": round(fps / world * GFLOP_FRAME / 1e3 / PEAK_F16_MFMA_TFLOPS, 4),
                "algorithmic_gflop_per_frame": GFLOP_FRAME, "executed_gflop_per_frame": GFLOP_EXECUTED_FRAME}
    roofline.update(roofline_evidence(roofline, dom, gflop[dom] * B, kernel_ms[dom], B))
    out = {"roofline": roofline, "cpu_baseline": None}
    single = world == 1 and B == 1
    if single and args.exact_frames > 0:
        # A/B: the exact-fp32 MFMA generation (v_mfma_f32_16x16x4_f32, THA4_STUDENT_EXACT_FP32) on the same stream
        face, body = split_flat_weights(work.w)
        exact = mode_14.create_poser_from_state_dicts(dev, face, body, exact_fp32=True)
        with torch.no_grad():
            for i in range(20):
                exact.pose(image, poses[i, 0])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(args.exact_frames):
                exact.pose(image, poses[W + (i % K), 0])
            torch.cuda.synchronize(dev)
            dte = time.perf_counter() - t0
        exact.free()
        roofline["exact_fp32_generation"] = {"fps": round(args.exact_frames / dte, 2), "frames": args.exact_frames,
                                             "achieved_tflops": round(args.exact_frames / dte * GFLOP_FRAME / 1e3, 2),
                                             "frac_of_fp32_mfma_peak": round(args.exact_frames / dte * GFLOP_FRAME / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                                             "what": "same stream on the exact-fp32 kernels (v_mfma_f32_16x16x4_f32, csrc/siren_kernels.h)"}
    if single and args.d2h_frames > 0:
        # SURVEY.md §8d config 2 also asks for the rate with the display epilogue + D2H of the RGBA8 frame included (what a
        # puppeteer actually consumes): pose with the epilogue FUSED into the composing kernel (tha4_display: the fp32 frame is
        # neither written nor re-read) -> async copy into a pinned ring.  PCIe-inclusive: never `value`.
        from tha4_amd import image_io
        R = 4
        host_ring = [torch.empty((1, 512, 512, 4), dtype=torch.uint8).pin_memory() for _ in range(R)]
        dev_ring = [torch.empty((1, 512, 512, 4), dtype=torch.uint8, device=dev) for _ in range(R)]
        side = torch.cuda.Stream(device=dev)
        ready = [torch.cuda.Event() for _ in range(R)]
        copied = [torch.cuda.Event() for _ in range(R)]

        def d2h_rate(make):
            """frame i -> device slot i % R on the compute stream; its copy to pinned host memory runs on a side stream (the copy
            engine), so frame i + 1 is composed while frame i crosses PCIe; a slot is reused when its copy has completed"""
            def one(i):
                j = i % R
                torch.cuda.current_stream(dev).wait_event(copied[j])
                make(i, dev_ring[j])
                ready[j].record()
                with torch.cuda.stream(side):
                    side.wait_event(ready[j])
                    host_ring[j].copy_(dev_ring[j], non_blocking=True)
                    copied[j].record(side)
            with torch.no_grad():
                for j in range(R):
                    copied[j].record(side)
                for i in range(8):
                    one(W + i)
                torch.cuda.synchronize(dev)
                t0d = time.perf_counter()
                for i in range(args.d2h_frames):
                    one(W + (i % K))
                torch.cuda.synchronize(dev)
                return args.d2h_frames / (time.perf_counter() - t0d)

        fused_make = lambda i, out: work.step_rgba8(i, out=out)                                       # noqa: E731
        unfused_make = lambda i, out: out.copy_(image_io.to_display_rgba8(work.step(i)))              # noqa: E731
        d2h_rate(fused_make)            # untimed: the first pass over a pinned ring pays for mapping it (measured: 2-3x slower than the second)
        fused = max(d2h_rate(fused_make), d2h_rate(fused_make))
        unfused = max(d2h_rate(unfused_make), d2h_rate(unfused_make))
        out["with_rgba8_d2h"] = {"fps": round(fused, 2), "frames": args.d2h_frames,
                                 "what": "pose with the sRGB/uint8 display epilogue fused into the composing kernel (tha4_display) + async D2H of the "
                                         "1 MiB RGBA8 frame into pinned host memory on a side stream, 4-slot ring (PCIe-inclusive)",
                                 "unfused_fps": round(unfused, 2),
                                 "unfused_what": "pose (fp32 frame) -> tha4_display_rgba8 as a second kernel -> the same D2H (round 2's path)"}
    if world == 1 and args.cpu_seconds > 0:
        from oracle import reference_runner as rr        # cpu_baseline leg only (oracle/ is test infrastructure)
        from oracle import student_oracle as so
        w, image_np, poses_cpu = work.w, work.image_np, work.poses_cpu
        if rr.available():                               # the unmodified reference itself (build container; never on the GPU box)
            ref_run, _ = rr.student_runner(work.character)
            out["cpu_baseline"] = cpu_baseline(lambda i: ref_run(poses_cpu[i % poses_cpu.shape[0]].numpy()),
                                               f"{work.character} student stream, the unmodified reference (tha4.poser.modes.mode_14 on CPU, fp32)",
                                               args.cpu_seconds, 48, kind="reference")
        else:
            out["cpu_baseline"] = cpu_baseline(lambda i: so.student_forward_torch(w, image_np, poses_cpu[i % poses_cpu.shape[0]].numpy(), "float32"),
                                               f"{work.character} student stream, oracle.student_forward_torch fp32", args.cpu_seconds, 48)
            out["cpu_baseline"]["port_fidelity"] = ("the oracle dispatches the same ATen ops as the reference and equals the unmodified reference bit for bit on "
                                                    "every committed fixture (tests/test_oracle_golden.py; /root/reference does not exist on the GPU box)")
    if single and args.full_frames > 0:
        try:      # secondary: configs[2] in the same process (the headline must not depend on it)
            poser.free()
            fw = FullWork(dev, 0, 1, args.full_frames + 3, steady=True)
            out["full_model"] = measure_full_b1(fw, dev, args.full_frames)
            fw.steady = True
            out["full_model"]["roofline"] = full_roofline(out["full_model"]["steady"]["fps"], GFLOP_FULL_STEADY, cold=False, batch1=True, work=fw)
            out["full_model"]["two_frames_in_flight"] = measure_full_two_in_flight(dev, args.full_frames, fw)
            fw.poser.free()
            # the strict-precision number next to it: the same frames on the exact-fp32 plan (THA4_FULL_EXACT_FP32: every convolution on
            # v_mfma_f32_16x16x4_f32 with fp32 operands - no 22-bit operand split to argue about)
            nx = max(6, min(12, args.full_frames // 2))
            fx = FullWork(dev, 0, 1, nx + 3, steady=True, exact_fp32=True)
            ex = measure_full_b1(fx, dev, nx)
            fx.poser.free()
            for k in ("steady", "cold"):
                ex[k]["frac_of_fp32_mfma_peak"] = round(ex[k]["achieved_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
                ex[k].pop("frac_of_f16_mfma_peak", None)
                ex[k].pop("frac_of_split_ceiling", None)
            ex["what"] = f"the same workload on the exact-fp32 plan (tha4_full_create_ex flags = THA4_FULL_EXACT_FP32), {nx} frames each"
            out["full_model"]["exact_fp32"] = ex
        except Exception as e:
            out["full_model"] = {"error": repr(e)}
    if single and args.batched_steps > 0:
        # the batched configurations of BASELINE.json, one GPU's share each, in the driver-visible line (the N = 8 runs of the
        # same commands are `--gpus 8 --batch 32` / `--gpus 8 --model full --batch 8`)
        for key, make, gflop, b in (("student_b32", lambda: StudentWork(dev, 0, 32, args.batched_steps + 4, "lambda_00"), GFLOP_FRAME, 32),
                                    ("full_b8", lambda: FullWork(dev, 0, 8, args.batched_steps + 3, steady=False), GFLOP_FULL_COLD, 8)):
            try:
                try:
                    poser.free()
                except Exception:
                    pass
                bw = make()
                n = args.batched_steps if key == "student_b32" else max(4, args.batched_steps // 2)
                with torch.no_grad():
                    for i in range(3):
                        bw.step(i)
                    torch.cuda.synchronize(dev)
                    t0b = time.perf_counter()
                    for i in range(n):
                        bw.step(3 + i if key == "full_b8" else 4 + i)
                    torch.cuda.synchronize(dev)
                    dtb = time.perf_counter() - t0b
                f = n * b / dtb
                out[key] = {"fps": round(f, 2), "batch": b, "steps": n, "ms_per_step": round(1e3 * dtb / n, 3),
                            "achieved_tflops": round(f * gflop / 1e3, 2), "frac_of_f16_mfma_peak": round(f * gflop / 1e3 / PEAK_F16_MFMA_TFLOPS, 4),
                            "frac_of_split_ceiling": round(f * gflop * MFMA_PASSES / 1e3 / PEAK_F16_MFMA_TFLOPS, 4),
                            "workload": ("configs[3], one GPU's share: lambda_00 student, 32 poses per Poser.pose() call" if key == "student_b32" else
                                         "configs[4], one GPU's share: full model, 8 distinct random images + poses per call, decomposer never cached")}
                bw.poser.free()
            except Exception as e:
                out[key] = {"error": repr(e)}
    return out


def measure_full_b1(fw, dev, frames):
    res = {}
    with torch.no_grad():
        for i in range(3):
            fw.step(i)
        for name, steady, gflop in (("steady", True, GFLOP_FULL_STEADY), ("cold", False, GFLOP_FULL_COLD)):
            fw.steady = steady
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(frames):
                fw.step(3 + i)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            f = frames / dt
            res[name] = {"fps": round(f, 2), "ms_per_frame": round(1e3 * dt / frames, 3), "achieved_tflops": round(f * gflop / 1e3, 2),
                         "frac_of_f16_mfma_peak": round(f * gflop / 1e3 / PEAK_F16_MFMA_TFLOPS, 4),
                         "frac_of_split_ceiling": round(f * gflop * MFMA_PASSES / 1e3 / PEAK_F16_MFMA_TFLOPS, 4)}
    return res


def measure_full_two_in_flight(dev, frames, first):
    """NOT the configs[2] number (that is one pose() after the other on one stream): the same batch-1 frames through TWO handles on two
    streams, frame i on stream i % 2 - what a throughput caller (offline rendering, a second character) gets from a chip that a single
    batch-1 frame leaves latency-bound (a frame is a chain of ~320 dependent launches).  Host-side only: no kernel differs."""
    try:
        first.steady = True
        works = [first, FullWork(dev, 0, 1, frames + 3, steady=True)]          # (`first`: the handle of the configs[2] measurement - one weight pack less)
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        with torch.no_grad():
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    for i in range(3):
                        works[k].step(i)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(frames):
                with torch.cuda.stream(streams[i % 2]):
                    works[i % 2].step(3 + i)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
        works[1].poser.free()
        f = frames / dt
        return {"fps": round(f, 2), "frames": frames, "achieved_tflops": round(f * GFLOP_FULL_STEADY / 1e3, 2),
                "what": "steady batch-1 frames alternating over two handles on two streams (two independent frames in flight); "
                        "never `value`, never the configs[2] figure"}
    except Exception as e:
        return {"error": repr(e)}


def full_live_classes(work, frames=8):
    """Per-launch-class durations of the full model measured LIVE: HIP events on the launch stream around every op of the schedule
    (tha4_full_set_timing, ABI v5) over `frames` steady frames; classes = ops with the same label (reference layer shape + kernel).
    Returns the classes sorted by total time per frame.  Events add 1-3 us of chain time per op: a conservative TFLOP/s."""
    poser = work.poser
    info = poser.op_info()
    tot, cnt = {}, {}
    poser.set_timing(True)
    try:
        with torch.no_grad():
            for i in range(frames):
                work.step(i)
                for (label, gf), ms in zip(info, poser.last_op_ms()):
                    if ms > 0.0:
                        tot[label] = tot.get(label, 0.0) + ms
                        cnt[label] = cnt.get(label, 0) + 1
    finally:
        poser.set_timing(False)
    gfl = {label: gf for label, gf in info}
    rows = [{"class": label, "launches_per_frame": round(cnt[label] / frames, 2), "ms_per_frame": round(tot[label] / frames, 4),
             "avg_us": round(1e3 * tot[label] / cnt[label], 2), "gflop_per_launch": round(gfl[label], 4),
             "tflops": round(gfl[label] / (tot[label] / cnt[label]), 1) if gfl[label] > 0 else None} for label in tot]
    rows.sort(key=lambda r: -r["ms_per_frame"])
    return rows, sum(tot.values()) / frames


def full_roofline(fps_per_gpu, gflop, cold, batch1, work=None):
    """Roofline object of the full model.  The frame is ~320 launches of a static schedule: `achieved` is the WHOLE FRAME (as-written FLOPs
    of the reference's five networks / measured frame time) against the dense fp16 MFMA peak of the instruction issued; `dominant_class` is
    the launch class (reference layer shape + kernel) with the largest share of the frame, its as-written FLOPs per launch / its average
    launch duration measured live with HIP events on the launch stream (tha4_full_set_timing), with the rocprofv3 figure of the newest
    committed capture beside it (profiles/r*_full_b1_layers.json, tools/conv_breakdown.py)."""
    ach = fps_per_gpu * gflop / 1e3
    prof, prof_file = newest_profile("r*_full_b1_traffic.json")
    per_frame = prof.get("cold_frame_bytes" if cold else "steady_frame_bytes") if (prof and batch1) else None
    dominant, top = None, None
    if work is not None:
        try:
            rows, event_ms = full_live_classes(work)
            convs = [r for r in rows if r["tflops"] is not None]
            if convs:
                d = convs[0]
                dominant = dict(d, frac_of_f16_mfma_peak=round(d["tflops"] / PEAK_F16_MFMA_TFLOPS, 4),
                                frac_of_split_ceiling=round(d["tflops"] * MFMA_PASSES / PEAK_F16_MFMA_TFLOPS, 4),
                                what="the convolution class with the largest total time per steady frame; tflops = as-written GFLOP of the layer / its "
                                     "average launch duration from HIP events on the launch stream (conservative: the events add chain time)")
            top = {"event_timed_frame_ms": round(event_ms, 3), "classes": rows[:8]}
        except Exception as e:                     # a measurement aid must not take the line down
            dominant = {"error": repr(e)}
    # what the dominant class's kernel waits for, from the newest committed SQ capture of the full model (tools/pmc_json.py --mode full): the template instance of
    # that kernel with the largest share of the capture
    evidence = None
    if isinstance(dominant, dict) and "class" in dominant and "[" in dominant["class"]:
        kname = dominant["class"].split("[")[-1].split("]")[0].split(",")[0].strip()
        pmc, pmc_file = newest_profile("r*_full_b1_pmc.json" if batch1 else "r*_full_b8_pmc.json")
        cands = {n: e for n, e in (pmc or {}).get("kernels", {}).items() if n.split("<")[0] == kname}
        if cands:
            n = max(cands, key=lambda x: cands[x].get("launches_per_pass", 0) * cands[x].get("avg_us", 0.0))
            evidence = dict(pmc_evidence(cands[n], pmc_file), kernel=n)
        else:
            evidence = pmc_evidence(None, None)
    layers, layers_file = newest_profile("r*_full_b1_layers.json")
    rocprof = None
    if layers and batch1 and layers.get("classes"):
        c = max(layers["classes"], key=lambda r: r["total_us"])
        rocprof = dict(c, frac_of_f16_mfma_peak=round(c["tflops"] / PEAK_F16_MFMA_TFLOPS, 4), source=layers_file)
    return {"bound": "mfma", "kernel": "whole frame (static schedule of conv_tile / conv_small / conv_point / attention / image kernels)",
            "achieved": round(ach, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F16_MFMA_TFLOPS, 4),
            "frac_of_split_ceiling": round(ach * MFMA_PASSES / PEAK_F16_MFMA_TFLOPS, 4),
            "vs_fp32_mfma_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
            "mfma": "v_mfma_f32_16x16x32_f16 on fp16 hi/lo operand halves, 3 per product block, fp32 accumulate",
            "traffic": per_frame, "traffic_unit": f"bytes/frame = sum over the frame's launches of 2 x FETCH_SIZE + WRITE_SIZE ({prof_file})",
            "dominant_class": dominant, "dominant_class_evidence": evidence, "dominant_class_rocprof": rocprof, "live_classes": top,
            "algorithmic_gflop_per_frame": gflop}


def full_extras(args, work, dev, world, fps, B):
    cold = args.cold or B > 1
    gflop = GFLOP_FULL_COLD if cold else GFLOP_FULL_STEADY
    roofline = full_roofline(fps / world, gflop, cold, B == 1, work=work if world == 1 else None)
    out = {"roofline": roofline, "cpu_baseline": None}
    if world == 1 and B == 1:
        other = FullWork.__new__(FullWork)
        other.__dict__.update(work.__dict__)
        out["steady_and_cold"] = measure_full_b1(other, dev, max(10, min(50, args.steps or 50)))
    if world == 1 and args.cpu_seconds > 0:
        from oracle import full_oracle as fo             # cpu_baseline leg only
        from oracle import reference_runner as rr
        w = synthetic.synth_full_weights()
        poses = make_poses(4, seed=77).numpy()
        img = work.image_np
        if rr.available():                               # the unmodified reference modules with the synthetic state_dicts (build container only)
            ref_run = rr.full_runner(w)
            out["cpu_baseline"] = cpu_baseline(lambda i: ref_run(img, poses[i % 4]),
                                               "cold frames of the full model, the unmodified reference (GeneralPoser02 + mode_07 protocol on CPU, fp32), "
                                               "synthetic weights", args.cpu_seconds, 12, kind="reference")
        else:
            out["cpu_baseline"] = cpu_baseline(lambda i: fo.full_forward_torch(w, img, poses[i % 4], "float32"),
                                               "cold frames of the full model, oracle.full_forward_torch fp32, synthetic weights", args.cpu_seconds, 12)
    return out


if __name__ == "__main__":
    main()
