#!/usr/bin/env python3
"""Benchmark of the student poser hot path (BASELINE.json configs[1]): lambda_00 distilled student,
batch=1 real-time stream of random 45-dim poses on 512x512 RGBA, through the drop-in Poser API
(tha4_amd.poser.modes.mode_14 -> include/tha4_hip.h -> gfx950 kernels).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one Poser.pose() call = one frame (batch 1), back-to-back on the rank's current stream,
image and poses resident in HBM before the timed region.  With N>1 every rank poses K frames of its
own (weak scaling, frames are independent) and finished frames are gathered to rank 0 in chunks
over RCCL on a side stream (the only exchange the path has).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import tha4_amd  # noqa: E402,F401
from tha4_amd.poser.modes import mode_14  # noqa: E402
from tha4_amd.sharding import FrameShardedStream  # noqa: E402
from tha4_amd.weights import split_flat_weights  # noqa: E402

# Algorithmic work of the reference's student forward pass as written (SURVEY.md §8d, 2*MAC),
# per 512x512 frame and per kernel of this implementation.
GFLOP_FRAME = 37.885
GFLOP_KERNEL = {"face": 3.947, "level0": 6.924, "level1": 11.726, "level2": 15.288}
# What the kernels actually execute after pose folding + commuting the x2 upsample with the next
# level's first layer: stated separately, never used for `roofline.achieved`.
GFLOP_EXECUTED_FRAME = 27.46
# The contractions issue v_mfma_f32_16x16x32_f16 on fp16 hi/lo halves of fp32 operands, three MFMAs per fp32-accurate
# product block (hi*hi + hi*lo + lo*hi, fp32 accumulate).  `roofline.peak` is the dense fp16 MFMA peak of the
# instruction actually issued (MI355X_MICROARCH.md: ~2.5 PFLOP/s); `roofline.achieved` stays ALGORITHMIC (every
# multiply-add of the reference counted once), so the ceiling of this arithmetic is peak/3 - reported next to it.
PEAK_F16_MFMA_TFLOPS = 2500.0
MFMA_PASSES = 3
PEAK_FP32_MFMA_TFLOPS = 157.3      # v_mfma_f32_16x16x4_f32 dense peak (what an exact-fp32 single pass could reach)
# HBM-side bytes per launch of each kernel from the PMC passes committed in profiles/r01_student_b1_profile.md
# (FETCH_SIZE x 2 [gfx950 wide-read correction, MI355X_MICROARCH.md §HBM] + WRITE_SIZE, KiB -> bytes).
# bench.py cannot run rocprofv3 on itself, so this is the profiled value for the same command line.
PMC_TRAFFIC_BYTES = {"face": (1920 * 2 + 256) * 1024, "level0": (4178 * 2 + 12290) * 1024,
                     "level1": (8516 * 2 + 24580) * 1024, "level2": (30730 * 2 + 4284) * 1024}
KERNEL_NAMES = ["posebias", "face", "level0", "level1", "level2"]
# full THA4 system (mode_07), SURVEY.md §8d: FlopCounterMode on the reference modules
GFLOP_FULL_COLD = 645.90
GFLOP_FULL_STEADY = 625.90

POSE_LO = np.array([0.0] * 37 + [-1.0] * 7 + [0.0], dtype=np.float32)
POSE_HI = np.ones(45, dtype=np.float32)


def make_poses(n, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 45, generator=g).numpy()
    return torch.from_numpy((POSE_LO + (POSE_HI - POSE_LO) * u).astype(np.float32))


def load_fixture():
    g = os.path.join(ROOT, "tests", "golden")
    w = dict(np.load(os.path.join(g, "student_lambda_00_weights.npz")))
    io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
    return w, io["image_f32"]


def cpu_baseline(w, image, poses, budget_s):
    """The CPU path timed beside the GPU path: the oracle's torch-functional restatement of the
    reference (same ATen ops), fp32, all host cores.  Bounded sample (~budget_s of CPU work)."""
    from oracle import student_oracle as so
    so.student_forward_torch(w, image, poses[0].numpy(), "float32")      # warm-up (thread pool, allocator)
    n = 0
    t0 = time.perf_counter()
    while True:
        so.student_forward_torch(w, image, poses[n % poses.shape[0]].numpy(), "float32")
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 64:
            break
    return {"value": round(n / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} frames of the same lambda_00 stream, oracle.student_forward_torch fp32 ({dt:.1f} s)",
            "ms_per_frame": round(1e3 * dt / n, 2)}


def measure_full(dev, image, frames, batch=1):
    """BASELINE.json configs[2]: full THA4 model (5 networks), batch 1, synthetic seeded weights (the reference
    checkout ships none), lambda_00 image.  Returns steady (eyebrow decomposer cached, mode_07.py:56-67) and cold fps."""
    from tha4_amd import synthetic
    from tha4_amd.poser.modes import mode_07
    poser = mode_07.create_poser_from_state_dicts(dev, synthetic.synth_full_weights(), max_batch=batch)
    poses = make_poses(8 * batch, seed=77).to(dev)
    if batch > 1:
        poses = poses.reshape(8, batch, 45)
    with torch.no_grad():
        for i in range(3):
            poser.pose(image, poses[i])
        out = {}
        for name, changed, gflop in (("steady", False, GFLOP_FULL_STEADY), ("cold", True, GFLOP_FULL_COLD)):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(frames):
                poser.pose(image, poses[i % 8], image_changed=changed)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            fps = frames * batch / dt
            out[name] = {"fps": round(fps, 2), "ms_per_frame": round(1e3 * dt / (frames * batch), 3),
                         "achieved_tflops": round(fps * gflop / 1e3, 2),
                         "frac_of_f16_mfma_peak": round(fps * gflop / 1e3 / PEAK_F16_MFMA_TFLOPS, 4),
                         "frac_of_split_ceiling": round(fps * gflop * MFMA_PASSES / 1e3 / PEAK_F16_MFMA_TFLOPS, 4)}
    poser.free()
    return out


def cpu_baseline_full(image, budget_s):
    """CPU path of the full model beside the GPU path: the oracle's torch-functional restatement of mode_07 (same ATen
    ops, every frame cold as the reference would be for a new image), fp32, all host cores, bounded sample."""
    from oracle import full_oracle as fo
    from tha4_amd import synthetic
    w = synthetic.synth_full_weights()
    poses = make_poses(4, seed=77).numpy()
    fo.full_forward_torch(w, image, poses[0], "float32")              # warm-up
    n = 0
    t0 = time.perf_counter()
    while True:
        fo.full_forward_torch(w, image, poses[n % 4], "float32")
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 16:
            break
    return {"value": round(n / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} cold frames, oracle.full_forward_torch fp32, synthetic weights ({dt:.1f} s)", "ms_per_frame": round(1e3 * dt / n, 1)}


def main_full(args):
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _, image_np = load_fixture()
    image = torch.from_numpy(image_np).to(dev)
    r = measure_full(dev, image, args.steps, max(1, args.batch))
    cpu = cpu_baseline_full(image_np, args.cpu_seconds) if args.cpu_seconds > 0 else None
    print(json.dumps({
        "metric": "frames/sec on 512x512 RGBA + 45-dim pose, full THA4 model", "value": r["steady"]["fps"], "unit": "frames/s",
        "n_gpus": 1, "steps": args.steps, "warmup": 3, "ms_per_step": round(r["steady"]["ms_per_frame"] * max(1, args.batch), 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic seeded weights (tha4_amd.synthetic, reference ships none); lambda_00 image fixture; random poses",
        "config": {"workload": f"configs[2]: THA4 full model (face_morpher+rotator+editor), batch={max(1, args.batch)}, steady state (eyebrow decomposer cached)",
                   "batch": max(1, args.batch)},
        "roofline": {"bound": "mfma", "achieved": r["steady"]["achieved_tflops"], "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": r["steady"]["frac_of_f16_mfma_peak"], "frac_of_split_ceiling": r["steady"]["frac_of_split_ceiling"],
                     "mfma": "v_mfma_f32_16x16x32_f16 on fp16 hi/lo operand halves, 3 per product block, fp32 accumulate (k > 1 convolutions)",
                     "traffic": None,
                     "algorithmic_gflop_per_frame": GFLOP_FULL_STEADY},
        "cold": r["cold"], "cpu_baseline": cpu}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["student", "full"], default="student",
                    help="student = BASELINE configs[1] (default, the headline metric); full = configs[2]")
    ap.add_argument("--full-frames", type=int, default=30, help="student run: frames for the appended full-model measurement (0 = skip)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="frames per Poser.pose() call (1 = configs[1]/[2]; 32 = configs[3]; 8..64 = configs[4] with --model full)")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--gather-chunk", type=int, default=32, help="frames per RCCL gather (N>1)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the gather of finished frames")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 disables)")
    ap.add_argument("--profile-frames", type=int, default=100, help="frames for the per-kernel HIP-event pass")
    ap.add_argument("--d2h-frames", type=int, default=500, help="frames for the secondary RGBA8 + D2H inclusive measurement (0 = skip)")
    args = ap.parse_args()
    if args.model == "full":
        if args.steps == 2000:
            args.steps = 100
        return main_full(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    w, image_np = load_fixture()
    face_sd, body_sd = split_flat_weights(w)
    B = max(1, args.batch)
    poser = mode_14.create_poser_from_state_dicts(dev, face_sd, body_sd, max_batch=max(B, 4))
    image = torch.from_numpy(image_np).to(dev)
    K, W = args.steps, args.warmup
    poses_cpu = make_poses((K + W) * B, seed=1234 + rank)
    poses = poses_cpu.to(dev)
    if B > 1:
        poses = poses.reshape(K + W, B, 45)           # one pose() call per step on a [B,45] batch, image shared

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    with torch.no_grad():
        for i in range(W):
            out = poser.pose(image, poses[i])
        if world > 1 and not args.no_gather and B == 1:
            chunk = args.gather_chunk

            def frame_fn(lo, hi):   # global frame ids of this rank start at rank*K
                base = rank * K
                blk = torch.empty((hi - lo, 4, 512, 512), dtype=torch.float32, device=dev)
                for i in range(lo, hi):
                    poser.pose(image, poses[W + i - base], out=blk[i - lo:i - lo + 1])    # straight into the gather block
                return blk

            stream = FrameShardedStream(frame_fn, total=K * world, frame_shape=(4, 512, 512), dtype=torch.float32,
                                        device=dev, chunk=chunk, gather=True)
            gathered = stream.allocate_result()       # rank 0: K*world frames (4 MiB each) - allocated outside the timed region
            barrier()
            t0 = time.perf_counter()
            gathered = stream.run(gathered)
            barrier()
            t1 = time.perf_counter()
            del gathered
        else:
            barrier()
            t0 = time.perf_counter()
            for i in range(K):
                out = poser.pose(image, poses[W + i])
            barrier()
            t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        total_frames = K * world * B
        fps = total_frames / elapsed
        # per-kernel durations from HIP events recorded on the launch stream inside the C ABI
        poser.set_timing(True)
        acc = np.zeros(len(KERNEL_NAMES))
        whole = 0.0
        nprof = max(1, args.profile_frames)
        with torch.no_grad():
            for i in range(nprof):
                poser.pose(image, poses[W + (i % K)])
                for k in range(len(KERNEL_NAMES)):
                    acc[k] += poser.last_kernel_ms(k)
                whole += poser.last_kernel_ms(-1)
        poser.set_timing(False)
        kernel_ms = {n: float(acc[k] / nprof) for k, n in enumerate(KERNEL_NAMES)}
        dom = max(GFLOP_KERNEL, key=lambda n: kernel_ms[n])
        achieved = GFLOP_KERNEL[dom] * B / kernel_ms[dom]        # GFLOP / ms = TFLOP/s
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 3), "peak": PEAK_F16_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4),
                    "mfma": "v_mfma_f32_16x16x32_f16 on fp16 hi/lo operand halves, 3 per product block, fp32 accumulate",
                    "frac_of_split_ceiling": round(achieved * MFMA_PASSES / PEAK_F16_MFMA_TFLOPS, 4),
                    "vs_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                    "traffic": PMC_TRAFFIC_BYTES.get(dom) if B == 1 else None, "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/r01_student_b1_profile.md)",
                    "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
                    "frame_event_ms": round(whole / nprof, 4),
                    "whole_frame_achieved_tflops": round(fps / world * GFLOP_FRAME / 1e3, 3),
                    "whole_frame_frac": round(fps / world * GFLOP_FRAME / 1e3 / PEAK_F16_MFMA_TFLOPS, 4),
                    "algorithmic_gflop_per_frame": GFLOP_FRAME, "executed_gflop_per_frame": GFLOP_EXECUTED_FRAME}
        # SURVEY.md §8d config 2 also asks for the rate with the display epilogue + D2H of the RGBA8 frame included (what a
        # puppeteer actually consumes): pose -> tha4_display_rgba8 -> async copy into a pinned ring, same stream.  Never `value`.
        d2h = None
        if world == 1 and B == 1 and args.d2h_frames > 0:
            from tha4_amd import image_io
            ring = [torch.empty((1, 512, 512, 4), dtype=torch.uint8).pin_memory() for _ in range(4)]
            with torch.no_grad():
                for i in range(8):
                    ring[i % 4].copy_(image_io.to_display_rgba8(poser.pose(image, poses[W + i])), non_blocking=True)
                torch.cuda.synchronize(dev)
                t0d = time.perf_counter()
                for i in range(args.d2h_frames):
                    ring[i % 4].copy_(image_io.to_display_rgba8(poser.pose(image, poses[W + (i % K)])), non_blocking=True)
                torch.cuda.synchronize(dev)
                dtd = time.perf_counter() - t0d
            d2h = {"fps": round(args.d2h_frames / dtd, 2), "frames": args.d2h_frames,
                   "what": "pose + sRGB/uint8 display epilogue + async D2H of the 1 MiB RGBA8 frame into pinned host memory (PCIe-inclusive)"}
        cpu = cpu_baseline(w, image_np, poses_cpu, args.cpu_seconds) if (args.cpu_seconds > 0 and world == 1) else None
        full = None
        if args.full_frames > 0 and world == 1:
            try:
                poser.free()
                full = measure_full(dev, image, args.full_frames)
            except Exception as e:      # the headline number must not depend on the secondary measurement
                full = {"error": repr(e)}
        result = {
            "metric": "frames/sec (whole job) on 512x512 RGBA + 45-dim pose, distilled student",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(1e3 * elapsed / K, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic pose stream (seed 1234+rank, pose_parameters ranges); lambda_00 student weights + image fixture (tests/golden)",
            "config": {"workload": ("configs[1]: lambda_00 distilled student, batch=1 real-time stream, 512x512 RGBA, one Poser.pose() per frame" if B == 1 else
                                    f"configs[3]-style: one lambda_00 character instance per GPU, batch={B} pose stream per Poser.pose() call"),
                       "frames_per_gpu": K * B, "batch": B, "parallelism": f"frame-parallel x{world}",
                       "gather": bool(world > 1 and not args.no_gather and B == 1)},
            "per_gpu_fps": round(fps / world, 2),
            "roofline": roofline, "cpu_baseline": cpu, "with_rgba8_d2h": d2h,
            "full_model": full,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
