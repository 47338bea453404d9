#!/usr/bin/env python3
"""bench.py -- frames/sec of the MI355X-native detect -> match -> pose hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run, one rank per GPU.  One JSON line on rank 0.

Workload (BASELINE.json configs[1]): synthetic 752x480 mono8 EuRoC-shaped stream ("S-752", integer-only
generator, see vi-slam_amd/csrc/geometry.cpp), ORB 1000 keypoints x 8 levels, BF-Hamming k=2 both
directions + ratio/symmetry/grid filter, essential RANSAC + recoverPose.  A "step" = one pass of the
whole hot path over one batch of B consecutive frames that are already resident in HBM.
Multi-GPU (configs[3]): rank r processes its own stream (seed + r); the only collective is one
broadcast of the parameter/intrinsics struct from rank 0 (RCCL); scaling is weak.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402  (device memory + torch.distributed plumbing; imported before the HIP library)
import vislam  # noqa: E402
from vislam import dist as vdist  # noqa: E402

W, H, NFEAT, LEVELS = 752, 480, 1000, 8
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)


def level_pixels(ctx):
    ws, hs, sc, q = ctx.level_geometry(W, H)
    return [int(a) * int(b) for a, b in zip(ws, hs)]


def algorithmic_bytes(px, n):
    """SURVEY.md section 8(d) per-frame figures, split per kernel family (DESIGN.md 'Roofline')."""
    ptot = sum(px)
    return {
        "k_fast": ptot,                                     # every pyramid pixel read once
        "k_resize": (ptot - px[-1]) + (ptot - px[0]),       # each level read once as source + levels>=1 written
        "k_describe": n * 1369 + n * 60,                    # 37x37 blurred footprint + 32 B desc + 28 B keypoint
        "total_detect_describe": ptot + (ptot - px[0]) + (ptot - px[-1]) + n * 1369 + n * 60,
    }


def make_frames(seed, count):
    canvas = vislam.synth_canvas(4096, seed)
    fr = np.empty((count, H, W), np.uint8)
    for t in range(count):
        vislam.synth_frame(canvas, t, W, H, seed, out=fr[t])
    return fr


def cpu_baseline(p, frames, budget_s=12.0):
    """oracle ('port' of the reference CPU path) timed on this host, 1 thread, bounded sample"""
    import oracle_bind as orc
    prev = None
    n = 0
    # 3 warm-up frames, then as many as fit the budget (at most len(frames))
    for t in range(3):
        k, d, r = orc.pipeline_frame(p, frames[t], prev)
        prev = (k, d)
    t0 = time.perf_counter()
    for t in range(3, len(frames)):
        k, d, r = orc.pipeline_frame(p, frames[t], prev)
        prev = (k, d)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} consecutive S-752 frames (752x480, N=1000, 8 levels), oracle pipeline_frame: Camera::Update + ORB + knn x2 + "
                      f"filters + essential RANSAC + recoverPose, g++ -O2, 1 thread; host has {os.cpu_count()} logical CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="frames per step (per GPU)")
    ap.add_argument("--ring", type=int, default=2, help="distinct batches resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stages", type=int, default=vislam.STAGE_ALL, help="debug: bitmask of stages (1 detect, 2 match, 4 pose); the reported metric needs all 7")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", world_size=world, rank=rank)   # "nccl" == RCCL on ROCm
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    # ---- parameters: rank 0 owns them; one RCCL broadcast of the POD struct ("intrinsics only")
    p = vislam.default_params()
    if rank == 0:
        p.nfeatures, p.nlevels, p.w_size, p.h_size = NFEAT, LEVELS, W, H
        p.fy = p.fx
    p = vdist.broadcast_params(p, dist, dev, rank)

    ctx = vislam.Context(local_rank if world > 1 else 0, p)
    B, R = a.batch, a.ring
    frames = make_frames(vdist.stream_seed(rank, world), B * R)
    dframes = torch.from_numpy(frames).to(dev)
    ctx.batch_plan(W, H, W, B)
    fbytes = W * H

    def step(i):
        ctx.batch_run(dframes.data_ptr() + (i % R) * B * fbytes, B, a.stages)

    for i in range(a.warmup):
        step(i)
    ctx.batch_sync()
    if ctx.batch_status() != 0:
        raise RuntimeError("device capacity flag set during warm-up")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    ctx.batch_sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = vdist.max_over_ranks(dt, dist, dev)
    if ctx.batch_status() != 0:
        raise RuntimeError("device capacity flag set during the timed region")

    # ---- per-kernel-family durations from HIP events on the library's stream (outside the timed region)
    fam = {k: 0.0 for k in ("ms_pyramid", "ms_fast", "ms_select", "ms_describe", "ms_knn", "ms_filter", "ms_pose", "ms_total")}
    nprof = 8
    launches_fast = LEVELS
    for i in range(nprof):
        step(i)
        ctx.batch_sync()
        t = ctx.timings()
        for k in fam:
            fam[k] += getattr(t, k) / nprof
        launches_fast = t.launches_fast

    # ---- SURVEY 8(f) N2 (outside the timed region, not part of `value`): Camera::Update half pyramid +
    # Camera::computeGradient over the same resident batch; pure streaming, reported against the HBM roofline
    aux = None
    if rank == 0:
        try:
            fe = vislam.gradient_frame_elems(W, H)
            gray = torch.empty(B * fe, dtype=torch.uint8, device=dev)
            gxb = torch.empty(B * fe, dtype=torch.int16, device=dev); gyb = torch.empty_like(gxb)
            gb = torch.empty(B * fe, dtype=torch.uint8, device=dev)
            def grad():
                ctx.gradient_batch(dframes.data_ptr(), W, H, W, B, gray.data_ptr(), gxb.data_ptr(), gyb.data_ptr(), gb.data_ptr())
            for _ in range(3):
                grad()
            torch.cuda.synchronize()
            tg = time.perf_counter()
            KG = 20
            for _ in range(KG):
                grad()
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg) / KG
            pg = sum((W >> l) * (H >> l) for l in range(5))
            p_half = sum((W >> l) * (H >> l) for l in range(4)) + sum((W >> l) * (H >> l) for l in range(1, 5))
            gbytes = 6 * pg + p_half                     # gradient: 1 B read + 5 B written per pixel; pyramid: read 4 levels, write 4
            aux = {"gradient_batch": {"what": "Camera::Update half pyramid + Camera::computeGradient (Scharr x/y int16 + blended u8), 5 levels",
                                      "ms_per_batch": tg * 1e3, "frames": B, "algorithmic_bytes_per_frame": gbytes,
                                      "achieved_GBps": gbytes * B / tg / 1e9, "peak_GBps": HBM_PEAK_GBS,
                                      "frac": gbytes * B / tg / 1e9 / HBM_PEAK_GBS, "frames_per_s": B / tg}}
            del gray, gxb, gyb, gb
        except Exception as e:                            # never let the side measurement break the contract line
            aux = {"gradient_batch": {"error": str(e)}}

    # ---- SURVEY 8(f) N3 (outside the timed region, never `value`): the same pipeline fed from pinned HOST memory through
    # the double-buffered feeder -- the PCIe-inclusive rate
    if rank == 0 and aux is not None:
        try:
            feed = vislam.Feeder(ctx, W, H, B)
            for k in range(2):
                feed.host_buffer(k)[:] = frames[(k % R) * B:(k % R + 1) * B]
            def fed_step(i):
                k = i & 1
                feed.host_buffer(k)                       # waits until the previous copy out of this buffer is done
                d = feed.submit(k, B)
                ctx.batch_run(d, B, a.stages)
                feed.release(k)
            ctx.batch_reset()
            for i in range(4):
                fed_step(i)
            ctx.batch_sync(); torch.cuda.synchronize()
            tf = time.perf_counter()
            KF = 20
            for i in range(KF):
                fed_step(i)
            ctx.batch_sync(); torch.cuda.synchronize()
            tf = (time.perf_counter() - tf) / KF
            aux["host_fed_pipeline"] = {"what": "same step, frames copied from pinned host memory by the double-buffered feeder (H2D overlapped with compute)",
                                        "frames_per_s": B / tf, "ms_per_step": tf * 1e3, "h2d_GBps": B * W * H / tf / 1e9}
            feed.close()
        except Exception as e:
            aux["host_fed_pipeline"] = {"error": str(e)}

    if rank == 0:
        px = level_pixels(ctx)
        alg = algorithmic_bytes(px, NFEAT)
        fam_bytes = {"ms_fast": ("k_fast", alg["k_fast"], launches_fast), "ms_pyramid": ("k_resize", alg["k_resize"], LEVELS - 1),
                     "ms_describe": ("k_describe", alg["k_describe"], 1)}
        dom = max(fam_bytes, key=lambda k: fam[k])
        kname, bytes_per_frame, nlaunch = fam_bytes[dom]
        per_launch_bytes = bytes_per_frame * B / nlaunch
        per_launch_s = fam[dom] * 1e-3 / nlaunch
        achieved = per_launch_bytes / per_launch_s / 1e9
        # HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/profile_workload.py, separate
        # FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE correction calibrated on a 256 MiB copy); measured at
        # profiles/pmc_traffic.json["batch_frames"] frames per launch and scaled linearly to this run's batch
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj[kname]["hbm_bytes_per_launch"] * (B / pj["batch_frames"])
            except Exception:
                traffic = None
        fps = vdist.aggregate_fps(world, a.steps, B, dt)
        # the detect/match kernels are bound by integer VALU issue, not by bytes: report that roofline too
        # (instruction counts from the committed SQ_INSTS_VALU pass, peak measured by tools/valu_peak.hip)
        valu = None
        try:
            pj = json.load(open(pmc))
            vi = pj["raw"][kname]["SQ_INSTS_VALU"]["mean"] * (B / pj["batch_frames"])
            peak = pj["valu_peak_measured"]["wave_insts_per_s"]
            valu = {"kernel": kname, "wave_insts_per_launch": vi, "achieved": vi / per_launch_s, "peak": peak,
                    "unit": "wave-instr/s", "frac": vi / per_launch_s / peak}
        except Exception:
            valu = None
        out = {
            "metric": "frames/sec detect+match+pose, 752x480 mono8", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "S-752: synthetic 752x480 mono8 EuRoC-shaped stream, 1000 ORB kps x 8 levels, BF-Hamming k=2 both "
                                   "directions + ratio/sym/grid filter, essential RANSAC (adaptive, max 1000) + recoverPose",
                       "frames_per_step_per_gpu": B, "parallelism": f"stream-per-gpu x{world}" if world > 1 else "single-gpu"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": per_launch_s * 1e3},
            "valu_roofline": valu,
            "aux_kernels": aux,
            "kernels_ms_per_step": {k: round(v, 4) for k, v in fam.items()},
            "detect_describe_GBps": alg["total_detect_describe"] * B / ((fam["ms_pyramid"] + fam["ms_fast"] + fam["ms_select"] + fam["ms_describe"]) * 1e-3) / 1e9,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, frames)
            # the same port frame-parallel on the host cores this GPU's share allows (OpenCV itself would
            # run TBB inside its calls); informational, `cpu_baseline` stays the single-thread figure of the reference's own code
            try:
                import oracle_bind as orc
                th = max(1, min(16, os.cpu_count() or 1))
                ns = min(len(frames), 24 * th)
                sec, _ = orc.pipeline_stream_mt(p, frames[:ns], th)
                out["cpu_baseline_multicore"] = {"value": ns / sec, "unit": "frames/s", "cores": th, "kind": "port",
                                                 "sample": f"{ns} consecutive S-752 frames, frame-parallel std::thread pool over the same oracle pipeline"}
            except Exception as e:
                out["cpu_baseline_multicore"] = {"error": str(e)}
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
