#!/usr/bin/env python3
"""BASELINE configs[0] on a SUPPLIED dataset: the detect -> match -> pose path over the first N frames of an image directory
(EuRoC layout: cam0/data/<timestamp ns>.png, 8-bit greyscale; .pgm / .raw likewise), the way the reference's CPU main walks it
(src/ImageReader.cpp:49-82: sorted listing, imread GRAYSCALE; src/Camera.cpp:127: ORB::create(200)).  No dataset ships with this image and
there is no network: tests/test_ingest.py drives this tool on a synthesised EuRoC-shaped directory; it is here so that a maintainer who HAS
MH_01 can run `python tools/run_directory.py /data/MH_01/mav0/cam0/data --frames 200 --check 20` and read one JSON line.

  frames -> vis_image_read into the feeder's pinned buffers (host decode) -> vis_feeder_submit (H2D on the copy stream)
         -> vis_batch_run(STAGE_FRAME) per batch; results of every frame downloaded.
  --check K: the first K frames also go through the CPU oracle's per-frame pipeline; keypoints, descriptors, good matches and the pose
             record are compared (bit-exact / 1e-7) -- the same checks as tests/test_configs_gpu.py::test_config1...
  --cpu-seconds S: times the oracle on the same frames for about S seconds (the `cpu_baseline` of this dataset).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (the process's HIP runtime: before the library)
import vislam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--frames", type=int, default=200, help="first N frames of the sorted listing (BASELINE configs[0]: 200)")
    ap.add_argument("--nfeatures", type=int, default=200, help="ORB::create(n): 200 = the reference's CPU main, 1000 = its GPU main")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--check", type=int, default=0, help="compare the first K frames against the CPU oracle")
    ap.add_argument("--cpu-seconds", type=float, default=0.0)
    ap.add_argument("--raw-size", default=None, help="WxH of headerless .raw files")
    a = ap.parse_args()

    names = vislam.image_list(a.directory)[:a.frames]
    if len(names) < 2:
        raise SystemExit(f"{a.directory}: {len(names)} image files (.png / .pgm / .raw)")
    paths = [os.path.join(a.directory, n) for n in names]
    if a.raw_size:
        w, h = (int(x) for x in a.raw_size.lower().split("x"))
    else:
        h, w = vislam.image_read(paths[0]).shape
    stamps = [vislam.image_time(n) for n in names]
    p = vislam.default_params()
    p.nfeatures, p.w_size, p.h_size = a.nfeatures, w, h
    p.fy = p.fx
    ctx = vislam.Context(0, p)
    B = min(a.batch, len(paths))
    ctx.batch_plan(w, h, w, B)
    feed = vislam.Feeder(ctx, w, h, B)
    n = len(paths)
    host = np.empty((n, h, w), np.uint8) if (a.check or a.cpu_seconds > 0) else None
    t_decode = 0.0
    results = []
    t0 = time.perf_counter()
    for bi, first in enumerate(range(0, n, B)):
        k, nb = bi & 1, min(B, n - first)
        buf = feed.host_buffer(k)                            # waits until the previous copy out of this buffer is done
        td = time.perf_counter()
        for i in range(nb):
            buf[i] = vislam.image_read(paths[first + i], w, h)
            if host is not None:
                host[first + i] = buf[i]
        t_decode += time.perf_counter() - td
        d = feed.submit(k, nb)
        ctx.batch_run(d, nb, vislam.STAGE_FRAME)
        feed.release(k)
        ctx.batch_sync()                                     # (results are fetched per batch below: this harness reports, it does not pipeline)
        if ctx.batch_status() != 0:
            raise SystemExit("device capacity flag set")
        for i in range(nb):
            kp, ds = ctx.batch_keypoints(i)
            g, nsym = ctx.batch_matches(i)
            results.append((kp, ds, g, nsym, ctx.batch_pose(i)))
    dt = time.perf_counter() - t0
    out = {"directory": a.directory, "frames": n, "width": w, "height": h, "nfeatures": a.nfeatures, "first_timestamp": stamps[0],
           "median_frame_interval_ns": int(np.median(np.diff(stamps))) if n > 1 else None,
           "frames_per_s_incl_decode_and_downloads": n / dt, "host_decode_s": t_decode, "frames_per_s_host_decode_alone": n / t_decode if t_decode > 0 else None,
           "keypoints_mean": float(np.mean([len(r[0]) for r in results])), "good_matches_mean": float(np.mean([len(r[2]) for r in results[1:]])),
           "inliers_mean": float(np.mean([r[4]["n_inliers"] for r in results[1:]]))}
    if a.check:
        import oracle_bind as orc
        prev, bad = None, []
        for t in range(min(a.check, n)):
            ok, od, r = orc.pipeline_frame(p, host[t], prev)
            kp, ds, g, nsym, pose = results[t]
            same = kp.tobytes() == ok.tobytes() and (ds == od).all() and nsym == r.n_sym and len(g) == r.n_good
            same = same and pose["n_inliers"] == r.n_inliers and pose["iters_run"] == r.iters_run
            if same and r.n_inliers:
                oE = np.array(r.E).reshape(3, 3)
                s = 1.0 if float((pose["E"] * oE).sum()) >= 0 else -1.0
                same = np.abs(pose["E"] - s * oE).max() <= 1e-9 and pose["n_pose_good"] == r.n_pose_good and np.abs(pose["R"] - np.array(r.R).reshape(3, 3)).max() <= 1e-7
            if not same:
                bad.append(t)
            prev = (ok, od)
        out["checked_frames"] = min(a.check, n)
        out["frames_differing_from_the_oracle"] = bad
    if a.cpu_seconds > 0:
        import oracle_bind as orc
        prev, m = None, 0
        tc = time.perf_counter()
        for t in range(n):
            ok, od, _r = orc.pipeline_frame(p, host[t], prev)
            prev = (ok, od)
            m += 1
            if time.perf_counter() - tc > a.cpu_seconds:
                break
        out["cpu_baseline"] = {"value": m / (time.perf_counter() - tc), "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"{m} frames of this directory"}
    feed.close()
    ctx.close()
    print(json.dumps(out, allow_nan=False))
    return 1 if out.get("frames_differing_from_the_oracle") else 0


if __name__ == "__main__":
    sys.exit(main())
