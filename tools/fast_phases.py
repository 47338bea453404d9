#!/usr/bin/env python3
"""Per-phase cycle shares of k_fast.  Needs the diagnostic build (`make -C vi-slam_amd/csrc clean all EXTRA=-DVIS_FAST_PROFILE`):
with VIS_FAST_STAMPS=1 thread 0 of every workgroup then stores its s_memtime deltas between the phase barriers into a
private record.  Shares only -- the stamped run is not a timing; the default build carries no stamp code."""
import ctypes as C
import os
import sys

os.environ["VIS_FAST_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vislam  # noqa: E402

W, H, B = 752, 480, 256
p = vislam.default_params()
ctx = vislam.Context(0, p)
cv = vislam.synth_canvas(4096, 0xE0C00001)
dcv = torch.from_numpy(cv).cuda()
fr = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
ctx.synth_frames_device(dcv.data_ptr(), 4096, 0xE0C00001, 0, B, W, H, W, fr.data_ptr())
ctx.batch_plan(W, H, W, B)
for _ in range(3):
    ctx.batch_run(fr.data_ptr(), B, 1)
ctx.batch_sync()
out = (C.c_ulonglong * 16)()
vislam.lib.vis_debug_fast_stamps(out)
ctx.batch_run(fr.data_ptr(), B, 1)
ctx.batch_sync()
assert vislam.lib.vis_debug_fast_stamps(out) == 0
v = np.array(list(out), dtype=np.float64)
names = ["load+clear", "pretest", "score", "nms", "-", "-"]
tot = v[:6].sum()
nwg = v[8]
print("workgroups", int(nwg), "queue px per tile", v[6] / nwg)
for n, x in zip(names, v[:6]):
    print(f"{n:8s} {x / nwg:9.0f} cycles/workgroup  {100 * x / tot:5.1f} %")
print(f"total    {tot / nwg:9.0f} cycles/workgroup (thread 0, includes barrier waits)")
