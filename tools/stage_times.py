import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"vi-slam_amd"))
import torch, vislam, bench
B, R = 1024, 2
p = vislam.default_params(); p.nfeatures, p.nlevels, p.w_size, p.h_size = bench.NFEAT, bench.LEVELS, bench.W, bench.H
p.fy = p.fx
ctx = vislam.Context(0, p)
stream = bench.Stream(ctx, "cuda:0", bench.W, bench.H, B * R, 0xE0C00001)
ctx.batch_plan(bench.W, bench.H, bench.W, B)
for st, name in ((1, "detect"), (3, "detect+match"), (7, "all")):
    dt, _ = bench.timed_steps(ctx, stream, B, R, st, 20, 3)
    print(name, round(dt / 20 * 1e3, 3), "ms/step", round(B * 20 / dt), "fps")
