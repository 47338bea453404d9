import os, sys, torch, torch.distributed as d
sys.path.insert(0, "vi-slam_amd")
d.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29511")
torch.cuda.set_device(0)
import vislam
from vislam import dist as vdist
p = vislam.default_params(); p.nfeatures = 777
q = vdist.broadcast_params(p, d, torch.device("cuda", 0), 0)
print("broadcast ok", q.nfeatures, "max", vdist.max_over_ranks(1.25, d, torch.device("cuda", 0)))
d.barrier(); d.destroy_process_group()
