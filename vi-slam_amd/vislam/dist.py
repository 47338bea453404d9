"""Multi-GPU plumbing for the frame-sharded path: one process per GPU under
torch.distributed ("nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The hot path has NO data-path collective: camera streams are independent (a frame only needs the
previous frame of its own stream, src/Camera.cpp:149-150), so rank r simply owns stream r.  The only
exchange is one broadcast of the parameter/intrinsics POD from rank 0 at start-up ("RCCL broadcast of
intrinsics only") and one MAX all-reduce of the elapsed time for the report."""
import ctypes as C

import torch

from . import Params

BASE_SEED = 0xE0C00010        # config 4: stream r uses seed BASE_SEED + r
SINGLE_SEED = 0xE0C00001      # S-752


def stream_seed(rank, world):
    return SINGLE_SEED if world == 1 else BASE_SEED + rank


def params_to_tensor(p):
    return torch.frombuffer(bytearray(bytes(p)), dtype=torch.uint8).clone()


def tensor_to_params(t):
    p = Params()
    raw = bytes(t.cpu().numpy().tobytes())
    assert len(raw) == C.sizeof(Params)
    C.memmove(C.byref(p), raw, C.sizeof(Params))
    return p


def broadcast_params(p, dist, device, rank):
    """rank 0's struct wins; every rank returns an identical copy (< 256 bytes, one collective)."""
    buf = params_to_tensor(p) if rank == 0 else torch.zeros(C.sizeof(Params), dtype=torch.uint8)
    if dist is None:
        return tensor_to_params(buf)
    buf = buf.to(device)
    dist.broadcast(buf, src=0)
    return tensor_to_params(buf)


def max_over_ranks(seconds, dist, device):
    if dist is None:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_fps(world, steps, frames_per_step, max_seconds):
    """whole-job throughput: the frames ALL ranks processed / the slowest rank's time"""
    return world * steps * frames_per_step / max_seconds
