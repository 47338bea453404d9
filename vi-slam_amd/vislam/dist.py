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


RANK_RECORD_BYTES = 64        # rank i32 | device i32 | frames_per_s f64 | seconds f64 | pci bus id char[32] | pad


def pack_rank_record(rank, device, frames_per_s, seconds, bus_id):
    import struct
    return struct.pack("<iidd32s8x", int(rank), int(device), float(frames_per_s), float(seconds), bus_id.encode()[:31])


def unpack_rank_record(raw):
    import struct
    r, d, f, s, b = struct.unpack("<iidd32s8x", bytes(raw))
    return {"rank": r, "device": d, "frames_per_s": f, "seconds": s, "pci_bus_id": b.split(b"\0", 1)[0].decode()}


def gather_rank_records(record, dist, device, world):
    """one all_gather of a 64-byte POD per rank AFTER the timed region: every rank's (rank, HIP device index, PCI bus id, frames/s, seconds),
    so that the rank-0 line proves that N ranks ran on N different devices (the driver computes the scaling itself)"""
    assert len(record) == RANK_RECORD_BYTES
    if dist is None:
        return [unpack_rank_record(record)]
    mine = torch.frombuffer(bytearray(record), dtype=torch.uint8).clone().to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [unpack_rank_record(t.cpu().numpy().tobytes()) for t in out]


def check_distinct_devices(records):
    """N ranks on N distinct PCI devices (ranks that could not read a bus id -- CPU tests -- are compared by device index)"""
    ids = [r["pci_bus_id"] or f"device-{r['device']}" for r in records]
    if len(set(ids)) != len(ids):
        raise RuntimeError(f"ranks share a device: {ids}")
    return ids
