"""CPU: the N>1 path of bench.py (stream sharding, intrinsics broadcast, max-over-ranks timing) under
torch.distributed gloo with world_size 2 -- the same functions bench.py runs over RCCL."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
    import vislam
    from vislam import dist as vd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = vislam.default_params()
    if rank == 0:
        p.nfeatures, p.fx, p.cx = 777, 123.5, 42.25          # only rank 0 knows the calibration
    else:
        p.nfeatures, p.fx = 1, 1.0
    dev = torch.device("cpu")
    got = vd.broadcast_params(p, dist, dev, rank)
    t = vd.max_over_ranks(1.0 + rank, dist, dev)
    # the per-rank record of the report: gathered by every rank (bench.py prints rank 0's copy)
    recs = vd.gather_rank_records(vd.pack_rank_record(rank, rank, 1000.0 * (rank + 1), 1.0 + rank, f"0000:0{rank}:00.0"), dist, dev, world)
    ids = vd.check_distinct_devices(recs)
    q.put((rank, got.nfeatures, got.fx, got.cx, got.ransac_seed, vd.stream_seed(rank, world), t, recs, ids))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    for r in res:
        assert (r[1], r[2], r[3]) == (777, 123.5, 42.25)     # every rank holds rank 0's intrinsics
        assert r[4] == 0xFFFFFFFFFFFFFFFF
        assert r[6] == 2.0                                   # MAX over ranks
    assert res[0][5] != res[1][5]                            # distinct, independent streams
    for r in res:                                            # every rank holds both records, in rank order
        assert [x["rank"] for x in r[7]] == [0, 1] and [x["device"] for x in r[7]] == [0, 1]
        assert [x["frames_per_s"] for x in r[7]] == [1000.0, 2000.0] and [x["seconds"] for x in r[7]] == [1.0, 2.0]
        assert r[8] == ["0000:00:00.0", "0000:01:00.0"]


def test_rank_records(built):
    sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
    from vislam import dist as vd
    import pytest
    rec = vd.pack_rank_record(3, 5, 123456.5, 0.25, "0000:c3:00.0")
    assert len(rec) == vd.RANK_RECORD_BYTES == 64                         # = struct RankRec of host/mgpu_main.cpp
    assert vd.unpack_rank_record(rec) == {"rank": 3, "device": 5, "frames_per_s": 123456.5, "seconds": 0.25, "pci_bus_id": "0000:c3:00.0"}
    one = vd.gather_rank_records(rec, None, None, 1)                      # world size 1: a one-entry list, no collective
    assert one == [vd.unpack_rank_record(rec)]
    with pytest.raises(RuntimeError):                                     # two ranks on one device are refused
        vd.check_distinct_devices([vd.unpack_rank_record(rec), vd.unpack_rank_record(rec)])
    assert vd.check_distinct_devices([{"pci_bus_id": "", "device": 0}, {"pci_bus_id": "", "device": 1}]) == ["device-0", "device-1"]


def test_aggregate_formula(built):
    sys.path.insert(0, os.path.join(ROOT, "vi-slam_amd"))
    from vislam import dist as vd
    assert vd.aggregate_fps(8, 10, 64, 2.0) == 8 * 10 * 64 / 2.0
    assert vd.stream_seed(0, 1) == 0xE0C00001 and vd.stream_seed(3, 8) == 0xE0C00013
    p = __import__("vislam").default_params()
    q = vd.tensor_to_params(vd.params_to_tensor(p))
    assert bytes(p) == bytes(q)
