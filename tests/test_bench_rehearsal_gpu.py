"""GPU: bench.py's N > 1 code path (per-rank streams, parameter broadcast, barriers, max over ranks, the gathered `ranks` records, the
rank-0 line) run for real with two ranks -- on the ONE GPU of the test box, over gloo, which is what `--rehearse-on-one-gpu` is for
(RCCL refuses two ranks on one device; the 1 -> 8 GPU runs are the driver's).  Not a scaling measurement: the line says "rehearsal"."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu(built, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29531",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--launches-per-step", "4", "--rehearse-on-one-gpu", "--no-legs",
           "--full-out", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-2000:]          # ONE line, from rank 0: the compact one
    c = json.loads(lines[0])
    assert len(lines[0].encode()) < 1900 and r.stdout.rstrip().endswith(lines[0])
    assert c["n_gpus"] == 2 and c["scaling"] == "weak" and c["rehearsal"] is True
    assert [x[:2] for x in c["ranks"]] == [[0, 0], [1, 0]]                                # [rank, device, frames/s]
    j = json.load(open(tmp_path / "full.json"))                                          # the full record, written beside it
    assert j["n_gpus"] == 2 and "rehearsal" in j and abs(j["value"] - c["value"]) <= 1e-5 * j["value"]
    assert [x["rank"] for x in j["ranks"]] == [0, 1] and all(x["device"] == 0 and x["pci_bus_id"] for x in j["ranks"])
    # whole-job value = the frames of BOTH ranks / the slowest rank's time
    frames = 2 * j["steps"] * j["config"]["frames_per_step_per_gpu"]
    assert abs(j["value"] - frames / (j["ms_per_step"] * 1e-3 * j["steps"])) < 1e-6 * j["value"]
    assert "roofline" in j
