"""Fixed-seed slices of the randomised parity tools (tools/stress_*.py: random problems through the C ABI against the oracle, bit for bit /
to the stated tolerances) so that the driver's `pytest -m gpu` record shows them: a few seconds each, the same seeds and the same NUMBER of cases every run.  The
tools themselves run for minutes with other seeds (earlier rounds: DESIGN_history.md); they found two real bugs in round 4."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# bounded by a CASE COUNT per seed: every box runs the same cases; the 150 s are a guard only (rates measured in round 5: pose 96,
# detect 26, batch 2, match 840, align 46 cases per second)
@pytest.mark.parametrize("tool,cases,seed", [("stress_pose.py", 250, 501), ("stress_detect.py", 100, 502), ("stress_batch.py", 8, 503),
                                             ("stress_match.py", 2000, 504), ("stress_align.py", 120, 505)])
def test_stress_slice(built, tool, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "150", str(seed), str(cases)], capture_output=True, text=True, timeout=300)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    last = [l for l in r.stdout.splitlines() if l.startswith("stress_")][-1]
    assert " 0 failures" in last and f"seed {seed}" in last, last
    n = int(last.split(":")[1].split()[0])
    assert n == cases, f"{tool} seed {seed}: ran {n} of {cases} cases before the 150 s guard ({last})"


@pytest.mark.parametrize("mode", ["parallax", "main"])
def test_soak_slice(built, mode):
    """tools/soak_pipeline.py for a few seconds: every pass over a ring of resident frames (launches queued back to back, results into
    pinned memory behind each launch) reproduces pass 0's records byte for byte (minutes of it: DESIGN.md section 4.3).
    "main" = the GPU main's sequence: gradients on the side stream into the plan's two sets in turn, vis_batch_align on the pose stream."""
    passes = 12                                                     # a pass count, not seconds: the same work on every box (150 s guard)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_pipeline.py"), "150", "64", "6", mode, str(passes)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
    last = [l for l in r.stdout.splitlines() if l.startswith("soak_pipeline:")][-1]
    assert " 0 passes differ" in last, last
    n = int(last.split(":")[1].split()[0])
    assert n == passes, f"soak {mode}: {n} of {passes} passes before the 150 s guard ({last})"
