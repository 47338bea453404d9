"""The compact LAST line of bench.py (benchline.py): short enough for a 2000-byte tail, strict JSON, the contract keys present.
BENCH_r05.json: the driver could not parse round 5's 23 KB line (`parsed: null`); the CPU half of this file runs the formatter on every
committed full record, the GPU half on the real stdout of a short bench run."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import benchline  # noqa: E402


def strict(s):
    def bad(x):
        raise ValueError(f"non-finite constant {x} in the line")
    return json.loads(s, parse_constant=bad)


def check_line(s, full):
    assert "\n" not in s and len(s.encode()) < 1900, len(s.encode())
    c = strict(s)
    for k in benchline.REQUIRED:
        if k in full:
            assert k in c, k
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "data", "higher_is_better", "vs_baseline"):
        assert c[k] == full[k], k
    assert abs(c["value"] - full["value"]) <= 1e-5 * full["value"] and abs(c["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    assert set(c["config"]) >= {"workload", "parallelism"} and len(c["config"]["workload"]) <= 200 and "model" not in c["config"]
    r = c["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # numbers only below the top level: no prose fields
    for blk in ("roofline", "issue_roofline", "matcher_roofline", "t1", "legs", "latency_ms", "clocks", "aux", "ms"):
        for k, v in (c.get(blk) or {}).items():
            assert v is None or isinstance(v, (int, float)) or k in ("bound", "kernel", "unit", "limited_by"), (blk, k, v)
    return c


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench*.json"))))
def test_committed_records_format(path):
    txt = open(path).read()
    full = strict(txt if txt.lstrip().startswith("{") else [l for l in txt.splitlines() if l.startswith("{")][-1])
    if "full" in full and len(txt) < 2500:
        # a committed COMPACT line (the stdout of a bench run, profiles/r06_bench.json): it must itself satisfy the contract, and the full
        # record it names travels beside it
        c = check_line(txt.strip(), full)
        assert c["roofline"]["kernel"].startswith("k_") and c["cpu_baseline"]["kind"] == "port" and c["t1"]["target"] == 0.4
        beside = os.path.join(os.path.dirname(path), os.path.basename(path).replace(".json", "_full.json"))
        assert os.path.exists(beside), beside
        fr = strict(open(beside).read())
        assert abs(fr["value"] - c["value"]) <= 1e-5 * fr["value"] and benchline.compact_line(fr, c["full"]) == txt.strip()
        return
    c = check_line(benchline.compact_line(full, "gpurun_out/bench_full.json"), full)
    if "cpu_baseline" in full:
        assert c["cpu_baseline"]["kind"] in ("port", "reference") and c["cpu_baseline"]["cores"] >= 1 and len(c["cpu_baseline"]["sample"]) <= 80
    if "detect_kernels" in full and full["detect_kernels"]:
        assert c["t1"]["target"] == 0.4 and c["t1"]["k_describe_frac"] is not None


def test_eight_ranks_and_every_block_fit():
    full = strict(open(os.path.join(ROOT, "profiles", "r05_bench.json")).read())
    full["n_gpus"] = 8
    full["ranks"] = [dict(full["ranks"][0], rank=i, device=i) for i in range(8)]
    full["clocks"] = {"sclk_mhz": 2345.0, "mclk_mhz": 2000.0, "watts": 1255.5, "samples": 9, "source": "x"}
    sf = full["legs"]["single_frame_api"]
    sf.update(ms_per_frame_p99=1.234, ms_per_frame_max=31.23)
    sf["add_frame_gpu"].update(ms_p99=2.25, ms_max=3.31)
    c = check_line(benchline.compact_line(full, "gpurun_out/bench_full.json"), full)
    assert len(c["ranks"]) == 8 and c["roofline"] and c["cpu_baseline"]            # the contract blocks survive whatever is dropped


def test_non_finite_values_become_null_and_overflow_is_an_error():
    full = strict(open(os.path.join(ROOT, "profiles", "r05_bench.json")).read())
    full["roofline"]["traffic"] = float("nan")
    full["legs"]["config3_s1080"]["frames_per_s"] = float("inf")
    c = strict(benchline.compact_line(full))
    assert c["roofline"]["traffic"] is None and c["legs"]["c3"] is None
    with pytest.raises(ValueError):
        benchline.compact_line(full, limit=300)


@pytest.mark.gpu
def test_real_bench_stdout_is_one_compact_line(built, tmp_path):
    fp = tmp_path / "full.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--launches-per-step", "4", "--no-legs", "--no-cpu-baseline",
                        "--full-out", str(fp)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    out = r.stdout.rstrip("\n").splitlines()
    assert out and out[-1].startswith("{") and sum(l.startswith("{") for l in out) == 1, r.stdout[-2000:]
    full = strict(open(fp).read())
    c = check_line(out[-1], full)
    assert c["roofline"]["kernel"].startswith("k_") and c["t1"]["chain_frac"] > 0 and c["clocks"] is not None
    assert c["ranks"][0][:2] == [0, 0]
