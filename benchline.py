"""benchline.py -- the compact LAST line of bench.py.

bench.py measures many things (legs, counter-derived figures, per-kernel tables).  The full record goes to a file
(`bench_full.json`); what the driver reads is ONE short strict-JSON line, numbers only, below 1900 bytes so that a 2000-byte
tail still holds all of it (BENCH_r05.json: the 23 KB line of round 5 could not be parsed).  Pure Python, no torch, no HIP:
tests/test_benchline.py runs it on the committed full records.
"""
import json
import math

LIMIT = 1900
T1_TARGET = 0.40          # north_star: >= 40 % of the HBM roofline for detect/describe
LEG_SHORT = {"align_n4": "align_pairs", "gpu_main_sequence": "gpu_main", "s752_fixed1000": "f1000", "s752_parallax": "parallax", "s752_results_d2h": "d2h",
             "s752_mispredicted_thresholds": "mispredict", "single_frame_api": "api_1frame", "config3_s1080": "c3", "config5_s2160": "c5"}
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def _r(x, sig=5):
    """numbers to `sig` significant digits; anything not finite becomes null (strict JSON has no NaN / Infinity)"""
    if x is None or isinstance(x, (bool, str)):
        return x
    if isinstance(x, int):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if not math.isfinite(x):
        return None
    if x == 0:
        return 0
    d = sig - 1 - int(math.floor(math.log10(abs(x))))
    y = round(x, d)
    return int(y) if d <= 0 else y


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _pick(d, keys, sig=5):
    return {k: _r(d.get(k), sig) for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else None


def t1_status(full):
    """north_star's second target as numbers: every detect/describe kernel and the whole chain against 8 TB/s"""
    dk = full.get("detect_kernels") or {}
    fr = lambda k: _r(_get(dk, k, "hbm_frac"), 3)                                  # noqa: E731
    chain = full.get("detect_describe_GBps")
    peak = _get(full, "roofline", "peak") or 8000.0
    return {"target": T1_TARGET, "chain_frac": _r(chain / peak, 3) if chain else None, "k_describe_frac": fr("k_describe"), "k_fast_frac": fr("k_fast"),
            "k_resize_frac": fr("k_resize"), "issue_frac": _r(_get(full, "issue_roofline", "frac"), 3)}


def _leg_value(leg):
    if not isinstance(leg, dict) or "error" in leg:
        return None
    for k in ("frames_per_s", "pairs_per_s", "frames_per_s_one_at_a_time"):
        if k in leg:
            return _r(leg[k], 4)
    return None


def compact(full, full_path=None):
    """the dict of the compact line (see compact_line)"""
    cfg = full.get("config") or {}
    out = {k: _r(full.get(k), 6) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                             "vs_baseline", "dtype", "data")}
    if "rehearsal" in full:
        out["rehearsal"] = True
    out["config"] = {"workload": str(cfg.get("workload_short") or cfg.get("workload") or "")[:140],
                     **{k: cfg.get(k) for k in ("frames_per_step_per_gpu", "launches_per_step", "frames_per_launch", "parallelism") if k in cfg}}
    out["roofline"] = _pick(full.get("roofline"), ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "limited_by"))
    iss = full.get("issue_roofline") or {}
    if "error" not in iss:
        tera = lambda k: _r(iss[k] / 1e12, 4) if iss.get(k) else None                    # noqa: E731
        out["issue_roofline"] = {"frac": _r(iss.get("frac"), 4), "achieved": tera("achieved"), "peak_half_rate": tera("peak_measured_half_rate_class"),
                                 "peak_full_rate": tera("peak_measured_full_rate_class"), "unit": "Twave-inst/s"}
    out["matcher_roofline"] = _pick(full.get("matcher_roofline"), ("frac", "l2_hit", "mfma_busy_frac", "traffic_over_algorithmic"), 3)
    out["t1"] = t1_status(full)
    ms = full.get("kernels_ms_per_launch") or {}
    if ms:
        out["ms"] = {k[3:]: _r(v, 3) for k, v in ms.items()}
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {**_pick(cb, ("value", "unit", "cores", "kind"), 4), "sample": str(cb.get("sample_short") or cb.get("sample") or "")[:64]}
    mt = full.get("cpu_baseline_multicore")
    if isinstance(mt, dict) and "value" in mt:
        out["cpu_mt"] = _pick(mt, ("value", "cores"), 4)
    ranks = full.get("ranks")
    if ranks:
        out["ranks"] = [[r.get("rank"), r.get("device"), _r(r.get("frames_per_s"), 4)] for r in ranks]
    legs = full.get("legs")
    if isinstance(legs, dict):
        out["legs"] = {LEG_SHORT.get(k, k): _leg_value(v) for k, v in legs.items()}
        sf = legs.get("single_frame_api") or {}
        lat = {}
        for k_out, k_in in (("api50", "ms_per_frame_p50"), ("api99", "ms_per_frame_p99"), ("api_max", "ms_per_frame_max")):
            if k_in in sf:
                lat[k_out] = _r(sf[k_in], 3)
        af = sf.get("add_frame_gpu") or {}
        for k_out, k_in in (("add50", "ms_p50"), ("add99", "ms_p99"), ("add_max", "ms_max")):
            if k_in in af:
                lat[k_out] = _r(af[k_in], 3)
        if lat:
            out["latency_ms"] = lat
        c5 = _get(legs, "config5_s2160", "matcher_roofline")
        if isinstance(c5, dict):
            out["c5_matcher"] = _pick(c5, ("frac", "l2_hit", "mfma_busy_frac", "traffic_over_algorithmic"), 3)
    aux = full.get("aux_kernels")
    if isinstance(aux, dict):
        out["aux"] = {"gradient_frac": _r(_get(aux, "gradient_batch", "frac"), 4), "host_fed_fps": _r(_get(aux, "host_fed_pipeline", "frames_per_s"), 4)}
    if isinstance(full.get("clocks"), dict):
        out["clocks"] = _pick(full["clocks"], ("sclk_mhz", "mclk_mhz", "watts"), 4)
    if full_path:
        out["full"] = full_path
    return out


def compact_line(full, full_path=None, limit=LIMIT):
    """one strict-JSON line below `limit` bytes.  Optional blocks are dropped from the back, least important first, should a line ever come
    out too long (it does not at N = 8 with every leg present: tests/test_benchline.py); the contract keys are never dropped."""
    out = compact(full, full_path)
    order = ["aux", "ms", "c5_matcher", "latency_ms", "cpu_mt", "clocks", "legs", "ranks", "t1", "matcher_roofline", "issue_roofline", "full"]
    while True:
        s = json.dumps(out, allow_nan=False, separators=(",", ":"))
        if len(s.encode()) <= limit or not order:
            break
        out.pop(order.pop(0), None)
    if len(s.encode()) > limit:
        raise ValueError(f"compact bench line is {len(s.encode())} bytes with every optional block dropped (limit {limit})")
    return s
