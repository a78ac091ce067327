"""The ONE JSON line bench.py prints, kept small enough for any reader of it.

bench.py measures a lot (fifteen secondary legs, each with its workload text, roofline, counter sources, CPU sample).
Round 5's line carrying all of it was 20 kB and the driver's parser gave up on it.  So there are two artefacts now:

  full      everything, prose included -> a side file (bench_extra.json beside bench.py, or --full-out) and stderr
  the line  compact(full): the contract keys, `roofline` and `cpu_baseline` reduced to numbers and one short sample
            string, `secondary` (the metric's SW half), one `[value, frac, traffic_ratio]` triple per secondary leg, the
            spot checks and the summary.  LIMIT bytes at most (tests/test_bench_line_cpu.py holds it to that for the
            one-GPU and the N-rank shape), floats cut to 6 significant digits, allow_nan=False.

Pure Python: nothing here touches the GPU or the oracle.
"""
from __future__ import annotations

import json
import math

LIMIT = 8192          # bytes of the printed line, newline included
_SIG = 6

# the one figure that stands for a leg in the compact line, first match wins
_VALUE_KEYS = ("cell_updates_per_s", "windows_per_s", "pairs_per_s_counts", "pairs_per_s", "bases_per_s", "file_GBs",
               "sequences_per_s", "kmers_per_s", "value")


def _num(x):
    """floats to 6 significant digits; NaN / infinities (which json would print as bare words) to None"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        if x == int(x) and abs(x) < 1e15:
            return int(x)
        return float(f"{x:.{_SIG}g}")
    if isinstance(x, dict):
        return {str(k): _num(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v) for v in x]
    try:                      # numpy scalars and the like
        return _num(x.item())
    except Exception:
        return str(x)


def _cut(s, n):
    if not isinstance(s, str) or len(s) <= n:
        return s
    return s[: n - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _leg_triple(leg):
    """[value, fraction of the leg's own bound, counter traffic / algorithmic bytes] of one secondary leg"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg:
        return {"error": _cut(str(leg["error"]), 120)}
    val = next((leg[k] for k in _VALUE_KEYS if isinstance(leg.get(k), (int, float))), None)
    rf = leg.get("roofline") if isinstance(leg.get("roofline"), dict) else {}
    frac = rf.get("frac", leg.get("hbm_frac", leg.get("frac_of_valu_issue_ceiling")))
    return [val, frac, rf.get("traffic_ratio")]


def _legs(extra):
    out = {}
    for name, leg in (extra or {}).items():
        if name == "e2e_host_pointers" and isinstance(leg, dict):
            # PCIe-inclusive host-pointer rates: never `value`; one number per entry point
            out["e2e"] = {k: next((v[kk] for kk in v if kk.endswith("_per_s") and isinstance(v[kk], (int, float))), None)
                          for k, v in leg.items() if isinstance(v, dict)}
        elif name == "mash_distance_allgather" and isinstance(leg, dict):
            o = _pick(leg, ("pairs_per_s", "ms_per_step", "self_pairs_share_all_hashes"))
            if "error" in leg:
                o["error"] = _cut(str(leg["error"]), 160)
            ag = leg.get("allgather") or {}
            o["allgather"] = _pick(ag, ("ms", "bus_GBs", "bytes_received_per_rank"))
            if isinstance(ag.get("via"), str):
                o["allgather"]["via"] = "rccl" if "RCCL" in ag["via"] and "gloo" not in ag["via"] else "gloo"
            for k in ("replicated_index", "sharded_index"):
                if isinstance(leg.get(k), dict):
                    o[k] = _pick(leg[k], ("ms_per_step", "row_block_equals_replicated", "index_bytes_received_per_rank"))
                    if "error" in leg[k] or "skipped" in leg[k]:
                        o[k]["note"] = _cut(str(leg[k].get("error", leg[k].get("skipped"))), 80)
            out[name] = o
        elif name.endswith("_check") and isinstance(leg, dict):
            out[name] = {k: v for k, v in leg.items() if isinstance(v, bool)}
        else:
            t = _leg_triple(leg)
            if t is not None:
                out[name] = t
    return out


def compact(full: dict) -> dict:
    """the printed line: contract keys first, then roofline / cpu_baseline / secondary / legs / checks / summary"""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                     "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = dict(full.get("config") or {})
    cfg["workload"] = _cut(cfg.get("workload"), 140)
    cfg["parallelism"] = _cut(cfg.get("parallelism"), 80)
    line["config"] = cfg
    la = dict(full.get("launch") or {})
    devs = la.pop("devices", None) or []
    la["devices"] = [d.get("device") for d in devs if isinstance(d, dict)]
    names = sorted({str(d.get("name")) for d in devs if isinstance(d, dict)})
    la["device_names"] = names[:2]
    la["pci_bus_ids"] = [d.get("pci_bus_id") for d in devs if isinstance(d, dict)]
    la["backend"] = _cut(la.get("backend"), 40)
    line["launch"] = la

    rf = full.get("roofline") or {}
    r = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "kernel", "kernel_ms",
                   "kernel_ms_median", "algorithmic_bytes_per_launch"))
    r.setdefault("traffic", None)
    # the counter figure comes from a committed rocprofv3 --pmc pass of this command, not from this run
    r["traffic_measured_by_this_run"] = False
    vi = rf.get("valu_issue")
    if isinstance(vi, dict):
        r["valu_issue"] = _pick(vi, ("instructions_per_kmer", "cycles_per_instruction", "clock_GHz", "simds",
                                      "ceiling_kmers_per_s", "frac"))
        if "error" in vi:
            r["valu_issue"]["error"] = _cut(str(vi["error"]), 100)
    r.update(_pick(rf, ("secondary_metric", "secondary_value", "secondary_ms_per_step", "secondary_bound",
                        "secondary_frac", "secondary_peak_T_cell_updates_per_s")))
    line["roofline"] = r

    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "secondary_value", "secondary_unit"))
        c["sample"] = _cut(cb.get("sample"), 200)
        for k in ("tight_variant", "all_cores", "all_cores_tight_variant"):
            if isinstance(cb.get(k), dict):
                c[k] = _pick(cb[k], ("value", "cores"))
        if isinstance(cb.get("config0_phix174"), dict):
            c["config0_phix174"] = _pick(cb["config0_phix174"], ("ms_per_sketch", "kmers_per_s", "cores", "sketch0"))
        line["cpu_baseline"] = c

    st = full.get("strong")
    if isinstance(st, dict):
        line["strong"] = _pick(st, ("scaling", "value", "unit", "ms_per_step", "this_rank_kernel_ms"))

    if isinstance(full.get("extra"), dict):
        line["legs"] = _legs(full["extra"])
        line["legs_format"] = "[value, frac_of_own_bound, counter_traffic/algorithmic_bytes]; everything else in `full`"

    sec = full.get("secondary")
    if isinstance(sec, dict):
        s = _pick(sec, ("value", "unit", "ms_per_step", "value_with_strings", "ms_per_step_with_strings"))
        s["metric"] = _cut(sec.get("metric"), 120)
        if isinstance(sec.get("roofline"), dict):
            s["roofline"] = _pick(sec["roofline"], ("bound", "achieved", "peak", "unit", "frac"))
            s["roofline"]["kernel"] = _cut(sec["roofline"].get("kernel"), 60)
        if isinstance(sec.get("cpu_baseline"), dict):
            s["cpu_baseline"] = _pick(sec["cpu_baseline"], ("value", "unit", "cores", "kind"))
            s["cpu_baseline"]["sample"] = _cut(sec["cpu_baseline"].get("sample"), 120)
        line["secondary"] = s

    line["parity_spot_check"] = full.get("parity_spot_check")
    sm = full.get("summary")
    if isinstance(sm, dict):
        s = _pick(sm, ("kmers_per_s", "hbm_frac", "sw_cell_updates_per_s", "sw_valu_frac"))
        tc = sm.get("traffic_checks")
        if isinstance(tc, dict):
            s["traffic_checks"] = {"checked": tc.get("checked"), "failed": [_cut(str(f), 80) for f in (tc.get("failed") or [])][:4]}
            if "error" in tc:
                s["traffic_checks"]["error"] = _cut(str(tc["error"]), 80)
        p = sm.get("parity_spot_check")
        s["parity_all_true"] = (all(v for v in p.values() if isinstance(v, bool)) and "error" not in p) if isinstance(p, dict) else p
        line["summary"] = s
    if full.get("full"):
        line["full"] = full["full"]
    return _num(line)


def dumps(line: dict) -> str:
    return json.dumps(line, allow_nan=False, separators=(",", ":"))


def render(full: dict) -> str:
    """the printed line (no newline): compact(full), and if a pathological input still overshoots, shed the optional parts
    (legs, then launch detail) rather than print something a reader cannot parse"""
    line = compact(full)
    text = dumps(line)
    for drop in ("legs", "legs_format", "strong", "launch"):
        if len(text) + 1 <= LIMIT:
            break
        if drop in line:
            line[drop] = "dropped: line over %d bytes; see `full`" % LIMIT if drop == "legs" else None
            text = dumps(line)
    return text


def render_full(full: dict) -> str:
    return json.dumps(_strip(full), allow_nan=False, default=str)


def _strip(x):
    """full-precision copy for the side file: only the NaN/inf guard of _num, no rounding"""
    if isinstance(x, float):
        return x if math.isfinite(x) else None
    if isinstance(x, dict):
        return {str(k): _strip(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_strip(v) for v in x]
    return x
