"""bench.py's printed line stays parseable: at most bench_line.LIMIT (8 kB) bytes, json round-trip, contract keys,
`roofline` and `cpu_baseline` present -- for the one-GPU shape and the N-rank shape (round-5 verdict: the 20 kB line of
that round came back from the driver with parsed = null)."""
import copy
import json
import os

import pytest

from poly_amd import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def _load(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden", name)))


def _inflate(x, times):
    """every prose string `times` as long, every list of rank records 8 long: the worst line a future bench could hand over"""
    if isinstance(x, str):
        return x * times
    if isinstance(x, dict):
        return {k: _inflate(v, times) for k, v in x.items()}
    if isinstance(x, list):
        return [_inflate(v, times) for v in x]
    return x


@pytest.mark.parametrize("fixture", ["bench_full_1gpu.json", "bench_full_2rank.json"])
def test_line_fits_and_round_trips(fixture):
    full = _load(fixture)
    assert len(json.dumps(full)) > 3000          # the fixture is a real full record, not a toy
    text = bench_line.render(full)
    assert len(text) + 1 <= bench_line.LIMIT
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["n_gpus"] == full["n_gpus"] and line["steps"] == full["steps"] and line["warmup"] == full["warmup"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    assert "workload" in line["config"] and "model" not in line["config"]
    if full["n_gpus"] == 1:
        cb = line["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in cb
        assert line["secondary"]["value"] == pytest.approx(full["secondary"]["value"], rel=1e-5)
        assert set(line["legs"]) >= {"smith_waterman", "mash_distance", "santalucia_scan", "seqhash", "least_rotation"}
        assert all(len(v) == 3 for k, v in line["legs"].items() if isinstance(v, list))
    else:
        assert line["strong"]["scaling"] == "strong"
        assert "mash_distance_allgather" in line["legs"]
        assert len(line["launch"]["devices"]) == full["n_gpus"]


@pytest.mark.parametrize("fixture", ["bench_full_1gpu.json", "bench_full_2rank.json"])
def test_line_fits_when_every_string_is_three_times_as_long_and_there_are_8_ranks(fixture):
    full = _inflate(_load(fixture), 3)
    full["n_gpus"] = 8
    dev = full["launch"]["devices"][0]
    full["launch"]["devices"] = [dict(dev, rank=r, device=r) for r in range(8)]
    text = bench_line.render(full)
    assert len(text) + 1 <= bench_line.LIMIT
    line = json.loads(text)
    assert "roofline" in line and line["launch"]["devices"] == list(range(8))


def test_nan_and_infinity_never_reach_the_line():
    full = copy.deepcopy(_load("bench_full_1gpu.json"))
    full["roofline"]["traffic"] = float("nan")
    full["extra"]["smith_waterman"]["cell_updates_per_s"] = float("inf")
    full["cpu_baseline"]["value"] = float("-inf")
    text = bench_line.render(full)
    assert "NaN" not in text and "Infinity" not in text
    line = json.loads(text)
    assert line["roofline"]["traffic"] is None and line["cpu_baseline"].get("value") is None
    assert "NaN" not in bench_line.render_full(full)


def test_a_leg_that_failed_is_reported_not_dropped():
    full = copy.deepcopy(_load("bench_full_1gpu.json"))
    full["extra"]["seqhash"] = {"error": "RuntimeError: " + "x" * 1000}
    line = json.loads(bench_line.render(full))
    assert line["legs"]["seqhash"]["error"].startswith("RuntimeError") and len(line["legs"]["seqhash"]["error"]) <= 120


def test_pathological_input_sheds_the_optional_parts_instead_of_overshooting():
    full = copy.deepcopy(_load("bench_full_1gpu.json"))
    full["extra"] = {f"leg{i}": {"value": float(i), "roofline": {"frac": 0.5}} for i in range(2000)}
    text = bench_line.render(full)
    assert len(text) + 1 <= bench_line.LIMIT
    line = json.loads(text)
    assert "roofline" in line and "cpu_baseline" in line and isinstance(line["legs"], str)


def test_bench_py_prints_through_the_compactor():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "bench_line.render(" in src
    # no other writer of the JSON descriptor
    assert src.count("os.write(json_fd") == 1 and "json.dumps(line)" not in src
