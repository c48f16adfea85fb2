"""bench.py's stdout line (round-4 verdict: a 21 KB line gave the driver `parsed: null`).  The headline is
built by bench.headline_line from the detailed record main() assembles; here from a canned record — the figures of
profiles/r04_bench.json reshaped to this round's keys — with no GPU."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned_record():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    d["roofline_syrk"] = d.pop("roofline")
    for k in ("c2", "c4", "loop_closures", "revisits", "all_points_eliminated", "revisits_retained_points_only", "revisits_plain_order", "c5"):
        d.pop(k, None)
    d["config"]["chain_steps"] = 64
    d["config"]["plan"] = "block envelope; 12 retained points; lock-step dissection head 751 | tail 800 | separator 176 cameras and pseudo-cameras; resident panel chain"
    d["config"]["parallelism_short"] = "one GPU"
    d["details_file"] = "gpurun_out/bench_details.json (and stderr: one JSON line per record)"
    return d


def check_contract(text):
    assert "\n" not in text and len(text.encode()) <= bench.HEADLINE_MAX_BYTES
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert "workload" in line["config"] and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert line["roofline"]["bound"] in ("hbm", "mfma")
    return line


def test_headline_is_one_short_json_line_with_roofline_and_cpu_baseline():
    d = canned_record()
    line = check_contract(bench.headline_line(d))
    assert line["value"] == pytest.approx(d["value"], rel=1e-4)
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], rel=1e-3)
    # the phase-level figure, the chain inside it, the SYRK launches as a sub-record
    assert line["roofline"]["chain"]["steps"] == 64
    assert line["roofline"]["chain"]["us_per_step"] == pytest.approx(1e3 * d["roofline_cholesky_phase"]["ms"] / 64, rel=1e-3)
    assert line["roofline"]["syrk"]["frac"] == pytest.approx(d["roofline_syrk"]["frac"], rel=1e-3)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["cpu_baseline"]["kind"] in ("port", "reference")


def test_headline_stays_under_the_cap_whatever_the_records_hold():
    d = canned_record()
    d["cpu_baseline"]["sample"] = "x" * 5000
    d["roofline_syrk"]["traffic_like_for_like"]["file"] = "y" * 3000
    d["independent_solves"] = {"value": 1.0, "unit": "u" * 2000, "scaling": "weak", "note": "n" * 4000}
    check_contract(bench.headline_line(d))


def test_headline_without_a_cpu_baseline_or_side_runs():
    d = canned_record()
    d["cpu_baseline"] = None
    d.pop("roofline_full", None)
    d["chain_model"] = None
    line = check_contract(bench.headline_line(d))
    assert line["cpu_baseline"] is None


def test_phases_are_those_of_the_timed_region():
    before = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    after = [1.1, 2.2, 3.6, 4.0, 5.05, 6.0]
    ph = bench.phases_per_step(before, after, 10)
    assert ph["cholesky"] == pytest.approx(60.0) and ph["jacobian_eval"] == pytest.approx(10.0) and ph["allreduce"] == 0.0
