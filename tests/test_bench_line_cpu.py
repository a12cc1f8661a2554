"""bench.py's driver-facing line (row (d) of SURVEY 8): compact, parseable, complete.

Round 4's run printed one 29 KB JSON line and the driver's record came back with `parsed: null`.  The line is now built
by bench.compact_line() from the complete record, which goes to a side file; these tests run that builder on a canned
complete record (profiles/r04_bench.json: a real run of the previous round) and on an adversarially long one."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "secondary")
ROOFLINE = ("kernel", "shape", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
            "avg_launch_ms", "backward_unit_frac")


@pytest.fixture(scope="module")
def canned():
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as f:
        return json.load(f)


def test_line_is_compact_and_complete(canned):
    assert len(json.dumps(canned)) > 20000                 # the record that did not parse when printed whole
    line = bench.compact_line(canned, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 4096 and "\n" not in text
    back = json.loads(text)
    for k in REQUIRED:
        assert k in back, k
    for k in ROOFLINE:
        assert k in back["roofline"], k
    assert back["roofline"]["frac"] == pytest.approx(back["roofline"]["achieved"] / back["roofline"]["peak"], rel=2e-3)
    # (the canned round-4 record predates `by_kernel`: the key is present, its value may be None there)
    assert "by_kernel" in back["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    for k in ("workload", "global_batch", "parallelism"):
        assert k in back["config"], k
    assert "model" not in back["config"]
    sec = back["secondary"]
    assert sec["metric"].endswith("ViL-Medium-Deep@384") and sec["value"] > 0 and sec["ms_per_step"] > 0
    assert 0 < sec["roofline"]["frac"] < 1
    # the numbers are the record's own, not re-derived
    assert back["value"] == canned["value"] and back["ms_per_step"] == canned["ms_per_step"]
    assert back["value"] == pytest.approx(back["config"]["global_batch"] * back["steps"] / (back["ms_per_step"] * back["steps"] / 1e3), rel=1e-3)
    # nothing bulky leaked into the line
    for k in ("roofline_by_shape", "kernels"):
        assert k not in back and k not in back["secondary"]


def test_line_stays_under_the_limit_with_hostile_strings(canned):
    big = copy.deepcopy(canned)
    big["config"]["workload"] = "w" * 5000
    big["config"]["precision"] = "p" * 5000
    big["cpu_baseline"]["sample"] = "s" * 5000
    big["tertiary"] = big["tertiary"] * 20
    big["comm"] = {"ranks_in_communicator": 8, "exposed_bytes": 1, "total_bytes": 2, "segments_bytes": [1, 2, 3],
                   "allreduce_alone_ms_per_segment": [0.1] * 200}
    line = bench.compact_line(big, None)
    assert len(json.dumps(line)) <= bench.LINE_LIMIT
    for k in REQUIRED:
        assert k in line


def test_multi_gpu_line_without_cpu_baseline(canned):
    multi = copy.deepcopy(canned)
    multi.pop("cpu_baseline")
    multi.pop("eval")
    multi["n_gpus"] = 8
    line = bench.compact_line(multi, None)
    assert "cpu_baseline" not in line and line["n_gpus"] == 8 and len(json.dumps(line)) < 4096


def test_detail_file_round_trip(tmp_path, canned):
    path = bench.write_detail(canned, str(tmp_path / "sub" / "detail.json"))
    assert path and json.load(open(path)) == canned
    assert bench.write_detail(canned, "/proc/definitely/not/writable.json") is None     # reported, never fatal


def test_roofline_names_the_worst_hot_kernel():
    """the line's `roofline` is the hot kernel furthest from its roof at the stage-1 shape, with all three
    fractions next to it."""
    shapes = {"B128_H3_M32_56x56_W7_G1_m0": {
        "k_mfma_fwd": {"GBps": 1555.0, "frac_hbm": 0.1944, "TFLOPs": 300.0, "frac_mfma": 0.12, "bytes_per_launch": 3.1e8, "avg_ms": 0.2, "launches": 2},
        "k_mfma_bwd_dq": {"GBps": 1221.0, "frac_hbm": 0.1526, "TFLOPs": 250.0, "frac_mfma": 0.1, "bytes_per_launch": 3.9e8, "avg_ms": 0.32, "launches": 2},
        "k_mfma_bwd_dkdv": {"GBps": 1546.0, "frac_hbm": 0.1933, "TFLOPs": 330.0, "frac_mfma": 0.13, "bytes_per_launch": 4.7e8, "avg_ms": 0.3, "launches": 2},
        "backward_unit": {"frac_hbm": 0.11}}}
    tags = {"B128_H3_M32_56x56_W7_G1_m0": (128, 3, 32, 56, 56, 7, 1, 0)}
    orig = bench.hot_shape
    bench.hot_shape = lambda t: "B128_H3_M32_56x56_W7_G1_m0"
    try:
        r = bench.roofline_of("vil_small_224", 128, shapes, tags)
    finally:
        bench.hot_shape = orig
    assert r["kernel"] == "k_mfma_bwd_dq" and r["frac"] == 0.1526
    assert r["by_kernel"] == {"mfma_fwd": 0.1944, "mfma_bwd_dq": 0.1526, "mfma_bwd_dkdv": 0.1933}
    short = bench._short_roofline(r)
    assert short["by_kernel"] == r["by_kernel"] and short["kernel"] == "k_mfma_bwd_dq"


def test_line_never_raises_on_an_oversized_value(canned):
    """an oversized field that the shortening rules do not know must not abort the run after the measurement."""
    big = copy.deepcopy(canned)
    big["secondary"]["metric"] = "m" * 9000
    big["roofline"]["shape"] = "s" * 9000
    line = bench.compact_line(big, "d" * 9000)
    assert len(json.dumps(line)) <= bench.LINE_LIMIT
    for k in bench.CONTRACT_KEYS:
        assert k in line
