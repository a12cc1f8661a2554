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
