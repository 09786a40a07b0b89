"""bench.py's stdout contract: the LAST line is a compact headline object the driver can parse from its 8 KB tail (the
single 30 KB line of r03 left BENCH_r03.parsed = null).  A canned full result — r03's own line, the largest the bench has produced —
goes through the printer."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r03", "bench.json")
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                 "roofline", "cpu_baseline")


def canned():
    r = json.load(open(CANNED))
    r["roofline_second"] = dict(r["roofline"], kernel="ra_analysis_kernel")
    r["roofline"]["issue_frac"] = 0.61
    return r


def test_last_line_is_compact_and_complete(tmp_path):
    full = canned()
    assert len(json.dumps(full)) > 20000          # the input really is the oversized line
    out = io.StringIO()
    line = bench.emit(full, str(tmp_path / "bench_details.json"), out=out)
    lines = out.getvalue().splitlines()
    assert lines[-1] == line and len(lines) == 2
    assert len(line.encode()) < bench.COMPACT_LIMIT == 4096
    c = json.loads(line)
    for k in HEADLINE_KEYS:
        assert k in c, k
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"]   # every digit: the driver checks one against the other
    assert c["config"]["workload"] == full["config"]["workload"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_launch", "issue_frac"):
        assert k in c["roofline"], k
    assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-4 * c["roofline"]["frac"] + 1e-12
    for k in ("value", "unit", "cores", "kind", "sample", "gpu_vs_oracle_max_rms"):
        assert k in c["cpu_baseline"], k
    assert c["cpu_baseline"]["kind"] in ("port", "reference")
    assert c["north_star"]["ms_per_step"] > 0 and c["north_star"]["x_realtime"] > 10
    assert set(c["other_ms_per_step"]) >= {"north_star", "stereo", "voc_fft", "voc_bank", "configs3", "voice_sweep", "voice_birds", "voice_robots"}
    # the earlier line and the file hold everything
    assert json.loads(lines[0])["bench_details"]["other_configs"] == full["other_configs"]
    assert json.load(open(tmp_path / "bench_details.json"))["kernels"] == full["kernels"]


def test_compact_line_survives_bloat_and_missing_blocks():
    full = canned()
    full["other_configs"] = [dict(o, workload=o.get("workload", "x") + "_" + "y" * 200 + str(i)) for i, o in enumerate(full["other_configs"] * 4) if "ms_per_step" in o]
    line = bench.compact_line(full, "bench_details.json")
    assert len(line.encode()) < bench.COMPACT_LIMIT
    c = json.loads(line)
    for k in HEADLINE_KEYS:
        assert k in c, k
    # a multi-rank line has no cpu_baseline / other_configs: still a valid headline
    for k in ("cpu_baseline", "other_configs", "roofline_second"):
        full.pop(k, None)
    c = json.loads(bench.compact_line(full))
    assert "cpu_baseline" not in c and c["roofline"]["kernel"] and c["value"] > 0
