"""The driver keeps only a tail of bench.py's stdout: the ONE result line has to stay small and parse (a 21.6 KB line left
BENCH_r05.json with parsed = null).  CPU-only: the line builder is run on a committed FULL-SIZE result dict."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def _full_dict():
    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:      # 21.6 KB: the line that did not parse
        return json.load(f)


def test_result_line_is_small_and_complete():
    import bench
    full = _full_dict()
    assert len(json.dumps(full)) > 20_000
    line = bench.compact_result(full)
    assert len(line) < 8192 and "\n" not in line
    o = json.loads(line)
    for k in CONTRACT:
        assert k in o, k
    assert o["value"] > 0 and o["ms_per_step"] > 0 and o["config"]["workload"].startswith("synthetic all-pairs DGG N=100000")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_bytes", "rocprof_kernel"):
        assert k in o["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in o["cpu_baseline"], k
    assert abs(o["roofline"]["frac"] - o["roofline"]["achieved"] / o["roofline"]["peak"]) < 1e-6
    assert set(o["configs"]) == set(full["configs"])
    for c in o["configs"].values():
        assert set(c) == {"ms_per_step", "value", "frac"}


def test_result_line_survives_a_bloated_dict(tmp_path, monkeypatch, capfd):
    """whatever a later round adds to the detail dict, the stdout line stays under the limit; the detail goes to the side file + stderr"""
    import bench
    full = _full_dict()
    full["variants"] = {f"v{i}": {"note": "x" * 500, "numbers": list(range(100))} for i in range(200)}
    full["config"]["workload"] = full["config"]["workload"] + " " + "y" * 5000
    for i in range(300):
        full["configs"][f"extra{i}"] = {"ms_per_step": 1.0 / 3, "value": 1e9 / 7, "roofline": {"frac": 0.1 / 3}, "junk": "z" * 200}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "_JSON_FD", None)
    bench.emit_json(full)
    cap = capfd.readouterr()
    lines = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 8192
    o = json.loads(lines[0])
    assert "roofline" in o and "cpu_baseline" in o and o["ms_per_step"] == float(f"{full['ms_per_step']:.6g}")
    assert cap.err.startswith("BENCH_DETAIL ")
    with open(tmp_path / "gpurun_out" / "bench_detail.json") as f:
        assert json.load(f)["variants"]["v7"]["numbers"][-1] == 99
