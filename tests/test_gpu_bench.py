"""The driver's contract with bench.py (one JSON line; metric / value / roofline / cpu_baseline objects), checked on
the small configs so that the suite stays short: config 1 (the reference's CPU-sized case) and config 2 (1M x 1M)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def run_bench(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *flags],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "bench.py prints exactly one JSON line"
    return json.loads(lines[0])


@pytest.mark.parametrize("config, flags", [(1, ["--steps", "20", "--warmup", "3"]),
                                           (2, ["--steps", "10", "--warmup", "3", "--cpu-seconds", "1", "--no-ceiling"])])
def test_bench_line_contract(config, flags):
    d = run_bench("--config", str(config), *flags)
    assert REQUIRED <= set(d), sorted(REQUIRED - set(d))
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f64" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes of one launch / its duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-2 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    assert c.get("gpu_agrees_with_cpu") is True
    if config == 2:
        assert c["gpu_equals_cpu_bit_for_bit"] is True          # every row of the stream path: the reference's order
        assert d["setup_s"]["autotune"] >= 0 and "place_vectors" in d["setup_s"]


def test_config5_record_carries_the_product_on_the_result():
    """config 5's second half ("then SpMV"): the assembled 5M x 5M matrix has its columns anywhere, so the product runs
    on the column-blocked kernel (entry-parallel form), and the record says so"""
    d = run_bench("--config", "5", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert d["unit"] == "Mentries/s" and d["value"] > 0 and d["roofline"]["bound"] == "hbm"
    s = d["spmv_on_result"]
    assert s["kernel"] == "csr_spmv_cblock" and s["plan"]["cblock_form"] == "entry", s
    assert s["ms"] > 0 and abs(s["roofline_frac"] - s["algorithmic_bytes_per_launch"] / (s["ms"] * 1e-3) / 8e12) < 1e-3
    assert d["config"]["plan"]["kernel"] == "cblock", d["config"]["plan"]     # (built by the first product on the assembled handle)


def test_config4_record_is_the_scatter_path_over_row_tiles():
    """config 4 names the atomic scatter path: `value` is that path (over row tiles on this band), the transposed route is
    reported beside it and agrees; the launches rotate over handles that own their arrays"""
    d = run_bench("--config", "4", "--steps", "50", "--warmup", "5", "--cpu-seconds", "1")
    assert d["roofline"]["kernel"].startswith("csc_spmv_rowtiles") and d["config"]["plan"]["row_tiles"] == 1, d["roofline"]
    assert d["transposed_route"]["agrees_with_scatter"] is True and d["cpu_baseline"]["gpu_agrees_with_cpu"] is True
    assert "3 independent copies" in d["config"]["workload"]
