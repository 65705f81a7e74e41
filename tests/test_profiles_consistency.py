"""The committed evidence must reproduce itself: `roofline.frac` of the committed bench lines from the committed rocprofv3 kernel
summaries (scripts/roofline_from_rocprof.py, what a reviewer runs), and the PMC traffic the bench line quotes from the committed
PMC summary."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


@pytest.mark.parametrize("csv,line", [("r05_bench_kernel_stats.csv", "r05_bench.json"),
                                      ("r05_bench_fp16_kernel_stats.csv", "r05_bench_fp16.json"),
                                      ("r05_config3_kernel_stats.csv", "r05_bench_config3.json")])
def test_roofline_reproduces_from_rocprof_summary(csv, line):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "roofline_from_rocprof.py"), os.path.join(PROF, csv),
                        os.path.join(PROF, line), "--tol", "0.05"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.strip().endswith("OK"), r.stdout[-500:]


def test_bench_line_is_internally_consistent():
    d = json.load(open(os.path.join(PROF, "r05_bench.json")))
    r = d["roofline"]
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert abs(r["bytes_per_launch"] / (r["avg_us_per_launch"] * 1e-6) / 1e9 - r["achieved"]) < 0.01 * r["achieved"]
    assert r["traffic"] is not None and 0.95 < r["traffic"] / r["bytes_per_launch"] < 1.10, "PMC traffic ~ algorithmic bytes"
    assert abs(d["value"] - d["config"]["tokens_per_sample"] * d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    ws = r["whole_step"]
    assert ws["frac"] <= 1.0 and abs(ws["achieved_GBps"] / 8000.0 - ws["frac"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and "full_run" in cb


def test_fast_mode_block_carries_its_own_evidence():
    """the fp16 fast mode has its own per-kind table, context sweep, attention fit and PMC traffic in the bench line."""
    d = json.load(open(os.path.join(PROF, "r05_bench.json")))
    f = d["fast_mode_fp16"]
    assert set(f["kernels"]) >= {"qkv_gemv", "attn_decode", "out_proj_gemv", "fc1_gemv", "fc2_gemv"}
    assert len(f["context_sweep"]["samples"]) == 16 and f["context_sweep"]["fit"]["intercept_us"] > 0
    a = f["attention"]
    assert "_Float16" in a["kernel_name"] and 0.95 < a["traffic"] / a["bytes_per_launch"] < 1.10
    assert abs(a["bytes_per_launch"] / (a["avg_us_per_launch"] * 1e-6) / 8e12 - a["frac"]) < 1e-3
    assert abs(sum(f["kernels"][k]["avg_us"] for k in ("qkv_gemv", "attn_decode", "attn_combine", "out_proj_gemv", "fc1_gemv", "fc2_gemv"))
               - f["per_layer_kernel_sum_us"]) < 0.05
