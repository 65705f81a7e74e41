"""The committed evidence must reproduce itself: `roofline.frac` of the committed bench lines from the committed rocprofv3 kernel
summaries (scripts/roofline_from_rocprof.py, what a reviewer runs), the PMC traffic the bench line quotes from the committed
PMC summary, the extra objects the driver's line carries (full-size configs[3] steps in both modes, detokenise cost), and the parity
figures DESIGN.md quotes from the committed GPU-suite log."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
HBM = 8000.0


@pytest.mark.parametrize("csv,line", [("r06_bench_kernel_stats.csv", "r06_bench.json"),
                                      ("r06_bench_fp16_kernel_stats.csv", "r06_bench_fp16.json"),
                                      ("r06_config3_kernel_stats.csv", "r06_bench_config3.json")])
def test_roofline_reproduces_from_rocprof_summary(csv, line):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "roofline_from_rocprof.py"), os.path.join(PROF, csv),
                        os.path.join(PROF, line), "--tol", "0.05"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.strip().endswith("OK"), r.stdout[-500:]


def test_bench_line_is_internally_consistent():
    d = json.load(open(os.path.join(PROF, "r06_bench.json")))
    r = d["roofline"]
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert abs(r["bytes_per_launch"] / (r["avg_us_per_launch"] * 1e-6) / 1e9 - r["achieved"]) < 0.01 * r["achieved"]
    assert r["traffic"] is not None and 0.95 < r["traffic"] / r["bytes_per_launch"] < 1.10, "PMC traffic ~ algorithmic bytes"
    assert "r06_pmc_hbm_summary.json" in r["traffic_source"], "the line quotes this round's PMC passes"
    assert abs(d["value"] - d["config"]["tokens_per_sample"] * d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    ws = r["whole_step"]
    assert ws["frac"] <= 1.0 and abs(ws["achieved_GBps"] / HBM - ws["frac"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and "full_run" in cb


def test_timed_step_is_the_whole_generate():
    """the timed step detokenises (a real meto engine, clean=True); the line says so and prices it."""
    d = json.load(open(os.path.join(PROF, "r06_bench.json")))
    assert "detokenise" in d["config"]["workload"] and "clean=True" in d["config"]["workload"]
    dt = d["detokenise_ms"]
    assert 0 < dt["per_sample_clean_false"] < dt["per_sample_clean_true"] < 0.01 * d["ms_per_step"], "a host scan of 4000 ids: far below 1 % of the step"


def test_fast_mode_block_carries_its_own_evidence():
    """the fp16 fast mode has its own per-kind table, context sweep, attention fit and PMC traffic in the bench line."""
    d = json.load(open(os.path.join(PROF, "r06_bench.json")))
    f = d["fast_mode_fp16"]
    assert set(f["kernels"]) >= {"qkv_gemv", "attn_decode", "out_proj_gemv", "fc1_gemv", "fc2_gemv"}
    assert len(f["context_sweep"]["samples"]) == 16 and f["context_sweep"]["fit"]["intercept_us"] > 0
    a = f["attention"]
    assert "_Float16" in a["kernel_name"] and 0.95 < a["traffic"] / a["bytes_per_launch"] < 1.10
    assert abs(a["bytes_per_launch"] / (a["avg_us_per_launch"] * 1e-6) / 8e12 - a["frac"]) < 1e-3
    assert abs(sum(f["kernels"][k]["avg_us"] for k in ("qkv_gemv", "attn_decode", "attn_combine", "out_proj_gemv", "fc1_gemv", "fc2_gemv"))
               - f["per_layer_kernel_sum_us"]) < 0.05


@pytest.mark.parametrize("key,precision,esz", [("config3_shard_fp16", "fp16", 2), ("config3_shard_exact_fp32", "fp32", 4)])
def test_driver_line_carries_full_size_config3_steps(key, precision, esz):
    """ONE full-size configs[3] step (B = 32, T = 4000) per mode under the driver's clock, each with a roofline block
    that follows from its own numbers and with this round's PMC traffic."""
    d = json.load(open(os.path.join(PROF, "r06_bench.json")))
    c = d[key]
    assert "error" not in c and c["steps"] == 1
    assert abs(c["value"] - 32 * 4000 / (c["ms_per_step"] * 1e-3)) < 0.01 * c["value"]
    assert c["decode_only_tokens_per_s"] >= c["value"]
    r = c["roofline"]
    assert r["kernel"] == "attn_decode" and "attn_stream_kernel" in r["kernel_name"] and ("_Float16" in r["kernel_name"]) == (precision == "fp16")
    assert abs(r["bytes_per_launch"] - 32 * 2 * r["context_len_at_measurement"] * 1536 * esz) < 1e-6 * r["bytes_per_launch"]
    assert abs(r["bytes_per_launch"] / (r["avg_us_per_launch"] * 1e-6) / 1e9 / HBM - r["frac"]) < 2e-3
    assert 0.5 < r["whole_step"]["frac"] < r["frac"] < 1.0
    assert r["traffic"] is not None and 0.95 < r["traffic"] / r["bytes_per_launch"] < 1.10 and "r06_pmc_hbm_config3" in r["traffic_source"]
    if precision == "fp16":
        g = c["greedy_ids_vs_exact_fp32"]
        fd = g["first_divergence_index_per_row"]
        assert g["rows"] == 32 and 0 <= fd["min"] <= fd["median"] <= fd["max"] <= 4000


def test_design_quotes_the_committed_parity_values():
    """the tolerances DESIGN.md quotes are the ones the committed GPU-suite log holds (scripts/parity_table.py
    regenerates the table of section 5 from profiles/r06_parity_values.log; DESIGN.md must contain it verbatim)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import parity_table
    table = parity_table.table(os.path.join(PROF, "r06_parity_values.log"))
    assert "NOT IN THE LOG" not in table, table
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert table in design, "DESIGN.md section 5 is out of date: paste the output of scripts/parity_table.py"
    log = open(os.path.join(PROF, "r06_parity_values.log")).read()
    assert " failed" not in log.split("\n")[-2] and "passed" in log
