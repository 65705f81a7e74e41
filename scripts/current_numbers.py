#!/usr/bin/env python3
"""The "Current numbers" table of DESIGN.md section 6, generated from the committed bench lines under profiles/ (round 6).
usage: current_numbers.py   -> markdown on stdout"""
import json
import os

PROF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def J(name):
    p = os.path.join(PROF, name)
    return json.load(open(p)) if os.path.exists(p) else None


def main():
    b, b16, b3, b3x, b2 = J("r06_bench.json"), J("r06_bench_fp16.json"), J("r06_bench_config3.json"), J("r06_bench_config3_exact.json"), J("r06_bench_config2.json")
    rows = ["| line | value | dominant kernel vs 8 TB/s | whole step vs 8 TB/s | evidence |", "|---|---|---|---|---|"]
    r = b["roofline"]
    rows.append(f"| **configs[1]** exact fp32, B = 1, greedy, T = 4000, whole `generate()` incl. detokenise (`bench.py`) | **{b['value']:.0f} tok/s** (decode only {b['decode_only_tokens_per_s']:.0f}); ids bit-exact vs the reference CPU run; detokenise {b['detokenise_ms']['per_sample_clean_true']:.1f} ms per sample inside the step | `{r['kernel_name']}` {r['bytes_per_launch'] / 1e6:.2f} MB / {r['avg_us_per_launch']:.2f} us = **{r['frac']:.3f}**; PMC traffic {r['traffic'] / 1e6:.2f} MB = {r['traffic'] / r['bytes_per_launch']:.3f}x | **{r['whole_step']['frac']:.3f}** | `r06_bench.json`, `r06_bench_kernel_stats.csv`, `r06_roofline_check_bench.log`, `r06_pmc_hbm_summary.json` |")
    f = b["fast_mode_fp16"]
    a = f["attention"]
    rows.append(f"| fast mode (fp16 storage), B = 1, T = 4000 (`fast_mode_fp16` in the line; own line: `bench.py --precision fp16`{', ' + format(b16['value'], '.0f') + ' tok/s whole step' if b16 else ''}) | **{f['decode_only_tokens_per_s']:.0f} tok/s** decode | `{a['kernel_name']}` {a['bytes_per_launch'] / 1e6:.2f} MB / {a['avg_us_per_launch']:.2f} us = **{a['frac']:.3f}**; fit {f['context_sweep']['fit']['intercept_us']:.2f} us + bytes / {f['context_sweep']['fit']['slope_TBps']:.2f} TB/s; PMC {a['traffic'] / a['bytes_per_launch']:.3f}x | **{f['hbm_frac']:.3f}** | `r06_bench.json`, `r06_bench_fp16.json`, `r06_bench_fp16_kernel_stats.csv`, `r06_pmc_hbm_fp16_summary.json` |")
    for key, own, tag in (("config3_shard_fp16", b3, "fp16 storage"), ("config3_shard_exact_fp32", b3x, "exact fp32 (ids bit-exact vs the CPU reference)")):
        c = b[key]
        r = c["roofline"]
        ownt = f"; own line `bench.py --config 3{' --precision fp32' if 'exact' in key else ''}`: {own['value']:.0f} tok/s" if own else ""
        tr = f"; PMC {r['traffic'] / r['bytes_per_launch']:.3f}x" if r.get("traffic") else ""
        rows.append(f"| **configs[3] shard** B = 32 greedy, T = 4000, {tag} (`{key}` in the driver's line: ONE full-size step{ownt}) | **{c['value']:.0f} tok/s** whole step incl. detokenise of 32 meshes, decode only {c['decode_only_tokens_per_s']:.0f} | `{r['kernel_name']}` {r['bytes_per_launch'] / 1e6:.0f} MB / {r['avg_us_per_launch']:.1f} us = **{r['frac']:.3f}**{tr} | **{r['whole_step']['frac']:.3f}** | `r06_bench.json`{', `r06_config3_kernel_stats.csv`, `r06_pmc_hbm_config3_summary.json`' if 'fp16' in key else ', `r06_pmc_hbm_config3_exact_summary.json`'} |")
    if b2:
        r = b2["roofline"]
        rows.append(f"| **configs[2]** full size, B = 32 sample (top-k 10), T = 16000, context to 18050 (`bench.py --config 2`) | **{b2['value']:.0f} tok/s** | streaming attention {r['bytes_per_launch'] / 1e9:.3f} GB / {r['avg_us_per_launch']:.0f} us = **{r['frac']:.3f}** | **{r['whole_step']['frac']:.3f}** | `r06_bench_config2.json` |")
    d = b["dit_front_end_fp16"]
    rows.append(f"| **configs[4]** DiT front-end, fp16 matrix cores, 20 guided DDIM steps (`dit_front_end_fp16`) | **{d['ms_per_cfg_forward']:.2f} ms** per guided forward = {d['roofline']['achieved']:.0f} TFLOP/s = **{d['roofline']['frac']:.3f}** of 2.5 PFLOP/s (same-box A/Bs this round: 8.47 -> 8.39, 8.17 -> 7.97, 7.70 -> 7.64, 7.60 -> 7.46 ms) | MFMA-busy: `gemm_hh256_kernel` 0.34, `gemm_hh_stream_kernel` 0.29, `gemm_hh_mfma_kernel` 0.12-0.23, `flash_attn_hh_kernel` 0.22 (`r06_pmc_sq_dit.json`) | | `r06_bench.json`, `r06_dit_fp16_kernel_stats.csv`, `r06_dit_trace_by_shape.json`, `r06_dit_ab_final.log`, `r06_dit_stream_adopted.log`, `r06_dit_time_embed_ab.log`, `r06_dit_fast_erf_in_situ.log` |")
    cb = b["cpu_baseline"]
    rows.append(f"| `cpu_baseline` (oracle = reference modules bit for bit, this box's host cores) | {cb['value']:.2f} tok/s on {cb['cores']} threads (bounded sample) | | | `r06_bench.json` |")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
