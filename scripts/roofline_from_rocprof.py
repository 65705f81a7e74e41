#!/usr/bin/env python3
"""Recompute bench.py's `roofline.frac` from a rocprofv3 `--kernel-trace --stats` summary of the SAME command and check
that the two agree (the bench line said 0.63, the committed profile implied 0.54).

    python scripts/roofline_from_rocprof.py profiles/r02_bench_kernel_stats.csv profiles/r02_bench.json [--tol 0.05]

bench.py quotes the dominant decode kernel at the MEAN context length of the timed run: algorithmic bytes of one launch at
that length / the launch duration at that length (HIP events, hipGraph replay).  Every decode kernel's duration is affine
in the context length and the context grows linearly over the run, so the rocprofv3 AVERAGE duration of that kernel over
the whole run is the duration at the mean length - the two must agree.  Also prints the per-layer kernel sum and every
kind's achieved GB/s from the rocprof averages.
"""
import argparse
import csv
import json
import sys

HBM_PEAK = 8.0e12


def load_stats(path):
    rows = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows[r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"])}
    return rows


def find(stats, needle):
    hits = [(n, v) for n, v in stats.items() if needle in n]
    if not hits:
        # rocprofv3 leaves kernels with _Float16 template arguments MANGLED (_ZN2er18attn_stream_kernelIDF16_Li96ELi2EEE...):
        # match the base name and the element type instead
        base = needle.split("<")[0]
        half = "_Float16" in needle
        hits = [(n, v) for n, v in stats.items() if n.startswith("_Z") and base in n and (("DF16_" in n) == half)]
    if not hits:
        return None, None
    return max(hits, key=lambda kv: kv[1]["calls"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats_csv")
    ap.add_argument("bench_json", help="file holding the bench.py JSON line of the same command (last line starting with '{')")
    ap.add_argument("--tol", type=float, default=0.05)
    a = ap.parse_args()
    stats = load_stats(a.stats_csv)
    line = [ln for ln in open(a.bench_json).read().splitlines() if ln.startswith("{")][-1]
    bench = json.loads(line)
    rf = bench["roofline"]
    name, row = find(stats, rf["kernel_name"])
    if row is None:
        raise SystemExit(f"kernel {rf['kernel_name']!r} not in {a.stats_csv}")
    frac_prof = rf["bytes_per_launch"] / (row["avg_ns"] * 1e-9) / HBM_PEAK
    print(f"dominant kernel  : {name}")
    print(f"  rocprofv3 avg  : {row['avg_ns'] / 1e3:.3f} us over {row['calls']} launches")
    print(f"  bench.py sweep : {rf['avg_us_per_launch']:.3f} us (mean over the sampled contexts of the run; mean context {rf['context_len_at_measurement']})")
    print(f"  bytes/launch   : {rf['bytes_per_launch'] / 1e6:.2f} MB (algorithmic, mean context length)")
    print(f"  frac (rocprof) : {frac_prof:.4f}    frac (bench line): {rf['frac']:.4f}")
    for kind, k in rf.get("kernels", {}).items():
        print(f"  bench sweep {kind:14s} {k['avg_us']:8.3f} us  {k['GBps']:8.1f} GB/s")
    print(f"  bench per-layer kernel sum: {rf.get('per_layer_kernel_sum_us')} us")
    top = sorted(((v["total_ns"], n) for n, v in stats.items()), reverse=True)[:8]
    tot = sum(v["total_ns"] for v in stats.values())
    print("top kernels by total time (rocprofv3):")
    for t, n in top:
        print(f"  {t / tot * 100:5.1f} %  {stats[n]['avg_ns'] / 1e3:9.3f} us x {stats[n]['calls']:7d}  {n[:110]}")
    rel = abs(frac_prof - rf["frac"]) / rf["frac"]
    print(f"relative difference {rel * 100:.2f} % (tolerance {a.tol * 100:.0f} %)")
    if rel > a.tol:
        raise SystemExit("MISMATCH: bench.py's roofline.frac does not follow from the rocprofv3 summary")
    print("OK")


if __name__ == "__main__":
    main()
