#!/usr/bin/env python3
"""Aggregate rocprofv3 CSV output into small JSON summaries that can be committed under profiles/.

  pmc_summary.py stats  <kernel_stats.csv>                 -> per-kernel calls / avg us / % (from --kernel-trace --stats)
  pmc_summary.py pmc [--attn-context L] <counter_collection.csv> [...]   -> per-kernel mean counter values per dispatch

HBM traffic per launch (MI355X_MICROARCH.md, section HBM): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so
read_bytes = 2 * FETCH_SIZE * 1024, write_bytes = WRITE_SIZE * 1024 (write side uncalibrated).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    """'void er::gemv_kernel<float, 1, 1, 2, 1, 1>(er::GemvArgs)' -> 'gemv_kernel<float, 1, 1, 2, 1, 1>'."""
    m = re.search(r"(?:er::)?([A-Za-z_0-9]+(?:<[^()]*>)?)\s*\(", name)
    return m.group(1) if m else re.sub(r"\(.*$", "", name)[:80]


def stats(path):
    rows = list(csv.DictReader(open(path)))
    out = []
    for r in rows:
        out.append({"kernel": short(r.get("Name", "")), "calls": int(r.get("Calls", 0)),
                    "total_ms": round(float(r.get("TotalDurationNs", 0)) / 1e6, 3),
                    "avg_us": round(float(r.get("AverageNs", 0)) / 1e3, 3),
                    "min_us": round(float(r.get("MinNs", 0)) / 1e3, 3), "max_us": round(float(r.get("MaxNs", 0)) / 1e3, 3),
                    "pct": float(r.get("Percentage", 0))})
    print(json.dumps({"source": path, "kernels": out}, indent=1))


def pmc(paths, attn_context=None):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in paths:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    out = {}
    for k, cs in acc.items():
        d = {c: {"mean": v[0] / v[1], "dispatches": v[1]} for c, v in cs.items()}
        if "FETCH_SIZE" in d:
            d["hbm_read_bytes_per_launch"] = 2.0 * d["FETCH_SIZE"]["mean"] * 1024.0     # gfx950 x2 correction
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes_per_launch"] = d["WRITE_SIZE"]["mean"] * 1024.0
        out[k] = d
    doc = {"sources": paths, "kernels": out}
    if attn_context is not None:      # mean number of keys the attention launches of this run covered (bench.py scales to its own)
        doc["attention_context_len_mean"] = attn_context
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        rest = sys.argv[2:]
        ctx = None
        if rest and rest[0] == "--attn-context":
            ctx, rest = float(rest[1]), rest[2:]
        pmc(rest, ctx)
