#!/usr/bin/env python3
"""Per-(kernel, grid) launch statistics from a rocprofv3 --kernel-trace CSV (``*_kernel_trace.csv``).

``--stats`` averages a kernel over every shape it ran on; the DiT front-end calls one GEMM template on four shapes
per layer, so the per-shape durations are grouped here by (kernel name, grid size, workgroup size).
usage: trace_by_shape.py <kernel_trace.csv> [name filter ...]   -> JSON on stdout
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    path, filters = sys.argv[1], sys.argv[2:]
    groups = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Name") or ""
            if filters and not any(t in name for t in filters):
                continue
            try:
                dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            except (KeyError, ValueError):
                continue
            grid = "x".join(str(row.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            wg = str(row.get("Workgroup_Size_X", ""))
            groups[(name.split("(")[0][:90], grid, wg)].append(dur)
    out = []
    for (name, grid, wg), d in groups.items():
        d.sort()
        out.append({"kernel": name, "grid_threads": grid, "workgroup": wg, "launches": len(d), "mean_us": round(sum(d) / len(d) / 1e3, 2),
                    "median_us": round(d[len(d) // 2] / 1e3, 2), "min_us": round(d[0] / 1e3, 2), "total_ms": round(sum(d) / 1e6, 3)})
    out.sort(key=lambda r: -r["total_ms"])
    tot = sum(r["total_ms"] for r in out)
    print(json.dumps({"total_ms": round(tot, 3), "groups": out}, indent=1))


if __name__ == "__main__":
    main()
