#!/usr/bin/env python3
"""The parity figures DESIGN.md section 5 quotes, as a markdown table generated from the committed GPU-suite log
(profiles/r06_parity_values.log: `pytest tests -m gpu -s -rA`, the value lines the tests print).  tests/test_profiles_consistency.py
requires DESIGN.md to contain this table verbatim, so a quoted tolerance cannot drift from what the suite measured.
usage: parity_table.py [log]   -> markdown on stdout
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [  # (label, regex with one or more groups, format of the groups)
    ("configs[1] exact fp32, T = 4000, single row: greedy ids vs the reference run", r"decode: [\d.]+ ms for 4000 tokens.*\n.*teacher-forced max\|dlogit\| over (\d+) recorded steps: ([\d.e+-]+)",
     "4000 / 4000 equal (test asserts); teacher-forced max abs dlogit over {0} recorded steps {1}"),
    ("2-layer golden, 96 teacher-forced steps, default knobs and the knob matrix", r"\{'ER_DECODE_V': '2'\}: teacher-forced max\|dlogit\| over 96 steps: ([\d.e+-]+)", "max abs dlogit {0} (ER_DECODE_V=2)"),
    ("long context (14050.. keys), 24 layers, single row / B = 18 / B = 6", r"24 layers, single row, context 14050\.\.: teacher-forced max\|dlogit\| ([\d.e+-]+)", "max abs dlogit {0}"),
    ("fp16 storage, 24 layers, B = 16, context 18050.. vs the fp16-storage emulation", r"24 layers, fp16 storage, B = 16, context 18050\.\.: teacher-forced max\|dlogit\| vs the fp16 emulation ([\d.e+-]+)", "max abs dlogit {0}"),
    ("fast mode (fp16 storage), single row", r"fast mode: max\|dlogit\| vs emulation ([\d.e+-]+); vs fp32 reference \(first (\d+) steps\) ([\d.e+-]+); (ids [^\n]*)",
     "vs emulation {0}; vs the fp32 reference over the first {1} steps {2}; {3}"),
    ("batched (B = 32) exact fp32, teacher-forced rows 0 / 13 / 31", r"batched fp32: max\|dlogit\| per row \[([^\]]+)\] over (\d+) steps", "max abs dlogit per row [{0}] over {1} steps"),
    ("batched (B = 32) fp16 storage", r"batched fp16: max\|dlogit\| per row \[([^\]]+)\] over (\d+) steps", "max abs dlogit per row [{0}] over {1} steps"),
    ("sample mode (configs[2] shape): top-10 sets and probabilities vs the oracle", r"configs\[2\]-shaped: worst \|dp\| over top-10 sets ([\d.e+-]+); near-tie set swaps (\d+); (\d+) device draws checked",
     "worst abs dp {0}; near-tie set swaps {1}; {2} device draws equal to the inverse-CDF draw"),
    ("exact prefill: tail rows through the GEMV kernels / key-range split", r"tail through GEMV: max\|dlogit\| vs golden ([\d.e+-]+), vs the all-GEMM prefill ([\d.e+-]+)", "vs golden {0}, vs the all-GEMM prefill {1}"),
    ("configs[3] shard, exact fp32, B = 32, T = 4000 (row 0 bit-exact vs the reference ids)", r"B=32 x T=4000: decode (\d+) ms -> (\d+) tok/s aggregate", "decode {0} ms = {1} tok/s aggregate in the test"),
    ("DiT forward (2 layers, fp32) vs the reference DiT module", r"DiT forward: max abs err on sampled rows ([\d.e+-]+)", "max abs err {0}"),
    ("DiT 6-step CFG / DDIM latents (fp32)", r"6-step CFG/DDIM latents: max abs err ([\d.e+-]+)", "max abs err {0}"),
    ("image -> cond / image -> latents (4 steps)", r"image -> cond: max abs err vs oracle ([\d.e+-]+)\s*\n\s*image -> latents \(4 steps\): max abs err ([\d.e+-]+)", "{0} / {1}"),
    ("fp16 matrix-core front-end vs the fp16 emulation", r"fp16 MFMA front-end: cond err vs emulation ([\d.e+-]+), latents err ([\d.e+-]+); latents vs fp32 path ([\d.e+-]+)",
     "cond {0}, latents {1}; latents vs the fp32 path {2}"),
    ("full depth (32 CLIP + 24 DiT layers, 3 guided steps) vs the reference-module golden", r"full depth \(32 CLIP \+ 24 DiT layers, 3 steps\): cond max abs err ([\d.e+-]+) .*latents max abs err ([\d.e+-]+)", "cond {0}, latents {1}"),
    ("GPU suite", r"=+ (\d+) passed, (\d+) deselected", "{0} passed ({1} CPU tests deselected)"),
]


def table(log_path):
    text = open(log_path).read()
    out = ["| check (tests/test_gpu_*.py) | measured, `profiles/" + os.path.basename(log_path) + "` |", "|---|---|"]
    for label, rx, fmt in ROWS:
        m = re.search(rx, text)
        out.append(f"| {label} | {fmt.format(*[' '.join(g.split()) for g in m.groups()]) if m else 'NOT IN THE LOG'} |")
    return "\n".join(out)


if __name__ == "__main__":
    print(table(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_parity_values.log")))
