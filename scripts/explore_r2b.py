#!/usr/bin/env python3
"""Round-2 exploration run (one process, one synthetic checkpoint): decode version 3 / 6-wave qkv A/B at B = 1 (fp32 and
fp16), the attention staircase at B = 1 with both versions, and the XCD-balance of the batched attention grid at B = 32.
GPU box only; every section prints JSON lines (scripts/tune_decode.py and scripts/attn_sweep.py do the work)."""
import dataclasses
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from edgerunner_amd import weights as W  # noqa: E402
from edgerunner_amd.options import config_defaults  # noqa: E402
import attn_sweep  # noqa: E402
import tune_decode  # noqa: E402

SECTIONS = sys.argv[1:] or ["tune32", "tune16", "sweep1", "sweep32"]


def main():
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=24, generate_mode="greedy")
    t0 = time.time()
    sd = W.make_state_dict(opt, 0, "perturbed")
    print(f"weights in {time.time() - t0:.1f}s", flush=True)
    v3 = {"ER_DECODE_V": 3}
    for sec in SECTIONS:
        print(f"=== {sec} (+{time.time() - t0:.0f}s)", flush=True)
        try:
            if sec == "tune32":
                tune_decode.run(opt, sd, [{}, v3, {**v3, "ER_NW_QKV": 6}], "fp32", 1000)
            elif sec == "tune16":
                tune_decode.run(opt, sd, [{}, {**v3, "ER_NW_QKV": 6}], "fp16", 1000)
            elif sec == "sweep1":
                ctx = [2060, 2426, 2926, 3426, 3926, 4050, 4176, 4676, 5176, 5676, 6040]
                attn_sweep.run(opt, sd, [{}, v3], "fp32", 1, 4000, ctx)
            elif sec == "stream32":
                ctx = [2176, 3176, 4176, 5176, 5926]
                attn_sweep.run(opt, sd, [{}, {"ER_ATTN_V_BATCHED": 3}], "fp16", 32, 4000, ctx)
                attn_sweep.run(opt, sd, [{}, {"ER_ATTN_V_BATCHED": 3}], "fp32", 32, 4000, [2176, 4176, 5926])
                attn_sweep.run(opt, sd, [{}, {"ER_ATTN_V_BATCHED": 3}], "fp16", 8, 4000, [2176, 4176, 5926])
            elif sec == "v3only":
                tune_decode.run(opt, sd, [v3], "fp32", 1000)
                attn_sweep.run(opt, sd, [v3], "fp32", 1, 4000, [2060, 2926, 3426, 4050, 4176, 5176, 6040])
            elif sec == "sweep32":
                ctx = [2176, 3176, 3926, 4176, 4426, 5176, 5926]
                attn_sweep.run(opt, sd, [{}, {"ER_ATTN_GRID_HS": 0}], "fp16", 32, 4000, ctx)
        except Exception:
            traceback.print_exc()
    print(f"=== done (+{time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
