#!/usr/bin/env python3
"""First exploration run of round 3 (one process, one synthetic checkpoint; GPU box only).  Sections print JSON lines.

  fast16   : fp16 single-row decode, rows-per-wave / waves-per-workgroup of the GEMVs (the defaults were tuned in fp32, where a
             row is 6 KB; in fp16 it is 3 KB, so two rows per wave carry the same bytes in flight)
  prefill  : encode + prefill per sample in fast mode with / without the staged split-fp16 prefix attention
             (ER_PREFILL_ATTN_F16S=1; unit-tested, never timed) at B = 1 / 8 / 32
  midbatch : streaming vs split batch attention at B = 8 / 12 (below the auto rule's edge of 16 rows)

Usage: python scripts/explore_r3.py [section ...]     (scripts/gpurun_call.sh 300 'python scripts/explore_r3.py fast16')"""
import dataclasses
import json
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from edgerunner_amd import weights as W  # noqa: E402
from edgerunner_amd.models import LMM  # noqa: E402
from edgerunner_amd.options import config_defaults  # noqa: E402
import attn_sweep  # noqa: E402
import tune_decode  # noqa: E402

SECTIONS = sys.argv[1:] or ["fast16", "prefill", "midbatch"]


def prefill_ms(opt, sd, precision, Bs, env):
    for k in ("ER_PREFILL_ATTN_F16S",):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    m = LMM(opt, "cuda:0", precision=precision)
    m.load_state_dict(sd, strict=True)
    out = {}
    for B in Bs:
        pcs = torch.cat([W.synthetic_point_cloud(i, 4096) for i in range(B)]).to("cuda:0")
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            _, toks = m.generate(pcs, 1000, tokenizer=object(), max_new_tokens=4, min_new_tokens=4)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) * 1e3)
        out[B] = {"ms_per_sample": round(best / B, 2), "first_ids": [int(x) for x in toks[0][:4]]}
    m.mesh_decoder.close()
    for k in env:
        os.environ.pop(k, None)
    return out


def main():
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=24, generate_mode="greedy")
    t0 = time.time()
    sd = W.make_state_dict(opt, 0, "perturbed")
    print(f"weights in {time.time() - t0:.1f}s", flush=True)
    for sec in SECTIONS:
        print(f"=== {sec} (+{time.time() - t0:.0f}s)", flush=True)
        try:
            if sec == "fast16":
                cfgs = [{}, {"ER_RW_QKV": 2}, {"ER_RW_QKV": 2, "ER_NW_QKV": 4}, {"ER_RW_FC1": 4}, {"ER_RW_FC2": 4},
                        {"ER_RW_FC1": 4, "ER_RW_FC2": 4, "ER_RW_QKV": 2}, {"ER_RW_OUT": 2}, {"ER_NW_OUT": 4}]
                tune_decode.run(opt, sd, cfgs, "fp16", 1000)
            elif sec == "prefill":
                for env in ({}, {"ER_PREFILL_ATTN_F16S": 1}):
                    print(json.dumps({"cfg": env, "fast_prefill": prefill_ms(opt, sd, "fp16", [1, 8, 32], env)}), flush=True)
            elif sec == "midbatch":
                for B in (8, 12):
                    attn_sweep.run(opt, sd, [{"ER_ATTN_V_BATCHED": 1}, {"ER_ATTN_V_BATCHED": 3}], "fp16", B, 4000, [2176, 4176, 5926])
        except Exception:
            traceback.print_exc()
    print(f"=== done (+{time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
