#!/usr/bin/env python3
"""In-process sweep of the decode tuning knobs (env vars read by er_create).  GPU box only.
Prints decode tok/s (T tokens after a 2050-token prefill) and the per-kernel-kind sweep per config."""
import dataclasses
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgerunner_amd import weights as W  # noqa: E402
from edgerunner_amd.models import LMM  # noqa: E402
from edgerunner_amd.options import config_defaults  # noqa: E402

T = int(os.environ.get("TUNE_TOKENS", "1000"))
CONFIGS = [dict()] + [dict(c) for c in json.loads(os.environ.get("TUNE_CONFIGS", "[]"))]
KNOBS = ["ER_NO_GRAPH", "ER_NW_QKV", "ER_NW_FC1", "ER_DECODE_V", "ER_ATTN_V_BATCHED", "ER_PREFILL_ATTN_F16S", "ER_FORCE_BATCHED", "ER_BATCHED_VALU"]
PRECISION = os.environ.get("TUNE_PRECISION", "fp32")


def main():
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=24, generate_mode="greedy")
    t0 = time.time()
    sd = W.make_state_dict(opt, 0, "perturbed")
    print(f"weights in {time.time() - t0:.1f}s", flush=True)
    run(opt, sd, CONFIGS, PRECISION, T)


def run(opt, sd, CONFIGS, PRECISION, T):
    pc = W.synthetic_point_cloud(0, 4096).to("cuda:0")
    ref = None
    for cfg in CONFIGS:
        for k in KNOBS:
            os.environ.pop(k, None)
        for k, v in cfg.items():
            os.environ[k] = str(v)
        lmm = LMM(opt, "cuda:0", precision=PRECISION)
        lmm.load_state_dict(sd, strict=True)
        best, pre_ms = 0.0, 1e9
        for rep in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, toks = lmm.generate(pc, 1000, tokenizer=object(), max_new_tokens=T, min_new_tokens=T)
            torch.cuda.synchronize()
            total_ms = (time.perf_counter() - t1) * 1e3
            best = max(best, T / lmm.mesh_decoder.last_decode_ms * 1e3)
            pre_ms = min(pre_ms, total_ms - lmm.mesh_decoder.last_decode_ms)     # encode_cond + prefill + host glue
        if ref is None:
            ref = toks[0].copy()
        same = bool((toks[0] == ref).all())
        prof = lmm.mesh_decoder.profile_decode_kernels(repeats=3, use_graph=True)
        per_tok = sum(v["avg_us"] * (24 if k not in ("lm_head_gemv", "sample_head") else 1) for k, v in prof.items())
        print(json.dumps({"cfg": cfg, "precision": PRECISION, "decode_tok_s": round(best, 1), "ids_equal_base": same,
                          "encode_prefill_ms": round(pre_ms, 1),
                          "sweep_us_per_token": round(per_tok, 1),
                          "kinds_us": {k: round(v["avg_us"], 2) for k, v in prof.items()},
                          "kinds_GBps": {k: round(v["bytes"] / max(v["avg_us"], 1e-9) / 1e3, 0) for k, v in prof.items()}}), flush=True)
        lmm.mesh_decoder.close()
        del lmm
        torch.cuda.empty_cache()
    for k in KNOBS:
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
