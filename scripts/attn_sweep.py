#!/usr/bin/env python3
"""Attention-kernel duration vs context length for a list of env-knob configurations (GPU box only).

The decode attention's duration is not linear in the context: the number of active chunks per head steps against the
XCD / CU counts.  This prints, per configuration and context, the hipGraph-replayed average of the attention partial
kernel, the merge and the out_proj launch (HIP events, all 24 layers' data), so staircases can be told apart from noise.

env: SWEEP_B (1), SWEEP_T (4000: reserved new tokens), SWEEP_PRECISION (fp32), SWEEP_CONTEXTS (json list),
     SWEEP_CONFIGS (json list of {env var: value}; the empty config always runs first)."""
import dataclasses
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgerunner_amd import weights as W  # noqa: E402
from edgerunner_amd.models import LMM  # noqa: E402
from edgerunner_amd.options import config_defaults  # noqa: E402

B = int(os.environ.get("SWEEP_B", "1"))
T = int(os.environ.get("SWEEP_T", "4000"))
PRECISION = os.environ.get("SWEEP_PRECISION", "fp32")
CONTEXTS = json.loads(os.environ.get("SWEEP_CONTEXTS", "[2176, 3176, 3926, 4050, 4176, 5176, 5926]"))
CONFIGS = [dict()] + [dict(c) for c in json.loads(os.environ.get("SWEEP_CONFIGS", "[]"))]
KNOBS = ["ER_ATTN_V_BATCHED", "ER_DECODE_V", "ER_NW_QKV"]


def main():
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=24, generate_mode="greedy")
    sd = W.make_state_dict(opt, 0, "perturbed")
    run(opt, sd, CONFIGS, PRECISION, B, T, CONTEXTS)


def run(opt, sd, CONFIGS, PRECISION, B, T, CONTEXTS):
    pcs = torch.cat([W.synthetic_point_cloud(i, 4096) for i in range(B)]).to("cuda:0")
    for cfg in CONFIGS:
        for k in KNOBS:
            os.environ.pop(k, None)
        for k, v in cfg.items():
            os.environ[k] = str(v)
        lmm = LMM(opt, "cuda:0", precision=PRECISION)
        lmm.load_state_dict(sd, strict=True)
        lmm.mesh_decoder.reserve(B, 2050 + T + 1)
        lmm.generate(pcs, 1000, tokenizer=object(), max_new_tokens=4, min_new_tokens=4)
        rows = []
        for L in CONTEXTS:
            p = lmm.mesh_decoder.profile_decode_kernels(repeats=3, context_len=int(L), use_graph=True)
            rows.append({"L": int(L), "attn_us": round(p["attn_decode"]["avg_us"], 2), "merge_us": round(p["attn_combine"]["avg_us"], 2),
                         "out_proj_us": round(p["out_proj_gemv"]["avg_us"], 2), "qkv_us": round(p["qkv_gemv"]["avg_us"], 2),
                         "attn_GBps": round(p["attn_decode"]["bytes"] / max(p["attn_decode"]["avg_us"], 1e-9) / 1e3, 0)})
        print(json.dumps({"cfg": cfg, "B": B, "precision": PRECISION, "sweep": rows}), flush=True)
        lmm.mesh_decoder.close()
        del lmm
        torch.cuda.empty_cache()
    for k in KNOBS:
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
