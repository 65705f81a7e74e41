#!/usr/bin/env python3
"""encode_cond + prefill wall time per sample at B = 1 / 8 / 32 (GPU box only).  Usage: prefill_time.py [fp16|fp32] [B,B,...]"""
import dataclasses
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgerunner_amd import weights as W  # noqa: E402
from edgerunner_amd.models import LMM  # noqa: E402
from edgerunner_amd.options import config_defaults  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
Bs = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else "1,8,32").split(",")]
opt = dataclasses.replace(config_defaults["ArAE"], num_layers=24, generate_mode="greedy")
m = LMM(opt, "cuda:0", precision=prec)
m.mesh_decoder.load_state_iter(W.iter_state_dict(opt, 0, "perturbed"), strict=True)
for B in Bs:
    pcs = torch.cat([W.synthetic_point_cloud(i, 4096) for i in range(B)]).to("cuda:0")
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        m.generate(pcs, 1000, tokenizer=object(), max_new_tokens=4, min_new_tokens=4)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) * 1e3)
    print(f"{prec} B {B}: encode + prefill + 4 steps {best:.1f} ms, {best / B:.2f} ms per sample", flush=True)
