#!/usr/bin/env python3
"""Where one bench.py step spends its wall time outside the decode loop (GPU box only): encode_cond, embedding, prefill,
decode (HIP events inside er_decode), ids copy-out, host glue.  Usage: python scripts/step_breakdown.py [fp32|fp16] [T]"""
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


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    dev = "cuda:0"
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=24, generate_mode="greedy")
    lmm = LMM(opt, dev, precision=prec)
    lmm.mesh_decoder.load_state_iter(W.iter_state_dict(opt, 0, "perturbed"), strict=True)
    dec = lmm.mesh_decoder

    def tick():
        torch.cuda.synchronize()
        return time.perf_counter()

    for rep in range(3):
        t0 = tick()
        pc = W.synthetic_point_cloud(rep, 4096).to(dev)
        t1 = tick()
        cond = lmm.encode_cond(pc, [1000])["cond_embeds"]
        t2 = tick()
        tok = dec.model.embd(torch.full((1, 1), opt.bos_token_id, dtype=torch.long))
        emb = torch.cat((cond, tok), dim=1)
        t3 = tick()
        dec.prefill(emb, T)
        t4 = tick()
        from edgerunner_amd import native
        ids = dec._decode_device(1, T, T, False, 10, native.ER_GRAMMAR_LR_ABSCO, None)
        t5 = tick()
        out = ids.detach().cpu().numpy()
        t6 = tick()
        t7 = tick()
        _, toks = lmm.generate(pc, 1000, tokenizer=object(), max_new_tokens=T, min_new_tokens=T)
        t8 = tick()
        print(f"rep {rep}: cloud {1e3 * (t1 - t0):.1f} ms | encode_cond {1e3 * (t2 - t1):.1f} | embd+cat {1e3 * (t3 - t2):.1f} | prefill {1e3 * (t4 - t3):.1f} | "
              f"er_decode wall {1e3 * (t5 - t4):.1f} (events {dec.last_decode_ms:.1f}) | ids to host {1e3 * (t6 - t5):.1f} | "
              f"whole generate() {1e3 * (t8 - t7):.1f} (events {dec.last_decode_ms:.1f})", flush=True)


if __name__ == "__main__":
    main()
