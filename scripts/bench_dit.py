#!/usr/bin/env python3
"""BASELINE configs[4] at full depth on one GPU: image -> CLIP ViT-H/14 (32 layers) -> proj/norm -> DiT (24 layers,
100 DDIM steps, CFG 7.5) -> latents -> ArAE greedy decode (24 layers, T tokens).  Synthetic weights; prints phase times."""
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
from edgerunner_amd.models_dit import MDiT  # noqa: E402
from edgerunner_amd.options import config_defaults  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    opt = dataclasses.replace(config_defaults["DiT"], generate_mode="greedy", cond_mode="point_latent")
    dev = "cuda:0"
    t0 = time.time()
    precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    mdit = MDiT(opt, dev, clip_layers=32, precision=precision)
    sd = W.make_dit_state_dict(opt, 0, "perturbed")
    sd.update(W.make_clip_state_dict(32, 0, "perturbed"))
    mdit.load_state_dict(sd, strict=True)
    del sd
    lmm = LMM(opt, dev)
    lmm.mesh_decoder.load_state_iter(W.iter_state_dict(opt, 0, "perturbed"), strict=True)
    print(f"weights in {time.time() - t0:.1f}s", flush=True)
    img = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        cond = mdit.get_cond(img)
        torch.cuda.synchronize(); res["clip_encode_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        lat = mdit.run(img, num_inference_steps=steps, guidance_scale=7.5)
        torch.cuda.synchronize(); res["dit_sample_ms_incl_encode"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        _, toks = lmm.generate(lat, 1000, tokenizer=object(), max_new_tokens=T, min_new_tokens=T)
        torch.cuda.synchronize(); res["arae_generate_ms"] = (time.perf_counter() - t) * 1e3
    res.update({"ddim_steps": steps, "tokens": T, "ms_per_dit_forward_cfg2": (res["dit_sample_ms_incl_encode"] - res["clip_encode_ms"]) / steps,
                "end_to_end_tokens_per_s": T / ((res["dit_sample_ms_incl_encode"] + res["arae_generate_ms"]) / 1e3),
                "finite": bool(torch.isfinite(lat).all())})
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
