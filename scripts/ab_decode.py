#!/usr/bin/env python3
"""A/B of library BUILDS on the single-row decode path (GPU box only): for every ER_LIB_PATH given, one subprocess that builds the
24-layer context, decodes T tokens twice and prints decode tok/s + the per-kernel-kind HIP-event sweep (24 graph-replayed launches of a
kind at the run's mean context).  Usage: python scripts/ab_decode.py [fp32|fp16] T label=[path][:ENV=V,...] ...   (empty path = the default build)"""
import dataclasses
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(prec, T):
    import torch
    sys.path.insert(0, ROOT)
    from edgerunner_amd import weights as W
    from edgerunner_amd.models import LMM
    from edgerunner_amd.options import config_defaults
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=24, generate_mode="greedy")
    lmm = LMM(opt, "cuda:0", precision=prec)
    lmm.mesh_decoder.load_state_iter(W.iter_state_dict(opt, 0, "perturbed"), strict=True)
    pc = W.synthetic_point_cloud(0, 4096).to("cuda:0")
    best, ids = 0.0, None
    for rep in range(2):
        _, toks = lmm.generate(pc, 1000, tokenizer=object(), max_new_tokens=T, min_new_tokens=T)
        torch.cuda.synchronize()
        best = max(best, T / lmm.mesh_decoder.last_decode_ms * 1e3)
        ids = toks[0]
    prof = lmm.mesh_decoder.profile_decode_kernels(repeats=5, context_len=4050, use_graph=True)
    import zlib
    print(json.dumps({"label": os.environ.get("AB_LABEL", ""), "precision": prec, "T": T, "decode_tok_s": round(best, 1),
                      "ids_crc": zlib.crc32(ids.tobytes()),
                      "layer_us": round(sum(v["avg_us"] for k, v in prof.items() if k not in ("lm_head_gemv", "sample_head")), 2),
                      "kinds_us": {k: round(v["avg_us"], 2) for k, v in prof.items()}}), flush=True)


if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        child(sys.argv[1], int(sys.argv[2]))
        sys.exit(0)
    prec, T = sys.argv[1], int(sys.argv[2])
    for spec in sys.argv[3:]:
        label, _, rest = spec.partition("=")
        path, _, envs = rest.partition(":")                     # label=path:ENV=V,ENV2=V  (path may be empty = the default build)
        env = dict(os.environ, AB_CHILD="1", AB_LABEL=label)
        if path:
            env["ER_LIB_PATH"] = os.path.join(ROOT, path)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__), prec, str(T)], env=env, capture_output=True, text=True)
        out = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(out[-1] if out else f"{label}: FAILED rc={r.returncode} {r.stderr[-400:]}", flush=True)
