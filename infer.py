#!/usr/bin/env python3
"""Drop-in for the reference's ``infer.py`` (point-conditioned ArAE inference) on MI355X.

    python infer.py ArAE --workspace out --resume model.safetensors --test_path mesh_or_dir \
        --generate_mode greedy --test_num_face 1000 --test_repeat 1 --seed 0

Same flags (``edgerunner_amd.options`` mirrors ``core/options.py``), same outputs
(``{name}_{i}_{n}f_tokens.npy`` = ids-3 cut at EOS, ``{name}_pc.obj``; reference infer.py:86-123).
Inputs: .obj/.ply meshes (surface-sampled to ``point_num`` points) or .npy point clouds [N,3].
With torchrun (one process per GPU) the (file x repeat x num_face) jobs are sharded over ranks.
"""
from __future__ import annotations

import glob
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from edgerunner_amd import dist as D  # noqa: E402
from edgerunner_amd import meshio  # noqa: E402
from edgerunner_amd.models import LMM  # noqa: E402
from edgerunner_amd.options import parse_cli  # noqa: E402
from edgerunner_amd.meto import get_tokenizer  # noqa: E402
from edgerunner_amd.utils import seed_everything, trim_tokens  # noqa: E402


def load_points(opt, path, rng):
    if path.endswith(".npy"):
        pts = np.load(path).astype(np.float32).reshape(-1, 3)
    else:
        v, f = meshio.load_mesh(path)
        v = meshio.normalize_mesh(v, bound=0.95)
        pts = meshio.sample_surface(v, f, opt.point_num, rng).astype(np.float32)
    return pts


def main(argv=None):
    opt = parse_cli(argv)
    rank, world, local = D.init_process_group()
    seed_everything(opt.seed)
    if opt.cond_mode != "point":
        raise SystemExit("this build serves cond_mode='point' (ArAE preset); see infer_dit for point_latent")
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device visible: this path has no CPU fallback")
    device = torch.device("cuda", local)
    # the reference runs fp16 on GPU (infer.py:56,105): 'fp16' = that storage precision, fp32 accumulate;
    # EDGERUNNER_PRECISION=fp32 selects the exact mode (greedy ids bit-exact vs the CPU path)
    model = LMM(opt, device, precision=os.environ.get("EDGERUNNER_PRECISION", "fp16"))
    if opt.resume is not None:
        if opt.resume.endswith("safetensors"):
            from safetensors.torch import load_file
            ckpt = load_file(opt.resume, device="cpu")
        else:
            ckpt = torch.load(opt.resume, map_location="cpu")
        model.load_state_dict(ckpt, strict=False)
        print(f"[INFO] Loaded checkpoint from {opt.resume}")
    else:
        from edgerunner_amd import weights as W
        print("[WARN] model randomly initialized, are you sane?")
        model.mesh_decoder.load_state_iter(W.iter_state_dict(opt, opt.seed, "reference"), strict=True)
    model = model.half().eval().to(device)

    tokenizer, _ = get_tokenizer(opt)

    assert opt.test_path is not None
    paths = sorted(glob.glob(os.path.join(opt.test_path, "*"))) if os.path.isdir(opt.test_path) else [opt.test_path]
    os.makedirs(opt.workspace, exist_ok=True)
    jobs = [(p, i, nf) for p in paths for i in range(opt.test_repeat) for nf in opt.test_num_face]
    rng = np.random.default_rng(opt.seed)
    clouds = {}
    for j in D.shard_indices(len(jobs), rank, world):
        path, i, num_faces = jobs[j]
        name = os.path.splitext(os.path.basename(path))[0]
        if path not in clouds:
            clouds[path] = load_points(opt, path, rng)
            meshio.save_points_obj(f"{opt.workspace}/{name}_pc.obj", clouds[path])
        cond = torch.from_numpy(clouds[path]).unsqueeze(0).float().to(device)
        t0 = time.time()
        meshes, tokens = model.generate(cond, num_faces=num_faces, max_new_tokens=opt.test_max_seq_length,
                                        tokenizer=tokenizer, clean=True,
                                        seed=opt.seed + 7919 * j)
        tokens = trim_tokens(tokens[0])
        filename = f"{name}_{i}" + (f"_{num_faces}f" if opt.use_num_face_cond else "")
        np.save(f"{opt.workspace}/{filename}_tokens.npy", tokens)
        if meshes[0] is not None:
            meshio.save_ply(f"{opt.workspace}/{filename}.ply", meshes[0][0], meshes[0][1])
        torch.cuda.synchronize()
        print(f"[INFO] Processing {path} --> {filename}, {len(tokens)} tokens, time = {time.time() - t0:.4f}s")
    D.barrier()


if __name__ == "__main__":
    main()
