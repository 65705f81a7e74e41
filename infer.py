#!/usr/bin/env python3
"""Drop-in for the reference's ``infer.py`` (point-conditioned ArAE inference) on MI355X.

    python infer.py ArAE --workspace out --resume model.safetensors --test_path mesh_or_dir \
        --generate_mode greedy --test_num_face 1000 --test_repeat 1 --seed 0

Same flags (``edgerunner_amd.options`` mirrors ``core/options.py``), same outputs
(``{name}_{i}_{n}f_tokens.npy`` = ids-3 cut at EOS, ``{name}_pc.obj``; reference infer.py:86-123).
Inputs: .obj/.ply meshes (surface-sampled to ``point_num`` points) or .npy point clouds [N,3].
With torchrun (one process per GPU) the (file x repeat x num_face) jobs are sharded block-cyclically over ranks;
inside a rank, jobs with the same face count run as ONE batched generate() call (the B > 1 decode path streams the
weights once for all rows), and the generated token streams are all-gathered over RCCL at the end
(``{workspace}/tokens_all.npz`` on rank 0).
"""
from __future__ import annotations

import glob
import os
import sys
import time
import zlib

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


def load_points(opt, path):
    """One cloud per input path, reused for every repeat and face count (reference infer.py:84-92).  The surface
    sampler is seeded by (opt.seed, path) so that every rank - and a world-size-1 run - derives the same cloud."""
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32).reshape(-1, 3)
    rng = np.random.default_rng([int(opt.seed) & 0xFFFFFFFF, zlib.crc32(os.path.basename(path).encode())])
    v, f = meshio.load_mesh(path)
    v = meshio.normalize_mesh(v, bound=0.95)
    return meshio.sample_surface(v, f, opt.point_num, rng).astype(np.float32)


def max_rows_per_call(opt, model, max_new_tokens, device) -> int:
    """How many independent jobs one generate() call may carry: bounded by the KV cache (+ prefill scratch) that fits
    in half of the free HBM, by 32 (one pass of the batched projections) and by ER_INFER_BATCH."""
    d = model.dims
    esz = 2 if model.precision == "fp16" else 4
    l_cap = d.num_cond_tokens + 2 + max_new_tokens
    kv_row = 2 * d.num_layers * d.hidden_dim * esz * l_cap
    scratch_row = (d.num_cond_tokens + 1) * (6 * d.hidden_dim + d.intermediate_dim) * 4
    free, _ = torch.cuda.mem_get_info(device)
    cap = int(os.environ.get("ER_INFER_BATCH", "32"))
    return max(1, min(cap, int(0.5 * free // (kv_row + scratch_row))))


def main(argv=None):
    opt = parse_cli(argv)
    rank, world, local = D.init_process_group()
    seed_everything(opt.seed)
    if opt.cond_mode != "point":
        raise SystemExit("this build serves cond_mode='point' (ArAE preset); see infer_dit for point_latent")
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device visible: this path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    model = LMM(opt, device, precision=None)          # module style: storage precision follows .half() / .float()
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
        model.load_state_dict(W.make_state_dict(opt, opt.seed, "reference"), strict=True)
    # the reference runs fp16 on the GPU (infer.py:56,105): .half() selects the fp16-storage context (fp32 accumulate);
    # EDGERUNNER_PRECISION=fp32 keeps the exact mode (greedy ids bit-exact vs the CPU path)
    if os.environ.get("EDGERUNNER_PRECISION", "fp16") == "fp32":
        model = model.float().eval().to(device)
    else:
        model = model.half().eval().to(device)

    tokenizer, _ = get_tokenizer(opt)

    assert opt.test_path is not None
    paths = sorted(glob.glob(os.path.join(opt.test_path, "*"))) if os.path.isdir(opt.test_path) else [opt.test_path]
    os.makedirs(opt.workspace, exist_ok=True)
    # the reference's serial loops (infer.py:99-101,136-137): path x repeat x num_face, here sharded block-cyclically over
    # ranks and, inside a rank, batched through the B > 1 decode path (rows are independent)
    jobs, mine, pc_owner = D.plan_jobs(paths, opt.test_repeat, opt.test_num_face, rank, world)
    clouds = {}
    for j in mine:
        path = jobs[j][0]
        if path not in clouds:
            clouds[path] = load_points(opt, path)
    for path in paths:                               # exactly one rank exports the cloud of a path
        if pc_owner[path] == rank:
            if path not in clouds:
                clouds[path] = load_points(opt, path)
            name = os.path.splitext(os.path.basename(path))[0]
            meshio.save_points_obj(f"{opt.workspace}/{name}_pc.obj", clouds[path])
    rows_max = max_rows_per_call(opt, model, opt.test_max_seq_length, device)
    local_streams = {}
    for num_faces, chunk in D.group_jobs(jobs, mine, lambda p: clouds[p].shape[0], rows_max):
        cond = torch.from_numpy(np.stack([clouds[jobs[j][0]] for j in chunk])).float().to(device)
        t0 = time.time()
        meshes, tokens = model.generate(cond, num_faces=num_faces, max_new_tokens=opt.test_max_seq_length,
                                        tokenizer=tokenizer, clean=True, seed=opt.seed + 7919 * chunk[0])
        torch.cuda.synchronize()
        dt = time.time() - t0
        for r, j in enumerate(chunk):
            path, i, _ = jobs[j]
            name = os.path.splitext(os.path.basename(path))[0]
            toks = trim_tokens(tokens[r])
            filename = f"{name}_{i}" + (f"_{num_faces}f" if opt.use_num_face_cond else "")
            np.save(f"{opt.workspace}/{filename}_tokens.npy", toks)
            if meshes[r] is not None:
                meshio.save_ply(f"{opt.workspace}/{filename}.ply", meshes[r][0], meshes[r][1])
            local_streams[j] = toks
            print(f"[INFO] Processing {path} --> {filename}.ply, {len(toks)} tokens, time = {dt:.4f}s "
                  f"({len(chunk)} jobs in this call)")
    # the one exchange of the sharded path: RCCL all-gather of the token streams (ids - 3, >= -3, so shift to >= 0)
    gathered = D.gather_token_streams([local_streams[j] + 3 for j in mine], len(jobs), device=device)
    if rank == 0:
        index = [f"{os.path.splitext(os.path.basename(p))[0]}_{i}" + (f"_{nf}f" if opt.use_num_face_cond else "")
                 for p, i, nf in jobs]
        np.savez(f"{opt.workspace}/tokens_all.npz", **{k: g - 3 for k, g in zip(index, gathered)})
        print(f"[INFO] {len(jobs)} jobs over {world} rank(s); token streams gathered into {opt.workspace}/tokens_all.npz")
    D.barrier()


if __name__ == "__main__":
    main()
