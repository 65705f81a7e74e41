#!/usr/bin/env python3
"""Drop-in for the reference's ``infer.py`` (point-conditioned ArAE inference) on MI355X.

    python infer.py ArAE --workspace out --resume model.safetensors --test_path mesh_or_dir \
        --generate_mode greedy --test_num_face 1000 --test_repeat 1 --seed 0

Same flags (``edgerunner_amd.options`` mirrors ``core/options.py``), same outputs
(``{name}_{i}_{n}f_tokens.npy`` = ids-3 cut at EOS, ``{name}_pc.obj``; reference infer.py:86-123).
Inputs: .obj/.ply meshes (surface-sampled to ``point_num`` points) or .npy point clouds [N,3].
Difference from the reference on MESH inputs: reference infer.py:86 first runs ``kiui.mesh_utils.clean_mesh(v, f, min_f=0, min_d=0,
remesh=False)`` (pymeshlab filters - both packages absent here and not restated) and samples with ``trimesh.sample``; this script
normalises the mesh as loaded and uses its own seeded area-weighted sampler, so the conditioning cloud is a different sample of the
same surface.  ``.npy`` clouds are passed through untouched (identical conditions for both code bases).  Meshes are written with
``mesh.export(...)`` on the ``edgerunner_amd.meto.Mesh`` objects generate() returns (reference infer.py:120 on trimesh objects).
``--cond_mode none`` (reference infer.py:96-97) generates from the face-count token alone, once per input path.
With torchrun (one process per GPU) the (file x repeat x num_face) jobs are sharded block-cyclically over ranks;
inside a rank, jobs with the same face count run as ONE batched generate() call (the B > 1 decode path streams the
weights once for all rows), and the generated token streams are all-gathered over RCCL at the end
(``{workspace}/tokens_all.npz`` on rank 0).

Reproducibility: in sample mode job j (its index in the reference's loop order) draws from the Philox stream
(--seed, step, j), whatever the world size, ER_INFER_BATCH or free memory made of the grouping.  In the fp16 (default) precision
the projections of calls with more than 4 rows run on the matrix cores and round differently from the 1..4-row kernels
(both within 1e-5 of the fp16-storage model), so a near-tie can still resolve differently when the grouping changes;
EDGERUNNER_PRECISION=fp32 with ER_INFER_BATCH <= 4 (or any fixed grouping) is bit-reproducible.
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
    if opt.cond_mode == "none":                      # reference infer.py:96-97: a [1, 0] dummy that only carries the batch size
        return np.zeros((0, 3), dtype=np.float32)
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32).reshape(-1, 3)
    rng = np.random.default_rng([int(opt.seed) & 0xFFFFFFFF, zlib.crc32(os.path.basename(path).encode())])
    v, f = meshio.load_mesh(path)
    v = meshio.normalize_mesh(v, bound=0.95)
    return meshio.sample_surface(v, f, opt.point_num, rng).astype(np.float32)


def max_rows_per_call(opt, model, max_new_tokens, device) -> int:
    """How many independent jobs one generate() call carries: ER_INFER_BATCH (default 32 = one pass of the batched
    projections) - a fixed number, so the grouping does not depend on what else occupies the device.  Memory is only a guard:
    the rows' KV cache + prefill / encoder scratch must fit in the HBM that is free once the weights are resident (the native
    context is created here, before the measurement)."""
    d = model.dims
    esz = 2 if model.precision == "fp16" else 4
    l_cap = d.num_cond_tokens + 2 + max_new_tokens
    kv_row = 2 * d.num_layers * d.hidden_dim * esz * l_cap
    scratch_row = (d.num_cond_tokens + 1) * (6 * d.hidden_dim + d.intermediate_dim) * 4
    if d.cond_mode == "point":                       # encoder scratch per sample at point_num points (K / V / x rows + the GEGLU buffers)
        scratch_row += (3 * opt.point_num + 14 * d.point_latent_size) * d.point_hidden_dim * 4
    _ = model.mesh_decoder                           # materialise the context: the weights are on the device from here on
    free, _total = torch.cuda.mem_get_info(device)
    cap = max(1, int(os.environ.get("ER_INFER_BATCH", "32")))
    fit = int(0.8 * free // (kv_row + scratch_row))
    if fit < 1:
        raise SystemExit(f"[ERROR] not enough free HBM for one job: {free / 2**30:.1f} GiB free, one row needs "
                         f"{(kv_row + scratch_row) / 2**30:.1f} GiB (lower --test_max_seq_length)")
    if fit < cap:
        print(f"[WARN] {free / 2**30:.1f} GiB free: {fit} jobs per call instead of {cap} (sample-mode results are unaffected, "
              "fp16 greedy near-ties may resolve differently - see the module docstring)")
    return min(cap, fit)


def main(argv=None):
    opt = parse_cli(argv)
    rank, world, local = D.init_process_group()
    seed_everything(opt.seed)
    if opt.cond_mode not in ("point", "none"):
        raise SystemExit("infer.py serves cond_mode='point' (ArAE preset) and 'none'; see infer_dit.py for the image -> point_latent path")
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
    model.release_checkpoint()        # the weights are on the device in their final precision: drop the host copy (~2.7 GB per rank)
    ckpt = None

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
        if pc_owner[path] == rank and opt.cond_mode == "point":
            if path not in clouds:
                clouds[path] = load_points(opt, path)
            name = os.path.splitext(os.path.basename(path))[0]
            meshio.save_points_obj(f"{opt.workspace}/{name}_pc.obj", clouds[path])
    rows_max = max_rows_per_call(opt, model, opt.test_max_seq_length, device)
    local_streams = {}
    for num_faces, chunk in D.group_jobs(jobs, mine, lambda p: clouds[p].shape[0], rows_max):
        cond = torch.from_numpy(np.stack([clouds[jobs[j][0]] for j in chunk])).float().to(device)
        t0 = time.time()
        # sample mode: job j draws from the Philox stream (opt.seed, step, j) whatever rows share its call
        meshes, tokens = model.generate(cond, num_faces=num_faces, max_new_tokens=opt.test_max_seq_length,
                                        tokenizer=tokenizer, clean=True, seed=opt.seed, row_streams=list(chunk))
        torch.cuda.synchronize()
        dt = time.time() - t0
        for r, j in enumerate(chunk):
            path, i, _ = jobs[j]
            name = os.path.splitext(os.path.basename(path))[0]
            toks = trim_tokens(tokens[r])
            filename = f"{name}_{i}" + (f"_{num_faces}f" if opt.use_num_face_cond else "")
            np.save(f"{opt.workspace}/{filename}_tokens.npy", toks)
            if meshes[r] is not None:
                meshes[r].export(f"{opt.workspace}/{filename}.ply")               # reference infer.py:120
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
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
