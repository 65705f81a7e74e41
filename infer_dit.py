#!/usr/bin/env python3
"""Drop-in for the reference's ``infer_dit.py`` (image -> mesh tokens) on MI355X.

    python infer_dit.py DiT --workspace out --resume lmm.safetensors --resume2 mdit.safetensors --test_path images/

Pipeline (reference infer_dit.py:40-140): image -> MDiT.run (CLIP ViT-H/14 -> DiT, 100 DDIM steps, CFG 7.5) ->
latents [1,2048,64] -> LMM.generate in cond_mode 'point_latent' -> ``{name}_{i}_{n}f_tokens.npy`` (+ .ply).
Inputs: RGBA / RGB images readable by PIL, or .npy arrays [H,W,3|4] in [0,1].  Background removal and
recentering (rembg / kiui in the reference) are not reproduced: supply a segmented, centred image.
"""
from __future__ import annotations

import glob
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from edgerunner_amd import dist as D  # noqa: E402
from edgerunner_amd import meshio  # noqa: E402
from edgerunner_amd.meto import get_tokenizer  # noqa: E402
from edgerunner_amd.models import LMM  # noqa: E402
from edgerunner_amd.models_dit import MDiT  # noqa: E402
from edgerunner_amd.options import parse_cli  # noqa: E402
from edgerunner_amd.utils import seed_everything, trim_tokens  # noqa: E402


def load_ckpt(path):
    if path.endswith("safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu")


def load_image(path):
    if path.endswith(".npy"):
        a = np.load(path).astype(np.float32)
    else:
        from PIL import Image
        a = np.asarray(Image.open(path)).astype(np.float32) / 255.0
    if a.ndim == 2:
        a = np.repeat(a[..., None], 3, axis=-1)
    if a.shape[-1] == 4:                      # composite on white (infer_dit.py:93)
        a = a[..., :3] * a[..., 3:4] + (1 - a[..., 3:4])
    return a[..., :3]


def main(argv=None):
    opt = parse_cli(argv)
    rank, world, local = D.init_process_group()
    seed_everything(opt.seed)
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device visible: this path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    from edgerunner_amd import weights as W
    # The reference builds LMM(opt) with the preset's cond_mode and only then flips it to 'point_latent'
    # (infer_dit.py:41,55): generate() thereafter skips the point encoder.  The native context is sized by cond_mode, so
    # the flip happens first here; the checkpoint's point_encoder.* tensors are then simply unexpected keys (strict=False).
    opt.cond_mode = "point_latent"
    model = LMM(opt, device, precision=None)          # module style: storage precision follows .half() below
    if opt.resume is not None:
        model.load_state_dict(load_ckpt(opt.resume), strict=False)
        print(f"[INFO] Loaded checkpoint from {opt.resume}")
    else:
        print("[WARN] model randomly initialized, are you sane?")
        model.load_state_dict(W.make_state_dict(opt, opt.seed, "reference"), strict=True)
    clip_layers = int(os.environ.get("ER_CLIP_LAYERS", "32"))      # test knob: the reference's encoder is ViT-H/14, 32 layers
    model_dit = MDiT(opt, device, clip_layers=clip_layers, precision=None)
    if opt.resume2 is not None:
        model_dit.load_state_dict(load_ckpt(opt.resume2), strict=False)
        print(f"[INFO] Loaded checkpoint from {opt.resume2}")
    else:
        sd = W.make_dit_state_dict(opt, opt.seed, "reference")
        sd.update(W.make_clip_state_dict(clip_layers, opt.seed, "reference"))
        model_dit.load_state_dict(sd, strict=True)
    # reference infer_dit.py:69-70: both models run fp16 on the GPU; EDGERUNNER_PRECISION=fp32 keeps the exact mode
    if os.environ.get("EDGERUNNER_PRECISION", "fp16") == "fp32":
        model, model_dit = model.float().eval().to(device), model_dit.float().eval().to(device)
    else:
        model, model_dit = model.half().eval().to(device), model_dit.half().eval().to(device)
    # the weights are on the device in their final precision: drop the host copies of both checkpoints
    model.release_checkpoint()
    model_dit.release_checkpoint()
    dit_steps = int(os.environ.get("ER_DIT_STEPS", "100"))         # test knob: MDiT.run's default is 100 (models_dit.py:187)
    tokenizer, _ = get_tokenizer(opt)

    assert opt.test_path is not None
    paths = sorted(glob.glob(os.path.join(opt.test_path, "*"))) if os.path.isdir(opt.test_path) else [opt.test_path]
    os.makedirs(opt.workspace, exist_ok=True)
    jobs = [(p, i, nf) for p in paths for i in range(opt.test_repeat) for nf in opt.test_num_face]
    for j in D.shard_indices(len(jobs), rank, world):
        path, i, num_faces = jobs[j]
        name = os.path.splitext(os.path.basename(path))[0]
        image = torch.from_numpy(load_image(path)).permute(2, 0, 1).contiguous().unsqueeze(0).float().to(device)
        cond = F.interpolate(image, (512, 512), mode="bilinear", align_corners=False)      # infer_dit.py:97
        t0 = time.time()
        latents = model_dit.run(cond, num_inference_steps=dit_steps)
        meshes, tokens = model.generate(latents, num_faces=num_faces, max_new_tokens=opt.test_max_seq_length,
                                        tokenizer=tokenizer, clean=True, seed=opt.seed + 7919 * j)
        tokens = trim_tokens(tokens[0])
        filename = f"{name}_{i}" + (f"_{num_faces}f" if opt.use_num_face_cond else "")
        np.save(f"{opt.workspace}/{filename}_tokens.npy", tokens)
        if meshes[0] is not None:
            meshes[0].export(f"{opt.workspace}/{filename}.ply")                   # reference infer_dit.py:126
        torch.cuda.synchronize()
        print(f"[INFO] Processing {path} --> {filename}, {len(tokens)} tokens, time = {time.time() - t0:.4f}s")
    D.barrier()


if __name__ == "__main__":
    main()
