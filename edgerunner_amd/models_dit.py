"""``MDiT`` - the image-conditioned diffusion front-end of the reference (core/models_dit.py:34-229) on
MI355X: projected image condition -> DiT denoiser (core/transformer/dit.py) sampled with DDIM +
classifier-free guidance -> latents ``[B, 2048, 64]`` that ``LMM.generate`` consumes in
``cond_mode='point_latent'`` (infer_dit.py:55,111-113).

Built: the CLIP ViT-H/14 image encoder (``image_encoder.vision_model.*``; architecture only - its pretrained
weights cannot be fetched here, so parity runs use synthetic weights), ``proj_cond``/``norm_cond``,
``DiT.forward``, ``run`` (from noise, and the img2img branch with ``latents`` / ``strength``; ``num_repeat``).
Out of scope: training ``forward``, background removal / recentering of the input photo (rembg, kiui: infer_dit.py:83-96).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import native

CLIP_DIM = 1280   # laion/CLIP-ViT-H-14 hidden width (core/models_dit.py:56)


def ddim_alphas_cumprod(num_train: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> torch.Tensor:
    """alphas_cumprod of the reference's DDIMScheduler (core/models_dit.py:79-98: scaled_linear betas 0.00085..0.012, fp32)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class MDiT:
    def __init__(self, opt, device="cuda:0", clip_layers: int = 32, precision: Optional[str] = "fp32"):
        """clip_layers: depth of the CLIP ViT image encoder to expect in the checkpoint (32 = ViT-H/14 as in the
        reference; 0 = no image encoder: get_cond then takes its last_hidden_state directly).
        precision: 'fp32' (exact), 'fp16' (every Linear on the fp16-input matrix cores, the reference's GPU dtype), or
        None = module style: fp32 until ``.half()`` is called (reference infer_dit.py:70), context created on first use."""
        self.opt = opt
        self.clip_layers = clip_layers
        if precision not in (None, "fp32", "fp16"):
            raise ValueError(precision)
        self._fp16 = precision == "fp16"
        if getattr(opt, "noise_scheduler_predtype", "v_prediction") != "v_prediction":
            # the reference forwards this option to DDIMScheduler (core/models_dit.py:91); the device sampler
            # (ddim_cfg_step_kernel) implements the v-prediction update only - refuse rather than sample wrong latents
            raise NotImplementedError(f"noise_scheduler_predtype={opt.noise_scheduler_predtype!r}: the DDIM step kernel "
                                      "implements 'v_prediction' (the released DiT checkpoints' target) only")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise native.NativeError("MDiT needs a HIP device; there is no CPU fallback")
        self.lib = native.load_library()
        self._ctx_h = C.c_void_p()
        self._sources = []                 # (state_dict reference, strict): replayed when .half()/.float() re-creates the context
        self._released = False
        self.stream = torch.cuda.Stream(device=self.device)
        if precision is not None:
            self._materialize()

    @property
    def precision(self) -> str:
        return "fp16" if self._fp16 else "fp32"

    def _materialize(self):
        if self._released and not self._sources:
            # close() after release_checkpoint(): the weights lived only in the native context that was just destroyed
            raise native.NativeError("MDiT: the native context was closed after release_checkpoint(); the checkpoint is gone - "
                                     "create a new MDiT and load the checkpoint again")
        opt = self.opt
        cfg = native.ErDitConfig(hidden_dim=opt.dit_hidden_dim, num_heads=opt.dit_num_heads, num_layers=opt.dit_num_layers,
                                 latent_size=opt.point_latent_size, latent_dim=opt.point_latent_dim, clip_dim=CLIP_DIM,
                                 clip_layers=self.clip_layers, clip_heads=16, clip_mlp_dim=5120, clip_image_size=224, clip_patch=14,
                                 weight_dtype=native.ER_F16 if self._fp16 else native.ER_F32)
        self._ctx_h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        native.check(self.lib.er_dit_create(C.byref(cfg), idx, C.byref(self._ctx_h)), "er_dit_create")
        for sd, strict in self._sources:
            self._load_now(sd, strict)
        return self._ctx_h

    @property
    def _ctx(self):
        return self._ctx_h if self._ctx_h else self._materialize()

    def close(self):
        if getattr(self, "_ctx_h", None) is not None and self._ctx_h:
            self.lib.er_dit_destroy(self._ctx_h)
            self._ctx_h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """The dict is kept by reference so that .half() / .float() can rebuild the context in the other precision.
        Returns (missing, unexpected); with ``strict`` a mismatch raises HERE, also while the context is not created yet
        (module style): the expected keys are known from the options alone."""
        if self._released:
            raise native.NativeError("MDiT.load_state_dict after release_checkpoint(): create a new MDiT")
        self._sources.append((sd, strict))
        if self._ctx_h:
            return self._load_now(sd, strict)
        from .weights import clip_tensor_specs, dit_tensor_specs
        want = {k for k, _, _ in dit_tensor_specs(self.opt)}
        if self.clip_layers > 0:
            want |= {k for k, _, _ in clip_tensor_specs(self.clip_layers)}
        def norm(k):                           # transformers >= 5 drops the "vision_model." level (er_dit.h accepts both)
            if k.startswith("image_encoder.") and not k.startswith("image_encoder.vision_model."):
                return "image_encoder.vision_model." + k[len("image_encoder."):]
            return k
        have = set()
        for src, _ in self._sources:           # several partial dicts (denoiser, image encoder) add up
            have |= {norm(k) for k, t in src.items() if isinstance(t, torch.Tensor)}
        missing, unexpected = sorted(want - have), sorted({k for k in sd if norm(k) not in want})
        if strict and unexpected:
            raise native.NativeError(f"MDiT.load_state_dict(strict=True): unexpected {unexpected[:5]}")
        return missing, unexpected

    def release_checkpoint(self):
        """Drop the retained state_dict references once the native context holds the weights (see LMM.release_checkpoint)."""
        _ = self._ctx
        self._sources.clear()
        self._released = True
        return self

    def _load_now(self, sd, strict):
        unexpected = []
        for key, t in sd.items():
            t = t.detach().float().contiguous() if t.dtype not in (torch.float32, torch.float16, torch.bfloat16) else t.detach().contiguous()
            dt = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[t.dtype]
            shape = (C.c_int64 * max(1, t.dim()))(*(list(t.shape) or [1]))
            rc = native.check(self.lib.er_dit_load_tensor(self._ctx_h, key.encode(), native.ptr(t), dt, max(1, t.dim()), shape,
                                                          1 if t.is_cuda else 0), f"er_dit_load_tensor({key})")
            if rc == 1:
                unexpected.append(key)
        rc = self.lib.er_dit_finalize_weights(self._ctx_h)
        missing = [self.lib.er_last_error().decode()] if rc < 0 else []
        if strict and (missing or unexpected):
            raise native.NativeError(f"missing={missing} unexpected={unexpected[:4]}")
        return missing, unexpected

    def _cast(self, fp16: bool):
        if fp16 != self._fp16 and self._released:
            raise native.NativeError("MDiT.half()/float() after release_checkpoint(): the weights cannot be re-stored")
        if fp16 != self._fp16:
            self._fp16 = fp16
            self.close()                   # rebuilt on next use from the retained checkpoints
        return self

    def half(self):
        """reference infer_dit.py:70: selects the fp16 matrix-core context."""
        return self._cast(True)

    def float(self):
        return self._cast(False)

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise native.NativeError("MDiT only runs on a HIP device")
        return self

    def _sync_in(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _sync_out(self):
        self.stream.synchronize()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def _sp(self):
        return C.c_void_p(self.stream.cuda_stream)

    @torch.no_grad()
    def get_cond(self, inputs: torch.Tensor) -> torch.Tensor:
        """core/models_dit.py:104-115.  inputs: images [B, 3, H, W] in [0, 1] (normalised, resized to 224 and
        encoded by the CLIP ViT here) or an already computed last_hidden_state [B, 257, 1280]."""
        if inputs.dim() == 4:
            if self.clip_layers <= 0:
                raise NotImplementedError("this MDiT was created without the image encoder (clip_layers=0)")
            img = inputs.to(self.device, torch.float32).contiguous()
            self._sync_in()
            with torch.cuda.stream(self.stream):
                hid = torch.empty((img.shape[0], 257, CLIP_DIM), dtype=torch.float32, device=self.device)
                native.check(self.lib.er_dit_encode_image(self._ctx, native.ptr(img), img.shape[0], img.shape[2], img.shape[3],
                                                          native.ptr(hid), self._sp()), "er_dit_encode_image")
            self._sync_out()
            inputs = hid
        if inputs.dim() != 3 or inputs.shape[-1] != CLIP_DIM:
            raise ValueError(f"expected images [B,3,H,W] or CLIP hidden states [B,257,{CLIP_DIM}], got {tuple(inputs.shape)}")
        x = inputs.to(self.device, torch.float32).contiguous()
        self._sync_in()
        with torch.cuda.stream(self.stream):
            out = torch.empty((x.shape[0], x.shape[1], self.opt.dit_hidden_dim), dtype=torch.float32, device=self.device)
            native.check(self.lib.er_dit_project_cond(self._ctx, native.ptr(x), x.shape[0], x.shape[1], native.ptr(out), self._sp()),
                         "er_dit_project_cond")
        self._sync_out()
        return out

    @torch.no_grad()
    def dit(self, x: torch.Tensor, c: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """DiT.forward(x, c, t) (core/transformer/dit.py:168-196)."""
        x = x.to(self.device, torch.float32).contiguous()
        c = c.to(self.device, torch.float32).contiguous()
        th = (C.c_float * x.shape[0])(*[float(v) for v in t.flatten().tolist()])
        self._sync_in()
        with torch.cuda.stream(self.stream):
            out = torch.empty_like(x)
            native.check(self.lib.er_dit_forward(self._ctx, native.ptr(x), native.ptr(c), th, x.shape[0], c.shape[1], native.ptr(out),
                                                 self._sp()), "er_dit_forward")
        self._sync_out()
        return out

    @torch.no_grad()
    def run(self, inputs, num_inference_steps=100, guidance_scale=7.5, num_repeat=1, latents=None, strength=0.5,
            noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """core/models_dit.py:184-229.  ``noise``: the Gaussian draw the reference takes from torch.randn /
        torch.randn_like (default: drawn on the device, like the reference); pass it explicitly to reproduce a CPU
        run.  ``latents`` given: the img2img branch (:207-209) - noise is added at timesteps[int(steps * strength)]
        and the loop starts there."""
        cond = self.get_cond(inputs)
        cond = cond.repeat_interleave(num_repeat, dim=0).contiguous()
        B = cond.shape[0]
        shape = (B, self.opt.point_latent_size, self.opt.point_latent_dim)
        if noise is None:
            noise = torch.randn(*shape, device=self.device, dtype=torch.float32)
        noise = noise.to(self.device, torch.float32)
        steps = int(num_inference_steps)
        if latents is None:
            init_step = 0
            lat = noise.contiguous().clone()
        else:
            init_step = int(steps * strength)
            if not 0 <= init_step < steps:
                raise IndexError(f"strength {strength} selects timestep index {init_step} of {steps}")   # timesteps[init_step]
            latents = latents.to(self.device, torch.float32)
            if tuple(latents.shape) != shape:
                raise ValueError(f"latents must be {shape}, got {tuple(latents.shape)}")
            # DDIMScheduler.add_noise at t = timesteps[init_step] (leading spacing, steps_offset 1)
            t = (steps - 1 - init_step) * (1000 // steps) + 1
            a_t = ddim_alphas_cumprod()[t].item()
            lat = ((a_t ** 0.5) * latents + ((1.0 - a_t) ** 0.5) * noise).contiguous()
        self._sync_in()
        with torch.cuda.stream(self.stream):
            native.check(self.lib.er_dit_sample(self._ctx, native.ptr(cond), B, cond.shape[1], native.ptr(lat), steps,
                                                float(guidance_scale), init_step, self._sp()), "er_dit_sample")
        self._sync_out()
        return lat
