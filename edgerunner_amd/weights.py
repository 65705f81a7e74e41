"""Checkpoint contract of the ArAE decode path and a deterministic synthetic
checkpoint generator.

The reference ships no pretrained checkpoint (readme.md:12-13), so every parity
and benchmark run uses seeded synthetic weights.  The contract is the
``state_dict`` the reference's ``infer.py`` loads through ``--resume``
(infer.py:44-50): same key names, same shapes (verified
against ``LMM(opt).state_dict()`` of the reference in the build container).

Because ``/root/reference`` does not exist on the GPU box, the synthetic
checkpoint cannot be "``torch.manual_seed(0); LMM(opt)``" (that needs the
reference's module construction order).  Instead every tensor is drawn from its
own generator seeded by ``(seed, crc32(key))`` with the reference's *init
distributions* (core/transformer/modeling_opt.py:443-458 for the decoder,
torch's default ``nn.Linear`` init for the encoder, ``randn/sqrt(dim)`` for
``query_embed`` core/transformer/point.py:177).  The same dict is loaded into
the reference (build container, to make the goldens), the CPU oracle and the
HIP path.  ``style='perturbed'`` additionally gives biases and LayerNorm affine
parameters non-trivial values so that bias/affine handling is actually tested
(the reference init leaves them at 0 / 1).
"""
from __future__ import annotations

import dataclasses
import math
import zlib
from typing import Dict, Iterator, List, Tuple

import numpy as np
import torch


@dataclasses.dataclass(frozen=True)
class ModelDims:
    """Sizes derived from ``Options`` exactly as the reference derives them
    (core/models.py:78-95 vocab / intermediate / max positions)."""
    hidden_dim: int
    num_heads: int
    num_layers: int
    intermediate_dim: int
    vocab_size: int
    max_positions: int
    num_cond_tokens: int
    point_hidden_dim: int
    point_num_heads: int
    point_latent_size: int
    point_latent_dim: int
    point_freq_dim: int      # columns of point_embed.basis (24 = 48/2)
    num_face_buckets: int    # rows of embed_num_face (10)
    cond_mode: str
    use_num_face_cond: bool

    @property
    def head_dim(self) -> int:
        return self.hidden_dim // self.num_heads


def dims_from_options(opt) -> ModelDims:
    if opt.use_meto:
        if opt.meto_backend == "LR":
            vocab = 2 * opt.discrete_bins + 3 + 3
        elif opt.meto_backend == "LR_ABSCO":
            vocab = opt.discrete_bins + 3 + 3
        else:
            raise ValueError(f"unknown meto backend {opt.meto_backend}")
    else:
        vocab = opt.discrete_bins + 3
    inter = opt.hidden_dim * 4 if opt.intermediate_dim is None else opt.intermediate_dim
    return ModelDims(
        hidden_dim=opt.hidden_dim, num_heads=opt.num_heads, num_layers=opt.num_layers,
        intermediate_dim=inter, vocab_size=vocab,
        max_positions=opt.max_seq_length + opt.num_cond_tokens + 10,
        num_cond_tokens=opt.num_cond_tokens,
        point_hidden_dim=opt.point_hidden_dim, point_num_heads=opt.point_num_heads,
        point_latent_size=opt.point_latent_size, point_latent_dim=opt.point_latent_dim,
        point_freq_dim=24, num_face_buckets=10,
        cond_mode=opt.cond_mode, use_num_face_cond=bool(opt.use_num_face_cond),
    )


# kind -> how the tensor is initialised
#   normal:<std>      N(0, std)
#   linear_w:<fan_in> U(-1/sqrt(fan_in), 1/sqrt(fan_in))   (torch nn.Linear default)
#   bias / ln_w / ln_b / basis / embed_pad
def tensor_specs(d: ModelDims) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) for every tensor the decode path reads."""
    specs: List[Tuple[str, Tuple[int, ...], str]] = []
    H, PH = d.hidden_dim, d.point_hidden_dim

    def linear(prefix, out_f, in_f, kind_w, bias=True):
        specs.append((f"{prefix}.weight", (out_f, in_f), kind_w))
        if bias:
            specs.append((f"{prefix}.bias", (out_f,), f"bias:{in_f}"))

    def ln(prefix, n):
        specs.append((f"{prefix}.weight", (n,), "ln_w"))
        specs.append((f"{prefix}.bias", (n,), "ln_b"))

    if d.cond_mode == "point":
        pe = "point_encoder"
        specs.append((f"{pe}.query_embed", (1, d.point_latent_size, PH), f"normal:{1.0 / math.sqrt(PH)}"))
        specs.append((f"{pe}.point_embed.basis", (3, d.point_freq_dim), "basis"))
        fin = 2 * d.point_freq_dim + 3
        linear(f"{pe}.point_embed.mlp", PH, fin, f"linear_w:{fin}")
        ln(f"{pe}.ln", PH)
        ln(f"{pe}.cross_att.ln1", PH)
        for p in ("q_proj", "k_proj", "v_proj", "out_proj"):
            linear(f"{pe}.cross_att.att.{p}", PH, PH, f"linear_w:{PH}")
        ln(f"{pe}.cross_att.ln2", PH)
        linear(f"{pe}.cross_att.mlp.net.0", PH * 8, PH, f"linear_w:{PH}")
        linear(f"{pe}.cross_att.mlp.net.2", PH, PH * 4, f"linear_w:{PH * 4}")
        linear(f"{pe}.linear", d.point_latent_dim, PH, f"linear_w:{PH}")
    if d.cond_mode in ("point", "point_latent"):
        linear("proj_cond", H, d.point_latent_dim, f"linear_w:{d.point_latent_dim}")
        ln("norm_cond", H)
    if d.use_num_face_cond:
        specs.append(("embed_num_face.weight", (d.num_face_buckets, H), "normal:1.0"))

    dec = "mesh_decoder.model"
    specs.append((f"{dec}.embd.weight", (d.vocab_size, H), "embed_pad:0.02"))
    specs.append((f"{dec}.embed_positions.weight", (d.max_positions, H), "normal:0.02"))
    out_std = 0.02 / math.sqrt(2 * d.num_layers)
    for i in range(d.num_layers):
        L = f"{dec}.layers.{i}"
        linear(f"{L}.self_attn.k_proj", H, H, "normal:0.02")
        linear(f"{L}.self_attn.v_proj", H, H, "normal:0.02")
        linear(f"{L}.self_attn.q_proj", H, H, "normal:0.02")
        linear(f"{L}.self_attn.out_proj", H, H, f"normal:{out_std}")
        ln(f"{L}.self_attn_layer_norm", H)
        linear(f"{L}.fc1", d.intermediate_dim, H, "normal:0.02")
        linear(f"{L}.fc2", H, d.intermediate_dim, "normal:0.02")
        ln(f"{L}.final_layer_norm", H)
    specs.append(("mesh_decoder.lm_head.weight", (d.vocab_size, H), "normal:0.02"))
    return specs


def point_basis(freq_dim: int = 24) -> torch.Tensor:
    """The fixed Fourier basis buffer (core/transformer/point.py:44-50):
    row a holds 2^k*pi (k = 0..freq_dim/3-1) in columns [a*n, (a+1)*n)."""
    n = freq_dim // 3
    e = torch.pow(2, torch.arange(n)).float() * np.pi
    z = torch.zeros(n)
    return torch.stack([torch.cat([e, z, z]), torch.cat([z, e, z]), torch.cat([z, z, e])])


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFFFFFFFFFF)
    return g


def make_tensor(key: str, shape, kind: str, seed: int, style: str) -> torch.Tensor:
    g = _gen(seed, key)
    name, _, arg = kind.partition(":")
    if name == "normal":
        return torch.randn(shape, generator=g, dtype=torch.float32) * float(arg)
    if name == "embed_pad":  # nn.Embedding(padding_idx=0): N(0,std), pad row zeroed (modeling_opt.py:455-458)
        w = torch.randn(shape, generator=g, dtype=torch.float32) * float(arg)
        w[0].zero_()
        return w
    if name == "linear_w":
        bound = 1.0 / math.sqrt(float(arg))
        return (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
    if name == "basis":
        return point_basis(shape[1])
    if name == "bias":
        if style == "reference":
            # decoder biases are zeroed by _init_weights; encoder biases keep torch's default U(+-1/sqrt(fan_in))
            if key.startswith("mesh_decoder"):
                return torch.zeros(shape, dtype=torch.float32)
            bound = 1.0 / math.sqrt(float(arg))
            return (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        return torch.randn(shape, generator=g, dtype=torch.float32) * 0.02
    if name == "ln_w":
        if style == "reference":
            return torch.ones(shape, dtype=torch.float32)
        return 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    if name == "ln_b":
        if style == "reference":
            return torch.zeros(shape, dtype=torch.float32)
        return 0.05 * torch.randn(shape, generator=g, dtype=torch.float32)
    raise ValueError(kind)


def iter_state_dict(opt, seed: int = 0, style: str = "perturbed") -> Iterator[Tuple[str, torch.Tensor]]:
    """Yield (key, fp32 CPU tensor) one at a time (peak memory = one tensor)."""
    assert style in ("perturbed", "reference")
    d = dims_from_options(opt)
    for key, shape, kind in tensor_specs(d):
        yield key, make_tensor(key, shape, kind, seed, style)


def make_state_dict(opt, seed: int = 0, style: str = "perturbed") -> Dict[str, torch.Tensor]:
    return dict(iter_state_dict(opt, seed, style))


def fingerprint(t: torch.Tensor) -> Tuple[float, float]:
    """Cheap content check used to pin the generator across machines."""
    t64 = t.double().flatten()
    return float(t64.sum()), float(t64[:: max(1, t64.numel() // 97)].abs().sum())


def synthetic_point_cloud(index: int, num_points: int = 4096) -> torch.Tensor:
    """Benchmark/parity point cloud *index* (SURVEY.md section 8d): uniform in the
    ``normalize_mesh(bound=0.95)`` cube (reference infer.py:88), [1, N, 3] fp32."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + int(index))
    return (torch.rand(num_points, 3, generator=g, dtype=torch.float32) * 1.9 - 0.95).unsqueeze(0)


def synthetic_resume_ids(seed: int, n: int):
    """n token ids that obey the LR_ABSCO layout (core/models.py:246-268: BOM + 9 coordinates, then (L | R) + 3 coordinates ...),
    as int64 numpy: the resumed prefix of the long-context parity cases (``resume_ids``, core/models.py:225-226)."""
    import numpy as np
    g = np.random.default_rng(seed)
    out = [5] + [int(v) for v in g.integers(6, 518, 9)]
    while len(out) < n:
        out += [int(g.integers(3, 5))] + [int(v) for v in g.integers(6, 518, 3)]
    return np.asarray(out[:n], dtype=np.int64)


# ------------------------------------------------------------------------------------ DiT front-end (f3)
def dit_tensor_specs(opt, clip_dim: int = 1280) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) of the MDiT checkpoint entries the denoiser reads (core/models_dit.py:43-63 /
    core/transformer/dit.py:143-166); init kinds follow the reference constructors (torch default Linear
    init, randn/sqrt(dim) tables)."""
    C, L, N = opt.dit_hidden_dim, opt.point_latent_dim, opt.point_latent_size
    specs: List[Tuple[str, Tuple[int, ...], str]] = []

    def linear(prefix, out_f, in_f):
        specs.append((f"{prefix}.weight", (out_f, in_f), f"linear_w:{in_f}"))
        specs.append((f"{prefix}.bias", (out_f,), f"bias:{in_f}"))

    specs.append(("dit.pos_embed", (1, N, C), f"normal:{1.0 / math.sqrt(C)}"))
    specs.append(("dit.scale_shift_table", (2, C), f"normal:{1.0 / math.sqrt(C)}"))
    linear("dit.proj_in", C, L)
    linear("dit.timestep_proj.linear_1", C, 256)
    linear("dit.timestep_proj.linear_2", C, C)
    linear("dit.adaln_linear", 6 * C, C)
    for i in range(opt.dit_num_layers):
        p = f"dit.layers.{i}"
        specs.append((f"{p}.scale_shift_table", (6, C), f"normal:{1.0 / math.sqrt(C)}"))
        linear(f"{p}.attn1.qkv_proj", 3 * C, C)
        linear(f"{p}.attn1.out_proj", C, C)
        for q in ("q_proj", "k_proj", "v_proj", "out_proj"):
            linear(f"{p}.attn2.{q}", C, C)
        linear(f"{p}.ff.net.0", 8 * C, C)
        linear(f"{p}.ff.net.2", C, 4 * C)
    linear("dit.proj_out", L, C)
    linear("proj_cond", C, clip_dim)
    specs.append(("norm_cond.weight", (C,), "ln_w"))
    specs.append(("norm_cond.bias", (C,), "ln_b"))
    return specs


def make_dit_state_dict(opt, seed: int = 0, style: str = "perturbed") -> Dict[str, torch.Tensor]:
    # encoder-style biases (torch default) in both styles: keys do not start with "mesh_decoder"
    return {k: make_tensor("mdit." + k, shape, kind, seed, style) for k, shape, kind in dit_tensor_specs(opt)}


def clip_tensor_specs(num_layers: int = 32, width: int = 1280, mlp: int = 5120, patch: int = 14, tokens: int = 257):
    """image_encoder.* entries (HF CLIPVisionModel, transformers 4.46.2 key names).  Synthetic init only - the
    pretrained laion/CLIP-ViT-H-14 weights are not available offline."""
    p = "image_encoder.vision_model"
    specs = [(f"{p}.embeddings.class_embedding", (width,), "normal:0.03"),
             (f"{p}.embeddings.patch_embedding.weight", (width, 3, patch, patch), "normal:0.02"),
             (f"{p}.embeddings.position_embedding.weight", (tokens, width), "normal:0.02"),
             (f"{p}.pre_layrnorm.weight", (width,), "ln_w"), (f"{p}.pre_layrnorm.bias", (width,), "ln_b")]
    for i in range(num_layers):
        L = f"{p}.encoder.layers.{i}"
        for q in ("k_proj", "v_proj", "q_proj", "out_proj"):
            specs += [(f"{L}.self_attn.{q}.weight", (width, width), "normal:0.02"), (f"{L}.self_attn.{q}.bias", (width,), f"bias:{width}")]
        specs += [(f"{L}.layer_norm1.weight", (width,), "ln_w"), (f"{L}.layer_norm1.bias", (width,), "ln_b"),
                  (f"{L}.mlp.fc1.weight", (mlp, width), "normal:0.02"), (f"{L}.mlp.fc1.bias", (mlp,), f"bias:{width}"),
                  (f"{L}.mlp.fc2.weight", (width, mlp), "normal:0.02"), (f"{L}.mlp.fc2.bias", (width,), f"bias:{mlp}"),
                  (f"{L}.layer_norm2.weight", (width,), "ln_w"), (f"{L}.layer_norm2.bias", (width,), "ln_b")]
    return specs


def make_clip_state_dict(num_layers: int = 32, seed: int = 0, style: str = "perturbed") -> Dict[str, torch.Tensor]:
    return {k: make_tensor("clip." + k, shape, kind, seed, style) for k, shape, kind in clip_tensor_specs(num_layers)}
