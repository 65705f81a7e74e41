"""Configuration surface of the ArAE decode path.

Keeps the reference's ``Options`` field names, defaults and the
``default`` / ``ArAE`` / ``DiT`` presets name-for-name so that command lines
and config objects written for the reference work unchanged
(reference: core/options.py:17-148 fields, :152-210 presets, :212 CLI type).

The reference builds its CLI with ``tyro`` (absent in this image); here the
same surface - ``python infer.py {default,ArAE,DiT} --flag value ...`` - is
produced from the field table with ``argparse`` (see :func:`parse_cli`).
"""
from __future__ import annotations

import argparse
import dataclasses
from typing import Dict, Optional, Sequence, Tuple

# (name, type, default, choices-or-None, help).  Order follows the reference's
# sections: tokenizer / point vae / dit / lmm / dataset / training / testing.
_FIELDS = [
    # --- tokenizer
    ("discrete_bins", int, 512, None, "coordinate quantisation bins (= number of coordinate tokens)"),
    ("use_meto", bool, True, None, "mesh tokens come from the meto tokenizer"),
    ("meto_backend", str, "LR_ABSCO", ("LR", "LR_ABSCO"), "meto engine variant"),
    ("bos_token_id", int, 1, None, "BOS id"),
    ("eos_token_id", int, 2, None, "EOS id"),
    ("pad_token_id", int, 0, None, "PAD id"),
    # --- point vae
    ("point_num", int, 8192, None, "points sampled per cloud"),
    ("point_hidden_dim", int, 1024, None, "point encoder width"),
    ("point_num_heads", int, 16, None, "point encoder heads"),
    ("point_latent_size", int, 2048, None, "number of latent (query) tokens"),
    ("point_latent_dim", int, 64, None, "latent channel count"),
    ("point_num_layers", int, 24, None, "(training) point decoder layers"),
    ("point_query_num", int, 81920, None, "(training) query points per iteration"),
    ("point_encoder_mode", str, "embed", ("downsample", "embed"), "point encoder variant"),
    ("kl_weight", float, 1e-8, None, "(training) latent penalty weight"),
    # --- dit
    ("dit_hidden_dim", int, 1024, None, "DiT width"),
    ("dit_num_heads", int, 16, None, "DiT heads"),
    ("dit_num_layers", int, 24, None, "DiT layers"),
    ("snr_gamma", Optional[float], 5.0, None, "(training) min-SNR gamma"),
    ("noise_scheduler_predtype", str, "v_prediction", ("epsilon", "v_prediction"), "diffusion target"),
    # --- lmm
    ("freeze_encoder", bool, True, None, "(training) freeze the conditioner"),
    ("max_seq_length", int, 10240, None, "max generated tokens (excl. BOS/EOS/COND)"),
    ("hidden_dim", int, 1024, None, "decoder width"),
    ("intermediate_dim", Optional[int], None, None, "decoder MLP width (default 4*hidden)"),
    ("num_layers", int, 24, None, "decoder layers"),
    ("num_heads", int, 16, None, "decoder heads"),
    ("cond_mode", str, "image", ("none", "image", "point", "point_latent"), "conditioning kind"),
    ("num_cond_tokens", int, 257, None, "length of the conditioning prefix"),
    ("generate_mode", str, "sample", ("greedy", "sample"), "decoding rule"),
    ("use_num_face_cond", bool, False, None, "append the face-count bucket token"),
    ("nof_dropout_ratio", float, 0.2, None, "(training) face-count dropout"),
    # --- dataset
    ("max_face_length", int, 1000, None, "(training) max faces"),
    ("dataset", str, "obj", ("obj", "objxl"), "(training) dataset"),
    ("num_workers", int, 64, None, "(training) loader workers"),
    ("testset_size", int, 32, None, "(training) test split size"),
    ("use_decimate_aug", bool, True, None, "(training) decimation augmentation"),
    ("use_scale_aug", bool, True, None, "(training) scale augmentation"),
    # --- training
    ("workspace", str, "./workspace", None, "output directory"),
    ("resume", Optional[str], None, None, "checkpoint (.safetensors or torch) to load"),
    ("resume2", Optional[str], None, None, "second checkpoint (DiT)"),
    ("resume_step_ratio", float, 0.0, None, "(training)"),
    ("align_posemb", str, "right", ("left", "right"), "(training) pos-emb alignment on resume"),
    ("batch_size", int, 4, None, "(training) per-GPU batch"),
    ("gradient_accumulation_steps", int, 1, None, "(training)"),
    ("num_epochs", int, 100, None, "(training)"),
    ("gradient_clip", float, 1.0, None, "(training)"),
    ("mixed_precision", str, "bf16", ("no", "fp8", "fp16", "fp32", "bf16"), "(training)"),
    ("lr", float, 1e-4, None, "(training)"),
    ("checkpointing", bool, True, None, "(training) gradient checkpointing"),
    ("seed", int, 0, None, "random seed"),
    ("eval_mode", str, "loss", ("none", "loss", "generate"), "(training)"),
    ("debug_eval", bool, False, None, "(training)"),
    ("warmup_ratio", float, 0.01, None, "(training)"),
    ("use_wandb", bool, False, None, "(training)"),
    # --- testing
    ("test_path", Optional[str], None, None, "input file or directory"),
    ("test_resume_tokens", Optional[str], None, None, "tokens to resume from"),
    ("test_repeat", int, 1, None, "generations per input"),
    ("test_num_face", Tuple[int, ...], (1000,), None, "target face counts (list)"),
    ("test_max_seq_length", Optional[int], None, None, "max_new_tokens override"),
]

Options = dataclasses.make_dataclass(
    "Options",
    [(n, t, dataclasses.field(default=d)) for (n, t, d, _c, _h) in _FIELDS],
)
Options.__doc__ = "Drop-in for the reference's ``core.options.Options`` (same fields and defaults)."

_ARAE = dict(
    point_encoder_mode="embed", kl_weight=1e-8, discrete_bins=512, use_num_face_cond=True,
    use_decimate_aug=True, cond_mode="point", num_cond_tokens=2049, freeze_encoder=False,
    use_meto=True, meto_backend="LR_ABSCO", max_face_length=4000, max_seq_length=40960,
    align_posemb="right", batch_size=4, hidden_dim=1536, num_heads=16, num_layers=24,
    gradient_accumulation_steps=1, lr=1e-5, warmup_ratio=0, num_epochs=100, eval_mode="loss",
)
_DIT = dict(
    point_encoder_mode="embed", kl_weight=1e-8, max_face_length=8000, discrete_bins=512,
    use_num_face_cond=True, use_decimate_aug=False, cond_mode="point", num_cond_tokens=2049,
    freeze_encoder=False, use_meto=True, meto_backend="LR_ABSCO", max_seq_length=40960,
    hidden_dim=1536, num_heads=16, num_layers=24, dit_hidden_dim=1024, dit_num_heads=16,
    dit_num_layers=24, snr_gamma=5.0, noise_scheduler_predtype="v_prediction", batch_size=8,
    gradient_accumulation_steps=1, lr=1e-5, num_epochs=300, eval_mode="none",
)

config_doc: Dict[str, str] = {"default": "the default settings", "ArAE": "ArAE", "DiT": "DiT"}
config_defaults: Dict[str, "Options"] = {
    "default": Options(),
    "ArAE": Options(**_ARAE),
    "DiT": Options(**_DIT),
}
AllConfigs = tuple(config_defaults)  # the subcommand names; see parse_cli


def _str2bool(s: str) -> bool:
    if s.lower() in ("1", "true", "yes", "y", "on"):
        return True
    if s.lower() in ("0", "false", "no", "n", "off"):
        return False
    raise argparse.ArgumentTypeError(f"expected a boolean, got {s!r}")


def _opt(cast):
    def f(s):
        return None if s in ("None", "none", "") else cast(s)
    return f


def build_parser() -> argparse.ArgumentParser:
    """``prog {default,ArAE,DiT} --field value`` - same shape as the tyro CLI.

    Booleans accept both the tyro spelling (``--use-meto`` / ``--no-use-meto``)
    and ``--use_meto True``; underscores and dashes are interchangeable.
    """
    top = argparse.ArgumentParser(description="EdgeRunner ArAE decode (MI355X-native)")
    sub = top.add_subparsers(dest="_preset", required=True)
    for preset, base in config_defaults.items():
        p = sub.add_parser(preset, help=config_doc[preset])
        for (name, typ, _d, choices, help_) in _FIELDS:
            default = getattr(base, name)
            flags = [f"--{name}"] + ([f"--{name.replace('_', '-')}"] if "_" in name else [])
            if typ is bool:
                p.add_argument(*flags, dest=name, nargs="?", const=True, default=default,
                               type=_str2bool, help=help_)
                neg = [f"--no-{name}", f"--no-{name.replace('_', '-')}"] if "_" in name else [f"--no-{name}"]
                p.add_argument(*neg, dest=name, action="store_false", help=argparse.SUPPRESS)
            elif typ is Tuple[int, ...]:
                p.add_argument(*flags, dest=name, nargs="+", type=int, default=default, help=help_)
            elif typ in (Optional[int], Optional[float], Optional[str]):
                cast = {Optional[int]: int, Optional[float]: float, Optional[str]: str}[typ]
                p.add_argument(*flags, dest=name, type=_opt(cast), default=default, help=help_)
            else:
                p.add_argument(*flags, dest=name, type=typ, default=default, choices=choices, help=help_)
    return top


def parse_cli(argv: Optional[Sequence[str]] = None) -> "Options":
    """Replacement for ``tyro.cli(AllConfigs)`` (reference: infer.py:36)."""
    ns = vars(build_parser().parse_args(argv))
    ns.pop("_preset")
    ns["test_num_face"] = tuple(ns["test_num_face"])
    return Options(**ns)
