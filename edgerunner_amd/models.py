"""``LMM`` - drop-in for the reference's ``core.models.LMM`` on the ArAE decode path
(reference: core/models.py:32-99 constructor, :101-144 encode_cond, :204-319 generate).

Same constructor argument (``Options``), same checkpoint keys, same
``generate(conds, num_faces, resume_ids, tokenizer, max_new_tokens, clean)``
signature and return value ``(meshes, all_tokens)``.  Differences, all additive:
``B > 1`` is allowed (independent rows, per-row grammar state), ``min_new_tokens``
can be passed (benchmark rule: EOS suppressed until T), and training-only members
(``forward``, image conditioner) are not part of this path.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import native
from .grammar import select_grammar
from .shape_opt import BuiltinGrammar, NativeShapeOPT
from .utils import quantize_num_faces
from .weights import dims_from_options


class _Embd:
    """``mesh_decoder.model.embd`` lookalike (core/models.py:228)."""

    def __init__(self, dec: NativeShapeOPT):
        self._dec = dec

    def __call__(self, input_ids):
        return self._dec.embd(input_ids)


class _DecoderModel:
    def __init__(self, dec):
        self.embd = _Embd(dec)


class LMM:
    def __init__(self, opt, device="cuda:0", precision: Optional[str] = "fp32"):
        """precision: 'fp32' = exact mode (fp32 weights + KV; greedy ids bit-exact vs the CPU path), 'fp16' = fast mode
        (decoder matrices and KV cache stored in fp16 like the reference's ``model.half()`` GPU path, fp32 accumulate),
        or None = module style: fp32 until ``.half()`` is called, the native context being created on first use, so
        that the reference's ``LMM(opt)`` -> ``load_state_dict`` -> ``.half().eval().to(device)`` (infer.py:41-56)
        selects the fp16 context exactly as it selects fp16 storage there."""
        self.opt = opt
        if precision not in (None, "fp32", "fp16"):
            raise ValueError(precision)
        if opt.cond_mode == "image":
            raise NotImplementedError("cond_mode='image' (CLIP conditioner) is outside the ArAE decode path")
        if opt.cond_mode == "point" and opt.point_encoder_mode != "embed":
            raise NotImplementedError("point_encoder_mode='downsample' needs torch_cluster FPS; ArAE uses 'embed'")
        self.dims = dims_from_options(opt)
        if (self.dims.hidden_dim, self.dims.intermediate_dim) != (1536, 6144) or self.dims.hidden_dim // max(self.dims.num_heads, 1) not in (96, 64):
            # the decode kernels stream 1536-wide rows in whole 1536-element slices (csrc/k_gemv.h); the reference's generic
            # ShapeOPTConfig (core/transformer/modeling_opt.py:86-134; Options() default hidden_dim 1024) is not built - say so
            # here, with the option names, instead of from er_create
            raise NotImplementedError(
                f"decoder shape hidden_dim={self.dims.hidden_dim} / intermediate_dim={self.dims.intermediate_dim} / "
                f"num_heads={self.dims.num_heads} is not built: this library serves the ArAE / DiT presets' decoder "
                "(hidden_dim=1536, intermediate_dim=6144, head_dim 96 or 64). Use config_defaults['ArAE'] (or 'DiT'), "
                "or pass --hidden_dim 1536 --num_heads 16 --intermediate_dim 6144.")
        self.vocab_size = self.dims.vocab_size
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise native.NativeError("this LMM only runs on a HIP device (cuda:N); there is no CPU fallback")
        self._dtype = torch.float16 if precision == "fp16" else torch.float32
        self._dec: Optional[NativeShapeOPT] = None
        self._sources: List[tuple] = []        # (state_dict reference, strict) of every load_state_dict call, for re-creation
        self.training = False
        self._released = False                 # release_checkpoint() was called: the context cannot be re-created
        if precision is not None:
            self._materialize()

    # -- native context ------------------------------------------------------------------------
    @property
    def precision(self) -> str:
        return "fp16" if self._dtype == torch.float16 else "fp32"

    def _materialize(self):
        if self._released:
            raise native.NativeError("LMM: the checkpoint was released (release_checkpoint()) and the native context is gone; "
                                     "create a new LMM and load the checkpoint again")
        dec = NativeShapeOPT(self.dims, self.opt, self.device, weight_dtype=self._dtype, kv_dtype=self._dtype)
        dec.model = _DecoderModel(dec)
        for sd, strict in self._sources:
            dec.load_state_dict(sd, strict=strict)
        dec.direct_loads = False               # everything loaded so far is replayable from self._sources
        self._dec = dec
        return dec

    @property
    def mesh_decoder(self) -> NativeShapeOPT:
        """The native decoder context (created on first access in module style)."""
        return self._dec if self._dec is not None else self._materialize()

    # -- nn.Module-shaped conveniences so infer.py reads like the reference's ----------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """Returns (missing, unexpected) like nn.Module.load_state_dict.  The dict is kept by reference (no copy) so a
        later ``.half()`` / ``.float()`` can rebuild the native context in the other storage precision."""
        self._sources.append((sd, strict))
        if self._dec is not None:
            prev = self._dec.direct_loads
            out = self._dec.load_state_dict(sd, strict=strict)
            self._dec.direct_loads = prev
            return out
        from .weights import tensor_specs
        want = {k for k, _, _ in tensor_specs(self.dims)}
        have = {k for k, t in sd.items() if isinstance(t, torch.Tensor)}
        missing, unexpected = sorted(want - have), sorted(have - want)
        if strict and (missing or unexpected):
            raise native.NativeError(f"load_state_dict(strict=True): missing {missing[:5]}, unexpected {unexpected[:5]}")
        return missing, unexpected

    def release_checkpoint(self):
        """Drop the retained state_dict references (a full ArAE checkpoint is ~2.7 GB of host memory per rank) once the native
        context holds the weights.  ``.half()`` / ``.float()`` can no longer re-store them afterwards and fail loudly."""
        _ = self.mesh_decoder                  # make sure everything retained so far is on the device
        self._sources.clear()
        self._dec.direct_loads = True
        self._released = True
        return self

    def _cast(self, dtype):
        if dtype == self._dtype:
            return self
        if self._dec is not None and self._dec.direct_loads:
            raise native.NativeError(
                f"LMM.{'half' if dtype == torch.float16 else 'float'}(): this context's weights were streamed directly into "
                f"the {self.precision} native context (mesh_decoder.load_state_iter / load_state_dict); they cannot be "
                f"re-stored. Create LMM(opt, device, precision='{'fp16' if dtype == torch.float16 else 'fp32'}') instead.")
        self._dtype = dtype
        if self._dec is not None:            # rebuild lazily from the retained checkpoints in the new storage precision
            self._dec.close()
            self._dec = None
        return self

    def half(self):
        """The reference casts to fp16 here (infer.py:56): selects the fp16-storage context (fp32 accumulate)."""
        return self._cast(torch.float16)

    def float(self):
        return self._cast(torch.float32)

    def eval(self):
        return self

    def to(self, device):
        d = torch.device(device)
        if d.type != "cuda":
            raise native.NativeError("this LMM only runs on a HIP device")
        if d.index is not None and self.device.index is not None and d.index != self.device.index:
            if self._dec is not None and self._dec.direct_loads:
                raise native.NativeError("LMM.to(): weights were streamed into the context of another device")
            if self._dec is not None:
                self._dec.close()
                self._dec = None
            self.device = d
        return self

    # -- conditioning -------------------------------------------------------------------------
    @torch.no_grad()
    def encode_cond(self, conds, num_faces):
        """core/models.py:101-144 (eval mode).  ``num_faces``: LongTensor[B] or list."""
        nf = num_faces.tolist() if isinstance(num_faces, torch.Tensor) else list(num_faces)
        buckets = [quantize_num_faces(int(n)) for n in nf]
        return {"cond_embeds": self.mesh_decoder.encode_cond(conds, buckets)}

    # -- generation ---------------------------------------------------------------------------
    @torch.no_grad()
    def generate_ids(self, conds, num_faces=1000, resume_ids=None, tokenizer=None, max_new_tokens=None,
                     min_new_tokens: int = 0, seed: Optional[int] = None, row_streams=None) -> torch.Tensor:
        """Everything of LMM.generate up to the HF call's return value (core/models.py:215-303)."""
        opt = self.opt
        B = conds.shape[0]
        cond_embeds = self.encode_cond(conds, [num_faces] * B)["cond_embeds"]
        input_ids = torch.full((B, 1), opt.bos_token_id, dtype=torch.long)
        if resume_ids is not None:
            input_ids = torch.cat((input_ids, resume_ids.to("cpu", torch.long)), dim=1)
        tokens_embeds = self.mesh_decoder.model.embd(input_ids)
        inputs_embeds = torch.cat((cond_embeds, tokens_embeds), dim=1) if cond_embeds is not None else tokens_embeds
        fn = BuiltinGrammar(select_grammar(opt, tokenizer is not None), self.vocab_size, opt.eos_token_id)
        if fn.er_grammar == native.ER_GRAMMAR_NONE:
            print("[WARN] prefix_allowed_tokens_fn is not defined for meto backend:", opt.meto_backend)
            fn = None
        max_new_tokens = opt.max_seq_length if max_new_tokens is None else max_new_tokens
        if num_faces < 0:
            num_tokens = torch.full((B,), -1, dtype=torch.long)
        else:
            num_tokens = torch.full((B,), num_faces * 4 + opt.num_cond_tokens, dtype=torch.long)
        kwargs = dict(inputs_embeds=inputs_embeds, num_tokens=num_tokens, pad_token_id=opt.pad_token_id,
                      bos_token_id=opt.bos_token_id, eos_token_id=opt.eos_token_id, max_new_tokens=max_new_tokens,
                      prefix_allowed_tokens_fn=fn, min_new_tokens=min_new_tokens, seed=seed, row_streams=row_streams)
        if opt.generate_mode == "greedy":
            kwargs["num_beams"] = 1
        elif opt.generate_mode == "sample":
            kwargs["do_sample"] = True
            kwargs["top_k"] = 10
        return self.mesh_decoder.generate(**kwargs)

    @torch.no_grad()
    def generate(self, conds, num_faces=1000, resume_ids=None, tokenizer=None, max_new_tokens=None, clean=True,
                 min_new_tokens: int = 0, seed: Optional[int] = None, row_streams=None):
        """-> (meshes, all_tokens) like core/models.py:204-319.  ``tokenizer`` is a
        ``edgerunner_amd.meto.Engine`` (or None for the 9-coordinate layout); each mesh is a
        ``edgerunner_amd.meto.Mesh`` - ``.vertices``, ``.faces``, ``.export(path)``, the members the reference's callers
        use of the trimesh objects it returns (trimesh itself is absent here); it still unpacks as ``v, f = mesh``."""
        output_ids = self.generate_ids(conds, num_faces, resume_ids, tokenizer, max_new_tokens, min_new_tokens, seed, row_streams)
        from .meto import Engine, save_mesh
        meshes: List[Optional[object]] = []
        all_tokens: List[np.ndarray] = []
        out = output_ids.detach().cpu().numpy()
        for b in range(out.shape[0]):
            tokens = out[b]
            if resume_ids is not None:
                tokens = np.concatenate((resume_ids[b].detach().cpu().numpy(), tokens), axis=0)
            # batch detokenize (core/models.py:315): a meto.Mesh instead of a trimesh object
            if tokenizer is None or isinstance(tokenizer, Engine):
                meshes.append(save_mesh(tokens, self.opt, tokenizer=tokenizer, clean=clean))
            else:
                meshes.append(None)      # opaque tokenizer marker (tests / benches): ids only
            all_tokens.append(tokens)
        return meshes, all_tokens
