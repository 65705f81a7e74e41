"""``NativeShapeOPT`` - the MI355X-native stand-in for the reference's ``ShapeOPT``
(core/transformer/modeling_opt.py:429-550) at the seam ``LMM.generate`` uses:

    output_ids = self.mesh_decoder.generate(**kwargs)            core/models.py:303

It accepts the kwargs of core/models.py:286-301 (``inputs_embeds, num_tokens,
pad/bos/eos_token_id, max_new_tokens, prefix_allowed_tokens_fn, num_beams |
do_sample + top_k``) and returns the same ``LongTensor[B, T]`` of newly generated
ids.  All arithmetic runs in the HIP library behind the C ABI
(include/edgerunner_hip.h); this class only moves pointers.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional

import torch

from . import native
from .weights import ModelDims

_COND = {"none": native.ER_COND_NONE, "point": native.ER_COND_POINT, "point_latent": native.ER_COND_POINT_LATENT}
_DT = {torch.float32: native.ER_F32, torch.float16: native.ER_F16, torch.bfloat16: native.ER_BF16}


class BuiltinGrammar:
    """A ``prefix_allowed_tokens_fn`` the device understands.  Callable with the
    reference's signature (so host-side code can still use it) and tagged with the
    ``er_grammar`` enum the sampling-head kernel evaluates."""

    def __init__(self, er_grammar: int, vocab_size: int, eos_token_id: int = 2):
        from .grammar import as_callable
        self.er_grammar = er_grammar
        self._fn = as_callable(er_grammar, vocab_size, eos_token_id)

    def __call__(self, batch_id, input_ids):
        return self._fn(batch_id, input_ids) if self._fn is not None else None


class NativeShapeOPT:
    def __init__(self, dims: ModelDims, opt, device: torch.device, weight_dtype=torch.float32,
                 kv_dtype=torch.float32):
        if dims.cond_mode not in _COND:
            raise NotImplementedError(
                f"cond_mode={dims.cond_mode!r}: only the point / point_latent / none conditioners are on this path")
        self.dims, self.opt = dims, opt
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise native.NativeError("NativeShapeOPT needs a HIP device (cuda:N); there is no CPU fallback")
        self.lib = native.load_library()
        cfg = native.ErConfig(
            hidden_dim=dims.hidden_dim, num_heads=dims.num_heads, num_layers=dims.num_layers,
            intermediate_dim=dims.intermediate_dim, vocab_size=dims.vocab_size, max_positions=dims.max_positions,
            num_cond_tokens=dims.num_cond_tokens, point_hidden_dim=dims.point_hidden_dim,
            point_num_heads=dims.point_num_heads, point_latent_size=dims.point_latent_size,
            point_latent_dim=dims.point_latent_dim, point_freq_dim=dims.point_freq_dim,
            num_face_buckets=dims.num_face_buckets if dims.use_num_face_cond else 0,
            cond_mode=_COND[dims.cond_mode], pad_token_id=opt.pad_token_id, bos_token_id=opt.bos_token_id,
            eos_token_id=opt.eos_token_id, weight_dtype=_DT[weight_dtype], kv_dtype=_DT[kv_dtype], ln_eps=1e-5)
        self._ctx = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        native.check(self.lib.er_create(C.byref(cfg), idx, C.byref(self._ctx)), "er_create")
        self.stream = torch.cuda.Stream(device=self.device)
        self._reserved = (0, 0)
        self.last_decode_ms = 0.0
        self.direct_loads = False     # weights loaded on this object directly (LMM.half()/float() cannot replay them)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self.lib.er_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- checkpoint --------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """Same keys as the reference checkpoint.  Returns the
        (missing, unexpected) lists like ``nn.Module.load_state_dict``."""
        self.direct_loads = True
        unexpected = []
        for key, t in sd.items():
            if not isinstance(t, torch.Tensor):
                continue
            self._load_one(key, t, unexpected)
        return self._finish_load(strict, unexpected)

    def load_state_iter(self, items, strict: bool = False):
        """Streaming variant (one tensor resident at a time)."""
        self.direct_loads = True
        unexpected = []
        for key, t in items:
            self._load_one(key, t, unexpected)
        return self._finish_load(strict, unexpected)

    def _load_one(self, key, t, unexpected):
        if t.dtype not in _DT:
            t = t.float()
        t = t.detach().contiguous()
        shape = (C.c_int64 * max(1, t.dim()))(*(list(t.shape) or [1]))
        rc = native.check(self.lib.er_load_tensor(self._ctx, key.encode(), native.ptr(t), _DT[t.dtype],
                                                  max(1, t.dim()), shape, 1 if t.is_cuda else 0),
                          f"er_load_tensor({key})")
        if rc == 1:
            unexpected.append(key)

    def _finish_load(self, strict, unexpected):
        rc = self.lib.er_finalize_weights(self._ctx)
        missing = []
        if rc < 0:
            msg = self.lib.er_last_error().decode()
            if strict:
                raise native.NativeError(msg)
            missing.append(msg)
        if strict and unexpected:
            raise native.NativeError(f"unexpected keys: {unexpected[:5]}...")
        return missing, unexpected

    # -- plumbing ----------------------------------------------------------------------------
    def _enter(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self.stream)

    def _exit(self):
        # inputs may have been allocated on the caller's stream: finish before they can be freed
        self.stream.synchronize()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def _sp(self):
        return C.c_void_p(self.stream.cuda_stream)

    def reserve(self, batch: int, max_len: int):
        native.check(self.lib.er_kv_reserve(self._ctx, batch, max_len), "er_kv_reserve")
        self._reserved = (batch, max_len)

    # -- pieces of LMM.generate --------------------------------------------------------------
    def encode_cond(self, conds: Optional[torch.Tensor], face_buckets) -> torch.Tensor:
        d = self.dims
        B = len(face_buckets)
        n_points = 0
        if d.cond_mode != "none":
            conds = conds.to(self.device, torch.float32).contiguous()
            n_points = conds.shape[1]
        with self._enter():
            out = torch.empty((B, d.num_cond_tokens, d.hidden_dim), dtype=torch.float32, device=self.device)
            native.check(self.lib.er_encode_cond(self._ctx, native.ptr(conds if d.cond_mode != "none" else None), B,
                                                 n_points, native.i32_array(face_buckets), native.ptr(out), self._sp()),
                         "er_encode_cond")
        self._exit()
        return out

    def embd(self, input_ids: torch.Tensor) -> torch.Tensor:
        ids = input_ids.detach().to("cpu", torch.int32).contiguous()
        B, R = ids.shape
        with self._enter():
            out = torch.empty((B, R, self.dims.hidden_dim), dtype=torch.float32, device=self.device)
            native.check(self.lib.er_embed_tokens(self._ctx, native.i32_array(ids.flatten().tolist()), B, R,
                                                  native.ptr(out), self._sp()), "er_embed_tokens")
        self._exit()
        return out

    def prefill(self, inputs_embeds: torch.Tensor, max_new_tokens: int):
        x = inputs_embeds.to(self.device, torch.float32).contiguous()
        B, S, _ = x.shape
        need = S + max_new_tokens + 1
        if self._reserved[0] != B or self._reserved[1] < need:
            self.reserve(B, need)
        with self._enter():
            native.check(self.lib.er_prefill(self._ctx, native.ptr(x), B, S, self._sp()), "er_prefill")
        self._exit()

    def logits(self) -> torch.Tensor:
        with self._enter():
            out = torch.empty((self._reserved[0], self.dims.vocab_size), dtype=torch.float32, device=self.device)
            native.check(self.lib.er_logits(self._ctx, native.ptr(out), self._sp()), "er_logits")
        self._exit()
        return out

    def feed(self, ids):
        ids = [int(v) for v in (ids.flatten().tolist() if isinstance(ids, torch.Tensor) else ids)]
        with self._enter():
            native.check(self.lib.er_feed(self._ctx, native.i32_array(ids), self._sp()), "er_feed")
        self._exit()

    # -- the seam ----------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, inputs_embeds: torch.Tensor, num_tokens=None, pad_token_id=None, bos_token_id=None,
                 eos_token_id=None, max_new_tokens: Optional[int] = None,
                 prefix_allowed_tokens_fn: Optional[Callable] = None, num_beams: int = 1, do_sample: bool = False,
                 top_k: int = 50, min_new_tokens: int = 0, seed: Optional[int] = None, row_streams=None,
                 **unused) -> torch.Tensor:
        """Drop-in for ``ShapeOPT.generate`` as called at core/models.py:303.
        ``num_tokens`` is accepted and ignored exactly like the reference decoder
        ignores it (modeling_opt.py:324,467,546)."""
        if num_beams != 1:
            raise NotImplementedError("beam search is not part of the reference's call (num_beams=1)")
        for name, got, want in (("pad", pad_token_id, self.opt.pad_token_id), ("bos", bos_token_id, self.opt.bos_token_id),
                                ("eos", eos_token_id, self.opt.eos_token_id)):
            if got is not None and got != want:
                raise ValueError(f"{name}_token_id={got} differs from the context's {want}")
        if max_new_tokens is None:
            max_new_tokens = self.opt.max_seq_length
        B = inputs_embeds.shape[0]
        self.prefill(inputs_embeds, max_new_tokens)
        # sample mode: the Philox stream of row b (default b; a sharding / batching caller passes global job indices)
        if row_streams is not None and len(row_streams) != B:
            raise ValueError(f"row_streams has {len(row_streams)} entries for a batch of {B}")
        arr = None if row_streams is None else (C.c_uint32 * B)(*[int(v) & 0xFFFFFFFF for v in row_streams])
        native.check(self.lib.er_set_row_streams(self._ctx, arr, B), "er_set_row_streams")
        if prefix_allowed_tokens_fn is None or isinstance(prefix_allowed_tokens_fn, BuiltinGrammar):
            grammar = native.ER_GRAMMAR_NONE if prefix_allowed_tokens_fn is None else prefix_allowed_tokens_fn.er_grammar
            return self._decode_device(B, max_new_tokens, min_new_tokens, do_sample, top_k, grammar, seed)
        return self._decode_stepwise(B, max_new_tokens, min_new_tokens, do_sample, top_k, prefix_allowed_tokens_fn)

    def _decode_device(self, B, T, min_new, do_sample, top_k, grammar, seed):
        if seed is None:
            # the reference draws with torch.multinomial on the GLOBAL generator, which advances on every call:
            # repeated generate() calls give different samples and torch.manual_seed() reproduces a whole script.
            # Same contract here: the Philox key of this call is itself drawn from torch's global CPU generator.
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if do_sample else 0
        p = native.ErDecodeParams(mode=native.ER_SAMPLE if do_sample else native.ER_GREEDY, top_k=int(top_k),
                                  grammar=int(grammar), max_new_tokens=int(T), min_new_tokens=int(min_new),
                                  seed=int(seed) & 0xFFFFFFFFFFFFFFFF)
        n = C.c_int32(0)
        with self._enter():
            out = torch.empty((B, T), dtype=torch.int64, device=self.device)
            native.check(self.lib.er_decode(self._ctx, C.byref(p), native.ptr(out), C.byref(n), self._sp()), "er_decode")
        self._exit()
        ms = C.c_float(0)
        self.lib.er_last_decode_ms(self._ctx, C.byref(ms))
        self.last_decode_ms = float(ms.value)
        return out[:, : n.value]

    def _decode_stepwise(self, B, T, min_new, do_sample, top_k, fn):
        """Arbitrary host callable: logits come back every step (like the reference's
        host loop); the forward passes still run in the HIP library."""
        eos, pad = self.opt.eos_token_id, self.opt.pad_token_id
        ids = torch.empty((B, 0), dtype=torch.long)
        unfinished = torch.ones(B, dtype=torch.long)
        for t in range(T):
            s = self.logits().float().cpu()
            if t < min_new:
                s[:, eos] = -float("inf")
            mask = torch.full_like(s, -float("inf"))
            for b in range(B):
                allowed = fn(b, ids[b])
                if len(allowed) == 0:
                    raise ValueError(f"`prefix_allowed_tokens_fn` returned an empty list for batch ID {b}.")
                mask[b, allowed] = 0
            s = s + mask
            if do_sample:
                k = min(top_k, s.shape[-1])
                kth = torch.topk(s, k)[0][..., -1, None]
                s = s.masked_fill(s < kth, -float("inf"))
                nxt = torch.multinomial(torch.softmax(s, dim=-1), 1).squeeze(1)
            else:
                nxt = torch.argmax(s, dim=-1)
            nxt = nxt * unfinished + pad * (1 - unfinished)
            ids = torch.cat([ids, nxt[:, None]], dim=-1)
            unfinished = unfinished & (nxt != eos).long()
            if unfinished.max() == 0 or t == T - 1:
                break
            self.feed(nxt)
        return ids.to(self.device)

    def plan(self):
        """Kernel selection of THIS context for its reserved cache (``er_ctx_plan``): dict of the ``er_decode_plan`` fields."""
        p = native.ErDecodePlan()
        native.check(self.lib.er_ctx_plan(self._ctx, C.byref(p)), "er_ctx_plan")
        return {n: int(getattr(p, n)) for n, _ in native.ErDecodePlan._fields_}

    # -- measurement -------------------------------------------------------------------------
    def profile_decode_kernels(self, repeats: int = 5, context_len: int = 0, use_graph: bool = False):
        """Per-kind average launch duration (HIP events on the launch stream) and algorithmic bytes per launch.
        context_len > 0: the attention kernels run over that many keys (default: the current context length);
        use_graph: the 24 launches of a kind are replayed from a hipGraph, as the generation loop does."""
        us = (C.c_float * native.ER_NUM_KERNEL_KINDS)()
        by = (C.c_double * native.ER_NUM_KERNEL_KINDS)()
        with self._enter():
            native.check(self.lib.er_profile_decode_kernels_at(self._ctx, repeats, int(context_len), 1 if use_graph else 0, us, by,
                                                               self._sp()), "er_profile")
        self._exit()
        return {self.lib.er_kernel_kind_name(k).decode(): {"avg_us": float(us[k]), "bytes": float(by[k])}
                for k in range(native.ER_NUM_KERNEL_KINDS)}
