"""ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by the product path
(``edgerunner_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

CPU (torch, fp32) restatement of the reference's ArAE decode path - the
"CPU/eager path" BASELINE.json names - written against the checkpoint
``state_dict`` instead of ``nn.Module`` objects so it needs nothing from
``/root/reference`` at run time.  It deliberately issues the SAME torch ops in
the SAME order as the reference modules (F.linear, bmm, softmax, layer_norm ...),
so on one machine its results are bit-identical to the reference's own modules;
``oracle/make_golden.py`` asserts exactly that (``torch.equal``) in the build
container, where the reference is importable.

Parity status: PINNED against the reference's own modules executed in the
build container (see tests/golden/MANIFEST.json: ``restatement_bit_identical``).
The generation loop itself lives in a third-party dependency that is NOT under
/root/reference - ``transformers==4.46.2`` (requirements.lock.txt:16),
``generation/utils.py::GenerationMixin.generate/_sample`` and
``generation/logits_process.py::{PrefixConstrainedLogitsProcessor,
TopKLogitsWarper,MinNewTokensLengthLogitsProcessor}``; the installed 5.15
cannot drive the reference's tuple KV cache (TypeError at modeling_opt.py:524).
Its published algorithm is restated in :func:`generate` and anchored on the
reference's call site (core/models.py:286-303) and patch (core/utils.py:118-141).
The reference holds no golden vectors for this path; the
restated loop is instead checked id for id (greedy and, under a shared
``torch.manual_seed``, sample mode; grammar closure, ``min_new_tokens``,
pad-after-EOS, early stop) against the INSTALLED transformers'
``GenerationMixin.generate`` driving a toy LM with the reference's
``prepare_inputs_for_generation`` protocol: tests/test_hf_loop_pin.py.

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]
LN_EPS = 1e-5  # nn.LayerNorm default, used everywhere in the reference


# ----------------------------------------------------------------------------- attention
def attention_naive(q, k, v, causal=False):
    """core/transformer/attention.py:27-62, the branch taken without flash_attn.
    q [B,N,H,D], k/v [B,M,H,D] -> [B,N,H,D]."""
    B, N, H, D = q.shape
    M = k.shape[1]
    if causal:
        assert N == 1 or N == M  # attention.py:40-41
    q = q.transpose(1, 2).reshape(B * H, N, D)
    k = k.transpose(1, 2).reshape(B * H, M, D)
    v = v.transpose(1, 2).reshape(B * H, M, D)
    if HEAD_CHUNK and N > 1 and B * H > HEAD_CHUNK:
        # memory knob of the golden generators only (contexts of 18 k keys: 16 x 18050^2 fp32 scores do not fit the build
        # container): the same per-(row, head) arithmetic, HEAD_CHUNK matrices at a time; every matrix of the batched
        # product is independent of its neighbours
        out = torch.cat([_attend(q[i:i + HEAD_CHUNK], k[i:i + HEAD_CHUNK], v[i:i + HEAD_CHUNK], causal, N, M, D)
                         for i in range(0, B * H, HEAD_CHUNK)])
    else:
        out = _attend(q, k, v, causal, N, M, D)
    return out.reshape(B, H, N, D).transpose(1, 2).contiguous()  # :61


HEAD_CHUNK = 0


def _attend(q, k, v, causal, N, M, D):
    w = torch.bmm(q, k.transpose(1, 2)) / (D ** 0.5)           # :52
    if causal and N > 1:                                        # :53-56
        mask = torch.full((N, M), float("-inf"), device=w.device, dtype=w.dtype)
        mask = torch.triu(mask, diagonal=1)
        w = w + mask.unsqueeze(0)
    w = F.softmax(w, dim=-1)                                    # :57
    return torch.bmm(w, v)                                      # :60


# ----------------------------------------------------------------------------- point encoder
def point_embed(sd: StateDict, x, prefix="point_encoder.point_embed"):
    """core/transformer/point.py:53-65 (PointEmbed.embed + forward)."""
    basis = sd[f"{prefix}.basis"]
    proj = torch.einsum("bnd,de->bne", x, basis.to(x.dtype))
    emb = torch.cat([proj.sin(), proj.cos()], dim=2)
    emb = torch.cat([emb, x], dim=2).to(x.dtype)
    return F.linear(emb, sd[f"{prefix}.mlp.weight"], sd[f"{prefix}.mlp.bias"])


def _ln(sd, prefix, x):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], LN_EPS)


_LINEAR_INPUT_ROUND = None     # test helper: emulate the fp16-input matrix path (see linear_input_rounding)


class linear_input_rounding:
    """Context manager for the fast-mode emulation of the DiT / CLIP front-end: inside it every Linear (and the patch
    convolution) sees its activation rounded through `dtype`, like the fp16-input MFMA GEMM does; arithmetic stays fp32.
    Pair it with weights rounded by :func:`round_linear_weights`."""

    def __init__(self, dtype=torch.float16):
        self.dtype = dtype

    def __enter__(self):
        global _LINEAR_INPUT_ROUND
        self._old, _LINEAR_INPUT_ROUND = _LINEAR_INPUT_ROUND, self.dtype

    def __exit__(self, *a):
        global _LINEAR_INPUT_ROUND
        _LINEAR_INPUT_ROUND = self._old


def round_linear_weights(sd: StateDict, dtype=torch.float16) -> StateDict:
    """Every >= 2-D '.weight' except lookup tables, rounded through `dtype` (kept fp32)."""
    return {k: (v.to(dtype).float() if (k.endswith(".weight") and v.dim() >= 2 and "position_embedding" not in k) else v)
            for k, v in sd.items()}


def _rin(x):
    return x if _LINEAR_INPUT_ROUND is None else x.to(_LINEAR_INPUT_ROUND).float()


def _lin(sd, prefix, x):
    return F.linear(_rin(x), sd[f"{prefix}.weight"], sd.get(f"{prefix}.bias"))


def point_encoder_embed(sd: StateDict, x, num_heads: int):
    """core/transformer/point.py:186-206 (PointEncoderEmbed.forward) with
    ResCrossAttBlock._forward :123-126, CrossAttention.forward attention.py:141-153,
    FeedForward/GEGLU point.py:68-84.  x [B,N,3] -> latent mean [B,L,latent_dim]."""
    pe = "point_encoder"
    B = x.shape[0]
    c = _ln(sd, f"{pe}.ln", point_embed(sd, x))                 # :194
    q = sd[f"{pe}.query_embed"].repeat(B, 1, 1)                 # :197
    # cross_att: x = q + att(ln1(q), c)
    a = f"{pe}.cross_att.att"
    xq = _ln(sd, f"{pe}.cross_att.ln1", q)
    N, M = xq.shape[1], c.shape[1]
    hd = xq.shape[2] // num_heads
    qq = _lin(sd, f"{a}.q_proj", xq).reshape(B, N, num_heads, hd)
    kk = _lin(sd, f"{a}.k_proj", c).reshape(B, M, num_heads, hd)
    vv = _lin(sd, f"{a}.v_proj", c).reshape(B, M, num_heads, hd)
    att = attention_naive(qq, kk, vv, causal=False)
    att = _lin(sd, f"{a}.out_proj", att.reshape(B, N, -1))
    l = q + att
    # mlp: x = x + net(ln2(x)), net = Linear -> GEGLU -> Linear
    u = _lin(sd, f"{pe}.cross_att.mlp.net.0", _ln(sd, f"{pe}.cross_att.ln2", l))
    xx, gates = u.chunk(2, dim=-1)
    l = l + _lin(sd, f"{pe}.cross_att.mlp.net.2", xx * F.gelu(gates))
    return _lin(sd, f"{pe}.linear", l)                          # :201, DummyLatent(mean)


def quantize_num_faces(n):
    """core/utils.py:89-116."""
    if isinstance(n, int):
        if n <= 0:
            return 0
        for bucket, hi in enumerate((1000, 2000, 4000, 8000), start=1):
            if n <= hi:
                return bucket
        return 5
    r = torch.zeros_like(n)
    r[(n > 0) & (n <= 1000)] = 1
    r[(n > 1000) & (n <= 2000)] = 2
    r[(n > 2000) & (n <= 4000)] = 3
    r[(n > 4000) & (n <= 8000)] = 4
    r[n > 8000] = 5
    return r


def encode_cond(sd: StateDict, opt, conds, num_faces):
    """core/models.py:101-144, eval mode (posterior.mode() == mean)."""
    cond_embeds = None
    if opt.cond_mode == "point":
        lat = point_encoder_embed(sd, conds, opt.point_num_heads)
        cond_embeds = _ln(sd, "norm_cond", _lin(sd, "proj_cond", lat))       # :124
    elif opt.cond_mode == "point_latent":
        cond_embeds = _ln(sd, "norm_cond", _lin(sd, "proj_cond", conds))     # :128-129
    elif opt.cond_mode != "none":
        raise NotImplementedError(opt.cond_mode)
    if opt.use_num_face_cond:                                                 # :135-141
        nf = F.embedding(quantize_num_faces(num_faces), sd["embed_num_face.weight"]).unsqueeze(1)
        cond_embeds = nf if cond_embeds is None else torch.cat((cond_embeds, nf), dim=1)
    return cond_embeds


# ----------------------------------------------------------------------------- decoder
Past = Optional[List[Tuple[torch.Tensor, torch.Tensor]]]


STREAMED_SUFFIXES = ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                     "self_attn.out_proj.weight", "fc1.weight", "fc2.weight", "lm_head.weight")


def round_streamed_weights(sd: StateDict, dtype=torch.float16) -> StateDict:
    """Test helper for the fp16 'fast' storage mode: the decoder matrices that are streamed per token,
    rounded through `dtype` (what the reference's ``model.half()`` does to them, infer.py:56) and kept as
    fp32 tensors, so the restated fp32 arithmetic below runs on exactly the values the device streams."""
    return {k: (v.to(dtype).float() if k.endswith(STREAMED_SUFFIXES) else v) for k, v in sd.items()}


def decoder_forward(sd: StateDict, opt, input_ids=None, inputs_embeds=None, past: Past = None, kv_round=None):
    """ShapeOPT.forward with use_cache=True (core/transformer/modeling_opt.py:464-517
    -> ShapeOPTDecoder.forward :321-426 -> OPTDecoderLayer.forward :253-298 ->
    OptFlashAttention2.forward :172-237).  Returns (logits [B,S,V], new past)."""
    dec = "mesh_decoder.model"
    H = opt.num_heads
    if input_ids is not None:                                                 # :340-342
        inputs_embeds = F.embedding(input_ids.view(-1, input_ids.shape[-1]), sd[f"{dec}.embd.weight"])
    B, S = inputs_embeds.shape[:2]
    past_len = past[0][0].shape[2] if past is not None else 0               # :345
    pos_ids = torch.arange(past_len, past_len + S, dtype=torch.long)        # :355
    h = inputs_embeds + F.embedding(pos_ids, sd[f"{dec}.embed_positions.weight"])  # :356-357
    D = h.shape[-1] // H
    new_past = []
    i = 0
    while f"{dec}.layers.{i}.fc1.weight" in sd:
        L = f"{dec}.layers.{i}"
        resid = h
        q = _lin(sd, f"{L}.self_attn.q_proj", h)                              # :185
        k = _lin(sd, f"{L}.self_attn.k_proj", h).view(B, -1, H, D).transpose(1, 2).contiguous()  # :169-170,189
        v = _lin(sd, f"{L}.self_attn.v_proj", h).view(B, -1, H, D).transpose(1, 2).contiguous()
        if kv_round is not None:   # fast-mode emulation: K/V are stored in the cache dtype (fp16 under infer.py:56,105)
            k, v = k.to(kv_round).float(), v.to(kv_round).float()
        if past is not None:                                                  # :191-192
            k = torch.cat([past[i][0], k], dim=2)
            v = torch.cat([past[i][1], v], dim=2)
        new_past.append((k, v))
        M = k.shape[-2]
        a = attention_naive(q.view(B, S, H, D), k.transpose(1, 2).view(B, M, H, D),
                            v.transpose(1, 2).view(B, M, H, D), causal=True)  # :206-229
        a = _lin(sd, f"{L}.self_attn.out_proj", a.reshape(B, S, H * D))       # :231-232
        h = _ln(sd, f"{L}.self_attn_layer_norm", resid + a)                   # :273-274 (post-LN)
        shape = h.shape
        h2 = h.reshape(-1, shape[-1])                                         # :277-279
        f = F.relu(_lin(sd, f"{L}.fc1", h2))                                  # :281-282
        f = _lin(sd, f"{L}.fc2", f)                                           # :284
        h = _ln(sd, f"{L}.final_layer_norm", (h2 + f).view(shape))            # :287-288
        i += 1
    logits = F.linear(h, sd["mesh_decoder.lm_head.weight"]).contiguous()      # :497
    return logits, new_past


# ----------------------------------------------------------------------------- grammar (integer state machine)
def make_allowed_fn(opt, vocab_size: int, use_tokenizer: bool = True) -> Optional[Callable]:
    """The ``prefix_allowed_tokens_fn`` LMM.generate builds (core/models.py:236-275).
    Returns fn(batch_id, ids_1d) -> list[int]; state is per returned closure, i.e.
    shared across batch rows exactly like the reference's single ``state`` dict
    (the reference asserts B == 1; use one closure per row for B > 1)."""
    eos = opt.eos_token_id
    if not use_tokenizer:                                                     # :237-242
        def fn(batch_id, ids):
            cand = list(range(3, vocab_size))
            if ids.shape[0] % 9 == 1:
                cand.append(eos)
            return cand
        return fn
    if opt.meto_backend not in ("LR", "LR_ABSCO"):                            # :272-274
        return None
    state = {"counter": 0}                                                    # :270

    def fn(batch_id, ids):                                                    # :246-268
        if ids.shape[0] == 0:
            return [5]
        last = int(ids[-1])
        if last == 5:
            state["counter"] = 9
        elif last in (3, 4):
            state["counter"] = 3
        elif last >= 6:
            state["counter"] -= 1
        if state["counter"] > 0:
            return list(range(6, vocab_size))
        return [3, 4, 5, eos]
    return fn


def prefix_constrained_scores(scores, ids, allowed_fns: Sequence[Optional[Callable]]):
    """Patched PrefixConstrainedLogitsProcessor.__call__ (core/utils.py:123-138):
    mask = -inf everywhere, 0 on allowed ids; scores + mask.  num_beams == 1."""
    mask = torch.full_like(scores, -math.inf)
    for b in range(scores.shape[0]):
        fn = allowed_fns[b]
        if fn is None:
            mask[b] = 0
            continue
        allowed = fn(b, ids[b])
        if len(allowed) == 0:
            raise ValueError(f"`prefix_allowed_tokens_fn` returned an empty list for batch ID {b}.")
        mask[b, allowed] = 0
    return scores + mask


def top_k_filter(scores, top_k: int):
    """transformers TopKLogitsWarper: remove scores < k-th largest (ties kept)."""
    k = min(top_k, scores.shape[-1])
    kth = torch.topk(scores, k)[0][..., -1, None]
    return scores.masked_fill(scores < kth, -math.inf)


def sample_from_uniform(filtered_scores_row, u: float) -> int:
    """Test helper for the device sampler (NOT part of the reference, whose
    ``torch.multinomial`` stream cannot be reproduced on another device): draws
    from softmax(filtered_scores_row) - the same categorical distribution HF
    samples - by inverse CDF over ascending token id, in fp32."""
    s = filtered_scores_row.float()
    m = s.max()
    e = torch.where(torch.isinf(s) & (s < 0), torch.zeros_like(s), torch.exp(s - m))
    total = float(e.sum())
    target = u * total
    acc = 0.0
    last = -1
    for i in range(e.shape[0]):
        if e[i] > 0:
            acc += float(e[i])
            last = i
            if acc > target:
                return i
    return last


# ----------------------------------------------------------------------------- generation loop
def make_forward(sd: StateDict, opt, kv_round=None) -> Callable:
    """fwd(input_ids=None, inputs_embeds=None, past=None) -> (logits, past) over this
    restatement; ``make_golden.py`` substitutes the reference's own ShapeOPT here."""
    def fwd(input_ids=None, inputs_embeds=None, past=None):
        return decoder_forward(sd, opt, input_ids=input_ids, inputs_embeds=inputs_embeds, past=past, kv_round=kv_round)
    return fwd


@torch.no_grad()
def generate(fwd: Callable, opt, inputs_embeds, max_new_tokens: int, mode: str = "greedy",
             allowed_fns: Optional[Sequence[Optional[Callable]]] = None, top_k: int = 10,
             min_new_tokens: int = 0, generator: Optional[torch.Generator] = None,
             forced_ids=None, record_logits: Optional[Callable[[int, torch.Tensor], None]] = None,
             step_timer: Optional[Callable[[int], None]] = None):
    """Restatement of what ``self.mesh_decoder.generate(**kwargs)`` does for the
    reference's call (core/models.py:286-303; transformers 4.46.2 ``_sample``):

      prefill on inputs_embeds (modeling_opt.py:536-538); input_ids starts EMPTY
      (only inputs_embeds was given), so the grammar sees idx == 0 first;
      each step: logits[:, -1].float() -> [MinNewTokens: EOS=-inf while t < min]
      -> prefix-constraint mask -> greedy argmax | top-k(10) -> softmax ->
      multinomial(1); next = next*unfinished + pad*(1-unfinished); append;
      unfinished &= next != eos; stop when none unfinished or max_new_tokens;
      later steps feed only the last id with the tuple cache (modeling_opt.py:523-541).

    ``forced_ids`` [B,T] (teacher forcing): feed these instead of the chosen
    ids (used for logits parity in sample mode).  Returns LongTensor [B, T'].
    """
    B = inputs_embeds.shape[0]
    if allowed_fns is None:
        allowed_fns = [None] * B
    eos, pad = opt.eos_token_id, opt.pad_token_id
    ids = torch.empty((B, 0), dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    logits, past = fwd(inputs_embeds=inputs_embeds)
    for t in range(max_new_tokens):
        s = logits[:, -1, :].float()
        if record_logits is not None:
            record_logits(t, s.clone())
        if t < min_new_tokens:
            s = s.clone()
            s[:, eos] = -math.inf
        if any(fn is not None for fn in allowed_fns):
            s = prefix_constrained_scores(s, ids, allowed_fns)
        if mode == "greedy":
            nxt = torch.argmax(s, dim=-1)
        else:
            probs = F.softmax(top_k_filter(s, top_k), dim=-1)
            nxt = torch.multinomial(probs, num_samples=1, generator=generator).squeeze(1)
        if forced_ids is not None:
            nxt = forced_ids[:, t]
        nxt = nxt * unfinished + pad * (1 - unfinished)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        unfinished = unfinished & (nxt != eos).long()
        if step_timer is not None:
            step_timer(t)
        if unfinished.max() == 0 or t == max_new_tokens - 1:
            break
        logits, past = fwd(input_ids=nxt[:, None], past=past)
    return ids


@torch.no_grad()
def lmm_generate_ids(sd: StateDict, opt, conds, num_faces: int = 1000, resume_ids=None,
                     use_tokenizer: bool = True, max_new_tokens: Optional[int] = None,
                     min_new_tokens: int = 0, fwd: Optional[Callable] = None,
                     encode_fn: Optional[Callable] = None, **kw):
    """LMM.generate up to (not including) detokenisation (core/models.py:204-303)."""
    B = conds.shape[0]
    nf = torch.full((B,), num_faces, dtype=torch.long)
    cond = (encode_fn or (lambda c, n: encode_cond(sd, opt, c, n)))(conds, nf)  # :219-221
    input_ids = torch.full((B, 1), opt.bos_token_id, dtype=torch.long)        # :224
    if resume_ids is not None:
        input_ids = torch.cat((input_ids, resume_ids), dim=1)
    tok = F.embedding(input_ids, sd["mesh_decoder.model.embd.weight"])        # :228
    emb = tok if cond is None else torch.cat((cond, tok), dim=1)              # :230-233
    vocab = sd["mesh_decoder.lm_head.weight"].shape[0]
    fns = [make_allowed_fn(opt, vocab, use_tokenizer) for _ in range(B)]
    if max_new_tokens is None:
        max_new_tokens = opt.max_seq_length                                   # :278
    return generate(fwd or make_forward(sd, opt), opt, emb, max_new_tokens, mode=opt.generate_mode, allowed_fns=fns,
                    min_new_tokens=min_new_tokens, **kw)


# ============================================================================= DiT front-end (scope row f3)
# Restatement of core/transformer/dit.py::DiT (adaLN-single, PixArt-alpha style) at state_dict level
# (prefix "dit.") and of MDiT.run's sampling loop (core/models_dit.py:184-229).  The scheduler is
# diffusers' DDIMScheduler (third-party, absent here: diffusers is in requirements.txt but not vendored);
# its published update rule is restated in ddim_* below for the reference's configuration
# (core/models_dit.py:91-102: v_prediction, scaled_linear betas 0.00085..0.012, 1000 train steps,
# "leading" spacing, steps_offset 1, set_alpha_to_one False, eta 0, no clipping) - unpinned by upstream tests.
def timestep_embedding(t, num_channels=256, max_period=10000):
    """dit.py:45-77 (Timesteps, flip_sin_to_cos=False, shift 0, scale 1)."""
    half = num_channels // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)


def _attention_f16_emulation(q, k, v, dtype):
    """Fast-mode emulation of the fused attention kernel (csrc/k_flash_attn.h): q, k, v and the unnormalised
    probabilities pass through `dtype`, everything else is fp32.  (The kernel rounds p against the running row
    maximum of its key tiles, this against the final one: same to within an fp16 ulp of p.)"""
    B, N, H, D = q.shape
    r = lambda t: t.to(dtype).float()
    qh, kh, vh = (r(t).transpose(1, 2) for t in (q, k, v))                # [B,H,*,D]
    s = (qh @ kh.transpose(-1, -2)) * (1.0 / D ** 0.5)
    p = torch.exp(s - s.amax(dim=-1, keepdim=True))
    o = (r(p) @ vh) / p.sum(dim=-1, keepdim=True)
    return o.transpose(1, 2).contiguous()


def _mha(sd, prefix, xq, ctx, num_heads, fused_qkv):
    """SelfAttention (attention.py:98-121, fused qkv_proj) / CrossAttention (:124-153)."""
    B, N, C = xq.shape
    D = C // num_heads
    if fused_qkv:
        qkv = _lin(sd, f"{prefix}.qkv_proj", xq).reshape(B, N, 3, num_heads, D).permute(2, 0, 1, 3, 4)
        q, k, v = qkv.chunk(3, dim=0)
        q, k, v = q[0], k[0], v[0]
    else:
        M = ctx.shape[1]
        q = _lin(sd, f"{prefix}.q_proj", xq).reshape(B, N, num_heads, D)
        k = _lin(sd, f"{prefix}.k_proj", ctx).reshape(B, M, num_heads, D)
        v = _lin(sd, f"{prefix}.v_proj", ctx).reshape(B, M, num_heads, D)
    if _LINEAR_INPUT_ROUND is not None and D == 64:
        a = _attention_f16_emulation(q, k, v, _LINEAR_INPUT_ROUND)
    else:
        a = attention_naive(q, k, v, causal=False)
    return _lin(sd, f"{prefix}.out_proj", a.reshape(B, N, -1))


def dit_forward(sd: StateDict, x, c, t, num_heads: int, prefix="dit"):
    """DiT.forward (dit.py:168-196) with DiTLayer._forward (:123-140).  x [B,N,latent], c [B,M,C], t [B]."""
    B = x.shape[0]
    x = _lin(sd, f"{prefix}.proj_in", x) + sd[f"{prefix}.pos_embed"]
    t_emb = timestep_embedding(t)
    t_emb = _lin(sd, f"{prefix}.timestep_proj.linear_2", F.silu(_lin(sd, f"{prefix}.timestep_proj.linear_1", t_emb)))
    t_adaln = _lin(sd, f"{prefix}.adaln_linear", F.silu(t_emb)).view(B, 6, -1)
    C = x.shape[-1]
    i = 0
    while f"{prefix}.layers.{i}.scale_shift_table" in sd:
        L = f"{prefix}.layers.{i}"
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = (sd[f"{L}.scale_shift_table"][None] + t_adaln).chunk(6, dim=1)
        h = F.layer_norm(x, (C,), None, None, 1e-6)
        h = h * (1 + sc_a) + sh_a
        h = h + g_a * _mha(sd, f"{L}.attn1", h, None, num_heads, True)       # NB: residual is the MODULATED input (dit.py:133-135)
        h = h + _mha(sd, f"{L}.attn2", h, c, num_heads, False)
        x = F.layer_norm(h, (C,), None, None, 1e-6)
        x = x * (1 + sc_m) + sh_m
        x = x + g_m * _lin(sd, f"{L}.ff.net.2", (lambda u: u[0] * F.gelu(u[1]))(_lin(sd, f"{L}.ff.net.0", x).chunk(2, dim=-1)))
        i += 1
    shift, scale = (sd[f"{prefix}.scale_shift_table"][None] + t_emb[:, None]).chunk(2, dim=1)
    x = F.layer_norm(x, (C,), None, None, 1e-6)
    x = x * (1 + scale) + shift
    return _lin(sd, f"{prefix}.proj_out", x)


def ddim_schedule(num_inference_steps=100, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
    """diffusers DDIMScheduler.__init__/set_timesteps for the reference's config: returns (timesteps list,
    alphas_cumprod fp32 tensor, final_alpha_cumprod)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    ratio = num_train // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + steps_offset
    return ts.tolist(), ac, ac[0]


def ddim_step_v(sample, v, t, ac, final_ac, ratio):
    """DDIMScheduler.step, prediction_type='v_prediction', eta=0, clip_sample=False."""
    prev = t - ratio
    a_t = ac[t]
    a_prev = ac[prev] if prev >= 0 else final_ac
    b_t = 1 - a_t
    pred_x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * v
    pred_eps = (a_t ** 0.5) * v + (b_t ** 0.5) * sample
    return (a_prev ** 0.5) * pred_x0 + ((1 - a_prev) ** 0.5) * pred_eps


def dit_project_cond(sd: StateDict, clip_hidden):
    """MDiT.get_cond after the CLIP encoder (core/models_dit.py:113)."""
    return _ln(sd, "norm_cond", _lin(sd, "proj_cond", clip_hidden))


@torch.no_grad()
def mdit_run(sd: StateDict, cond, init_latents, num_heads: int, num_inference_steps=100, guidance_scale=7.5,
             forward_fn=None, latents=None, strength=0.5):
    """MDiT.run (core/models_dit.py:184-229) from projected cond [B,M,C] and the Gaussian draw the reference takes
    from torch.randn / randn_like (`init_latents`, passed in: RNG streams do not transfer across devices);
    num_repeat = 1.  `latents` given: the img2img branch (:207-209), DDIMScheduler.add_noise at
    timesteps[int(steps * strength)] and the loop over timesteps[init_step:]."""
    fwd = forward_fn or (lambda x, c, t: dit_forward(sd, x, c, t, num_heads))
    ts, ac, final = ddim_schedule(num_inference_steps)
    ratio = 1000 // num_inference_steps
    B = cond.shape[0]
    if latents is None:
        init_step = 0
        latents = init_latents.clone()
    else:
        init_step = int(num_inference_steps * strength)
        a_t = ac[ts[init_step]]
        latents = (a_t ** 0.5) * latents + ((1 - a_t) ** 0.5) * init_latents        # scheduler.add_noise
    c2 = torch.cat([torch.zeros_like(cond), cond], dim=0)
    for t in ts[init_step:]:
        x2 = torch.cat([latents] * 2, dim=0)
        t_in = torch.tensor([t] * B * 2, dtype=latents.dtype)
        pred = fwd(x2, c2, t_in)
        u, cnd = pred.chunk(2)
        pred = u + guidance_scale * (cnd - u)
        latents = ddim_step_v(latents, pred, t, ac, final, ratio)
    return latents


# ----------------------------------------------------------------------------- CLIP ViT-H/14 image encoder (f3)
# MDiT.get_cond (core/models_dit.py:104-115) runs a frozen HuggingFace CLIPVisionModel
# ('laion/CLIP-ViT-H-14-laion2B-s32B-b79K': 32 layers, width 1280, 16 heads, MLP 5120, gelu, patch 14, 224 px) and
# takes .last_hidden_state.  The model code is third-party (transformers==4.46.2 models/clip/modeling_clip.py:
# CLIPVisionEmbeddings / CLIPEncoderLayer / CLIPAttention / CLIPMLP), restated here at state_dict level with the
# 4.46.2 key names ("image_encoder.vision_model.*"); oracle/make_golden.py checks it against the installed
# transformers' CLIPVisionModel (random init - the pretrained weights cannot be fetched).
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(images):
    """core/models_dit.py:108-109: TF.normalize(mean, std) then bilinear resize to 224 (align_corners=False)."""
    mean = torch.tensor(CLIP_MEAN, dtype=images.dtype).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=images.dtype).view(1, 3, 1, 1)
    x = (images - mean) / std
    return F.interpolate(x, (224, 224), mode="bilinear", align_corners=False)


def clip_vision_forward(sd: StateDict, pixel_values, num_heads: int = 16, prefix="image_encoder.vision_model", eps=1e-5):
    """CLIPVisionTransformer.forward -> last_hidden_state [B, 257, width] (no post_layernorm)."""
    B = pixel_values.shape[0]
    w = sd[f"{prefix}.embeddings.patch_embedding.weight"]
    x = F.conv2d(_rin(pixel_values), w, stride=w.shape[-1]).flatten(2).transpose(1, 2)
    cls = sd[f"{prefix}.embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[f"{prefix}.embeddings.position_embedding.weight"]
    C = x.shape[-1]
    D = C // num_heads
    x = F.layer_norm(x, (C,), sd[f"{prefix}.pre_layrnorm.weight"], sd[f"{prefix}.pre_layrnorm.bias"], eps)
    i = 0
    while f"{prefix}.encoder.layers.{i}.mlp.fc1.weight" in sd:
        L = f"{prefix}.encoder.layers.{i}"
        h = F.layer_norm(x, (C,), sd[f"{L}.layer_norm1.weight"], sd[f"{L}.layer_norm1.bias"], eps)
        N = h.shape[1]
        q = (_lin(sd, f"{L}.self_attn.q_proj", h) * (D ** -0.5)).view(B, N, num_heads, D).transpose(1, 2)
        k = _lin(sd, f"{L}.self_attn.k_proj", h).view(B, N, num_heads, D).transpose(1, 2)
        v = _lin(sd, f"{L}.self_attn.v_proj", h).view(B, N, num_heads, D).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
        x = x + _lin(sd, f"{L}.self_attn.out_proj", a.transpose(1, 2).reshape(B, N, C))
        h = F.layer_norm(x, (C,), sd[f"{L}.layer_norm2.weight"], sd[f"{L}.layer_norm2.bias"], eps)
        x = x + _lin(sd, f"{L}.mlp.fc2", F.gelu(_lin(sd, f"{L}.mlp.fc1", h)))
        i += 1
    return x


def mdit_get_cond(sd: StateDict, images, num_heads: int = 16):
    """MDiT.get_cond (core/models_dit.py:104-115) from images [B,3,H,W] in [0,1]."""
    return dit_project_cond(sd, clip_vision_forward(sd, clip_preprocess(images), num_heads))
