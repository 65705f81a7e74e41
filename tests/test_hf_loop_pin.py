"""Pins the oracle's restated generation loop (oracle/arae_oracle.py::generate - what `self.mesh_decoder.generate(**kwargs)`
does at core/models.py:286-303) against an EXECUTABLE HuggingFace loop: the installed transformers' own
`GenerationMixin.generate` driving a toy causal LM whose logits are a fixed function of (last token, sequence length).
Same call shape as the reference's: `inputs_embeds` only (so `input_ids` starts zero-width and the grammar sees idx == 0
first), `prefix_allowed_tokens_fn` with the reference's stateful closure, pad / bos / eos ids, `max_new_tokens`,
`num_beams=1` | `do_sample=True, top_k=10`, plus `min_new_tokens`.  The model side mirrors
`ShapeOPT.prepare_inputs_for_generation` (core/transformer/modeling_opt.py:519-550: step 0 feeds `inputs_embeds`, later
steps the last id only).  Under one `torch.manual_seed` both loops call `torch.multinomial` the same number of times on the
same probabilities, so sample mode is compared id for id as well.

The reference pins transformers == 4.46.2 (requirements.lock.txt:16), which is not installed; this checks the restatement
against the installed release instead (the loop's semantics - processor order, pad-after-EOS, stop rule - are the same
published behaviour)."""
import dataclasses

import pytest
import torch

import arae_oracle as O

transformers = pytest.importorskip("transformers")
from transformers import GenerationMixin, PretrainedConfig, PreTrainedModel  # noqa: E402
from transformers.modeling_outputs import CausalLMOutputWithPast  # noqa: E402

V, C, S0, B = 518, 8, 5, 3


class ToyConfig(PretrainedConfig):
    model_type = "toy_loop_pin"

    def __init__(self, **kw):
        super().__init__(**kw)
        self.vocab_size, self.hidden_size, self.num_hidden_layers, self.num_attention_heads = V, C, 1, 1


def toy_logits(table, last, length):
    """[B, V] logits from the previous token (-1 at the prefill step) and the total sequence length; EOS gets a
    length-dependent bonus so rows finish at different steps."""
    rows = table[(last + 1) % table.shape[0]]
    out = torch.roll(rows, shifts=int(length) % V, dims=-1) + 0.01 * (length % 7)
    out[:, 2] += 3.0 if length % 5 == 0 else -1.0
    return out


class ToyLM(PreTrainedModel, GenerationMixin):
    config_class = ToyConfig
    main_input_name = "input_ids"

    def __init__(self, cfg):
        super().__init__(cfg)
        self.table = torch.nn.Parameter(torch.randn(V + 1, V, generator=torch.Generator().manual_seed(7)) * 2)

    def forward(self, input_ids=None, inputs_embeds=None, attention_mask=None, past_key_values=None, use_cache=None, **kw):
        length = attention_mask.shape[1]
        if input_ids is None:
            n, s = inputs_embeds.shape[:2]
            lg = toy_logits(self.table, torch.full((n,), -1, dtype=torch.long), length)[:, None, :].expand(n, s, V).contiguous()
        else:
            lg = toy_logits(self.table, input_ids[:, -1], length)[:, None, :]
        return CausalLMOutputWithPast(logits=lg, past_key_values=past_key_values)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kw):
        if inputs_embeds is not None and input_ids.shape[1] == 0:           # modeling_opt.py:536-538
            model_inputs = {"inputs_embeds": inputs_embeds, "input_ids": None}
        else:                                                                 # modeling_opt.py:523-534
            model_inputs = {"input_ids": input_ids[:, -1:], "inputs_embeds": None}
        model_inputs.update({"attention_mask": attention_mask, "past_key_values": past_key_values, "use_cache": True})
        return model_inputs


def reference_grammar():
    """The stateful closure of core/models.py:246-268 (one state per row, as `partial(..., state=state)` gives B = 1)."""
    state = {"counter": 0}

    def fn(batch_id, ids):
        if ids.shape[0] == 0:
            return [5]
        if ids[-1] == 5:
            state["counter"] = 9
        elif ids[-1] in (3, 4):
            state["counter"] = 3
        elif ids[-1] >= 6:
            state["counter"] -= 1
        return list(range(6, V)) if state["counter"] > 0 else [3, 4, 5, 2]
    return fn


@pytest.fixture(scope="module")
def toy():
    model = ToyLM(ToyConfig()).eval()
    table = model.table.detach()

    def fwd(input_ids=None, inputs_embeds=None, past=None):                  # the oracle's forward protocol
        if inputs_embeds is not None:
            n, s = inputs_embeds.shape[:2]
            return toy_logits(table, torch.full((n,), -1, dtype=torch.long), s)[:, None, :], s
        return toy_logits(table, input_ids[:, -1], past + 1)[:, None, :], past + 1
    return model, fwd


CASES = [("greedy", 0, True, None), ("greedy", 12, True, None), ("greedy", 0, False, None), ("greedy", 25, False, None)] + \
        [("sample", mn, fn, seed) for seed in range(6) for mn, fn in ((0, True), (8, False))]


@pytest.mark.parametrize("mode,min_new,use_fn,seed", CASES)
def test_restated_loop_equals_installed_hf_generate(toy, mode, min_new, use_fn, seed):
    model, fwd = toy
    opt = dataclasses.make_dataclass("Ids", [("eos_token_id", int, 2), ("pad_token_id", int, 0)])()
    emb = torch.zeros(B, S0, C)
    fns_hf, fns_me = [reference_grammar() for _ in range(B)], [reference_grammar() for _ in range(B)]
    kw = dict(inputs_embeds=emb, max_new_tokens=48, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    if min_new:
        kw["min_new_tokens"] = min_new
    if use_fn:
        kw["prefix_allowed_tokens_fn"] = lambda b, ids: fns_hf[b](b, ids)
    kw.update(dict(num_beams=1, do_sample=False) if mode == "greedy" else dict(do_sample=True, top_k=10))
    with torch.no_grad():
        if seed is not None:
            torch.manual_seed(seed)
        want = model.generate(**kw)
        if seed is not None:
            torch.manual_seed(seed)
        got = O.generate(fwd, opt, emb, 48, mode=mode, allowed_fns=fns_me if use_fn else None, min_new_tokens=min_new)
    assert got.shape == want.shape and torch.equal(got, want), (got, want)
    if min_new:
        assert (want[:, :min_new] != 2).all()


def test_cases_cover_early_stop_and_pad_after_eos(toy):
    """The toy must actually exercise the interesting branches: a row that finishes early and is padded while another
    continues, and a run HF cuts short because every row finished."""
    model, _ = toy
    emb = torch.zeros(B, S0, C)
    padded = short = False
    with torch.no_grad():
        for seed in range(6):
            torch.manual_seed(seed)
            out = model.generate(inputs_embeds=emb, max_new_tokens=48, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                                 do_sample=True, top_k=10, prefix_allowed_tokens_fn=(lambda fns: lambda b, ids: fns[b](b, ids))(
                                     [reference_grammar() for _ in range(B)]))
            short |= out.shape[1] < 48
            for row in out.tolist():
                if 2 in row and row.index(2) + 1 < len(row):
                    padded |= all(t == 0 for t in row[row.index(2) + 1:])
    assert padded and short
