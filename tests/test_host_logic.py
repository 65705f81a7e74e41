"""Host-side logic of the drop-in surface: Options/CLI, face buckets, grammars, token post-processing."""
import dataclasses

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

import arae_oracle as O
from edgerunner_amd import native, weights as W
from edgerunner_amd.grammar import GrammarState, as_callable, select_grammar
from edgerunner_amd.options import Options, config_defaults, parse_cli
from edgerunner_amd.utils import quantize_num_faces, trim_tokens


def test_presets_match_reference_values():
    a = config_defaults["ArAE"]
    assert (a.hidden_dim, a.num_heads, a.num_layers, a.num_cond_tokens, a.max_seq_length) == (1536, 16, 24, 2049, 40960)
    assert (a.cond_mode, a.point_encoder_mode, a.meto_backend, a.use_num_face_cond, a.generate_mode) == \
           ("point", "embed", "LR_ABSCO", True, "sample")
    d = config_defaults["default"]
    assert (d.hidden_dim, d.cond_mode, d.num_cond_tokens, d.max_seq_length, d.point_num) == (1024, "image", 257, 10240, 8192)
    assert config_defaults["DiT"].dit_num_layers == 24 and config_defaults["DiT"].batch_size == 8
    dims = W.dims_from_options(a)
    assert (dims.vocab_size, dims.intermediate_dim, dims.max_positions, dims.head_dim) == (518, 6144, 43019, 96)
    assert len(W.tensor_specs(dims)) == 416          # 416 checkpoint keys


def test_cli_surface():
    o = parse_cli(["ArAE", "--workspace", "w", "--resume", "ck.safetensors", "--test_path", "in", "--generate_mode",
                   "greedy", "--test_num_face", "1000", "4000", "--test_repeat", "3", "--seed", "7",
                   "--test-max-seq-length", "4000", "--no-use-num-face-cond"])
    assert (o.workspace, o.resume, o.test_path, o.generate_mode, o.test_num_face, o.test_repeat, o.seed) == \
           ("w", "ck.safetensors", "in", "greedy", (1000, 4000), 3, 7)
    assert o.test_max_seq_length == 4000 and o.use_num_face_cond is False and o.hidden_dim == 1536
    assert parse_cli(["default"]).hidden_dim == 1024
    with pytest.raises(SystemExit):
        parse_cli(["ArAE", "--generate_mode", "beam"])


@given(st.integers(min_value=-5, max_value=20000))
def test_quantize_num_faces_int(n):
    assert quantize_num_faces(n) == O.quantize_num_faces(n)


def test_quantize_num_faces_tensor_and_edges():
    n = torch.tensor([-1, 0, 1, 1000, 1001, 2000, 2001, 4000, 4001, 8000, 8001, 10 ** 6])
    assert torch.equal(quantize_num_faces(n), O.quantize_num_faces(n))
    assert quantize_num_faces(n).tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5]
    assert quantize_num_faces(np.array([5, 3000])).tolist() == [1, 3]


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(min_value=0, max_value=517), min_size=0, max_size=60), st.booleans())
def test_grammar_matches_reference_closure(ids, use_tok):
    """Feed the same (arbitrary, even illegal) id history to the oracle's closure and to the host
    mirror of the device automaton: allowed sets must agree at every step."""
    opt = config_defaults["ArAE"]
    ref = O.make_allowed_fn(opt, 518, use_tokenizer=use_tok)
    mine = GrammarState(select_grammar(opt, use_tok), 518)
    hist = torch.empty(0, dtype=torch.long)
    last = None
    for t in ids + [0]:
        assert sorted(ref(0, hist)) == sorted(mine.allowed(last))
        hist = torch.cat([hist, torch.tensor([t])])
        last = t


def test_grammar_callable_and_selection():
    opt = config_defaults["ArAE"]
    assert select_grammar(opt, True) == native.ER_GRAMMAR_LR_ABSCO
    assert select_grammar(opt, False) == native.ER_GRAMMAR_NAIVE9
    fn = as_callable(native.ER_GRAMMAR_LR_ABSCO, 518)
    assert fn(0, torch.empty(0, dtype=torch.long)) == [5]
    assert fn(0, torch.tensor([5])) == list(range(6, 518))
    assert as_callable(native.ER_GRAMMAR_NONE, 518) is None


def test_trim_tokens():
    t = np.array([5, 10, 11, 2, 0, 0])
    assert trim_tokens(t).tolist() == [2, 7, 8]
    assert trim_tokens(np.array([5, 6])).tolist() == [2, 3]


def test_synthetic_inputs_are_deterministic():
    a, b = W.synthetic_point_cloud(3, 100), W.synthetic_point_cloud(3, 100)
    assert torch.equal(a, b) and a.shape == (1, 100, 3) and float(a.abs().max()) <= 0.95
    opt = dataclasses.replace(config_defaults["ArAE"], num_layers=1)
    k, t = next(iter(W.iter_state_dict(opt, 0, "perturbed")))
    k2, t2 = next(iter(W.iter_state_dict(opt, 0, "perturbed")))
    assert k == k2 and torch.equal(t, t2)
    assert torch.equal(W.point_basis(24)[1, 8:16], torch.pow(2, torch.arange(8)).float() * np.pi)
