"""Host-side mirrors of the drop-in surface compared LIVE with the reference's own functions, in the build container
where /root/reference is importable (oracle/ref_stubs.py supplies its absent non-arithmetic dependencies and
oracle/_ref the compiled meto extension).  Skipped wherever the reference is not present (e.g. the GPU box):
nothing here is needed for the GPU tests."""
import dataclasses
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_stubs  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_stubs.reference_available(), reason="/root/reference not present")


@pytest.fixture(scope="module")
def ref():
    ref_stubs.install()
    import core.options as ro
    import core.utils as ru
    return ro, ru


def test_options_presets_field_for_field(ref):
    """core/options.py:17-212: every preset, every field, same names, same order, same values."""
    ro, _ = ref
    from edgerunner_amd.options import Options, config_defaults
    assert sorted(ro.config_defaults) == sorted(config_defaults)
    assert [f.name for f in dataclasses.fields(ro.Options)] == [f.name for f in dataclasses.fields(Options)]
    for name, want in ro.config_defaults.items():
        assert dataclasses.asdict(want) == dataclasses.asdict(config_defaults[name]), name
    assert dataclasses.asdict(ro.Options()) == dataclasses.asdict(Options())


def test_quantize_num_faces_live(ref):
    _, ru = ref
    from edgerunner_amd.utils import quantize_num_faces
    for n in list(range(-3, 12)) + [999, 1000, 1001, 1999, 2000, 2001, 3999, 4000, 4001, 7999, 8000, 8001, 10 ** 7]:
        assert quantize_num_faces(n) == ru.quantize_num_faces(n), n
    t = torch.randint(-10, 20000, (500,), generator=torch.Generator().manual_seed(0))
    assert torch.equal(quantize_num_faces(t), ru.quantize_num_faces(t))
    # (numpy input is an extension of this mirror: the reference accepts ints and tensors only)


def _ref_meto():
    refdir = os.path.join(ROOT, "oracle", "_ref")
    if not glob.glob(os.path.join(refdir, "_meto*.so")):
        pytest.skip("oracle/_ref not built")
    sys.path.insert(0, refdir)
    sys.path.insert(0, os.path.join(ref_stubs.REFERENCE_ROOT, "meto"))
    import meto
    return meto


def _meshes():
    rng = np.random.default_rng(5)
    k = 6
    xs = np.linspace(-0.9, 0.9, k)
    grid_v = np.array([[x, y, 0.2 * np.sin(3 * x) * np.cos(2 * y)] for y in xs for x in xs])
    grid_f = np.array([[j * k + i, j * k + i + 1, (j + 1) * k + i + 1] for j in range(k - 1) for i in range(k - 1)] +
                      [[j * k + i, (j + 1) * k + i + 1, (j + 1) * k + i] for j in range(k - 1) for i in range(k - 1)])
    soup_v = rng.random((30, 3)) * 1.9 - 0.95
    soup_f = np.stack([rng.choice(30, 3, replace=False) for _ in range(40)])
    return {"grid": (grid_v, grid_f), "soup": (soup_v, soup_f)}


@pytest.mark.parametrize("backend", [None, "LR_ABSCO", "LR"])
def test_tokenize_detokenize_mesh_live(ref, backend):
    """core/provider.py:69-147 with and without a meto tokenizer, plus meto's sort_mesh / normalize_mesh helpers."""
    rmeto = _ref_meto()
    import core.provider as rp
    from edgerunner_amd import meshio, meto
    r_tok = rmeto.Engine(512, backend=backend) if backend else None
    m_tok = meto.Engine(512, backend=backend) if backend else None
    for name, (v, f) in _meshes().items():
        want = rp.tokenize_mesh(v.copy(), f.copy(), 512, r_tok)
        got = meto.tokenize_mesh(v.copy(), f.copy(), 512, m_tok)
        assert np.array_equal(np.asarray(want), np.asarray(got)), (backend, name)
        wv, wf = rp.detokenize_mesh(np.asarray(want), 512, r_tok)
        gv, gf = meto.detokenize_mesh(np.asarray(got), 512, m_tok)
        assert np.allclose(np.asarray(wv, np.float64), gv, atol=0, rtol=0) and np.array_equal(np.asarray(wf), gf), (backend, name)
        sv, sf = rmeto.sort_mesh(v.copy(), f.copy())
        mv, mf = meto.sort_mesh(v.copy(), f.copy())
        assert np.array_equal(sv, mv) and np.array_equal(sf, mf)
        assert np.array_equal(rmeto.normalize_mesh(v.copy(), 0.95), meshio.normalize_mesh(v.copy(), 0.95))
    # a stream that is not a multiple of 9 without a tokenizer: trimmed with a warning in both
    if backend is None:
        raw = np.arange(22) + 3
        wv, wf = rp.detokenize_mesh(raw.copy(), 512, None)
        gv, gf = meto.detokenize_mesh(raw.copy(), 512, None)
        assert np.array_equal(wv, gv) and np.array_equal(wf, gf)


def test_get_tokenizer_vocabulary_sizes(ref):
    """core/utils.py:78-86 / core/models.py:78-84: tokenizer choice and vocabulary size per Options."""
    _ref_meto()
    _, ru = ref
    from edgerunner_amd.meto import get_tokenizer
    from edgerunner_amd.options import config_defaults
    from edgerunner_amd import weights as W
    for kw, vocab in (({}, 518), ({"meto_backend": "LR"}, 1030), ({"use_meto": False}, 515)):
        opt = dataclasses.replace(config_defaults["ArAE"], **kw)
        r_tok, r_n = ru.get_tokenizer(opt)
        m_tok, m_n = get_tokenizer(opt)
        assert r_n == m_n and (r_tok is None) == (m_tok is None)
        assert W.dims_from_options(opt).vocab_size == vocab


def test_oracle_restatement_bit_identical_to_reference_modules_live():
    """The claim the goldens rest on, re-checked on every CPU run where the reference is importable: the state_dict-level
    restatement (oracle/arae_oracle.py) gives torch.equal results to the reference's own LMM modules - encode_cond,
    prefill logits, cached decode steps and the KV cache (2 layers, full width) - for both tokenizer vocabularies."""
    import make_golden as MG                      # build-container helpers (imports the reference via ref_stubs)
    from edgerunner_amd import weights as W
    for kw in ({}, {"meto_backend": "LR"}):
        opt, ref_opt = MG.opts(2, **kw)
        sd = W.make_state_dict(opt, 0, "perturbed")
        model = MG.build_reference(ref_opt, sd)
        conds = torch.stack([W.synthetic_point_cloud(i, 256)[0] for i in range(2)])
        assert MG.validate_restatement(model, sd, opt, conds), kw


def test_dit_restatement_bit_identical_to_reference_module_live():
    import arae_oracle as O
    from core.transformer.dit import DiT as RefDiT
    from edgerunner_amd import weights as W
    from edgerunner_amd.options import config_defaults
    opt = dataclasses.replace(config_defaults["ArAE"], dit_num_layers=2)
    sd = W.make_dit_state_dict(opt, 0, "perturbed")
    dit = RefDiT(hidden_dim=opt.dit_hidden_dim, num_heads=opt.dit_num_heads, latent_size=opt.point_latent_size,
                 latent_dim=opt.point_latent_dim, num_layers=2, gradient_checkpointing=False).eval()
    dit.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("dit.")}, strict=True)
    g = torch.Generator().manual_seed(11)
    x, c, t = torch.randn(2, 2048, 64, generator=g), torch.randn(2, 257, 1024, generator=g), torch.tensor([701.0, 33.0])
    with torch.no_grad():
        assert torch.equal(dit(x, c, t), O.dit_forward(sd, x, c, t, opt.dit_num_heads))
