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


# ------------------------------------------------------------------ mesh / point-cloud I/O used by infer.py
def test_meshio_formats_and_surface_sampling(tmp_path):
    import struct
    from edgerunner_amd import meshio
    v = np.array([[x, y, z] for x in (0.0, 2.0) for y in (0.0, 1.0) for z in (0.0, 1.0)])
    quads = [[0, 1, 3, 2], [4, 6, 7, 5], [0, 4, 5, 1], [2, 3, 7, 6], [0, 2, 6, 4], [1, 5, 7, 3]]
    # OBJ: 1-based, v/vt/vn triplets, negative (relative) indices, polygons are fan-triangulated
    with open(tmp_path / "box.obj", "w") as fh:
        for p in v:
            fh.write(f"v {p[0]} {p[1]} {p[2]}\n")
        for q in quads[:5]:
            fh.write("f " + " ".join(f"{i + 1}/1/1" for i in q) + "\n")
        fh.write("f " + " ".join(str(i - 8) for i in quads[5]) + "\n")
    ov, of = meshio.load_mesh(str(tmp_path / "box.obj"))
    assert ov.shape == (8, 3) and of.shape == (12, 3) and of.min() == 0 and of.max() == 7
    assert of[-2].tolist() == [1, 5, 7] and of[-1].tolist() == [1, 7, 3]
    # ASCII PLY round trip, binary little-endian PLY with an extra vertex property
    meshio.save_ply(str(tmp_path / "a.ply"), ov, of)
    av, af = meshio.load_mesh(str(tmp_path / "a.ply"))
    assert np.allclose(av, ov) and np.array_equal(af, of)
    with open(tmp_path / "b.ply", "wb") as fh:
        fh.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 8\nproperty float x\nproperty float y\nproperty float z\n"
                 b"property uchar red\nelement face 6\nproperty list uchar int vertex_indices\nend_header\n")
        for p in v:
            fh.write(struct.pack("<fffB", *p, 7))
        for q in quads:
            fh.write(struct.pack("<Biiii", 4, *q))
    bv, bf = meshio.load_mesh(str(tmp_path / "b.ply"))
    assert np.allclose(bv, v) and bf.shape == (12, 3)
    with pytest.raises(ValueError):
        meshio.load_mesh(str(tmp_path / "x.stl"))
    # normalize_mesh (core/utils.py:69-75): longest side spans [-bound, bound], centred
    n = meshio.normalize_mesh(ov, bound=0.95)
    assert np.isclose(n[:, 0].max(), 0.95) and np.isclose(n[:, 0].min(), -0.95) and np.allclose(n.max(0) + n.min(0), 0)
    # area-weighted surface sampling: deterministic per generator, on the surface, proportional to face area
    pts = meshio.sample_surface(ov, of, 20000, np.random.default_rng(0))
    assert np.array_equal(pts, meshio.sample_surface(ov, of, 20000, np.random.default_rng(0)))
    on_face = (np.isclose(pts[:, 0], 0) | np.isclose(pts[:, 0], 2) | np.isclose(pts[:, 1], 0) | np.isclose(pts[:, 1], 1)
               | np.isclose(pts[:, 2], 0) | np.isclose(pts[:, 2], 1))
    assert on_face.all() and pts.min() >= 0 and pts[:, 0].max() <= 2
    frac_small = (np.isclose(pts[:, 0], 0) | np.isclose(pts[:, 0], 2)).mean()      # the two 1x1 ends: 2 of 10 area units
    assert abs(frac_small - 0.2) < 0.02
    meshio.save_points_obj(str(tmp_path / "pc.obj"), pts[:5])
    assert len(open(tmp_path / "pc.obj").read().splitlines()) == 5


def test_philox_host_replica_known_answers():
    """The host replica of the device sampler's generator against the Random123 known-answer vectors for
    philox4x32-10 (kat_vectors); the GPU tests then compare the device draw with this replica."""
    from edgerunner_amd.kernels import philox4x32_10, philox_uniform
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for counter, key, want in kat:
        assert tuple(philox4x32_10(counter, key)) == want
    assert philox_uniform(0, 0, 0) == (0x6627e8d5 >> 8) / 2 ** 24
    us = [philox_uniform(1234, t, b) for t in range(200) for b in range(4)]
    assert 0.0 <= min(us) and max(us) < 1.0 and 0.4 < sum(us) / len(us) < 0.6 and len(set(us)) == len(us)


# ------------------------------------------------------------------ save_mesh(clean=True): restated trimesh clean-up
def _cube():
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=np.float64)
    # outward-facing winding (positive signed volume)
    f = np.array([[0, 2, 1], [0, 3, 2], [0, 5, 4], [0, 1, 5], [1, 6, 5], [1, 2, 6], [2, 7, 6], [2, 3, 7], [3, 4, 7], [3, 0, 4],
                  [4, 6, 7], [4, 5, 6]], dtype=np.int64)
    return v, f


def _signed_volume(v, f):
    t = v[f]
    return float(np.einsum("ij,ij->", t[:, 0], np.cross(t[:, 1], t[:, 2])) / 6.0)


def _winding_consistent(f):
    from edgerunner_amd import meto
    adj, ev = meto.face_adjacency(f)
    return all(meto._traverses(f[a], u, w) != meto._traverses(f[b], u, w) for (a, b), (u, w) in zip(adj.tolist(), ev.tolist()))


def test_clean_like_trimesh_merge_unique_and_winding():
    """core/provider.py:52-58 semantics (trimesh restated, see edgerunner_amd/meto.py): triangle soup -> shared vertices,
    duplicate faces dropped (any winding), DEGENERATE faces kept, winding made consistent, volume made positive."""
    from edgerunner_amd import meto
    v, f = _cube()
    assert _signed_volume(v, f) == pytest.approx(1.0)
    rng = np.random.default_rng(0)
    # soup: every face gets private copies of its vertices (jittered below the 1e-8 merge tolerance), 5 faces flipped,
    # one face duplicated with the other winding, one degenerate face, one unreferenced vertex
    flip = [1, 4, 5, 8, 11]
    soup_f = f.copy()
    soup_f[flip] = soup_f[flip][:, ::-1]
    sv = v[soup_f].reshape(-1, 3) + rng.uniform(-2e-9, 2e-9, size=(36, 3))
    sf = np.arange(36).reshape(12, 3)
    sv = np.vstack([sv, sv[[0]], sv[[2]], sv[[1]], [[5.0, 5.0, 5.0]], sv[[3]], sv[[3]], sv[[4]]])
    sf = np.vstack([sf, [[36, 37, 38]], [[40, 41, 42]]])       # duplicate of face 0 (reversed) ; degenerate (two copies of one point)
    cv, cf = meto.clean_like_trimesh(sv, sf)
    assert len(cv) == 8                                           # merged; the unreferenced vertex is gone
    assert len(cf) == 13                                          # 12 cube faces + the degenerate one; the duplicate is gone
    degenerate = [i for i, t in enumerate(cf.tolist()) if len(set(t)) < 3]
    assert len(degenerate) == 1
    solid = np.delete(cf, degenerate, axis=0)
    assert _winding_consistent(solid)
    assert _signed_volume(cv, solid) == pytest.approx(1.0, abs=1e-6)
    # a fully inverted cube comes back outward; an already clean mesh is left alone
    iv, if_ = meto.clean_like_trimesh(v, f[:, ::-1])
    assert _signed_volume(iv, if_) == pytest.approx(1.0)
    kv, kf = meto.clean_like_trimesh(v, f)
    assert np.array_equal(kv, v) and np.array_equal(kf, f) and _signed_volume(kv, kf) == pytest.approx(1.0)    # vertex ORDER kept too
    # merged vertices keep the order of their first occurrence (trimesh 4.0.5 unique_rows(..., keep_order=True)), not a sorted one
    pv = np.array([[3.0, 0, 0], [1.0, 0, 0], [9.0, 9, 9], [2.0, 0, 0], [1.0, 0, 0], [3.0, 0, 0], [0.5, 1, 0]])
    pf = np.array([[0, 1, 3], [4, 5, 6]])
    mv, mf = meto.merge_vertices(pv, pf)
    assert mv.tolist() == [[3.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0], [0.5, 1, 0]] and mf.tolist() == [[0, 1, 2], [1, 0, 3]]


def test_merge_vertices_properties_on_random_soups():
    """merge_vertices (trimesh 4.0.5 restated): idempotent, geometry-preserving (every face keeps its three corner positions), survivors in
    first-occurrence order, unreferenced vertices dropped - on random triangle soups with duplicated corners."""
    from edgerunner_amd import meto
    rng = np.random.default_rng(5)
    for trial in range(20):
        base = rng.integers(0, 6, size=(int(rng.integers(4, 30)), 3)).astype(np.float64) / 5.0     # few distinct positions -> many duplicates
        nf = int(rng.integers(1, 40))
        faces = rng.integers(0, len(base), size=(nf, 3))
        verts = np.vstack([base, rng.random((3, 3)) + 7.0])                                            # three unreferenced vertices
        mv, mf = meto.merge_vertices(verts, faces)
        assert np.array_equal(mv[mf], verts[faces])                                                    # geometry of every face unchanged
        assert len(np.unique(np.round(mv * 1e8).astype(np.int64), axis=0)) == len(mv)                  # no duplicate survivors
        assert not (mv > 6.0).any()                                                                    # unreferenced ones are gone
        # first-occurrence order: scanning the ORIGINAL vertices in index order, a position is appended when a referenced vertex shows it first
        seen, order = set(), []
        ref = set(faces.reshape(-1).tolist())
        for i, vtx in enumerate(verts):
            key = tuple(np.round(vtx * 1e8).astype(np.int64).tolist())
            if i in ref and key not in seen:
                seen.add(key); order.append(vtx)
        assert np.array_equal(mv, np.asarray(order))
        mv2, mf2 = meto.merge_vertices(mv, mf)
        assert np.array_equal(mv2, mv) and np.array_equal(mf2, mf)                                      # idempotent


def test_mesh_object_exports_like_the_reference_callers_expect(tmp_path):
    """LMM.generate returns objects with .vertices / .faces / .export(path) (what infer.py:120 uses of trimesh.Trimesh);
    they still unpack as (vertices, faces)."""
    from edgerunner_amd import meshio, meto
    v, f = _cube()
    m = meto.Mesh(v, f)
    vv, ff = m
    assert vv is m.vertices and ff is m.faces and m[0] is m.vertices and len(m) == 2
    for ext in ("ply", "obj"):
        path = str(tmp_path / f"cube.{ext}")
        assert m.export(path) == path
        rv, rf = meshio.load_mesh(path)
        assert np.allclose(rv, v, atol=1e-6) and np.array_equal(rf, f)


def test_clean_like_trimesh_multibody_and_reference_fixtures():
    from edgerunner_amd import meto
    v, f = _cube()
    # two bodies, the second one inside-out with one extra inconsistent face: each body is repaired on its own
    f2 = f[:, ::-1].copy()
    f2[3] = f2[3][::-1]
    vv = np.vstack([v, v + 3.0])
    ff = np.vstack([f, f2 + 8])
    cv, cf = meto.clean_like_trimesh(vv, ff)
    assert len(cv) == 16 and len(cf) == 24 and _winding_consistent(cf)
    lo = cf[(cv[cf][:, :, 0] < 2).all(axis=1)]
    hi = cf[(cv[cf][:, :, 0] > 2).all(axis=1)]
    assert _signed_volume(cv, lo) == pytest.approx(1.0) and _signed_volume(cv, hi) == pytest.approx(1.0)
    # reference fixture 'lRlre' (meto/tests/engine.py:66-71): a strip whose second triangle is flipped -> consistent after the clean-up
    sv = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [-1, 1, 0], [-1, 2, 0], [-2, 2, 0]], dtype=np.float64)
    sfaces = np.array([[0, 1, 2], [0, 3, 2], [0, 3, 4], [4, 3, 5], [5, 4, 6]])
    assert not _winding_consistent(sfaces)
    _, cfaces = meto.clean_like_trimesh(sv, sfaces)
    assert len(cfaces) == 5 and _winding_consistent(cfaces)
    # empty input
    ev, ef = meto.clean_like_trimesh(np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64))
    assert len(ev) == 0 and len(ef) == 0


def test_unbuilt_decoder_shapes_are_refused_with_the_option_names():
    """config_defaults['default'] = Options() is hidden_dim 1024 (core/options.py:71,155-156) and ShapeOPTConfig is generic
    (modeling_opt.py:86-134); the decode kernels stream 1536-wide rows only.  LMM says so before any device is touched."""
    import dataclasses
    from edgerunner_amd.models import LMM
    from edgerunner_amd.options import config_defaults
    with pytest.raises(NotImplementedError, match=r"hidden_dim=1024.*--hidden_dim 1536"):
        LMM(dataclasses.replace(config_defaults["ArAE"], hidden_dim=1024), "cuda:0", precision=None)
    with pytest.raises(NotImplementedError, match="intermediate_dim=4096"):
        LMM(dataclasses.replace(config_defaults["ArAE"], intermediate_dim=4096), "cuda:0", precision=None)
    with pytest.raises(NotImplementedError):
        LMM(config_defaults["default"], "cuda:0", precision=None)          # image-conditioned, hidden 1024


# ------------------------------------------------------------------ kernel selection rules (pure host logic behind the C ABI)
def _plan(batch, l_cap, heads=16, head_dim=96, hidden=1536):
    from edgerunner_amd import native
    lib = native.load_library()
    p = native.ErDecodePlan()
    native.check(lib.er_plan_decode(batch, heads, head_dim, hidden, l_cap, p), "er_plan_decode")
    return p


def test_decode_plan_single_row_versions(monkeypatch):
    from edgerunner_amd import native
    for k in ("ER_DECODE_V", "ER_ATTN_V_BATCHED", "ER_FORCE_BATCHED"):
        monkeypatch.delenv(k, raising=False)
    p = _plan(1, 2050 + 4000 + 1)                         # BASELINE configs[1]
    assert (p.batched, p.decode_version, p.attn_kernel, p.merge_launch, p.launches_per_layer) == (0, 3, native.ER_ATTN_BALANCED, 0, 5)
    assert p.attn_chunks == 16
    assert _plan(1, 8192).decode_version == 3             # 16 chunks x 512 keys
    q = _plan(1, 8193)                                    # does not fit: fixed chunks + merge kernel
    assert (q.decode_version, q.attn_kernel, q.merge_launch, q.launches_per_layer) == (2, native.ER_ATTN_SPLIT2, 1, 6)
    assert _plan(3, 4096).decode_version == 2             # rows 2..4 share the weights through the NB groups of version 2
    assert _plan(1, 4096, heads=24, head_dim=64).decode_version == 2
    monkeypatch.setenv("ER_DECODE_V", "2")
    assert _plan(1, 6051).decode_version == 2
    monkeypatch.delenv("ER_DECODE_V")
    monkeypatch.setenv("ER_FORCE_BATCHED", "1")
    f = _plan(1, 6051)
    assert (f.batched, f.decode_version, f.attn_kernel) == (1, 2, native.ER_ATTN_SPLIT1)


def test_decode_plan_batched_attention(monkeypatch):
    from edgerunner_amd import native
    for k in ("ER_DECODE_V", "ER_ATTN_V_BATCHED", "ER_FORCE_BATCHED"):
        monkeypatch.delenv(k, raising=False)
    assert _plan(32, 6051).attn_kernel == native.ER_ATTN_STREAM          # 512 (row, head) pairs: two workgroups per CU
    assert _plan(32, 18051).attn_kernel == native.ER_ATTN_STREAM         # configs[2]
    assert _plan(16, 6051).attn_kernel == native.ER_ATTN_STREAM          # 256 pairs: one workgroup per CU, still no merge launch
    p = _plan(15, 6051)
    assert (p.batched, p.attn_kernel, p.merge_launch) == (1, native.ER_ATTN_SPLIT1, 1)
    assert _plan(5, 6051).attn_kernel == native.ER_ATTN_SPLIT1
    assert _plan(4, 6051).batched == 0
    monkeypatch.setenv("ER_ATTN_V_BATCHED", "3")
    assert _plan(8, 6051).attn_kernel == native.ER_ATTN_STREAM
    monkeypatch.setenv("ER_ATTN_V_BATCHED", "1")
    assert _plan(32, 6051).attn_kernel == native.ER_ATTN_SPLIT1
    monkeypatch.setenv("ER_ATTN_V_BATCHED", "2")                         # not a value any more: auto
    assert _plan(32, 6051).attn_kernel == native.ER_ATTN_STREAM


def test_gemm_tile_choice_follows_workgroups_per_cu():
    """128x128 while it leaves >= 3 workgroups per CU (768), else 64x128, else 64x64."""
    from edgerunner_amd import native
    import os
    if os.environ.get("ER_GEMM_TILE"):
        pytest.skip("ER_GEMM_TILE forces a shape")
    lib = native.load_library()
    tile = lambda m, n, b=1: lib.er_plan_gemm_tile(m, n, b)
    assert tile(2050, 6144) == 1          # prefill fc1: 17 x 48 = 816 tiles of 128x128
    assert tile(2050, 4608) == 2          # prefill qkv: 612 -> 33 x 36 = 1188 tiles of 64x128
    assert tile(2050, 1536) == 3          # prefill out_proj / fc2: 204 -> 396 -> 64x64
    assert tile(4096, 1024) == 3          # DiT width-1024 Linears on the CFG batch
    assert tile(4096, 3072) == 1          # 32 x 24 = 768
    assert tile(32 * 2050, 1536) == 1     # a 32-sample prefill
    assert tile(2048, 2048, 16) == 1      # batched over heads
    assert lib.er_plan_gemm_tile(0, 5, 1) < 0


def test_geglu_erf_formula_accuracy():
    """The erf of the fused GEGLU epilogue (k_gemm.h erf_as7126, Abramowitz & Stegun 7.1.26) restated in float32 numpy: its truncation
    error against scipy's erf stays below 1e-6 everywhere, and the relative error of 1 + erf - what the GEGLU multiplies by - below fp16's
    half ulp for every gate value above -3.
    (The device evaluates the reciprocal and the exponential on v_rcp_f32 / v_exp_f32, ~1 ulp each: tests/test_gpu_kernels.py checks it.)"""
    import numpy as np
    from scipy.special import erf
    # (the formula's own bound is 1.5e-7; evaluated in float32, 1 - p e loses a few ulps of 1: 6e-7 worst case)
    v = np.linspace(-6.0, 6.0, 2_000_001, dtype=np.float32)
    a = np.abs(v)
    t = (np.float32(1.0) / (np.float32(0.3275911) * a + np.float32(1.0))).astype(np.float32)
    c = [np.float32(x) for x in (0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429)]
    p = t * (c[0] + t * (c[1] + t * (c[2] + t * (c[3] + t * c[4]))))
    r = np.copysign(np.float32(1.0) - p * np.exp2((-a * a * np.float32(1.4426950408889634)).astype(np.float32)), v).astype(np.float32)
    err = np.abs(r.astype(np.float64) - erf(v.astype(np.float64)))
    assert err.max() < 1e-6, err.max()
    # what the GEGLU needs is 1 + erf: its relative error stays below fp16's half ulp (2.4e-4) down to gate values of -3 (gelu = -0.004)
    sel = v > -3.0 * np.float32(0.70710678)
    rel = err[sel] / (1.0 + erf(v[sel].astype(np.float64)))
    assert rel.max() < 2.4e-4, rel.max()
