"""``meto`` mesh tokenizer surface for the decode path (reference: meto/meto/__init__.py:21-54).

The ``LR_ABSCO`` (ArAE preset) and ``LR`` backends - the two ``Options.meto_backend`` admits - are native C++ behind
``er_meto_decode`` (the step right after the decode loop, row f1) and ``er_meto_encode`` (training data /
partial-mesh completion, row f4), both bit-exact against the reference's own engines.
Also the reference's ``detokenize_mesh`` / ``save_mesh`` (core/provider.py:39-66,112-147)
without trimesh: meshes are ``(vertices float64 [V,3], faces int64 [F,3])`` tuples.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import native


class Engine:
    def __init__(self, discrete_bins: int, verbose: bool = False, backend: str = "LR_ABSCO"):
        if backend not in ("LR_ABSCO", "LR"):
            raise NotImplementedError(f"meto backend {backend!r}: LR_ABSCO and LR (the values of Options.meto_backend) are built")
        self.backend = backend
        self._backend_id = native.ER_METO_LR if backend == "LR" else native.ER_METO_LR_ABSCO
        self.discrete_bins = discrete_bins
        self.verbose = verbose
        self.num_base_tokens = discrete_bins * 2 if backend == "LR" else discrete_bins      # meto/meto/__init__.py:30-37
        self.num_special_tokens = 3
        self.num_tokens = self.num_base_tokens + self.num_special_tokens
        self._lib = native.load_library()

    def encode(self, vertices, faces) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """vertices [N,3] float in [-1,1], faces [M,3] int -> (tokens, face_order, face_type)."""
        v = np.ascontiguousarray(np.asarray(vertices, dtype=np.float32).reshape(-1, 3))
        f = np.ascontiguousarray(np.asarray(faces, dtype=np.int32).reshape(-1, 3))
        nf = len(f)
        per = 2 if self.backend == "LR" else 1
        tok = np.empty((max(1, 10 * per * nf),), np.int32)
        order = np.empty((max(1, per * nf),), np.int32)
        ftype = np.empty((max(1, per * nf),), np.int32)
        nt, no = C.c_int32(), C.c_int32()
        i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        native.check(self._lib.er_meto_encode(v.ctypes.data_as(f32p), len(v), f.ctypes.data_as(i32p), nf, self.discrete_bins, self._backend_id,
                                              tok.ctypes.data_as(i32p), C.byref(nt), order.ctypes.data_as(i32p),
                                              ftype.ctypes.data_as(i32p), C.byref(no)), "er_meto_encode")
        return (tok[: nt.value].astype(np.int64), order[: no.value].astype(np.int64), ftype[: no.value].astype(np.int64))

    def decode(self, tokens) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """tokens: meto ids [N] -> (vertices [V,3], faces [F,3], face_type [K])."""
        tok = np.ascontiguousarray(np.asarray(tokens).reshape(-1), dtype=np.int32)
        n = len(tok)
        v = np.empty((n // 3 + 3, 3), np.float32)
        f = np.empty((n // 4 + 2, 3), np.int32)
        t = np.empty((n // 4 + 3,), np.int32)
        nv, nf, nt = C.c_int32(), C.c_int32(), C.c_int32()
        i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        native.check(self._lib.er_meto_decode(tok.ctypes.data_as(i32p), n, self.discrete_bins, self._backend_id, v.ctypes.data_as(f32p),
                                              f.ctypes.data_as(i32p), t.ctypes.data_as(i32p), C.byref(nv), C.byref(nf),
                                              C.byref(nt)), "er_meto_decode")
        return (v[: nv.value].astype(np.float64), f[: nf.value].astype(np.int64), t[: nt.value].astype(np.int64))


def get_tokenizer(opt):
    """core/utils.py:78-86."""
    if opt.use_meto:
        tok = Engine(discrete_bins=opt.discrete_bins, backend=opt.meto_backend)
        return tok, tok.num_tokens + 3
    return None, opt.discrete_bins + 3


def _canonical_faces(vertices: np.ndarray, faces: np.ndarray, keys):
    """Shared by the tokenizer-less layout: sort vertices by `keys` (np.lexsort order), re-index the faces,
    rotate each face so its lowest vertex comes first, sort the faces lexicographically."""
    sort_inds = np.lexsort(keys)
    vertices = vertices[sort_inds]
    faces = np.argsort(sort_inds)[faces]
    start = faces.argmin(axis=1)
    rolled = np.take_along_axis(np.concatenate([faces, faces[:, :2]], axis=1), start[:, None] + np.arange(3)[None, :], axis=1)
    return vertices, np.array(sorted(rolled.tolist()), dtype=faces.dtype).reshape(-1, 3)


def tokenize_mesh(vertices, faces, discrete_bins: int, tokenizer: Optional[Engine] = None) -> np.ndarray:
    """core/provider.py:69-110 (mesh -> model ids, without BOS/EOS): meto stream, or 9 coordinates per
    face (vertices sorted z-y-x, faces canonicalised) when there is no tokenizer; ids are offset by 3."""
    vertices, faces = np.asarray(vertices), np.asarray(faces)
    if tokenizer is None:
        vertices, faces = _canonical_faces(vertices, faces, vertices.T)
        vertices = vertices[:, [2, 1, 0]]
        coords = ((vertices[faces] + 1) * 0.5 * discrete_bins).clip(0, discrete_bins - 1).astype(np.int32)
        tokens = coords.reshape(-1)
    else:
        tokens, _, _ = tokenizer.encode(vertices, faces)
    return tokens + 3


def sort_mesh(vertices, faces):
    """meto/meto/__init__.py:93-115: vertices in y-z-x order, faces canonicalised."""
    vertices, faces = np.asarray(vertices), np.asarray(faces)
    return _canonical_faces(vertices, faces, (vertices[:, 0], vertices[:, 2], vertices[:, 1]))


def detokenize_mesh(tokens, discrete_bins: Optional[int] = None, tokenizer: Optional[Engine] = None):
    """core/provider.py:112-147 (model ids -> vertices/faces)."""
    tokens = np.asarray(tokens) - 3
    if tokenizer is None:
        if len(tokens) % 9 != 0:
            print(f"[WARN] tokens len is {len(tokens)} % 9 != 0, trimming...")
            tokens = tokens[: -(len(tokens) % 9)]
        invalid = (tokens < 0).reshape(-1, 9).any(axis=1)
        coords = tokens.reshape(-1, 3)
        if discrete_bins is None:
            vertices = coords / coords.max() * 2 - 1
        else:
            vertices = (coords + 0.5) / discrete_bins * 2 - 1
        faces = np.arange(len(vertices)).reshape(-1, 3)[~invalid]
        return vertices[:, [2, 1, 0]], faces
    vertices, faces, _ = tokenizer.decode(tokens)
    return vertices, faces


def merge_and_dedupe(vertices: np.ndarray, faces: np.ndarray):
    """The index-level part of save_mesh's clean-up (core/provider.py:55-58): merge identical
    vertices, drop duplicate and degenerate faces.  (trimesh's ``fix_normals`` winding repair is
    not reproduced.)"""
    if len(vertices) == 0:
        return vertices, faces
    uniq, inv = np.unique(np.round(vertices, 8), axis=0, return_inverse=True)
    faces = inv.reshape(-1)[faces]
    keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    faces = faces[keep]
    _, first = np.unique(np.sort(faces, axis=1), axis=0, return_index=True)
    return uniq, faces[np.sort(first)]


def save_mesh(tokens, opt, path=None, tokenizer=None, clean=True, verbose=False):
    """core/provider.py:39-66: ids -> (vertices, faces) [-> .ply when path is given]."""
    tokens = np.asarray(tokens)
    eos = np.nonzero(tokens == opt.eos_token_id)[0]
    if len(eos) > 0:
        tokens = tokens[: eos[0]]
    vertices, faces = detokenize_mesh(tokens, opt.discrete_bins, tokenizer=tokenizer)
    if verbose:
        print(f"[INFO] vertices: {vertices.shape[0]}, faces: {faces.shape[0]}")
    if clean:
        vertices, faces = merge_and_dedupe(vertices, faces)
        if verbose:
            print(f"[INFO] cleaned vertices: {vertices.shape[0]}, faces: {faces.shape[0]}")
    if path is not None:
        from .meshio import save_ply
        save_ply(path, vertices, faces)
    return vertices, faces
