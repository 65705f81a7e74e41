"""TEST INFRASTRUCTURE ONLY (build container).  Produces tests/golden/meto_lr_absco.npz and meto_lr.npz with the
reference's own meto engines (compiled by oracle/build_ref.sh into oracle/_ref/): token streams of
procedurally generated meshes (encode) and the decode of those and of random / malformed streams."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
subprocess.run(["bash", os.path.join(HERE, "build_ref.sh")], check=True)
sys.path.insert(0, os.path.join(HERE, "_ref"))
import _meto  # noqa: E402  (the reference's pybind module)


def grid(n):
    xs = np.linspace(-0.9, 0.9, n)
    v = np.array([[x, y, 0.1 * np.sin(3 * x) * np.cos(2 * y)] for y in xs for x in xs])
    f = []
    for j in range(n - 1):
        for i in range(n - 1):
            a, b, c, d = j * n + i, j * n + i + 1, (j + 1) * n + i, (j + 1) * n + i + 1
            f += [[a, b, d], [a, d, c]]
    return v, np.array(f)


def cube():
    v = np.array([[x, y, z] for x in (-0.5, 0.5) for y in (-0.5, 0.5) for z in (-0.5, 0.5)], float)
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6],
                  [0, 6, 4], [1, 5, 7], [1, 7, 3]])
    return v, f


def torus(nu, nv):
    v, f = [], []
    for i in range(nu):
        for j in range(nv):
            a, b = 2 * np.pi * i / nu, 2 * np.pi * j / nv
            v.append([(0.6 + 0.25 * np.cos(b)) * np.cos(a), (0.6 + 0.25 * np.cos(b)) * np.sin(a), 0.25 * np.sin(b)])
    for i in range(nu):
        for j in range(nv):
            a, b = i * nv + j, i * nv + (j + 1) % nv
            c, d = ((i + 1) % nu) * nv + j, ((i + 1) % nu) * nv + (j + 1) % nv
            f += [[a, c, d], [a, d, b]]
    return np.array(v), np.array(f)


def random_stream(rng, n_ops, n_coord):
    t = []
    for k in range(n_ops):
        if k == 0 or rng.random() < 0.15:
            t += [2] + list(rng.integers(3, 3 + n_coord, 9))
        else:
            t += [int(rng.integers(0, 2))] + list(rng.integers(3, 3 + n_coord, 3))
    return np.array(t, np.int64)


def main(backend="LR_ABSCO"):
    out = {}
    bins = 512
    make = {"LR_ABSCO": _meto.Engine_LR_ABSCO, "LR": _meto.Engine_LR}[backend]
    n_coord = bins if backend == "LR_ABSCO" else 2 * bins          # coordinate alphabet (meto/meto/__init__.py:30-37)
    eng = make(bins, False)
    meshes = {"cube": cube(), "grid7": grid(7), "torus": torus(12, 8),
              "two_parts": (np.concatenate([cube()[0] * 0.5 - 0.4, cube()[0] * 0.5 + 0.4]),
                            np.concatenate([cube()[1], cube()[1] + 8]))}
    streams = {}
    for name, (v, f) in meshes.items():
        tokens, _, _ = eng.encode(v.astype(np.float32).tolist(), f.tolist())
        streams[name] = np.asarray(tokens, np.int64)
    rng = np.random.default_rng(0)
    for k in range(6):
        streams[f"rand{k}"] = random_stream(rng, int(rng.integers(1, 400)), n_coord)
    base = streams["grid7"]
    streams["trunc_mid_vertex"] = base[:-2]
    streams["trunc_in_bom"] = base[:5]
    streams["coord_where_op"] = np.concatenate([base[:10], [77], base[10:]])
    streams["negative_ids"] = np.concatenate([base[:14], [-3, -1, -2, -3], base[14:]])
    streams["empty"] = np.zeros((0,), np.int64)
    streams["starts_with_op"] = np.concatenate([[0, 10, 11, 12], base])[4:]  # well-formed after trimming
    for name, t in streams.items():
        v, f, ft = make(bins, False).decode(t.tolist())
        out[f"{name}.tokens"] = t
        out[f"{name}.vertices"] = np.asarray(v, np.float64).reshape(-1, 3)
        out[f"{name}.faces"] = np.asarray(f, np.int64).reshape(-1, 3)
        out[f"{name}.face_type"] = np.asarray(ft, np.int64)
        print(name, len(t), out[f"{name}.vertices"].shape, out[f"{name}.faces"].shape)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"meto_{backend.lower()}.npz"), **out)


if __name__ == "__main__":
    for b in (sys.argv[1:] or ["LR_ABSCO", "LR"]):
        main(b)
