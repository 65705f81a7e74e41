"""Minimal mesh / point-cloud I/O for ``infer.py`` (the reference uses trimesh + kiui,
which are absent in this image): OBJ and ASCII/binary-little-endian PLY triangle meshes,
``normalize_mesh`` (core/utils.py:69-75) and area-weighted surface sampling
(what ``trimesh.Trimesh.sample`` does, reference infer.py:89-90)."""
from __future__ import annotations

import os
import struct
from typing import Tuple

import numpy as np


def normalize_mesh(vertices: np.ndarray, bound: float = 0.95) -> np.ndarray:
    vmin, vmax = vertices.min(0), vertices.max(0)
    center = (vmax + vmin) / 2
    scale = 2 * bound / np.max(vmax - vmin)
    return (vertices - center) * scale


def load_obj(path: str) -> Tuple[np.ndarray, np.ndarray]:
    v, f = [], []
    with open(path) as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                v.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                idx = [int(t.split("/")[0]) for t in p[1:]]
                idx = [i - 1 if i > 0 else len(v) + i for i in idx]
                for k in range(1, len(idx) - 1):          # fan-triangulate polygons
                    f.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(v, np.float64), np.asarray(f, np.int64)


def load_ply(path: str) -> Tuple[np.ndarray, np.ndarray]:
    with open(path, "rb") as fh:
        assert fh.readline().strip() == b"ply"
        fmt, nv, nf, vprops, in_vertex = None, 0, 0, [], False
        while True:
            line = fh.readline().strip().decode()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element vertex"):
                nv, in_vertex = int(line.split()[-1]), True
            elif line.startswith("element face"):
                nf, in_vertex = int(line.split()[-1]), False
            elif line.startswith("element"):
                in_vertex = False
            elif line.startswith("property") and in_vertex:
                vprops.append(line.split()[1:])
            elif line == "end_header":
                break
        if fmt == "ascii":
            rows = [fh.readline().split() for _ in range(nv)]
            v = np.asarray([[float(r[0]), float(r[1]), float(r[2])] for r in rows], np.float64)
            f = []
            for _ in range(nf):
                p = [int(x) for x in fh.readline().split()]
                for k in range(2, p[0]):
                    f.append([p[1], p[k], p[k + 1]])
            return v, np.asarray(f, np.int64)
        assert fmt == "binary_little_endian", fmt
        code = {"float": "f", "float32": "f", "double": "d", "float64": "d", "uchar": "B", "uint8": "B", "int": "i",
                "int32": "i", "uint": "I", "uint32": "I", "short": "h", "ushort": "H", "char": "b"}
        vfmt = "<" + "".join(code[p[0]] for p in vprops)
        vsz = struct.calcsize(vfmt)
        names = [p[1] for p in vprops]
        ix, iy, iz = names.index("x"), names.index("y"), names.index("z")
        v = np.empty((nv, 3), np.float64)
        for i in range(nv):
            rec = struct.unpack(vfmt, fh.read(vsz))
            v[i] = (rec[ix], rec[iy], rec[iz])
        f = []
        for _ in range(nf):
            n = struct.unpack("<B", fh.read(1))[0]
            idx = struct.unpack("<" + "i" * n, fh.read(4 * n))
            for k in range(1, n - 1):
                f.append([idx[0], idx[k], idx[k + 1]])
        return v, np.asarray(f, np.int64)


def load_mesh(path: str) -> Tuple[np.ndarray, np.ndarray]:
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        return load_obj(path)
    if ext == ".ply":
        return load_ply(path)
    raise ValueError(f"unsupported mesh format {ext!r} (obj, ply; or pass a .npy point cloud)")


def sample_surface(vertices: np.ndarray, faces: np.ndarray, count: int, rng=None) -> np.ndarray:
    """Uniform (area-weighted) surface samples."""
    rng = np.random.default_rng() if rng is None else rng
    tri = vertices[faces]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    fi = rng.choice(len(faces), size=count, p=area / area.sum())
    u = rng.random((count, 2))
    flip = u.sum(1) > 1
    u[flip] = 1 - u[flip]
    t = tri[fi]
    return t[:, 0] + u[:, :1] * (t[:, 1] - t[:, 0]) + u[:, 1:] * (t[:, 2] - t[:, 0])


def save_ply(path: str, vertices: np.ndarray, faces: np.ndarray) -> None:
    with open(path, "w") as fh:
        fh.write("ply\nformat ascii 1.0\n")
        fh.write(f"element vertex {len(vertices)}\nproperty float x\nproperty float y\nproperty float z\n")
        fh.write(f"element face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n")
        for v in vertices:
            fh.write(f"{v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
        for f in faces:
            fh.write(f"3 {int(f[0])} {int(f[1])} {int(f[2])}\n")


def save_obj(path: str, vertices: np.ndarray, faces: np.ndarray) -> None:
    with open(path, "w") as fh:
        for v in vertices:
            fh.write(f"v {v[0]:.8f} {v[1]:.8f} {v[2]:.8f}\n")
        for f in faces:
            fh.write(f"f {int(f[0]) + 1} {int(f[1]) + 1} {int(f[2]) + 1}\n")


def save_points_obj(path: str, points: np.ndarray) -> None:
    with open(path, "w") as fh:
        for p in points:
            fh.write(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}\n")
