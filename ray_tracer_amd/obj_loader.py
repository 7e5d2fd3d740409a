"""Minimal Wavefront OBJ reader for mesh.rs:101-167 / obj_loader.rs:21-83.

The reference loads OBJ files through the third-party crate `tobj` (not vendored) with
`triangulate: true, single_index: true`; what reaches the hot path is a list of
triangles whose positions went through f32 (tobj stores f32, mesh.rs:137-153 widens to
f64).  This reader keeps exactly that: `v` lines parsed to f32, faces fan-triangulated
(v0, v_i, v_i+1), negative indices relative to the end.  Parity with tobj on real assets
is unpinned: the reference ships no model file.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np


def load_obj(path: str, scale: Optional[float] = None, offset: Optional[Sequence[float]] = None) -> np.ndarray:
    """(N, 3, 3) float64 triangle vertices read by the library's own reader (rt_obj_load).  Without `scale` the
    positions are kept as Mesh::from_obj keeps them (mesh.rs:137-153); with it every component becomes
    p * scale + offset in f64, as ObjModel::load computes it (obj_loader.rs:51-67)."""
    from . import _ffi as F
    tris = C.POINTER(F.rt_triangle)()
    n = C.c_uint64()
    so = None
    if scale is not None:
        off = (0.0, 0.0, 0.0) if offset is None else offset
        so = (C.c_double * 4)(float(scale), float(off[0]), float(off[1]), float(off[2]))
    F.check(F.lib.rt_obj_load(path.encode(), so, 0, C.byref(tris), C.byref(n)))
    try:
        if n.value == 0:
            return np.zeros((0, 3, 3), dtype=np.float64)
        raw = np.ctypeslib.as_array(C.cast(tris, C.POINTER(C.c_double)), shape=(n.value, 10))  # 9 doubles + mat/_pad
        return np.ascontiguousarray(raw[:, :9].reshape(n.value, 3, 3).copy())
    finally:
        F.lib.rt_obj_free(tris)


def load_obj_triangles(path: str) -> np.ndarray:
    """The same reader in Python, kept as an independent cross-check of rt_obj_load in the tests (it parses decimals
    to f64 and then rounds to f32, which can differ from a direct decimal -> f32 conversion in the last bit of a
    coordinate in rare cases).  Returns an (N, 3, 3) float64 array of triangle vertices."""
    positions = []
    faces = []
    with open(path, "r", encoding="utf-8", errors="replace") as fh:
        for line in fh:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v":
                if len(parts) < 4:
                    raise ValueError(f"{path}: malformed vertex line: {line!r}")
                positions.append((float(parts[1]), float(parts[2]), float(parts[3])))
            elif parts[0] == "f":
                idx = []
                for tok in parts[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(positions) + i)
                if len(idx) < 3:
                    continue
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    if not faces:
        return np.zeros((0, 3, 3), dtype=np.float64)
    pos = np.asarray(positions, dtype=np.float32).astype(np.float64)  # f32 storage, widened
    f = np.asarray(faces, dtype=np.int64)
    if f.min() < 0 or f.max() >= len(pos):
        raise ValueError(f"{path}: face index out of range")
    return np.ascontiguousarray(pos[f])
