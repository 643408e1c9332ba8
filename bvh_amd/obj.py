"""Wavefront OBJ -> triangle soup with the semantics of the reference's loader (test/load_obj.cpp:57-96), which is what its
benchmark and examples feed the builders: only `v` and `f` records count, an index may be `i`, `i/t`, `i//n` or `i/t/n` (the
position index is the first field), negative indices count back from the vertices read so far, and a polygon with k vertices
becomes the fan (p0, p[i-1], p[i]), i = 2 .. k-1. Host-side text parsing (the reference does it on the host too); the result
goes to the device as one AoS array of `Tri` {p0, p1, p2} (tri.h:16)."""
from __future__ import annotations

import numpy as np


def load_obj(path: str, dtype=np.float32) -> np.ndarray:
    """(n, 9) array of triangles {p0, p1, p2}; empty (0, 9) when the file has no faces (the reference returns an empty vector)."""
    verts = []
    tris = []
    with open(path, "r", errors="replace") as f:
        for line in f:
            s = line.strip()
            if not s or s[0] == "#":
                continue
            if s[0] == "v" and len(s) > 1 and s[1].isspace():
                tok = s.split()
                verts.append((float(np.float32(tok[1])), float(np.float32(tok[2])), float(np.float32(tok[3]))))   # strtof
            elif s[0] == "f" and len(s) > 1 and s[1].isspace():
                idx = []
                for t in s.split()[1:]:
                    head = t.split("/", 1)[0]
                    if not head or not (head[0].isdigit() or head[0] == "-"):
                        break                                      # read_index() stops the face at the first non-index token
                    k = int(head)
                    j = len(verts) + k if k < 0 else k - 1
                    if not 0 <= j < len(verts):
                        raise ValueError(f"{path}: face index {k} outside the {len(verts)} vertices read so far")
                    idx.append(j)
                for i in range(2, len(idx)):
                    tris.append(verts[idx[0]] + verts[idx[i - 1]] + verts[idx[i]])
    return np.asarray(tris, dtype=np.float64).astype(dtype).reshape(-1, 9)


def save_obj(path: str, tris9: np.ndarray) -> None:
    """Writes a triangle soup as OBJ (three `v` and one `f` per triangle, 9 significant digits: float32 round-trips exactly)."""
    t = np.asarray(tris9).reshape(-1, 3, 3)
    with open(path, "w") as f:
        for tri in t:
            for p in tri:
                f.write(f"v {float(p[0]):.9g} {float(p[1]):.9g} {float(p[2]):.9g}\n")
            f.write("f -3 -2 -1\n")
