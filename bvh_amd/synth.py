"""Deterministic synthetic meshes and ray batches.

No OBJ scene other than the 36-triangle Cornell box ships with the reference and there is no network, so
the bench and the parity tests run on procedural meshes generated here with a counter-based splitmix64
generator (identical bytes on every machine, any chunking).

Meshes are returned as float32/float64 arrays of shape (n, 9) = ``bvh::v2::Tri`` {p0, p1, p2}
(reference test/load_obj.cpp:104-117 produces the same layout from an OBJ file).
Rays are (n, 8) = ``bvh::v2::Ray`` {org, dir, tmin, tmax} (reference src/bvh/v2/ray.h:16-27).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
FLT_MAX = float(np.finfo(np.float32).max)


def splitmix64(seed: int, count: int, stream: int = 0) -> np.ndarray:
    """count 64-bit values; value i depends only on (seed, stream, i)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream) * np.uint64(0xDA942042E4DD58B5)
        z = base + (np.arange(1, count + 1, dtype=np.uint64)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, count: int, stream: int = 0) -> np.ndarray:
    """float64 values k / 2^24, k in [0, 2^24): exactly representable in float32."""
    return (splitmix64(seed, count, stream) >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


# ----------------------------------------------------------------------------------------------
# Meshes
# ----------------------------------------------------------------------------------------------

def soup(n: int, seed: int = 7, jitter: float = 0.005, dtype=np.float32) -> np.ndarray:
    """M3 "soup": n random triangles, centre uniform in the unit cube, vertices within +-jitter."""
    c = uniform01(seed, 3 * n, 0).reshape(n, 1, 3)
    d = (uniform01(seed, 9 * n, 1).reshape(n, 3, 3) * 2.0 - 1.0) * jitter
    return (c + d).astype(dtype).reshape(n, 9)


def terrain(n: int, dtype=np.float32) -> np.ndarray:
    """M2 "terrain": regular height-field, 2 triangles per cell, side = floor(sqrt(n/2)); tie-heavy."""
    side = int(np.floor(np.sqrt(n / 2)))
    g = np.arange(side + 1, dtype=np.float64) / side
    x, z = np.meshgrid(g, g, indexing="ij")
    y = 0.05 * np.sin(40 * x) * np.cos(37 * z) + 0.3 * np.sin(3 * x + 2 * z)
    p = np.stack([x, y, z], axis=-1)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    t0 = np.stack([a, b, c], axis=-2).reshape(-1, 9)
    t1 = np.stack([a, c, d], axis=-2).reshape(-1, 9)
    return np.stack([t0, t1], axis=1).reshape(-1, 9).astype(dtype)


def _grid_patch(origin, du, dv, nu, nv, height=None):
    """(nu x nv) quad patch -> 2*nu*nv triangles; optional per-vertex displacement height(u,v)->(.., 3)."""
    u = np.arange(nu + 1, dtype=np.float64) / nu
    v = np.arange(nv + 1, dtype=np.float64) / nv
    uu, vv = np.meshgrid(u, v, indexing="ij")
    p = np.asarray(origin, dtype=np.float64) + uu[..., None] * np.asarray(du, dtype=np.float64) \
        + vv[..., None] * np.asarray(dv, dtype=np.float64)
    if height is not None:
        p = p + height(uu, vv)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    t0 = np.stack([a, b, c], axis=-2).reshape(-1, 9)
    t1 = np.stack([a, c, d], axis=-2).reshape(-1, 9)
    return np.concatenate([t0, t1], axis=0)


def _column(cx, cz, y0, y1, radius, nseg, nring):
    th = np.arange(nseg + 1, dtype=np.float64) / nseg * 2 * np.pi
    yy = y0 + (y1 - y0) * np.arange(nring + 1, dtype=np.float64) / nring
    t, y = np.meshgrid(th, yy, indexing="ij")
    r = radius * (1.0 + 0.08 * np.sin(8 * t) + 0.05 * np.cos(6.0 * (y - y0) / (y1 - y0) * np.pi))
    p = np.stack([cx + r * np.cos(t), y, cz + r * np.sin(t)], axis=-1)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
    return np.concatenate([np.stack([a, b, c], axis=-2).reshape(-1, 9),
                           np.stack([a, c, d], axis=-2).reshape(-1, 9)], axis=0)


def sponza_proxy(n: int = 262144, dtype=np.float32) -> np.ndarray:
    """M1: architectural mix in a 30 x 12 x 18 hall standing in for Sponza (not available offline).

    Large coarsely tessellated walls/floor/ceiling + two rows of finely tessellated fluted columns +
    arches + wavy drapes: strongly non-uniform triangle sizes, long thin boxes, heavy overlap — the
    features that make Sponza a harder SAH case than a uniform soup. Exactly n triangles (the drapes
    absorb the remainder; the tail is padded with small floor tiles).
    """
    parts = []
    L, H, W = 30.0, 12.0, 18.0
    parts.append(_grid_patch((0, 0, 0), (L, 0, 0), (0, 0, W), 40, 24))            # floor
    parts.append(_grid_patch((0, H, 0), (0, 0, W), (L, 0, 0), 12, 20))            # ceiling
    parts.append(_grid_patch((0, 0, 0), (0, H, 0), (L, 0, 0), 8, 20))             # walls
    parts.append(_grid_patch((0, 0, W), (L, 0, 0), (0, H, 0), 20, 8))
    parts.append(_grid_patch((0, 0, 0), (0, 0, W), (0, H, 0), 12, 8))
    parts.append(_grid_patch((L, 0, 0), (0, H, 0), (0, 0, W), 8, 12))
    used = sum(len(p) for p in parts)
    ncol = 2 * 10
    col_budget = int(0.45 * n)
    nseg = 48
    nring = max(4, col_budget // (ncol * nseg * 2))
    for side_z in (4.5, 13.5):
        for i in range(10):
            parts.append(_column(2.0 + i * 2.9, side_z, 0.0, 7.5, 0.45, nseg, nring))
    used = sum(len(p) for p in parts)
    # arches between columns: half-tori sections
    arch_budget = int(0.20 * n)
    na = 2 * 9
    nu = 32
    nv = max(4, arch_budget // (na * nu * 2))
    for side_z in (4.5, 13.5):
        for i in range(9):
            x0 = 2.0 + i * 2.9
            def arch(uu, vv, x0=x0, side_z=side_z):
                ang = uu * np.pi
                tube = vv * 2 * np.pi
                R, r = 1.45, 0.28
                x = x0 + 1.45 - (R + r * np.cos(tube)) * np.cos(ang)
                y = 7.5 + (R + r * np.cos(tube)) * np.sin(ang)
                z = side_z + r * np.sin(tube)
                return np.stack([x, y, z], axis=-1)
            parts.append(_grid_patch((0, 0, 0), (0, 0, 0), (0, 0, 0), nu, nv, height=arch))
    used = sum(len(p) for p in parts)
    # drapes: wavy hanging sheets across the nave absorb the remainder
    remaining = n - used
    ndrape = 6
    per = remaining // ndrape
    for i in range(ndrape):
        cnt = per if i + 1 < ndrape else remaining - per * (ndrape - 1)
        nu_d = max(1, int(np.sqrt(cnt / 2)))
        nv_d = max(1, cnt // (2 * nu_d))
        x = 4.0 + i * 4.2
        def wave(uu, vv, i=i):
            return np.stack([0.35 * np.sin(9 * vv + i) * (0.3 + uu), 0.0 * uu, 0.15 * np.sin(14 * uu + 5 * vv)], axis=-1)
        parts.append(_grid_patch((x, 10.5, 5.2), (0, -5.0, 0), (0, 0, 7.6), nu_d, nv_d, height=wave))
    tris = np.concatenate(parts, axis=0)
    if len(tris) < n:                                                             # pad with small floor tiles
        k = n - len(tris)
        u = uniform01(11, 2 * k).reshape(k, 2)
        x = u[:, 0] * (L - 0.2)
        z = u[:, 1] * (W - 0.2)
        y = np.full(k, 0.01)
        pad = np.stack([x, y, z, x + 0.2, y, z, x, y, z + 0.2], axis=-1)
        tris = np.concatenate([tris, pad], axis=0)
    return tris[:n].astype(dtype)


def spheres(n: int, seed: int = 21, rmin: float = 0.002, rmax: float = 0.006, dtype=np.float64) -> np.ndarray:
    """M4: n spheres {center, radius}, centre uniform in the unit cube, radius in [rmin, rmax]."""
    c = uniform01(seed, 3 * n, 0).reshape(n, 3)
    r = rmin + (rmax - rmin) * uniform01(seed, n, 1).reshape(n, 1)
    return np.concatenate([c, r], axis=1).astype(dtype)


def circles(n: int, seed: int = 33, rmin: float = 0.0005, rmax: float = 0.004, dtype=np.float32) -> np.ndarray:
    """2D families: n circles {center.x, center.y, radius} (Sphere<T, 2>), centre uniform in the unit square."""
    c = uniform01(seed, 2 * n, 0).reshape(n, 2)
    r = rmin + (rmax - rmin) * uniform01(seed, n, 1).reshape(n, 1)
    return np.ascontiguousarray(np.concatenate([c, r], axis=1).astype(dtype))


def rays_2d(n: int, lo=(0.0, 0.0), hi=(1.0, 1.0), seed: int = 1234, dtype=np.float32, scale: float = 1.1, segment: bool = False) -> np.ndarray:
    """Ray<T, 2> {org[2], dir[2], tmin, tmax}: origin uniform in the box scaled about its centre, direction uniform on the
    circle; with `segment` the shadow-ray form (unnormalised direction between two points, t in [1e-4, 1 - 1e-4])."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    out = np.empty((n, 6), dtype=dtype)
    if segment:
        a = lo + uniform01(seed, 2 * n, 0).reshape(n, 2) * (hi - lo)
        b = lo + uniform01(seed, 2 * n, 1).reshape(n, 2) * (hi - lo)
        out[:, 0:2], out[:, 2:4], out[:, 4], out[:, 5] = a, b - a, 1e-4, 1.0 - 1e-4
        return out
    ctr, half = 0.5 * (lo + hi), 0.5 * (hi - lo) * scale
    org = ctr + (2.0 * uniform01(seed, 2 * n, 0).reshape(n, 2) - 1.0) * half
    phi = 2.0 * np.pi * uniform01(seed, n, 1)
    out[:, 0:2], out[:, 2], out[:, 3], out[:, 4], out[:, 5] = org, np.cos(phi), np.sin(phi), 0.0, np.finfo(dtype).max
    return out


# the reference's own scene (test/scenes/cornell_box.obj: 18 quads = 36 triangles, light + floor + ceiling + walls + two boxes), as
# vertex-index quads so that no file is needed at run time
_CORNELL_QUADS = [
    ((-0.24, 1.98, 0.16), (-0.24, 1.98, -0.22), (0.23, 1.98, -0.22), (0.23, 1.98, 0.16)),          # light
    ((-1.01, 0.0, 0.99), (1.0, 0.0, 0.99), (1.0, 0.0, -1.04), (-0.99, 0.0, -1.04)),                # floor
    ((-1.02, 1.99, 0.99), (-1.02, 1.99, -1.04), (1.0, 1.99, -1.04), (1.0, 1.99, 0.99)),            # ceiling
    ((-0.99, 0.0, -1.04), (1.0, 0.0, -1.04), (1.0, 1.99, -1.04), (-1.02, 1.99, -1.04)),            # back wall
    ((1.0, 0.0, -1.04), (1.0, 0.0, 0.99), (1.0, 1.99, 0.99), (1.0, 1.99, -1.04)),                  # right wall
    ((-1.01, 0.0, 0.99), (-0.99, 0.0, -1.04), (-1.02, 1.99, -1.04), (-1.02, 1.99, 0.99)),          # left wall
    ((0.53, 0.6, 0.75), (0.7, 0.6, 0.17), (0.13, 0.6, 0.0), (-0.05, 0.6, 0.57)),                   # short box
    ((-0.05, 0.0, 0.57), (-0.05, 0.6, 0.57), (0.13, 0.6, 0.0), (0.13, 0.0, 0.0)),
    ((0.53, 0.0, 0.75), (0.53, 0.6, 0.75), (-0.05, 0.6, 0.57), (-0.05, 0.0, 0.57)),
    ((0.7, 0.0, 0.17), (0.7, 0.6, 0.17), (0.53, 0.6, 0.75), (0.53, 0.0, 0.75)),
    ((0.13, 0.0, 0.0), (0.13, 0.6, 0.0), (0.7, 0.6, 0.17), (0.7, 0.0, 0.17)),
    ((-0.53, 1.2, 0.09), (0.04, 1.2, -0.09), (-0.14, 1.2, -0.67), (-0.71, 1.2, -0.49)),            # tall box
    ((-0.53, 0.0, 0.09), (-0.53, 1.2, 0.09), (-0.71, 1.2, -0.49), (-0.71, 0.0, -0.49)),
    ((-0.71, 0.0, -0.49), (-0.71, 1.2, -0.49), (-0.14, 1.2, -0.67), (-0.14, 0.0, -0.67)),
    ((-0.14, 0.0, -0.67), (-0.14, 1.2, -0.67), (0.04, 1.2, -0.09), (0.04, 0.0, -0.09)),
    ((0.04, 0.0, -0.09), (0.04, 1.2, -0.09), (-0.53, 1.2, 0.09), (-0.53, 0.0, 0.09)),
]


def cornell_tessellated(n: int = 1_000_000, dtype=np.float32) -> np.ndarray:
    """A Cornell-box-like room (the layout of the reference's test scene: light, floor, ceiling, three walls, a short and a tall
    box) with every quad tessellated into k x k cells of two triangles, k chosen so that the total is close to n: large planar
    regions meeting at right angles around mostly EMPTY space — structurally unlike the soup (volume-filling), the terrain (one
    sheet) and the Sponza proxy (columns and drapes). Used to check the traversal heuristics outside the scenes they were fitted on."""
    k = max(1, int(round(np.sqrt(n / (2.0 * len(_CORNELL_QUADS))))))
    parts = []
    for q in _CORNELL_QUADS:
        p0, p1, p2, p3 = (np.asarray(v, np.float64) for v in q)
        parts.append(_grid_patch(p0, p1 - p0, p3 - p0, k, k, height=lambda uu, vv, p0=p0, p1=p1, p2=p2, p3=p3:
                                 (uu * vv)[..., None] * (p2 - p1 - p3 + p0)))        # bilinear: exact for the (slightly) non-planar quads
    return np.concatenate(parts, axis=0).astype(dtype)


def clusters(n: int = 1_000_000, seed: int = 19, n_clusters: int = 400, dtype=np.float32) -> np.ndarray:
    """n small triangles in `n_clusters` blobs of very different size and density (cluster radius 0.002 .. 0.08, populations
    following a power law) scattered in the unit cube: vegetation / debris-like, heavy overlap inside a blob, empty space between."""
    u = uniform01(seed, 5 * n_clusters, 0).reshape(n_clusters, 5)
    centre = u[:, :3]
    radius = 0.002 * (40.0 ** u[:, 3])
    weight = (0.02 + u[:, 4]) ** 3
    counts = np.floor(weight / weight.sum() * n).astype(np.int64)
    counts[0] += n - counts.sum()
    owner = np.repeat(np.arange(n_clusters), counts)
    g = uniform01(seed, 3 * n, 1).reshape(n, 3) + uniform01(seed, 3 * n, 2).reshape(n, 3) + uniform01(seed, 3 * n, 3).reshape(n, 3) - 1.5   # ~gaussian
    c = centre[owner] + g * radius[owner, None]
    d = (uniform01(seed, 9 * n, 4).reshape(n, 3, 3) * 2.0 - 1.0) * (0.15 * radius[owner])[:, None, None]
    return (c[:, None, :] + d).astype(dtype).reshape(n, 9)


def procedural_10m(n: int = 10_000_000, seed: int = 7, dtype=np.float32) -> np.ndarray:
    """Config 4's "10M-triangle procedural mesh" = the soup at 10M."""
    return soup(n, seed=seed, dtype=dtype)


# ----------------------------------------------------------------------------------------------
# Rays
# ----------------------------------------------------------------------------------------------

def scene_bounds(prims: np.ndarray):
    if prims.shape[1] == 9:
        p = prims.reshape(-1, 3)
        return p.min(axis=0).astype(np.float64), p.max(axis=0).astype(np.float64)
    c, r = prims[:, :3].astype(np.float64), prims[:, 3:4].astype(np.float64)
    return (c - r).min(axis=0), (c + r).max(axis=0)


def rays_closest(n: int, lo, hi, seed: int = 1234, dtype=np.float32, scale: float = 1.1) -> np.ndarray:
    """R-closest: origin uniform in the bbox scaled 1.1x about its centre, direction uniform on the sphere."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    ctr, half = 0.5 * (lo + hi), 0.5 * (hi - lo) * scale
    u = uniform01(seed, 3 * n, 0).reshape(n, 3)
    org = ctr + (2.0 * u - 1.0) * half
    w = uniform01(seed, 2 * n, 1).reshape(n, 2)
    z = 2.0 * w[:, 0] - 1.0
    phi = 2.0 * np.pi * w[:, 1]
    s = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    d = np.stack([s * np.cos(phi), s * np.sin(phi), z], axis=1)
    out = np.empty((n, 8), dtype=dtype)
    out[:, 0:3] = org
    out[:, 3:6] = d
    out[:, 6] = 0.0
    out[:, 7] = np.finfo(dtype).max
    return out


def rays_shadow(n: int, lo, hi, seed: int = 4321, dtype=np.float32) -> np.ndarray:
    """R-shadow: segment between two uniform points of the bbox, dir unnormalised, t in [1e-4, 1-1e-4]."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    a = lo + uniform01(seed, 3 * n, 0).reshape(n, 3) * (hi - lo)
    b = lo + uniform01(seed, 3 * n, 1).reshape(n, 3) * (hi - lo)
    out = np.empty((n, 8), dtype=dtype)
    out[:, 0:3] = a
    out[:, 3:6] = b - a
    out[:, 6] = 1e-4
    out[:, 7] = 1.0 - 1e-4
    return out


def rays_pinhole(width: int, height: int, eye, direction, up, dtype=np.float32) -> np.ndarray:
    """Primary rays of the reference's benchmark camera (test/benchmark.cpp:343-359): no fov term,
    ``dir + u * right + v * up`` with u = 2x/w - 1, v = 2y/h - 1, all in float32, row-major (y, x)."""
    f = np.float32

    def normalize(a):
        ln = np.sqrt(((f(0) + a[0] * a[0]) + a[1] * a[1]) + a[2] * a[2], dtype=f)
        return a * (f(1) / ln)

    def cross(a, b):
        return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f)

    eye = np.asarray(eye, f)
    d = normalize(np.asarray(direction, f))
    right = normalize(cross(d, np.asarray(up, f)))
    upv = cross(right, d)
    us = f(2) * np.arange(width, dtype=f) / f(width) - f(1)
    vs = f(2) * np.arange(height, dtype=f) / f(height) - f(1)
    U, V = np.meshgrid(us, vs, indexing="xy")
    dirs = (d[None, None] + U[..., None] * right[None, None]) + V[..., None] * upv[None, None]
    out = np.empty((height * width, 8), dtype=dtype)
    out[:, 0:3] = eye
    out[:, 3:6] = dirs.reshape(-1, 3)
    out[:, 6] = 0
    out[:, 7] = np.finfo(dtype).max
    return out
