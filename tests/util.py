import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """bit-for-bit equality of two f32 arrays (NaNs with equal payload compare equal)"""
    return np.array_equal(bits(a), bits(b))


def ulp_diff(a, b):
    """max distance in units in the last place between two finite f32 arrays"""
    ia = bits(a).astype(np.int64)
    ib = bits(b).astype(np.int64)
    ia = np.where(ia & 0x80000000, 0x80000000 - ia, ia)
    ib = np.where(ib & 0x80000000, 0x80000000 - ib, ib)
    return np.abs(ia - ib)


def psnr(a, b, peak=255.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def same_values(a, b):
    """bit-for-bit equality except that any NaN equals any NaN (x86 and gfx950 produce different
    default-NaN payloads for inf*0)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all())


def random_scene(seed):
    """A seeded scene recipe (raytracer_amd.scenes directives) built to stress the kd walk and its culling: 1 … 3000 loose
    triangles of very different sizes — slivers, zero-area and axis-aligned ones, exact duplicates, a few that span the
    whole scene —, up to three spheres and up to two planes.  Returns (lines, rays_o, rays_d): the rays start inside and
    around the scene, ON primitives (like bounce rays), on the scene's bounds, and run along axes, along triangle edges and
    at random."""
    rng = np.random.default_rng(1000 + seed)
    n_tri = [1, 2, 7, 40, 300, 3000][seed % 6]
    lines = ["aspect 1.5", "skybox color 0.5 0.7 1.0", "camera 30 -30 30  0 0 0  0 0 1  40 0 0", "texture grey color 0.5 0.5 0.5",
             "material m diffuse grey"]
    centre = rng.uniform(-10, 10, (n_tri, 3))
    size = np.exp(rng.normal(-1.0, 1.2, (n_tri, 1)))
    tri = centre[:, None, :] + size[:, None, :] * rng.normal(size=(n_tri, 3, 3))
    kind = rng.integers(0, 12, n_tri)
    for i in range(n_tri):
        if kind[i] == 0:  # axis-aligned: a flat box
            tri[i, :, rng.integers(0, 3)] = np.float32(centre[i, 0])
        elif kind[i] == 1:  # zero area: two equal vertices
            tri[i, 2] = tri[i, 1]
        elif kind[i] == 2:  # collinear
            tri[i, 2] = tri[i, 0] + 2.0 * (tri[i, 1] - tri[i, 0])
        elif kind[i] == 3 and i:  # an exact duplicate of the previous one
            tri[i] = tri[i - 1]
        elif kind[i] == 4 and seed % 4 == 1:  # spans the scene
            tri[i] = rng.uniform(-12, 12, (3, 3))
    tri = tri.astype(np.float32)
    for t in tri:
        lines.append("triangle " + " ".join(repr(float(x)) for x in t.reshape(9)) + " m")
    spheres = [(rng.uniform(-8, 8, 3), float(np.exp(rng.normal(0, 1)))) for _ in range(seed % 4)]
    for c, r in spheres:
        lines.append("sphere %r %r %r %r m" % (float(c[0]), float(c[1]), float(c[2]), r))
    if seed % 3 >= 1:
        lines.append("plane 0 0 -11 0 0 1 m")
    if seed % 3 == 2:
        lines.append("plane 0 13 0 0.3 -1 0.1 m")
    n = 6000
    o = rng.uniform(-14, 14, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    k = n // 6
    pick = rng.integers(0, n_tri, k)  # on a triangle, like the rays a bounce starts
    w = rng.dirichlet((1, 1, 1), k).astype(np.float32)
    o[:k] = (w[:, :, None] * tri[pick]).sum(axis=1)
    o[k:2 * k] = tri[rng.integers(0, n_tri, k), rng.integers(0, 3, k)]  # on a vertex
    d[k:2 * k] = tri[pick, 1] - tri[pick, 0] + np.float32(1e-3) * d[k:2 * k]  # nearly along an edge
    axis = rng.integers(0, 3, k)  # exactly along an axis, either way
    d[2 * k:3 * k] = 0.0
    d[np.arange(2 * k, 3 * k), axis] = np.where(rng.random(k) < 0.5, 1.0, -1.0)
    par = rng.integers(0, 3, k)  # exactly parallel to an axis plane, with +0 and -0
    d[np.arange(3 * k, 4 * k), par] = np.where(rng.random(k) < 0.5, 0.0, -0.0)
    lo, hi = tri.reshape(-1, 3).min(axis=0), tri.reshape(-1, 3).max(axis=0)
    face = rng.integers(0, 3, k)  # origins exactly on the bounds of all triangles
    o[np.arange(4 * k, 5 * k), face] = np.where(rng.random(k) < 0.5, lo[face], hi[face])
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    nrm[nrm == 0] = 1
    d = (d / nrm).astype(np.float32)
    return lines, o, d
