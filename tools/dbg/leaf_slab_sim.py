"""CPU estimate (statistics only, not a parity tool): how many of the triangle tests of a scene's leaf visits a 1-D
bound along the leaf's longest axis could remove, if a leaf's references were sorted along that axis and a ray
tested only the sub-range its segment inside the leaf can reach.
    python tools/dbg/leaf_slab_sim.py pear [n_rays] [bins]"""
import sys
import numpy as np

sys.path.insert(0, ".")
from raytracer_amd import scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pear"
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 8
b = scenes.build(name)
d = b.desc.contents if hasattr(b.desc, "contents") else b.desc
n = d.n_nodes
w0 = np.ctypeslib.as_array(d.node_w0, (n,)).copy()
w1 = np.ctypeslib.as_array(d.node_w1, (n,)).copy()
refs = np.ctypeslib.as_array(d.leaf_ref, (d.n_refs,)).copy()
T = d.n_tris
tri = np.ctypeslib.as_array(d.tri_pos, (9 * T,)).reshape(3, 3, T).transpose(2, 0, 1).astype(np.float64)
split = w0.view(np.float32).astype(np.float64)
typ = (w1 >> 30).astype(np.int64)
idx = (w1 & 0x3FFFFFFF).astype(np.int64)
root = int(d.root_index)
tlo, thi = tri.min(axis=1), tri.max(axis=1)
slo, shi = tlo.min(axis=0), thi.max(axis=0)
edge = np.linalg.norm(tri - np.roll(tri, 1, axis=1), axis=2).max(axis=1)
D = 4.0 * np.linalg.norm(shi - slo) / 2
beta = 1.25 * 7.5 * 2.0 ** -24 * D * edge / 1e-4           # DESIGN §5 (barycentric units)
pad = (beta[:, None] * (thi - tlo)) + 32 * 2.0 ** -24 * D   # what a safe box adds per side, roughly
plo, phi = tlo - pad, thi + pad
print(f"{name}: {T} triangles, median edge {np.median(edge):.3f}, median beta {np.median(beta):.2f}, median pad / extent "
      f"{np.median(pad / np.maximum(thi - tlo, 1e-9)):.2f}")

rng = np.random.default_rng(1)
cam = b.rtx_camera()
co = np.array(cam.origin[:], dtype=np.float64)
bl, hz, vt = (np.array(x[:], dtype=np.float64) for x in (cam.bottom_left, cam.horizontal, cam.vertical))
rays = []
for _ in range(NR // 2):   # camera rays
    s, t = rng.random(2)
    dd = bl + s * hz + t * vt - co
    rays.append((co, dd / np.linalg.norm(dd)))
for _ in range(NR - NR // 2):   # rays from points on the surface, any direction (stand-ins for bounce rays)
    k = rng.integers(T)
    u, v = rng.random(2)
    if u + v > 1:
        u, v = 1 - u, 1 - v
    p = tri[k, 0] + u * (tri[k, 1] - tri[k, 0]) + v * (tri[k, 2] - tri[k, 0])
    dd = rng.normal(size=3)
    rays.append((p, dd / np.linalg.norm(dd)))


def tri_hits(o, dv, ids):
    v0, e1, e2 = tri[ids, 0], tri[ids, 1] - tri[ids, 0], tri[ids, 2] - tri[ids, 0]
    h = np.cross(dv, e2)
    a = (e1 * h).sum(1)
    ok = np.abs(a) > 1e-4
    f = 1.0 / np.where(ok, a, 1.0)
    s = o - v0
    u = f * (s * h).sum(1)
    q = np.cross(s, e1)
    v = f * (q * dv).sum(1)
    t = f * (e2 * q).sum(1)
    return np.where(ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 1e-4), t, np.inf)


tot = dict(visits=0, tests=0, bins=0, slab=0, box=0)
by_size = {}
for o, dv in rays:
    inv = 1.0 / np.where(dv == 0, 1e-300, dv)
    t0 = (slo - o) * inv
    t1 = (shi - o) * inv
    tn, tf = max(0.0, np.minimum(t0, t1).max()), np.maximum(t0, t1).min()
    if tn > tf:
        continue
    stack = [(root, tn, tf)]
    done = False
    while stack and not done:
        node, a, z = stack.pop()
        while typ[node] != 3:
            ax = typ[node]
            tp = (split[node] - o[ax]) * inv[ax]
            left, right = idx[node], node + 1
            near, far = (right, left) if (o[ax] > split[node] or (o[ax] == split[node] and dv[ax] > 0)) else (left, right)
            if tp > z or tp <= 0:
                node = near
            elif tp < a:
                node = far
            else:
                stack.append((far, tp, z))
                node, z = near, tp
        c, first = int(w0[node]), int(idx[node])
        if c == 0:
            continue
        r = refs[first:first + c]
        r = (r[(r >> 31) == 0] & 0x3FFFFFFF).astype(np.int64)
        if len(r) == 0:
            continue
        # the ray's segment inside the box of the leaf's (padded) content, as the culled walk has it
        ulo, uhi = plo[r].min(axis=0), phi[r].max(axis=0)
        q0, q1 = (ulo - o) * inv, (uhi - o) * inv
        te, tx = max(0.0, np.minimum(q0, q1).max()), min(np.maximum(q0, q1).min(), z * 1.0001)
        if te > tx:
            continue   # the culled walk does not enter this leaf
        t = tri_hits(o, dv, r)
        k = len(r)
        A = int(np.argmax(uhi - ulo))
        a0, a1 = sorted((o[A] + te * dv[A], o[A] + tx * dv[A]))
        inslab = (phi[r, A] >= a0) & (plo[r, A] <= a1)
        p0, p1 = o + te * dv, o + tx * dv
        inbox = np.all((phi[r] >= np.minimum(p0, p1)) & (plo[r] <= np.maximum(p0, p1)), axis=1)
        # bins: references sorted by the bin of their lower bound; a ray tests [first index that reaches bin b0, last index with lower bin <= b1)
        w = (uhi[A] - ulo[A]) / NB
        lb = np.clip(np.floor((plo[r, A] - ulo[A]) / w), 0, NB - 1).astype(int)
        hb = np.clip(np.floor((phi[r, A] - ulo[A]) / w), 0, NB - 1).astype(int)
        order = np.argsort(lb, kind="stable")
        lb, hb = lb[order], hb[order]
        b0 = int(np.clip(np.floor((a0 - ulo[A]) / w), 0, NB - 1))
        b1 = int(np.clip(np.floor((a1 - ulo[A]) / w), 0, NB - 1))
        reach = np.nonzero(hb >= b0)[0]
        lo_i = int(reach[0]) if len(reach) else k
        hi_i = int((lb <= b1).sum())
        nb_tests = max(0, hi_i - lo_i)
        for key, val in (("visits", 1), ("tests", k), ("bins", nb_tests), ("slab", int(inslab.sum())), ("box", int(inbox.sum()))):
            tot[key] += val
        g = by_size.setdefault(min(k, 64) // 8 * 8, [0, 0, 0, 0, 0])
        for j, val in enumerate((1, k, nb_tests, int(inslab.sum()), int(inbox.sum()))):
            g[j] += val
        if np.isfinite(t).any() and t.min() <= z * 1.0001:
            done = True
print(f"{len(rays)} rays: leaf visits {tot['visits']}, triangle tests {tot['tests']} ({tot['tests'] / len(rays):.1f} per ray); "
      f"with {NB} bins {tot['bins']} ({tot['bins'] / max(1, tot['tests']):.2f}); exact 1-D slab {tot['slab']} ({tot['slab'] / max(1, tot['tests']):.2f}); "
      f"per-reference 3-D box against the segment's box {tot['box']} ({tot['box'] / max(1, tot['tests']):.2f})")
print("leaf size: visits, tests, with bins, 1-D slab, 3-D box")
for ksz in sorted(by_size):
    print(f"  {ksz:2d}+: ", by_size[ksz])
b.close()
