"""Subtree-box culling cannot change results, checked on the CPU.

The boxes come from the product (rtx_cull_data_build: host-side preprocessing of librt_hip.so, no device
needed); the oracle replays the kernel's culling test with them and must return exactly what its own
literal recursion — pinned to the reference by test_oracle_pinned.py — returns, on the golden ray sets
and on rays built to sit on the awkward spots: axis-parallel directions, origins on box faces, origins
outside the culling radius, non-finite components.
"""
import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import hip
from util import bits, golden

FIELDS = ("t", "position", "normal", "uv")


def _same(a, b):
    assert np.array_equal(a["prim_id"], b["prim_id"]) and np.array_equal(a["front"], b["front"])
    for k in FIELDS:
        assert np.array_equal(bits(a[k]), bits(b[k])), k


@pytest.mark.parametrize("scene", ["teapot", "bunny", "pear", "materials"])
def test_culled_equals_literal_on_golden_rays(built_scenes, scene):
    sc = built_scenes(scene)
    cull = hip.cull_data(sc.desc)
    g = golden("trace_%s.npz" % scene)  # incoherent rays, not normalised
    lit = O.trace(sc.desc, g["rays"][:, :3], g["rays"][:, 3:])
    cul = O.trace(sc.desc, g["rays"][:, :3], g["rays"][:, 3:], cull=cull)
    _same(lit, cul)
    assert np.array_equal(lit["prim_id"], g["rec"]["id"])  # and the literal side is the reference's answer
    rec = golden("primary_%s_B.npz" % scene)["rec"]
    lit = O.trace(sc.desc, rec["o"], rec["d"], unit_dir=True)
    cul = O.trace(sc.desc, rec["o"], rec["d"], unit_dir=True, cull=cull)
    _same(lit, cul)
    assert cul["visits"][:3].sum() < lit["visits"].sum()


def test_culled_equals_literal_on_awkward_rays(built_scenes):
    sc = built_scenes("teapot")
    cull = hip.cull_data(sc.desc)
    info = cull[1]
    boxes = hip.cull_boxes_world(*cull).astype(np.float32)
    rng = np.random.default_rng(7)
    n = 20000
    node = rng.integers(0, len(boxes), n)
    lo, hi = boxes[node, :3], boxes[node, 3:]
    inside = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    o = inside.copy()
    face = rng.integers(0, 6, n)
    for a in range(3):  # origins exactly on a face of some node's box
        o[face == a, a] = lo[face == a, a]
        o[face == 3 + a, a] = hi[face == 3 + a, a]
    d = rng.normal(size=(n, 3)).astype(np.float32)
    par = rng.integers(0, 4, n)
    for a in range(3):  # a third of the rays exactly parallel to an axis plane, some with -0.0
        d[par == a, a] = np.where(rng.random((par == a).sum()) < 0.5, 0.0, -0.0)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    axis = rng.integers(0, 3, n // 10)  # and some exactly along an axis
    d[: n // 10] = 0.0
    d[np.arange(n // 10), axis] = np.where(rng.random(n // 10) < 0.5, 1.0, -1.0)
    _same(O.trace(sc.desc, o, d, unit_dir=True), O.trace(sc.desc, o, d, unit_dir=True, cull=cull))

    # origins beyond the culling radius, huge coordinates, non-unit and non-finite directions: the walk
    # must fall back to the literal one ray by ray (checker counts no culled node for them)
    far = (info[:3] + 3.0 * np.sqrt(info[3]) * d[:2000]).astype(np.float32)
    back = -d[:2000]
    a, b = O.trace(sc.desc, far, back, unit_dir=True), O.trace(sc.desc, far, back, unit_dir=True, cull=cull)
    _same(a, b)
    assert b["visits"][3] == 0 and (a["prim_id"] != 0xFFFFFFFF).any()
    bad_d = d[:3000].copy()
    bad_d[:1000, 0] = np.nan
    bad_d[1000:2000] *= 1.5
    bad_d[2000:3000, 1] = np.inf
    a, b = O.trace(sc.desc, inside[:3000], bad_d, unit_dir=True), O.trace(sc.desc, inside[:3000], bad_d, unit_dir=True, cull=cull)
    _same(a, b)
    assert b["visits"][3] == 0


def test_boxes_contain_their_primitives(built_scenes):
    """Structure of the cull data: a parent's box contains both children's, a leaf's box its primitives' vertices."""
    sc = built_scenes("teapot")
    d = sc.desc.contents
    qbox, info = hip.cull_data(sc.desc)
    boxes = hip.cull_boxes_world(qbox, info)
    w0 = np.ctypeslib.as_array(d.node_w0, (d.n_nodes,))
    w1 = np.ctypeslib.as_array(d.node_w1, (d.n_nodes,))
    refs = np.ctypeslib.as_array(d.leaf_ref, (d.n_refs,))
    pos = np.ctypeslib.as_array(d.tri_pos, (9 * d.n_tris,)).reshape(9, d.n_tris)
    inner = np.nonzero((w1 >> 30) != 3)[0]
    reach = np.zeros(d.n_nodes, bool)  # nodes reachable from the root
    stack = [d.root_index]
    while stack:
        k = stack.pop()
        reach[k] = True
        if (w1[k] >> 30) != 3:
            stack += [k + 1, int(w1[k] & 0x3FFFFFFF)]
    for k in inner[reach[inner]]:
        for c in (k + 1, int(w1[k] & 0x3FFFFFFF)):
            if (w1[c] >> 30) == 3 and w0[c] == 0:
                continue  # empty leaf: no box
            assert (boxes[k, :3] <= boxes[c, :3]).all() and (boxes[k, 3:] >= boxes[c, 3:]).all()
    leaves = np.nonzero(((w1 >> 30) == 3) & (w0 > 0) & reach)[0]
    for k in leaves[:: max(1, len(leaves) // 500)]:
        for r in refs[(w1[k] & 0x3FFFFFFF): (w1[k] & 0x3FFFFFFF) + w0[k]]:
            if r & 0x80000000:
                continue
            v = pos[:, r].reshape(3, 3)
            assert (v >= boxes[k, :3]).all() and (v <= boxes[k, 3:]).all()
    assert info[4] >= 1.0 and info[3] > 0 and not hip.cull_never(qbox[reach]).any()
