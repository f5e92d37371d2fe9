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
from util import bits, golden, random_scene

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
    # the speculative replay (checker of k_trace_long): every reachable subtree walked, the accepting leaf
    # that comes first in the visiting order answers — with and without the culling
    _same(lit, O.trace_speculative(sc.desc, rec["o"], rec["d"], cull=cull))
    _same(lit, O.trace_speculative(sc.desc, rec["o"], rec["d"]))
    _same(O.trace(sc.desc, g["rays"][:, :3], g["rays"][:, 3:]), O.trace_speculative(sc.desc, g["rays"][:, :3], g["rays"][:, 3:], cull=cull, unit_dir=False))


def test_culled_equals_literal_on_awkward_rays(built_scenes):
    sc = built_scenes("teapot")
    cull = hip.cull_data(sc.desc)
    info = cull[1]
    boxes = hip.cull_boxes_world(cull[0], cull[1]).astype(np.float32)
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
    _same(O.trace(sc.desc, o, d, unit_dir=True), O.trace_speculative(sc.desc, o, d, cull=cull))

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
    c = O.trace_speculative(sc.desc, inside[:3000], bad_d, cull=cull)
    assert np.array_equal(a["prim_id"], c["prim_id"])


@pytest.mark.parametrize("seed", range(12))
def test_culled_equals_literal_on_random_scenes(seed):
    """Seeded soups — 1 to 3000 triangles from slivers to scene-spanning ones, degenerate and duplicate ones, spheres,
    planes — and rays that start on primitives, on vertices, on the bounds, and run along axes and edges: the replay of
    the culled walk (boxes, folded cut-off nodes and all) returns the literal recursion's hits bit for bit."""
    from raytracer_amd import scenes
    lines, o, d = random_scene(seed)
    sc = scenes.build(lines)
    try:
        cull = hip.cull_data(sc.desc)
        lit = O.trace(sc.desc, o, d, unit_dir=True)
        cul = O.trace(sc.desc, o, d, unit_dir=True, cull=cull)
        _same(lit, cul)
        _same(lit, O.trace_speculative(sc.desc, o, d, cull=cull))  # k_trace_long's rule: the first accepting leaf in the visiting order
        if seed % 6 >= 2 or seed % 3 >= 1:  # (a lone triangle or two are rarely hit)
            assert (lit["prim_id"] != 0xFFFFFFFF).sum() > len(o) // 100
        if seed % 6 >= 3:
            assert cul["visits"][3] > 0 and cul["visits"][:3].sum() < lit["visits"].sum()  # (it does cull on the larger ones)
    finally:
        sc.close()


def test_boxes_contain_their_primitives(built_scenes):
    """Structure of the cull data: a parent's box contains both children's, a leaf's box its primitives' vertices."""
    sc = built_scenes("teapot")
    d = sc.desc.contents
    qbox, info, _ = hip.cull_data(sc.desc)
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


def test_boxes_do_not_depend_on_the_node_numbering(built_scenes):
    """rtx_scene_create accepts any numbering with right child = node + 1.  KDTree_New's (node before both subtrees)
    lets the subtree boxes be filled by one backward scan; any other takes the order from a walk.  The teapot's
    tree renumbered so that every left subtree comes BEFORE its parent's right spine must get the same boxes —
    and the oracle must find the same hits in it."""
    import ctypes as C
    sc = built_scenes("teapot")
    d = sc.desc.contents
    n = d.n_nodes
    w0 = np.ctypeslib.as_array(d.node_w0, (n,)).copy()
    w1 = np.ctypeslib.as_array(d.node_w1, (n,)).copy()
    new_of_old = np.full(n, -1, np.int64)
    nxt = 0
    stack = [("spine", int(d.root_index))]  # order(node) = order(left(s)) for s on node's right spine, then the spine itself
    while stack:
        what, k = stack.pop()
        if what == "emit":
            for s in k:
                new_of_old[s] = nxt
                nxt += 1
            continue
        spine = [k]
        while (w1[spine[-1]] >> 30) != 3:
            spine.append(spine[-1] + 1)
        stack.append(("emit", spine))
        for s in spine[:-1]:
            stack.append(("spine", int(w1[s] & 0x3FFFFFFF)))
    reach = new_of_old >= 0
    assert nxt == reach.sum() == n
    p0, p1 = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    inner = (w1 >> 30) != 3
    p0[new_of_old] = w0
    p1[new_of_old] = np.where(inner, (w1 & 0xC0000000) | new_of_old[np.where(inner, w1 & 0x3FFFFFFF, 0)].astype(np.uint32), w1)
    assert (new_of_old[np.nonzero(inner)[0] + 1] == new_of_old[inner] + 1).all()            # right child = node + 1 still
    assert (new_of_old[(w1 & 0x3FFFFFFF)[inner]] < new_of_old[inner]).any()                 # and it is no preorder
    d2 = type(d)()
    C.memmove(C.byref(d2), C.byref(d), C.sizeof(d))
    d2.node_w0 = p0.ctypes.data_as(C.POINTER(C.c_uint32))
    d2.node_w1 = p1.ctypes.data_as(C.POINTER(C.c_uint32))
    d2.root_index = int(new_of_old[d.root_index])
    q1, i1, f1 = hip.cull_data(sc.desc)
    q2, i2, f2 = hip.cull_data(C.pointer(d2))
    assert np.array_equal(i1.view(np.uint32), i2.view(np.uint32)) and np.array_equal(q2[new_of_old], q1) and np.array_equal(f2[new_of_old], f1)
    rays = golden("trace_teapot.npz")["rays"][:4000]
    _same(O.trace(C.pointer(d2), rays[:, :3], rays[:, 3:], cull=(q2, i2, f2)), O.trace(sc.desc, rays[:, :3], rays[:, 3:]))


def _desc_with_nodes(d, w0, w1, root=None, max_depth=None):
    import ctypes as C
    d2 = type(d)()
    C.memmove(C.byref(d2), C.byref(d), C.sizeof(d))
    d2.node_w0 = w0.ctypes.data_as(C.POINTER(C.c_uint32))
    d2.node_w1 = w1.ctypes.data_as(C.POINTER(C.c_uint32))
    d2.n_nodes = len(w0)
    if root is not None:
        d2.root_index = root
    if max_depth is not None:
        d2.max_depth = max_depth
    return d2


def test_malformed_trees_are_refused_on_the_host(built_scenes):
    """validate_desc (every entry point that takes an rtx_scene_desc goes through it) checks the kd-tree's SHAPE, not
    only its indices: a child that points back at an ancestor (the device walk would spin, the host walk would never
    end), a subtree with two parents, and a tree deeper than the traversal stack or than desc.max_depth promises are
    refused with RTX_E_INVALID / RTX_E_LIMIT before anything reaches the GPU."""
    import ctypes as C
    sc = built_scenes("teapot")
    d = sc.desc.contents
    n = d.n_nodes
    w0 = np.ctypeslib.as_array(d.node_w0, (n,)).copy()
    w1 = np.ctypeslib.as_array(d.node_w1, (n,)).copy()
    hip.cull_data(sc.desc)  # the tree as built is fine
    inner = np.nonzero((w1 >> 30) != 3)[0]
    # a cycle: some inner node's left child becomes the root
    deep = int(inner[len(inner) // 2])
    c1 = w1.copy()
    c1[deep] = (c1[deep] & 0xC0000000) | int(d.root_index)
    with pytest.raises(hip.RtxError) as e:
        hip.cull_data(C.pointer(_desc_with_nodes(d, w0, c1)))
    assert e.value.code == -1 and "reached twice" in str(e.value)
    # a shared subtree: the root's left child is also its right child's left child (if that is an inner node)
    r = int(d.root_index)
    if (w1[r + 1] >> 30) != 3:
        c2 = w1.copy()
        c2[r + 1] = (c2[r + 1] & 0xC0000000) | (w1[r] & 0x3FFFFFFF)
        with pytest.raises(hip.RtxError) as e:
            hip.cull_data(C.pointer(_desc_with_nodes(d, w0, c2)))
        assert e.value.code == -1
    # deeper than promised: the real tree with a smaller max_depth
    with pytest.raises(hip.RtxError) as e:
        hip.cull_data(C.pointer(_desc_with_nodes(d, w0, w1, max_depth=5)))
    assert e.value.code == -5
    # deeper than the traversal stack (RTX_STACK = 64) although max_depth says 64: a right spine of `k` inner nodes
    # 0..k-1 (right child = node + 1), node k = the leaf that ends it, nodes k+1..2k = each inner node's own empty left leaf
    def spine(k):
        t0, t1 = np.zeros(2 * k + 1, np.uint32), np.zeros(2 * k + 1, np.uint32)
        t1[:k] = (0 << 30) | (k + 1 + np.arange(k, dtype=np.uint32))
        t1[k:] = (3 << 30) | 0  # len 0
        return _desc_with_nodes(d, t0, t1, root=0, max_depth=64), t0, t1

    d80, *keep = spine(80)
    with pytest.raises(hip.RtxError) as e:
        hip.cull_data(C.pointer(d80))
    assert e.value.code == -5 and "deeper" in str(e.value)
    d40, *keep2 = spine(40)  # the same shape within the limit is accepted
    hip.cull_data(C.pointer(d40))
