"""SAH kd-tree build on the device (rtx_kd_build) against the host build.

The host build (KDTree_New of librtx_host.so) is pinned to dumps of the real reference's trees by
test_host_build.py (tests/golden/kd.json); the device build has to produce the same node words and the
same leaf lists, for every parity scene and for triangle soups that exercise the corner cases (lists of
at most MIN_LEAF_LOAD primitives, identical boxes that no plane separates, flat boxes).
"""
import hashlib

import numpy as np
import pytest

from raytracer_amd import scenes
from test_host_build import KD, _tree_arrays

HEAD = [scenes.CAMERA, "texture grey color 0.5 0.5 0.5", "material m diffuse grey"]
TAIL = scenes.LIGHT  # one sphere, so that every pool of the flattened scene exists


def _soup(n, seed, size=0.5, spread=10.0, flat_axis=None):
    rng = np.random.default_rng(seed)
    c = (rng.random((n, 1, 3)) * spread).astype(np.float32)
    v = c + (rng.random((n, 3, 3)).astype(np.float32) - 0.5) * size
    if flat_axis is not None:
        v[:, :, flat_axis] = 1.0
    return HEAD + ["triangle " + " ".join("%.9g" % x for x in t.reshape(-1)) + " m" for t in v] + TAIL


def _same_tree(a, b):
    sa, sb = a.kd_stats(), b.kd_stats()
    assert sa == sb
    na, ra, _ = _tree_arrays(a)
    nb, rb, _ = _tree_arrays(b)
    assert np.array_equal(na, nb) and np.array_equal(ra, rb)


def test_external_builder_fails_loudly_without_a_device():
    """not gpu: on a box without a HIP device the device build reports RTX_E_NO_DEVICE and Scene_Prepare_External
    fails — it does not fall back to the host recursion behind the caller's back."""
    from raytracer_amd import hip
    try:
        if hip.device_count() > 0:
            pytest.skip("a HIP device is visible")
    except hip.RtxError:
        pass
    with pytest.raises(RuntimeError, match="no HIP device|external builder failed"):
        scenes.build(_soup(50, 1), kd_device=0)


@pytest.mark.gpu
@pytest.mark.parametrize("scene", list(KD))
def test_device_build_is_the_reference_tree(built_scenes, scene):
    dev = scenes.build(scene, kd_device=0)
    try:
        _same_tree(built_scenes(scene), dev)
        nodes, refs, d = _tree_arrays(dev)
        want = KD[scene]
        assert hashlib.sha256(nodes.tobytes()).hexdigest() == want["sha256_nodes"]
        assert hashlib.sha256(refs.tobytes()).hexdigest() == want["sha256_refs"]
        assert dev.kd_ms_device is not None and dev.kd_ms_device > 0.0
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("recipe", [
    _soup(1, 2), _soup(4, 3), _soup(5, 4), _soup(300, 5), _soup(5000, 6), _soup(3000, 7, size=4.0, spread=3.0),
    _soup(2000, 8, flat_axis=2), HEAD + _soup(1, 9)[3:4] * 40 + TAIL,  # 40 copies of one triangle: no plane separates them
], ids=["one", "four", "five", "300", "5000", "overlapping", "flat", "copies"])
def test_device_build_on_triangle_soups(recipe):
    a, b = scenes.build(recipe), scenes.build(recipe, kd_device=0)
    try:
        _same_tree(a, b)
    finally:
        a.close(), b.close()


@pytest.mark.gpu
def test_device_build_raw_entry_point():
    """rtx_kd_build called directly on boxes (Surface_BoundingBox of the host library, object.c:26): the arrays it
    returns are the node words and leaf lists of the scene built from the same triangles."""
    import ctypes as C
    from raytracer_amd import hip, host
    recipe = _soup(2500, 11)
    sc = scenes.build(recipe)
    try:
        L = host.lib()
        tris = [np.array(ln.split()[1:10], np.float32).reshape(3, 3) for ln in recipe if ln.startswith("triangle")]
        boxes = np.zeros((len(tris) + 1, 6), np.float32)
        s = L.Surface_Sphere_Make(host.point3(20, 20, 10), 8.0)  # scenes.LIGHT, added after the triangles
        bb = L.Surface_BoundingBox(C.byref(s))
        boxes[len(tris)] = [bb.min.x, bb.min.y, bb.min.z, bb.max.x, bb.max.y, bb.max.z]
        for i, v in enumerate(tris):
            s = host.Surface()
            s.type = 1
            s.triangle = L.Triangle_MakeSimple(host.point3(*v[0]), host.point3(*v[1]), host.point3(*v[2]))
            bb = L.Surface_BoundingBox(C.byref(s))
            boxes[i] = [bb.min.x, bb.min.y, bb.min.z, bb.max.x, bb.max.y, bb.max.z]
        got = hip.kd_build(boxes, int(8.0 + 1.8 * np.log2(len(boxes))))
        nodes, refs, d = _tree_arrays(sc)
        assert got["root"] == d.root_index and got["ms_device"] > 0
        assert np.array_equal(np.stack([got["w0"], got["w1"]], axis=1), nodes) and np.array_equal(got["refs"], refs)
        assert got["depth_reached"] == sc.kd_stats()["depth_reached"]
    finally:
        sc.close()
