"""Long walks handed to k_trace_long: one ray per wave, the pending far subtrees of the walk taken over by
free lanes and walked ahead of time, the answer being the report that comes first in the literal visiting order
(KDTree_HitAt, kdtree.c:453-577: the first leaf that accepts a hit ends the walk).

The results must be the literal walk's, bit for bit.  Knob LONG_ALL sends EVERY ray through that kernel (instead of only
the walks a draining launch has left), LONG_AFTER is the hand-over point of the persistent kernel (-1: never, the default —
measured: it does not pay, experiments/README.md).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import hip, scenes
from util import GOLDEN, golden, random_scene, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture
def tune():
    def set_(**kw):
        hip.tune(**kw)
    yield set_
    hip.tune()


def _same_hits(a, b, what):
    assert np.array_equal(a["prim_id"], b["prim_id"]), what
    assert np.array_equal(a["front"], b["front"]), what
    for k in ("t", "position", "normal", "uv"):
        assert same_bits(a[k], b[k]), (what, k)


@pytest.mark.parametrize("scene", ["teapot", "bunny", "pear", "materials", "dragon1", "teapotdof"])
def test_every_ray_through_the_range_kernel_matches_the_reference(built_scenes, device_scenes, tune, scene):
    """Golden primary rays (ids and t of the reference itself) and the incoherent ray sets, every ray walked by a whole wave:
    the reference's hits."""
    ds = device_scenes(scene)
    sets = [golden("primary_%s_%s.npz" % (scene, tag))["rec"] for tag in ("A", "B")]
    for seg in (0,):
        tune(LONG_ALL=1)
        for rec in sets:
            got = ds.trace_closest(rec["o"], rec["d"], flags=hip.RTX_TRACE_UNIT_DIR)
            assert np.array_equal(got["prim_id"], rec["id"]), seg
            hit = rec["id"] != 0xFFFFFFFF
            assert same_bits(got["t"][hit], rec["t"][hit]) and same_bits(got["normal"][hit], rec["n"][hit]), seg
        if scene in ("teapot", "bunny", "pear", "materials"):
            g = golden("trace_%s.npz" % scene)
            got = ds.trace_closest(g["rays"][:, :3], g["rays"][:, 3:])
            assert np.array_equal(got["prim_id"], g["rec"]["id"]), seg
            hit = g["rec"]["id"] != 0xFFFFFFFF
            assert same_bits(got["t"][hit], g["rec"]["t"][hit]), seg


@pytest.mark.parametrize("scene", ["teapot", "dragon1"])
def test_full_1080p_checksums_through_the_range_kernel(built_scenes, device_scenes, tune, scene):
    """All 2 073 600 jittered primary rays of the 1080p frame, each walked by a whole wave: the digests of the reference's own dump."""
    sums = json.load(open(os.path.join(GOLDEN, "primary_checksums.json")))
    W, H = 1920, 1080
    pix = np.arange(W * H, dtype=np.uint32)
    want = sums["%s_B" % scene]
    ds = device_scenes(scene)
    o, d, _ = ds.camera_rays(built_scenes(scene).rtx_camera(), hip.make_params(W, H, 1, 1, seed=tuple(want["seed"])), pix, np.zeros_like(pix))
    tune(LONG_ALL=1)
    got = ds.trace_closest(o, d, flags=hip.RTX_TRACE_UNIT_DIR)
    assert hashlib.sha256(got["prim_id"].tobytes()).hexdigest() == want["sha256_id"]
    assert hashlib.sha256(got["t"].tobytes()).hexdigest() == want["sha256_t"]


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_through_the_range_kernel(tune, seed):
    """The seeded soups (degenerate, duplicate, huge and tiny triangles, spheres, planes; rays on primitives, vertices and
    bounds, along axes and edges): literal kernel == oracle == every ray through k_trace_long."""
    lines, o, d = random_scene(seed)
    sc = scenes.build(lines)
    try:
        with hip.DeviceScene(sc.desc, 0) as ds:
            want = O.trace(sc.desc, o, d, unit_dir=True)
            lit = ds.trace_closest(o, d, flags=hip.RTX_TRACE_UNIT_DIR | hip.RTX_TRACE_LITERAL)
            assert np.array_equal(lit["prim_id"], want["prim_id"])
            for seg in (0,):
                tune(LONG_ALL=1)
                _same_hits(ds.trace_closest(o, d, flags=hip.RTX_TRACE_UNIT_DIR), lit, (seed, seg))
            tune()
    finally:
        sc.close()


@pytest.mark.parametrize("scene", ["teapot", "materials"])
def test_awkward_rays_through_the_range_kernel(built_scenes, device_scenes, tune, scene):
    """Axis-parallel directions with +-0, origins on box faces and outside the culling radius, NaN / inf / non-unit
    directions (no boxes to go by for those): the literal kernel's ids, and its t wherever the hit is finite."""
    sc, ds = built_scenes(scene), device_scenes(scene)
    qbox, info, _ = hip.cull_data(sc.desc)
    boxes = hip.cull_boxes_world(qbox, info)
    boxes = boxes[np.isfinite(boxes).all(axis=1)].astype(np.float32)
    rng = np.random.default_rng(23)
    n = 20000
    node = rng.integers(0, len(boxes), n)
    lo, hi = boxes[node, :3], boxes[node, 3:]
    o = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    face = rng.integers(0, 12, n)
    for a in range(3):
        o[face == a, a] = lo[face == a, a]
        o[face == 3 + a, a] = hi[face == 3 + a, a]
    d = rng.normal(size=(n, 3)).astype(np.float32)
    par = rng.integers(0, 4, n)
    for a in range(3):
        d[par == a, a] = np.where(rng.random((par == a).sum()) < 0.5, 0.0, -0.0)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 10
    d[:k] = 0.0
    d[np.arange(k), rng.integers(0, 3, k)] = np.where(rng.random(k) < 0.5, 1.0, -1.0)
    o[k:2 * k] = (info[:3] + 4.0 * np.sqrt(info[3]) * d[k:2 * k]).astype(np.float32)
    d[k:2 * k] *= -1.0
    d[2 * k:2 * k + 300, 0] = np.nan
    d[2 * k + 300:2 * k + 600] *= 1.7
    d[2 * k + 600:2 * k + 900, 1] = np.inf
    d[2 * k + 900:2 * k + 1200] = 0.0
    o[2 * k + 1200:2 * k + 1400, 2] = np.nan
    lit = ds.trace_closest(o, d, flags=hip.RTX_TRACE_UNIT_DIR | hip.RTX_TRACE_LITERAL)
    ok = (lit["prim_id"] != 0xFFFFFFFF) & np.isfinite(lit["t"])
    for seg in (0,):
        tune(LONG_ALL=1)
        got = ds.trace_closest(o, d, flags=hip.RTX_TRACE_UNIT_DIR)
        assert np.array_equal(got["prim_id"], lit["prim_id"]), seg
        assert same_bits(got["t"][ok], lit["t"][ok]), seg
    assert ok.mean() > 0.2


@pytest.mark.parametrize("scene,w,h,spp,depth", [("dragon1", 240, 135, 4, 8), ("pear", 160, 90, 4, 8), ("materials", 200, 112, 4, 6)])
def test_frames_with_walks_handed_over_are_the_same_frames(built_scenes, device_scenes, tune, scene, w, h, spp, depth):
    """Whole frames, every bounce depth (TAIL_MAX = 0: a pass per depth, so that every queue goes through the kernel that
    hands over): never handing over, handing over at once / after 8 / after 200 iterations, and every ray through
    k_trace_long give the same f32 sums bit for bit, the same rays per depth and shaded hits."""
    sc, ds = built_scenes(scene), device_scenes(scene)
    p = hip.make_params(w, h, spp, depth, seed=(41, 42))
    out = {}
    for name, kw in (("never", dict(LONG_AFTER=-1)), ("at_once", dict(LONG_AFTER=0)), ("after_8", dict(LONG_AFTER=8)), ("after_200", dict(LONG_AFTER=200)),
                     ("all", dict(LONG_ALL=1)), ("default_loop", dict())):
        tune(**(kw if name == "default_loop" else dict(kw, TAIL_MAX=0)))
        out[name] = ds.render(sc.rtx_camera(), p)
    ref = out["never"]
    for name, (bgr, acc, st) in out.items():
        assert np.array_equal(bgr, ref[0]) and same_bits(acc, ref[1]), name
        assert st["rays_at_depth"] == ref[2]["rays_at_depth"] and st["hits_shaded"] == ref[2]["hits_shaded"], name
