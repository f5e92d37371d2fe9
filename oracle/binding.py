"""ctypes binding of oracle/liboracle.so and runner for oracle/_ref — TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package (raytracer_amd/).
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from raytracer_amd.host import rtx_camera, rtx_render_params, rtx_scene_desc  # noqa: E402  (plain ctypes structs)

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: make -C oracle oracle")
        L = C.CDLL(path)
        f32p, u32p, u8p, u64p = (C.POINTER(t) for t in (C.c_float, C.c_uint32, C.c_uint8, C.c_uint64))
        dp = C.POINTER(rtx_scene_desc)
        L.oracle_trace.argtypes = [dp, C.c_size_t, f32p, f32p, C.c_int, u32p, f32p, f32p, f32p, f32p, u8p, u64p, u32p]
        L.oracle_trace_culled.argtypes = [dp, C.POINTER(C.c_uint16), f32p, u8p, C.c_size_t, f32p, f32p, C.c_int, u32p, f32p, f32p, f32p, f32p, u8p, u64p, u32p]
        L.oracle_trace_speculative.argtypes = [dp, C.POINTER(C.c_uint16), f32p, u8p, C.c_size_t, f32p, f32p, C.c_int, u32p, f32p, f32p, f32p, f32p, u8p, u32p]
        L.oracle_render.argtypes = [dp, C.POINTER(rtx_camera), C.POINTER(rtx_render_params), f32p, u8p, u64p, C.c_int]
        L.oracle_render_pixels.argtypes = [dp, C.POINTER(rtx_camera), C.POINTER(rtx_render_params), C.c_size_t, u32p, f32p, u64p, C.c_int]
        L.oracle_camera_rays.argtypes = [C.POINTER(rtx_camera), C.POINTER(rtx_render_params), C.c_size_t, u32p, u32p,
                                         f32p, f32p, u32p]
        L.oracle_rng_unilateral.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, f32p]
        L.oracle_material_bounce.argtypes = [dp, C.c_uint32, C.c_size_t, f32p, f32p, f32p, f32p, u8p, u32p, f32p, f32p,
                                             f32p, f32p, u8p]
        L.oracle_skybox_color.argtypes = [dp, C.c_size_t, f32p, f32p]
        L.oracle_resolve.argtypes = [f32p, C.c_size_t, C.c_uint32, u8p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def trace(desc, origin, direction, unit_dir=False, cull=None):
    """Scene_ClosestHit per ray.  cull = (qbox, info14[, node_flags]) replays the GPU's subtree-box culling (checker mode)."""
    o = np.ascontiguousarray(origin, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(direction, np.float32).reshape(-1, 3)
    n = len(o)
    out = dict(prim_id=np.empty(n, np.uint32), t=np.empty(n, np.float32), position=np.empty((n, 3), np.float32),
               normal=np.empty((n, 3), np.float32), uv=np.empty((n, 2), np.float32), front=np.empty(n, np.uint8))
    vis = np.zeros(4 if cull is not None else 3, np.uint64)
    steps = np.zeros(n, np.uint32)
    qbox = np.ascontiguousarray(cull[0], np.uint16) if cull is not None else None
    info = np.ascontiguousarray(cull[1], np.float32) if cull is not None else None
    flags = np.ascontiguousarray(cull[2], np.uint8) if cull is not None and len(cull) > 2 else None
    rc = lib().oracle_trace_culled(desc, _p(qbox, C.c_uint16), _p(info, C.c_float), _p(flags, C.c_uint8), n, _p(o, C.c_float), _p(d, C.c_float), int(unit_dir), _p(out["prim_id"], C.c_uint32),
                            _p(out["t"], C.c_float), _p(out["position"], C.c_float), _p(out["normal"], C.c_float),
                            _p(out["uv"], C.c_float), _p(out["front"], C.c_uint8), _p(vis, C.c_uint64),
                            _p(steps, C.c_uint32))
    assert rc == 0
    out["visits"] = vis
    out["steps"] = steps
    return out


def trace_speculative(desc, origin, direction, cull=None, unit_dir=True):
    """The speculative replay (checker of k_trace_long): every reachable subtree is walked, the accepting leaf that comes first
    in the literal visiting order answers.  cull as in trace()."""
    o = np.ascontiguousarray(origin, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(direction, np.float32).reshape(-1, 3)
    n = len(o)
    out = dict(prim_id=np.empty(n, np.uint32), t=np.empty(n, np.float32), position=np.empty((n, 3), np.float32),
               normal=np.empty((n, 3), np.float32), uv=np.empty((n, 2), np.float32), front=np.empty(n, np.uint8), steps=np.zeros(n, np.uint32))
    qbox = np.ascontiguousarray(cull[0], np.uint16) if cull is not None else None
    info = np.ascontiguousarray(cull[1], np.float32) if cull is not None else None
    flags = np.ascontiguousarray(cull[2], np.uint8) if cull is not None and len(cull) > 2 else None
    rc = lib().oracle_trace_speculative(desc, _p(qbox, C.c_uint16), _p(info, C.c_float), _p(flags, C.c_uint8), n, _p(o, C.c_float), _p(d, C.c_float), int(unit_dir),
                                        _p(out["prim_id"], C.c_uint32), _p(out["t"], C.c_float), _p(out["position"], C.c_float), _p(out["normal"], C.c_float),
                                        _p(out["uv"], C.c_float), _p(out["front"], C.c_uint8), _p(out["steps"], C.c_uint32))
    assert rc == 0
    return out


def render(desc, cam, params, threads=8, want_bgr=True):
    cum = np.zeros((params.height, params.width, 3), np.float32)
    bgr = np.zeros((params.height, params.width, 3), np.uint8) if want_bgr else None
    rays = np.zeros(64, np.uint64)
    rc = lib().oracle_render(desc, C.byref(cam), C.byref(params), _p(cum, C.c_float), _p(bgr, C.c_uint8),
                             _p(rays, C.c_uint64), threads)
    assert rc == 0
    return cum, bgr, rays


def render_pixels(desc, cam, params, pixel_index, threads=8):
    """RenderPixel for a list of pixels of the frame `params` describes: (n, 3) f32 sums, rays per depth."""
    pi = np.ascontiguousarray(pixel_index, np.uint32)
    cum = np.zeros((len(pi), 3), np.float32)
    rays = np.zeros(64, np.uint64)
    rc = lib().oracle_render_pixels(desc, C.byref(cam), C.byref(params), len(pi), _p(pi, C.c_uint32), _p(cum, C.c_float), _p(rays, C.c_uint64), threads)
    assert rc == 0
    return cum, rays


def camera_rays(cam, params, pixel_index, sample):
    pi = np.ascontiguousarray(pixel_index, np.uint32)
    sa = np.ascontiguousarray(sample, np.uint32)
    n = len(pi)
    o, d, s = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty((n, 4), np.uint32)
    lib().oracle_camera_rays(C.byref(cam), C.byref(params), n, _p(pi, C.c_uint32), _p(sa, C.c_uint32),
                             _p(o, C.c_float), _p(d, C.c_float), _p(s, C.c_uint32))
    return o, d, s


def rng_unilateral(lo, hi, n):
    out = np.empty(n, np.float32)
    lib().oracle_rng_unilateral(lo, hi, n, _p(out, C.c_float))
    return out


def rng_state_after_seed(lo, hi):
    """xoshiro state right after Random_Seed(lo, hi) — via camera_rays' state dump is awkward, so
    recompute it here in pure Python (integer arithmetic only; src/math/random.c:31-85)."""
    M = 0xFFFFFFFF
    s = [lo & M, (lo >> 32) & M, hi & M, (hi >> 32) & M]

    def nxt():
        t = (s[1] << 9) & M
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 11) | (s[3] >> 21)) & M

    j = [0, 0, 0, 0]
    for w in (0x8764000b, 0xf542d2d3, 0x6fa035c3, 0x77f2db5b):
        for b in range(32):
            if w & (1 << b):
                for k in range(4):
                    j[k] ^= s[k]
            nxt()
    return j


def material_bounce(desc, mat_index, ray_dir, pos, normal, uv, front, rng_state):
    f = lambda a, w: np.ascontiguousarray(a, np.float32).reshape(-1, w)
    rd, po, no, u2 = f(ray_dir, 3), f(pos, 3), f(normal, 3), f(uv, 2)
    fr = np.ascontiguousarray(front, np.uint8)
    stt = np.ascontiguousarray(rng_state, np.uint32).reshape(-1, 4).copy()
    n = len(rd)
    surf, emit, oo, od = (np.empty((n, 3), np.float32) for _ in range(4))
    cont = np.empty(n, np.uint8)
    rc = lib().oracle_material_bounce(desc, mat_index, n, _p(rd, C.c_float), _p(po, C.c_float), _p(no, C.c_float),
                                      _p(u2, C.c_float), _p(fr, C.c_uint8), _p(stt, C.c_uint32), _p(surf, C.c_float),
                                      _p(emit, C.c_float), _p(oo, C.c_float), _p(od, C.c_float), _p(cont, C.c_uint8))
    assert rc == 0
    return dict(surface=surf, emitted=emit, out_origin=oo, out_dir=od, cont=cont, rng_state=stt)


def skybox_color(desc, dirs):
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    out = np.empty_like(d)
    lib().oracle_skybox_color(desc, len(d), _p(d, C.c_float), _p(out, C.c_float))
    return out


def resolve(cum, spp):
    c = np.ascontiguousarray(cum, np.float32).reshape(-1, 3)
    out = np.empty((len(c), 3), np.uint8)
    lib().oracle_resolve(_p(c, C.c_float), len(c), spp, _p(out, C.c_uint8))
    return out.reshape(np.shape(cum))


# ---- the real reference, where its binaries were built (oracle/_ref) ----
def ref_binary(kind="parity"):
    p = os.path.join(HERE, "_ref", "ref_" + kind)
    return p if os.path.exists(p) else None


def ref_run(kind, scene_path, base_dir, *args, timeout=3600):
    exe = ref_binary(kind)
    if not exe:
        raise RuntimeError("oracle/_ref/ref_%s not built" % kind)
    r = subprocess.run([exe, scene_path, base_dir, *map(str, args)], capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("ref_%s failed: %s" % (kind, r.stderr[-1000:]))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") or ln.startswith("[")]
    return json.loads(lines[-1]) if lines else None
