"""ctypes binding of librt_hip.so (the C ABI of include/rtx.h) with numpy-friendly wrappers.

There is no fallback: if the library is missing, or no HIP device is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

from .host import (LIB_DIR, rtx_camera, rtx_kd_tree, rtx_render_params, rtx_scene_desc, rtx_stats)

RTX_RENDER_COUNT_VISITS = 1
RTX_RENDER_PIXEL_CENTRES = 2
RTX_TRACE_UNIT_DIR = 4
RTX_TRACE_LITERAL = 8
RTX_ID_MISS = 0xFFFFFFFF
RTX_ID_UNBOUNDED = 0x80000000

EXPORTS = ["rtx_abi_version", "rtx_last_error", "rtx_device_count", "rtx_scene_create", "rtx_scene_destroy",
           "rtx_scene_device_bytes", "rtx_cull_data_build", "rtx_trace_closest", "rtx_trace_closest_device", "rtx_render",
           "rtx_render_progressive", "rtx_render_device", "rtx_resolve_device", "rtx_camera_rays", "rtx_rng_unilateral", "rtx_material_bounce",
           "rtx_skybox_color", "rtx_kd_build", "rtx_kd_free", "rtx_closest_hit_noalloc", "rtx_cull_node_flags", "rtx_scene_create_ex", "rtx_tune_set", "rtx_tune_get", "rtx_tune_reset", "rtx_render_multi"]
RTX_SPLIT_TILE, RTX_SPLIT_SAMPLE = 0, 1
RTX_SCENE_F32_TEXTURES = 1
RTX_SCENE_SKY_LOOKUP = 2
RTX_NOALLOC_MAX_RAYS = 1 << 18

_lib = None


class RtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtx error {code}: {msg}")
        self.code = code


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("RTX_HIP_LIB") or os.path.join(LIB_DIR, "librt_hip.so")  # (RTX_HIP_LIB: another build of it, tools/ab2.sh)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: the HIP extension was not built (python -m raytracer_amd.build)")
    # ONE HIP runtime per process.  PyTorch's wheel bundles its own libamdhip64.so (soname libamdhip64.so.7) and
    # libhsa-runtime64.so; librt_hip.so needs libamdhip64.so.7.  If torch is loaded first, librt_hip.so binds to the
    # runtime already in the process.  The other way round /opt/rocm's runtime comes in first and a later `import torch`
    # starts a SECOND one, which finds no GPU (the driver gives a process's GPU VM to one runtime: "No HIP GPUs are
    # available").  Callers that share device memory with torch (bench.py, the multi-GPU path, the tests) therefore need
    # torch first — done here, so that the order of imports in user code does not matter.  RTX_NO_TORCH_PRELOAD=1 skips it.
    import sys
    if os.environ.get("RTX_NO_TORCH_PRELOAD") != "1" and "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        except Exception as e:  # a torch that is installed but cannot load must not take librt_hip.so down with it
            import warnings
            warnings.warn("raytracer_amd.hip: torch is installed but failed to import (%s: %s); loading librt_hip.so without "
                          "the preload — importing torch later in this process may then find no GPU" % (type(e).__name__, e))
    L = C.CDLL(path)
    vp, u32p, f32p, u8p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_uint8)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("rtx_abi_version", C.c_int)
    sig("rtx_last_error", C.c_char_p)
    sig("rtx_device_count", C.c_int, C.POINTER(C.c_int))
    sig("rtx_scene_create", C.c_int, C.POINTER(rtx_scene_desc), C.c_int, C.POINTER(vp))
    sig("rtx_scene_create_ex", C.c_int, C.POINTER(rtx_scene_desc), C.c_int, C.c_uint32, C.POINTER(vp))
    sig("rtx_tune_set", C.c_int, C.c_char_p, C.c_longlong)
    sig("rtx_tune_get", C.c_int, C.c_char_p, C.POINTER(C.c_longlong))
    sig("rtx_tune_reset", None)
    sig("rtx_scene_destroy", C.c_int, vp)
    sig("rtx_scene_device_bytes", C.c_int, vp, C.POINTER(C.c_uint64))
    sig("rtx_cull_data_build", C.c_int, C.POINTER(rtx_scene_desc), C.POINTER(C.c_uint16), f32p)
    sig("rtx_cull_node_flags", C.c_int, C.POINTER(rtx_scene_desc), u8p)
    sig("rtx_trace_closest", C.c_int, vp, C.c_size_t, f32p, f32p, u32p, f32p, f32p, f32p, f32p, u8p,
        C.POINTER(rtx_stats), C.c_uint32)
    sig("rtx_trace_closest_device", C.c_int, vp, C.c_size_t, vp, vp, vp, C.c_uint32, vp)
    sig("rtx_closest_hit_noalloc", C.c_int, vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32)
    sig("rtx_render", C.c_int, vp, C.POINTER(rtx_camera), C.POINTER(rtx_render_params), u8p, f32p,
        C.POINTER(rtx_stats))
    sig("rtx_render_progressive", C.c_int, vp, C.POINTER(rtx_camera), C.POINTER(rtx_render_params), C.c_uint32, u8p, f32p,
        vp, vp, C.POINTER(rtx_stats))
    sig("rtx_render_device", C.c_int, vp, C.POINTER(rtx_camera), C.POINTER(rtx_render_params), vp, vp,
        C.POINTER(rtx_stats))
    sig("rtx_render_multi", C.c_int, C.POINTER(vp), C.c_uint32, C.POINTER(rtx_camera), C.POINTER(rtx_render_params), C.c_uint32, C.c_uint32,
        u8p, f32p, C.POINTER(rtx_stats))
    sig("rtx_resolve_device", C.c_int, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp)
    sig("rtx_camera_rays", C.c_int, vp, C.POINTER(rtx_camera), C.POINTER(rtx_render_params), C.c_size_t, u32p, u32p,
        f32p, f32p, u32p)
    sig("rtx_rng_unilateral", C.c_int, vp, C.c_uint64, C.c_uint64, C.c_size_t, f32p)
    sig("rtx_material_bounce", C.c_int, vp, C.c_uint32, C.c_size_t, f32p, f32p, f32p, f32p, u8p, u32p, f32p, f32p,
        f32p, f32p, u8p)
    sig("rtx_skybox_color", C.c_int, vp, C.c_size_t, f32p, f32p)
    sig("rtx_kd_build", C.c_int, C.c_int, f32p, C.c_uint64, C.c_uint64, f32p, C.POINTER(rtx_kd_tree))
    sig("rtx_kd_free", None, C.POINTER(rtx_kd_tree))
    _lib = L
    return L


def tune(**knobs):
    """rtx_tune_reset, then rtx_tune_set for every NAME=value given (names as in include/rtx.h, any case)."""
    lib().rtx_tune_reset()
    for k, v in knobs.items():
        _check(lib().rtx_tune_set(k.upper().encode(), int(v)))


def tune_get(name: str) -> int:
    v = C.c_longlong(0)
    _check(lib().rtx_tune_get(name.upper().encode(), C.byref(v)))
    return v.value


def last_error() -> str:
    return lib().rtx_last_error().decode(errors="replace")


def kd_build(box_min_max: np.ndarray, max_depth: int, device: int = 0) -> dict:
    """rtx_kd_build on n x 6 f32 boxes (min xyz, max xyz): node words, leaf lists and the timings."""
    b = np.ascontiguousarray(box_min_max, np.float32).reshape(-1, 6)
    world = np.concatenate([b[:, :3].min(axis=0), b[:, 3:].max(axis=0)]).astype(np.float32)
    t = rtx_kd_tree()
    _check(lib().rtx_kd_build(device, _p(b, C.c_float), len(b), max_depth, _p(world, C.c_float), C.byref(t)))
    try:
        return {"w0": np.ctypeslib.as_array(t.node_w0, (t.n_nodes,)).copy(), "w1": np.ctypeslib.as_array(t.node_w1, (t.n_nodes,)).copy(),
                "refs": np.ctypeslib.as_array(t.leaf_ref, (max(t.n_refs, 1),))[:t.n_refs].copy(), "root": int(t.root_index),
                "depth_reached": int(t.depth_reached), "ms_device": t.ms_device, "ms_total": t.ms_total}
    finally:
        lib().rtx_kd_free(C.byref(t))


def _check(rc):
    if rc != 0:
        raise RtxError(rc, lib().rtx_last_error().decode(errors="replace"))


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


def device_count() -> int:
    n = C.c_int(0)
    _check(lib().rtx_device_count(C.byref(n)))
    return n.value


def stats_dict(st: rtx_stats) -> dict:
    d = {n: getattr(st, n) for n, _ in st._fields_ if n not in ("rays_at_depth", "ms_trace_at_depth", "wave_stats")}
    d["wave_stats"] = [int(x) for x in st.wave_stats]
    d["rays_at_depth"] = [int(x) for x in st.rays_at_depth if x]
    d["ms_trace_at_depth"] = [float(x) for x in st.ms_trace_at_depth][:len(d["rays_at_depth"])]
    return d


def make_params(width, height, spp, max_depth, seed=(1, 2), sample_range=None, band=None, flags=0,
                max_paths_in_flight=0) -> rtx_render_params:
    p = rtx_render_params()
    p.width, p.height, p.spp, p.max_depth = width, height, spp, max_depth
    p.seed_lo, p.seed_hi = seed
    if sample_range:
        p.sample_begin, p.sample_end = sample_range
    if band:
        p.band_rows, p.band_rank, p.band_world = band
    p.flags = flags
    p.max_paths_in_flight = max_paths_in_flight
    return p


def render_multi(device_scenes, cam, params, split=RTX_SPLIT_TILE, band_rows=0, want_accum=True, want_bgr=True):
    """rtx_render_multi: one frame over the devices of `device_scenes` (DeviceScene handles of the same scene), inside
    the library — no torch.distributed, no Python thread per device.  Returns (bgr, accum, [stats per device])."""
    n = len(device_scenes)
    handles = (C.c_void_p * n)(*[ds.handle for ds in device_scenes])
    bgr = np.zeros((params.height, params.width, 3), np.uint8) if want_bgr else None
    acc = np.zeros((params.height, params.width, 3), np.float32) if want_accum else None
    st = (rtx_stats * n)()
    _check(lib().rtx_render_multi(handles, n, C.byref(cam), C.byref(params), split, band_rows, _p(bgr, C.c_uint8), _p(acc, C.c_float), st))
    return bgr, acc, [stats_dict(x) for x in st]


def cull_data(desc):
    """Quantised conservative subtree boxes + culling constants of a scene (host-side preprocessing of
    librt_hip.so, no device needed): (qbox[n_nodes, 6] u16, info[14] f32, node_flags[n_nodes] u8), see rtx_cull_data_build
    and rtx_cull_node_flags."""
    qbox = np.zeros((desc.contents.n_nodes, 6), np.uint16)
    info = np.zeros(14, np.float32)
    flags = np.zeros(desc.contents.n_nodes, np.uint8)
    _check(lib().rtx_cull_data_build(desc, _p(qbox, C.c_uint16), _p(info, C.c_float)))
    _check(lib().rtx_cull_node_flags(desc, _p(flags, C.c_uint8)))
    return qbox, info, flags


def cull_never(qbox):
    """nodes marked 'never cull' (unbounded box)"""
    return (qbox[:, 0] == 0xFFFF) & (qbox[:, 3] == 0)


def cull_boxes_world(qbox, info):
    """Decode (qbox, info) to world-space boxes [n, 6] (min xyz, max xyz); 'never cull' nodes come out infinite."""
    lo, scale = info[5:8].astype(np.float64), info[8:11].astype(np.float64)  # the grid x(q) = lo + q * scale
    b = np.concatenate([lo + qbox[:, :3].astype(np.float64) * scale, lo + qbox[:, 3:].astype(np.float64) * scale], axis=1)
    never = cull_never(qbox)
    b[never, :3], b[never, 3:] = -np.inf, np.inf
    return b


class DeviceScene:
    """rtx_scene handle (one per scene and device)."""

    def __init__(self, desc, device=0, flags=0):
        self.handle = C.c_void_p()
        _check(lib().rtx_scene_create_ex(desc, device, flags, C.byref(self.handle)))
        self.device = device

    def close(self):
        if self.handle:
            lib().rtx_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_bytes(self) -> int:
        b = C.c_uint64(0)
        _check(lib().rtx_scene_device_bytes(self.handle, C.byref(b)))
        return b.value

    # ---- Scene_ClosestHit, batched ----
    def trace_closest(self, origin, direction, flags=0):
        o = np.ascontiguousarray(origin, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(direction, np.float32).reshape(-1, 3)
        n = len(o)
        out = dict(prim_id=np.empty(n, np.uint32), t=np.empty(n, np.float32), position=np.empty((n, 3), np.float32),
                   normal=np.empty((n, 3), np.float32), uv=np.empty((n, 2), np.float32),
                   front=np.empty(n, np.uint8))
        st = rtx_stats()
        _check(lib().rtx_trace_closest(self.handle, n, _p(o, C.c_float), _p(d, C.c_float), _p(out["prim_id"], C.c_uint32),
                                       _p(out["t"], C.c_float), _p(out["position"], C.c_float),
                                       _p(out["normal"], C.c_float), _p(out["uv"], C.c_float),
                                       _p(out["front"], C.c_uint8), C.byref(st), flags))
        out["stats"] = stats_dict(st)
        return out

    # ---- Render_Start equivalent, blocking ----
    def render(self, cam, params, want_accum=True, want_bgr=True):
        npx = params.width * params.height
        bgr = np.zeros((params.height, params.width, 3), np.uint8) if want_bgr else None
        acc = np.zeros((params.height, params.width, 3), np.float32) if want_accum else None
        st = rtx_stats()
        _check(lib().rtx_render(self.handle, C.byref(cam), C.byref(params), _p(bgr, C.c_uint8), _p(acc, C.c_float),
                                C.byref(st)))
        return bgr, acc, stats_dict(st)

    def render_progressive(self, cam, params, spp_per_pass, on_pass=None, want_accum=True):
        """rtx_render_progressive: on_pass(samples_done, samples_total, bgr) is called after every pass with the
        image so far (a view of the output buffer); a true return value stops the frame there."""
        bgr = np.zeros((params.height, params.width, 3), np.uint8)
        acc = np.zeros((params.height, params.width, 3), np.float32) if want_accum else None
        st = rtx_stats()
        cb_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)
        cb = cb_t(lambda user, done, total: 1 if (on_pass and on_pass(done, total, bgr)) else 0)
        _check(lib().rtx_render_progressive(self.handle, C.byref(cam), C.byref(params), spp_per_pass, _p(bgr, C.c_uint8),
                                            _p(acc, C.c_float), C.cast(cb, C.c_void_p), None, C.byref(st)))
        return bgr, acc, stats_dict(st)

    def render_device(self, cam, params, d_accum_ptr, stream=0):
        st = rtx_stats()
        _check(lib().rtx_render_device(self.handle, C.byref(cam), C.byref(params), C.c_void_p(d_accum_ptr),
                                       C.c_void_p(stream), C.byref(st)))
        return stats_dict(st)

    def resolve_device(self, d_accum_ptr, width, height, spp, d_bgr_ptr, stream=0):
        _check(lib().rtx_resolve_device(self.handle, C.c_void_p(d_accum_ptr), width, height, spp,
                                        C.c_void_p(d_bgr_ptr), C.c_void_p(stream)))

    def trace_closest_device(self, n, d_rays_ptr, d_hits_ptr, stream=0, flags=0, d_counters_ptr=0):
        _check(lib().rtx_trace_closest_device(self.handle, n, C.c_void_p(d_rays_ptr), C.c_void_p(d_hits_ptr),
                                              C.c_void_p(stream), flags, C.c_void_p(d_counters_ptr)))

    def closest_hit_noalloc(self, n, d_origin, d_dir, d_prim_id=0, d_t=0, d_position=0, d_normal=0, d_uv=0, d_front=0, stream=0, flags=0):
        """rtx_closest_hit_noalloc: device pointers (ints) in rtx_trace_closest's layouts; asynchronous on `stream`."""
        _check(lib().rtx_closest_hit_noalloc(self.handle, n, *(C.c_void_p(x) for x in (d_origin, d_dir, d_prim_id, d_t, d_position, d_normal, d_uv,
                                                                                        d_front, stream)), flags))

    # ---- unit seams ----
    def camera_rays(self, cam, params, pixel_index, sample):
        pi = np.ascontiguousarray(pixel_index, np.uint32)
        sa = np.ascontiguousarray(sample, np.uint32)
        n = len(pi)
        o, d, s = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty((n, 4), np.uint32)
        _check(lib().rtx_camera_rays(self.handle, C.byref(cam), C.byref(params), n, _p(pi, C.c_uint32),
                                     _p(sa, C.c_uint32), _p(o, C.c_float), _p(d, C.c_float), _p(s, C.c_uint32)))
        return o, d, s

    def rng_unilateral(self, seed_lo, seed_hi, n):
        out = np.empty(n, np.float32)
        _check(lib().rtx_rng_unilateral(self.handle, seed_lo, seed_hi, n, _p(out, C.c_float)))
        return out

    def material_bounce(self, mat_index, ray_dir, pos, normal, uv, front, rng_state):
        f = lambda a, w: np.ascontiguousarray(a, np.float32).reshape(-1, w)
        rd, po, no, u2 = f(ray_dir, 3), f(pos, 3), f(normal, 3), f(uv, 2)
        fr = np.ascontiguousarray(front, np.uint8)
        stt = np.ascontiguousarray(rng_state, np.uint32).reshape(-1, 4).copy()
        n = len(rd)
        surf, emit, oo, od = (np.empty((n, 3), np.float32) for _ in range(4))
        cont = np.empty(n, np.uint8)
        _check(lib().rtx_material_bounce(self.handle, mat_index, n, _p(rd, C.c_float), _p(po, C.c_float),
                                         _p(no, C.c_float), _p(u2, C.c_float), _p(fr, C.c_uint8), _p(stt, C.c_uint32),
                                         _p(surf, C.c_float), _p(emit, C.c_float), _p(oo, C.c_float),
                                         _p(od, C.c_float), _p(cont, C.c_uint8)))
        return dict(surface=surf, emitted=emit, out_origin=oo, out_dir=od, cont=cont, rng_state=stt)

    def skybox_color(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.empty_like(d)
        _check(lib().rtx_skybox_color(self.handle, len(d), _p(d, C.c_float), _p(out, C.c_float)))
        return out
