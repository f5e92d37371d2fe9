"""The C-ABI libraries load without a GPU and export every symbol include/*.h declares."""
import ctypes as C
import os
import re

import pytest

from raytracer_amd import hip, host

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix_re):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w*)\s*\(" % prefix_re, text)))


def test_rtx_h_symbols_exported():
    names = _declared("rtx.h", "rtx_")
    assert len(names) >= 15 and set(names) == set(hip.EXPORTS)
    L = C.CDLL(os.path.join(host.LIB_DIR, "librt_hip.so"))
    for n in names:
        assert hasattr(L, n), n
    assert hip.lib().rtx_abi_version() == 1


def test_rt_host_h_symbols_exported():
    names = _declared("rt_host.h", r"(?:Scene|SceneFlat|Mesh|Material|Texture|Skybox|Camera|Render|KDTree|Surface|Triangle|ImageRGB|RtHost)_")
    L = C.CDLL(os.path.join(host.LIB_DIR, "librtx_host.so"))
    assert len(names) > 50
    for n in names:
        assert hasattr(L, n), n


def test_hip_library_fails_loudly_without_a_device_or_with_bad_arguments(built_scenes):
    L = hip.lib()
    n = C.c_int(-1)
    rc = L.rtx_device_count(C.byref(n))
    h = C.c_void_p()
    if rc != 0:  # CPU-only container: every entry point must refuse, not fall back
        assert rc == -2 and n.value == 0 and b"no HIP device" in L.rtx_last_error()
        assert L.rtx_scene_create(built_scenes("materials").desc, 0, C.byref(h)) == -2
    assert L.rtx_scene_create(None, 0, C.byref(h)) < 0
    assert L.rtx_render(None, None, None, None, None, None) == -1
    # the entry points added for the §8f rows refuse bad arguments before they look for a device
    assert L.rtx_render_progressive(None, None, None, 0, None, None, None, None, None) == -1
    assert L.rtx_render_progressive(None, None, None, 4, None, None, None, None, None) == -1
    t = host.rtx_kd_tree()
    assert L.rtx_kd_build(0, None, 0, 10, None, C.byref(t)) == -1 and b"rtx_kd_build" in L.rtx_last_error()
    import numpy as np
    bad = np.array([[0, 0, 0, 1, 1, 1], [0, 0, 2, 1, 1, 1], [0, 0, 0, 1, np.nan, 1]], np.float32)  # min > max, NaN
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    for k in (1, 2):
        assert L.rtx_kd_build(0, fp(bad[[0, k]].copy()), 2, 9, fp(np.array([0, 0, 0, 1, 1, 2], np.float32)), C.byref(t)) == -1
        assert b"not finite or has min > max" in L.rtx_last_error()
    L.rtx_kd_free(C.byref(t))  # an empty tree may be freed
    L.rtx_kd_free(None)
    if rc != 0:
        box = np.array([[0, 0, 0, 1, 1, 1]], np.float32)
        assert L.rtx_kd_build(0, box.ctypes.data_as(C.POINTER(C.c_float)), 1, 8, box.ctypes.data_as(C.POINTER(C.c_float)), C.byref(t)) == -2


def test_product_never_touches_the_oracle():
    """raytracer_amd/ must not import, link or load anything under oracle/ (the checker is not the product)."""
    pkg = os.path.join(REPO, "raytracer_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in text and "oracle." not in text.replace("oracle.cpp", "").replace("oracle.h", "") \
                    or f == "build.py", os.path.join(root, f)
