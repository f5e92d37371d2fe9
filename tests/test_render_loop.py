"""The frame-level drop-in boundary: Render_New / Render_Start / Render_Done / Render_Delete
(src/rt/renderer.h:8-22) as exported by librtx_host.so, which flattens the scene and drives librt_hip.so.

Contract kept from the reference: Render_Start returns at once, a waiter thread does the work and flips
ctx->finished, the caller polls Render_Done; the image is ImageRGB (BGR, bottom-up).  Without a HIP device
the render fails loudly (status RTX_E_NO_DEVICE, image untouched) — there is no CPU renderer here.
"""
import ctypes as C
import time

import numpy as np
import pytest

from raytracer_amd import hip, host
from util import golden, psnr


def _render_through_host(sc, w, h, spp, depth, seed=(1, 2), timeout=120.0):
    L = host.lib()
    img = host.ImageRGB()
    assert L.ImageRGB_Load_Empty(C.byref(img), w, h)
    C.memset(img.pix, 0x5A, w * h * 3)
    ctx = L.Render_New(sc.scene, C.byref(img), sc.camera)
    assert ctx and L.Render_Done(ctx)  # idle context reports done
    L.Render_Set_Seed(ctx, *seed)
    t0 = time.time()
    L.Render_Start(ctx, spp, depth)
    started_in = time.time() - t0
    while not L.Render_Done(ctx):
        assert time.time() - t0 < timeout, "Render_Done never became true"
        time.sleep(0.005)
    st = host.rtx_stats()
    status = L.Render_Status(ctx, C.byref(st))
    pix = np.ctypeslib.as_array(C.cast(img.pix, C.POINTER(C.c_uint8)), (h, w, 3)).copy()
    L.Render_Delete(ctx)
    L.ImageRGB_Unload(C.byref(img))
    return status, pix, st, started_in


def test_render_start_without_a_device_fails_loudly(built_scenes):
    if hip.lib() and _has_device():
        pytest.skip("a HIP device is present")
    status, pix, _, _ = _render_through_host(built_scenes("materials"), 32, 18, 1, 2)
    assert status < 0, "rendering without a GPU must not succeed"
    assert (pix == 0x5A).all(), "the image must stay untouched"


def _has_device():
    try:
        return hip.device_count() > 0
    except hip.RtxError:
        return False


@pytest.mark.gpu
def test_render_loop_matches_the_c_abi_and_the_reference(built_scenes, device_scenes):
    g = golden("render_teapot_96x54_8spp_d8.npz")
    w, h, spp, depth, lo, hi = map(int, g["meta"])
    sc = built_scenes("teapot")
    status, pix, st, started_in = _render_through_host(sc, w, h, spp, depth, seed=(lo, hi))
    assert status == 0, host.last_error()
    assert st.rays_at_depth[0] == w * h * spp
    # the same frame straight through librt_hip.so: identical bytes
    bgr, _, _ = device_scenes("teapot").render(sc.rtx_camera(), hip.make_params(w, h, spp, depth, seed=(lo, hi)))
    assert np.array_equal(pix, bgr)
    # and the reference's own render of it (same seeds)
    fin = np.isfinite(g["cum"]).all(axis=2)
    assert psnr(pix[fin], g["bgr"][fin]) >= 30.0
    assert started_in < 1.0  # asynchronous: Render_Start does not wait for the frame
