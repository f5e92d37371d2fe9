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
from util import golden, psnr, same_bits


def _render_through_host(sc, w, h, spp, depth, seed=(1, 2), timeout=120.0, devices=None, split=0, band_rows=0):
    L = host.lib()
    img = host.ImageRGB()
    assert L.ImageRGB_Load_Empty(C.byref(img), w, h)
    C.memset(img.pix, 0x5A, w * h * 3)
    ctx = L.Render_New(sc.scene, C.byref(img), sc.camera)
    assert ctx and L.Render_Done(ctx)  # idle context reports done
    L.Render_Set_Seed(ctx, *seed)
    if devices:
        assert L.Render_Set_Devices(ctx, (C.c_int * len(devices))(*devices), len(devices), split, band_rows)
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


def hip_no_device():
    return -2  # RTX_E_NO_DEVICE


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
    # the frame shared by several devices inside the library (Render_Set_Devices -> rtx_render_multi; here: two scene
    # handles on device 0): the tile split gives the same bytes, the sample split the same image to a rounding step
    status, pix2, st2, _ = _render_through_host(sc, w, h, spp, depth, seed=(lo, hi), devices=[0, 0], split=hip.RTX_SPLIT_TILE, band_rows=8)
    assert status == 0, host.last_error()
    assert np.array_equal(pix2, bgr) and st2.rays_total == st.rays_total and st2.rays_at_depth[0] == w * h * spp
    status, pix3, st3, _ = _render_through_host(sc, w, h, spp, depth, seed=(lo, hi), devices=[0, 0, 0], split=hip.RTX_SPLIT_SAMPLE)
    assert status == 0, host.last_error()
    assert np.abs(pix3.astype(int) - bgr.astype(int)).max() <= 1 and st3.camera_paths == w * h * spp
    status, _, _, _ = _render_through_host(sc, w, h, spp, depth, seed=(lo, hi), devices=[0, 99])
    assert status == hip_no_device(), "a device that does not exist is an error, not a frame"


# ---- progressive frame: the image fills in while it renders, and can be stopped and exported ----
from util import bits  # noqa: E402


@pytest.mark.gpu
def test_progressive_frame_is_byte_identical_to_the_frame_rendered_at_once(built_scenes, device_scenes):
    sc, ds = built_scenes("teapot"), device_scenes("teapot")
    p = hip.make_params(96, 54, 8, 8, seed=(1, 2))
    bgr, acc, st = ds.render(sc.rtx_camera(), p)
    seen = []

    def on_pass(done, total, img):
        seen.append((done, total, img.copy()))

    bgr2, acc2, st2 = ds.render_progressive(sc.rtx_camera(), p, 3, on_pass)
    assert [(d, t) for d, t, _ in seen] == [(3, 8), (6, 8), (8, 8)]
    assert np.array_equal(bgr, bgr2) and np.array_equal(bits(acc), bits(acc2))  # the sequential sum, continued
    assert np.array_equal(seen[-1][2], bgr) and not np.array_equal(seen[0][2], bgr)
    assert st2["samples_done"] == 8 and st2["rays_total"] == st["rays_total"] and st2["passes"] == 3 * st["passes"]
    # the image after the first pass is the 3-sample image: same sum as the sample range [0, 3) rendered alone
    _, acc3, _ = ds.render(sc.rtx_camera(), hip.make_params(96, 54, 8, 8, seed=(1, 2), sample_range=(0, 3)))
    stopped = ds.render_progressive(sc.rtx_camera(), p, 3, lambda done, total, img: True)
    assert stopped[2]["samples_done"] == 3 and np.array_equal(bits(stopped[1]), bits(acc3))
    assert np.array_equal(stopped[0], seen[0][2])  # sum * (1/3): the preview of three samples


@pytest.mark.gpu
def test_render_loop_progressive_and_stop(built_scenes, device_scenes):
    sc, L = built_scenes("teapot"), host.lib()
    w, h, spp, depth = 96, 54, 8, 8
    status, whole, _, _ = _render_through_host(sc, w, h, spp, depth)
    assert status == 0

    def run(spp, per_pass, stop_after_first_update):
        img = host.ImageRGB()
        assert L.ImageRGB_Load_Empty(C.byref(img), w, h)
        C.memset(img.pix, 0, w * h * 3)
        ctx = L.Render_New(sc.scene, C.byref(img), sc.camera)
        L.Render_Set_Progressive(ctx, per_pass)
        L.Render_Start(ctx, spp, depth)
        t0, partial_seen = time.time(), False
        while not L.Render_Done(ctx):
            assert time.time() - t0 < 120.0
            done = L.Render_Samples_Done(ctx)
            if 0 < done < spp:
                partial_seen = True
                if stop_after_first_update:
                    L.Render_Stop(ctx)
            time.sleep(0.001)
        status, done = L.Render_Status(ctx, None), L.Render_Samples_Done(ctx)
        pix = np.ctypeslib.as_array(C.cast(img.pix, C.POINTER(C.c_uint8)), (h, w, 3)).copy()
        L.Render_Delete(ctx)
        L.ImageRGB_Unload(C.byref(img))
        return status, done, pix, partial_seen

    status, done, pix, _ = run(spp, 2, False)
    assert status == 0 and done == spp and np.array_equal(pix, whole)
    # the SIGINT export (main.c:47-58): stop a long frame once something is there; what is there is a whole image
    status, done, pix, partial_seen = run(4096, 1, True)
    assert status == 0 and partial_seen and 0 < done < 4096
    # ... namely the frame of the first `done` samples, as the C ABI renders it when stopped at the same point
    same = device_scenes("teapot").render_progressive(sc.rtx_camera(), hip.make_params(w, h, 4096, depth), 1, lambda d, t, img: d >= done)
    assert same[2]["samples_done"] == done and np.array_equal(pix, same[0])


@pytest.mark.gpu
def test_render_multi_behind_the_c_abi(built_scenes):
    """rtx_render_multi: the frame's partition over devices lives in librt_hip.so — one host thread per
    scene handle, as Render_Waiter spawns and joins its workers (renderer.c:186-237) — not in Python.  On the one-GPU box the
    handles are n scenes on device 0 (the shares then serialise on the device, the peer copies are device-to-device copies).
    Tile split: bgr and f32 sums are rtx_render's bit for bit, for 1, 2, 3 handles, band heights that leave a partial last
    band, and the per-device stats add up to the frame's.  Sample split: another association of the same terms (1e-5), and
    the same bits from run to run.  Bad arguments are errors; the tuning table is read-only only while a call runs."""
    sc = built_scenes("dragon1")
    cam = sc.rtx_camera()
    W, H, spp, depth = 200, 110, 8, 6
    p = hip.make_params(W, H, spp, depth, seed=(5, 6))
    dss = [hip.DeviceScene(sc.desc, 0) for _ in range(3)]
    try:
        bgr1, acc1, st1 = dss[0].render(cam, p)
        for n, rows in ((1, 0), (2, 0), (3, 16), (2, 7), (3, 200)):
            bgr, acc, sts = hip.render_multi(dss[:n], cam, p, hip.RTX_SPLIT_TILE, rows)
            assert np.array_equal(bgr, bgr1) and same_bits(acc, acc1), (n, rows)
            assert sum(s["rays_total"] for s in sts) == st1["rays_total"] and sum(s["camera_paths"] for s in sts) == W * H * spp
            assert [sum(s["rays_at_depth"][k] for s in sts if k < len(s["rays_at_depth"])) for k in range(len(st1["rays_at_depth"]))] == st1["rays_at_depth"]
        frames = [hip.render_multi(dss[:n], cam, p, hip.RTX_SPLIT_SAMPLE) for n in (1, 2, 3, 3)]
        assert same_bits(frames[0][1], acc1) and np.array_equal(frames[0][0], bgr1)  # one device: the sequential sum itself
        for bgr, acc, sts in frames[1:]:
            assert np.allclose(acc, acc1, rtol=1e-5, atol=1e-5)
            assert sum(s["camera_paths"] for s in sts) == W * H * spp
            assert (np.abs(bgr.astype(int) - bgr1.astype(int)) > 1).mean() == 0.0  # a rounding step at most
        assert same_bits(frames[2][1], frames[3][1]) and np.array_equal(frames[2][0], frames[3][0])  # reproducible
        # outputs are optional (what a benchmark times)
        _, acc, _ = hip.render_multi(dss[:2], cam, p, hip.RTX_SPLIT_TILE, want_bgr=False)
        assert same_bits(acc, acc1)
        hip.render_multi(dss[:2], cam, p, hip.RTX_SPLIT_TILE, want_bgr=False, want_accum=False)
        # errors
        with pytest.raises(hip.RtxError):
            hip.render_multi([dss[0], dss[0]], cam, p)  # the same handle twice
        with pytest.raises(hip.RtxError):
            hip.render_multi(dss[:2], cam, hip.make_params(W, H, spp, depth, band=(8, 0, 2)))  # params must describe the whole frame
        with pytest.raises(hip.RtxError):
            hip.render_multi(dss[:2], cam, p, 7)
        with pytest.raises(hip.RtxError):
            hip.render_multi(dss[:3], cam, hip.make_params(W, H, 2, depth), hip.RTX_SPLIT_SAMPLE)  # fewer samples than devices
        hip.tune(TAIL_MAX=0)  # allowed again once no multi call runs
        with pytest.raises(hip.RtxError):
            hip.tune(TRACE_CHUNK=1 << 32)  # a value that truncates to 0 in the kernel's 32-bit argument
        with pytest.raises(hip.RtxError):
            hip.tune(OVERLAP=9)
        hip.tune()
    finally:
        for ds in dss:
            ds.close()
