#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REAL REFERENCE.

Needs oracle/_ref/ref_parity (the unmodified reference sources + oracle/ref_driver.c, built by
`make -C oracle ref` in the build container with the pinned flags -O2 -ffp-contract=off).  The
fixtures are data only — inputs and the reference's outputs — and are what pins both the CPU
restatement (oracle/oracle.cpp) and the HIP path on machines where /root/reference is absent.

    python tests/golden/make_golden.py            # everything
    python tests/golden/make_golden.py primary    # one group
"""
import hashlib
import json
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from raytracer_amd import assets, scenes  # noqa: E402

REF = os.path.join(REPO, "oracle", "_ref", "ref_parity")
TMP = os.path.join(assets.cache_dir(), "golden_tmp")
os.makedirs(TMP, exist_ok=True)

HITREC = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("id", "<u4"), ("t", "<f4"), ("pos", "<f4", 3), ("n", "<f4", 3),
                   ("uv", "<f4", 2), ("front", "<u4")])
BOUNCE_IN = np.dtype([("mat", "<u4"), ("dir", "<f4", 3), ("pos", "<f4", 3), ("n", "<f4", 3), ("uv", "<f4", 2),
                      ("front", "<u4"), ("seed_lo", "<u8"), ("seed_hi", "<u8")])
BOUNCE_OUT = np.dtype([("cont", "<u4"), ("surf", "<f4", 3), ("emit", "<f4", 3), ("oo", "<f4", 3), ("od", "<f4", 3),
                       ("draws", "<u4", 4)])


def ref(scene, *args):
    path = scenes.write_recipe(scene)
    r = subprocess.run([REF, path, assets.cache_dir(), *map(str, args)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("{") or ln.startswith("[")]
    return json.loads(last[-1]) if last else None


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items()})


def g_rng():
    out = {}
    for lo, hi in [(1, 2), (0x123456789ABCDEF, 0xFEDCBA9876543210), (7, 0)]:
        r = subprocess.run([REF, "rng", str(lo), str(hi), "64"], capture_output=True, text=True, check=True)
        out["%d_%d" % (lo, hi)] = json.loads(r.stdout)
    json.dump(out, open(os.path.join(HERE, "rng.json"), "w"))
    print("wrote rng.json")


def g_kd():
    out = {}
    for sc in ["teapot", "bunny", "pear", "materials", "dragon1"]:
        f = os.path.join(TMP, "kd.bin")
        st = ref(sc, "kd", f)
        raw = open(f, "rb").read()
        _, nn, nr, root, _ = struct.unpack_from("<5I", raw, 0)
        nodes = np.frombuffer(raw, np.uint32, nn * 2, 20)
        refs = np.frombuffer(raw, np.uint32, nr, 20 + nn * 8)
        st.update(root=root, sha256_nodes=hashlib.sha256(nodes.tobytes()).hexdigest(),
                  sha256_refs=hashlib.sha256(refs.tobytes()).hexdigest())
        out[sc] = st
        if sc == "materials":  # tiny tree: keep it whole
            save("kd_materials.npz", nodes=nodes.reshape(-1, 2), refs=refs)
    json.dump(out, open(os.path.join(HERE, "kd.json"), "w"), indent=1)
    print("wrote kd.json")


def g_primary(only=None):
    W, H = 1920, 1080
    path = os.path.join(HERE, "primary_checksums.json")
    sums = json.load(open(path)) if only and os.path.exists(path) else {}
    for sc, step in [("teapot", 12), ("bunny", 12), ("pear", 12), ("materials", 12), ("dragon1", 24), ("teapotdof", 12)]:
        if only and sc != only:
            continue
        f = os.path.join(TMP, "p.bin")
        for jitter, tag in [(0, "A"), (1, "B")]:
            full = ref(sc, "primary", W, H, 1, jitter, 1, 2, 1, f)  # full 1080p: checksums only
            rec = np.fromfile(f, HITREC)
            sums["%s_%s" % (sc, tag)] = dict(full, width=W, height=H, seed=[1, 2],
                                             sha256_id=hashlib.sha256(rec["id"].tobytes()).hexdigest(),
                                             sha256_t=hashlib.sha256(rec["t"].tobytes()).hexdigest())
            sub = rec.reshape(H, W)[::step, ::step].reshape(-1)
            save("primary_%s_%s.npz" % (sc, tag), rec=sub, step=np.int32(step), width=np.int32(W), height=np.int32(H))
    json.dump(sums, open(path, "w"), indent=1)
    print("wrote primary_checksums.json")


def g_trace():
    """Incoherent secondary-style rays: origins on/near the mesh, random directions."""
    rng = np.random.default_rng(20241111)
    for sc in ["teapot", "bunny", "pear", "materials"]:
        prim = np.load(os.path.join(HERE, "primary_%s_B.npz" % sc))["rec"]
        hitp = prim["pos"][prim["id"] != 0xFFFFFFFF]
        n = 4096
        o = hitp[rng.integers(0, len(hitp), n)].astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[: n // 8] *= np.float32(1e-3)  # un-normalised inputs exercise Ray_Make
        d[n // 8: n // 4, 0] = 0.0       # axis-parallel components exercise the |dir| < eps branch
        d[n // 4: n // 4 + 64, 1] = np.float32(5e-5)
        rays = np.concatenate([o, d], axis=1).astype("<f4")
        fi, fo = os.path.join(TMP, "rays.bin"), os.path.join(TMP, "hits.bin")
        rays.tofile(fi)
        ref(sc, "trace", fi, fo)
        save("trace_%s.npz" % sc, rays=rays, rec=np.fromfile(fo, HITREC))


def ref_render(sc, w, h, spp, depth, lo, hi):
    f = os.path.join(TMP, "r.bin")
    meta = ref(sc, "render", w, h, spp, depth, lo, hi, 8, f)
    raw = open(f, "rb").read()
    cum = np.frombuffer(raw, "<f4", w * h * 3, 16).reshape(h, w, 3)
    bgr = np.frombuffer(raw, np.uint8, w * h * 3, 16 + w * h * 12).reshape(h, w, 3)
    return cum, bgr, meta


RENDERS = [("teapot", 96, 54, 8, 8), ("bunny", 96, 54, 8, 8), ("dragon1", 96, 54, 8, 8), ("pear", 96, 54, 8, 16),
           ("materials", 128, 72, 16, 8), ("materials", 64, 36, 4, 1),
           # the bench configuration's sampling (BASELINE configs[2]: 64 spp, depth 8) and config 5's (1024 spp)
           ("dragon1", 240, 135, 64, 8), ("dragon1", 192, 108, 1024, 8),
           # Camera_GetRay with a lens (aperture 2, focus 25): no BASELINE config has one
           ("teapotdof", 96, 54, 8, 8)]


def g_render(only=None):
    for sc, w, h, spp, depth in RENDERS:
        name = "render_%s_%dx%d_%dspp_d%d.npz" % (sc, w, h, spp, depth)
        if only and only not in name:
            continue
        cum, bgr, meta = ref_render(sc, w, h, spp, depth, 1, 2)
        save(name, cum=cum, bgr=bgr, meta=np.array([w, h, spp, depth, 1, 2], np.int64))
        print(meta)


def ref_render_pixels(sc, w, h, spp, depth, lo, hi, pix):
    fi, fo = os.path.join(TMP, "px_in.bin"), os.path.join(TMP, "px_out.bin")
    np.ascontiguousarray(pix, "<u4").tofile(fi)
    meta = ref(sc, "render_pixels", w, h, spp, depth, lo, hi, 8, fi, fo)
    raw = open(fo, "rb").read()
    n = len(pix)
    assert struct.unpack_from("<5I", raw, 0) == (w, h, spp, depth, n)
    assert np.array_equal(np.frombuffer(raw, "<u4", n, 20), pix)
    cum = np.frombuffer(raw, "<f4", 3 * n, 20 + 4 * n).reshape(n, 3)
    bgr = np.frombuffer(raw, np.uint8, 3 * n, 20 + 16 * n).reshape(n, 3)
    return cum, bgr, meta


def sparse_pixels(sc, W, H, n, seed):
    """n pixels of the W x H frame: half on a jittered grid (ground, sky, light and model in the proportions of the
    picture), half in the cells of the 1080p primary-ray golden where the kind of the thing hit — model, sphere, plane,
    nothing — differs from a neighbour's: silhouettes, contact lines and the light's rim."""
    rng = np.random.default_rng(seed)
    gx = int(round((n // 2 * W / H) ** 0.5))
    gy = (n // 2 + gx - 1) // gx
    ys, xs = np.meshgrid(np.arange(gy), np.arange(gx), indexing="ij")
    x = ((xs + rng.random(xs.shape)) * W / gx).astype(np.int64).clip(0, W - 1)
    y = ((ys + rng.random(ys.shape)) * H / gy).astype(np.int64).clip(0, H - 1)
    grid = (y * W + x).reshape(-1)[: n // 2]
    g = np.load(os.path.join(HERE, "primary_%s_A.npz" % sc))
    step, gw, gh = int(g["step"]), int(g["width"]), int(g["height"])
    ids = g["rec"]["id"].reshape((gh + step - 1) // step, (gw + step - 1) // step)
    kind = np.where(ids == 0xFFFFFFFF, 0, np.where(ids >= 0x80000000, 1, 2)).astype(np.int64)
    big = np.bincount(ids[(kind == 2)].astype(np.int64)).argmax() if (kind == 2).any() else -1
    kind[(kind == 2) & (ids == ids.max(where=kind == 2, initial=0))] = 3  # the light sphere is the last kd object
    edge = np.zeros_like(kind, bool)
    edge[:, 1:] |= kind[:, 1:] != kind[:, :-1]
    edge[1:, :] |= kind[1:, :] != kind[:-1, :]
    edge[:, :-1] |= edge[:, 1:].copy()
    edge[:-1, :] |= edge[1:, :].copy()
    cy, cx = np.nonzero(edge)
    pick = rng.integers(0, len(cy), n - len(grid))
    ex = ((cx[pick] * step + rng.integers(-step, step, len(pick))) * W // gw).clip(0, W - 1)
    ey = ((cy[pick] * step + rng.integers(-step, step, len(pick))) * H // gh).clip(0, H - 1)
    pix = np.unique(np.concatenate([grid, ey * W + ex]).astype(np.uint32))
    del big
    return pix


# BASELINE configs 2-5 at their FULL size, a pixel subset each: the per-path seeding makes pixel p of
# the list pixel p of the whole frame
SPARSE = [("bunny", 1920, 1080, 64, 8, 8192), ("dragon1", 1920, 1080, 64, 8, 8192), ("pear", 1920, 1080, 64, 16, 8192),
          ("dragon1", 3840, 2160, 1024, 8, 1024)]


def g_sparse(only=None):
    for k, (sc, w, h, spp, depth, n) in enumerate(SPARSE):
        name = "sparse_%s_%dx%d_%dspp_d%d.npz" % (sc, w, h, spp, depth)
        if only and only not in name:
            continue
        pix = sparse_pixels(sc, w, h, n, 31 + k)
        cum, bgr, meta = ref_render_pixels(sc, w, h, spp, depth, 1, 2, pix)
        cum2, bgr2, _ = ref_render_pixels(sc, w, h, spp, depth, 3, 4, pix)  # the noise floor of exactly these pixels
        fin = np.isfinite(cum).all(axis=1) & np.isfinite(cum2).all(axis=1)
        floor = float(np.sqrt(np.mean((bgr[fin].astype(np.float64) - bgr2[fin].astype(np.float64)) ** 2)))
        save(name, pix=pix, cum=cum, bgr=bgr, meta=np.array([w, h, spp, depth, 1, 2], np.int64), noise_rmse_8bit=np.float64(floor))
        print(meta, "noise floor (8 bit, seeds (1,2) vs (3,4)):", floor)


def g_noise():
    """CPU-vs-CPU noise floor: the reference against ITSELF with another seed, same scene and sampling.
    Two frames that agree path for path differ by far less than this; two frames that are merely both correct
    differ by about this much.  The numbers go to noise.json; the GPU tests' image bounds cite them."""
    out = {}
    for sc, w, h, spp, depth in [("dragon1", 240, 135, 64, 8), ("dragon1", 96, 54, 8, 8), ("materials", 128, 72, 16, 8), ("bunny", 96, 54, 8, 8),
                                 ("teapot", 96, 54, 8, 8), ("pear", 96, 54, 8, 16), ("teapotdof", 96, 54, 8, 8)]:
        a_cum, a_bgr, _ = ref_render(sc, w, h, spp, depth, 1, 2)
        b_cum, b_bgr, _ = ref_render(sc, w, h, spp, depth, 3, 4)
        fin = np.isfinite(a_cum).all(axis=2) & np.isfinite(b_cum).all(axis=2)
        d8 = a_bgr[fin].astype(np.float64) - b_bgr[fin].astype(np.float64)
        rmse8 = float(np.sqrt(np.mean(d8 ** 2)))
        dl = (a_cum[fin].astype(np.float64) - b_cum[fin].astype(np.float64)) / spp
        out["%s_%dx%d_%dspp_d%d" % (sc, w, h, spp, depth)] = {
            "seeds": [[1, 2], [3, 4]], "rmse_8bit": rmse8, "psnr_8bit_db": float(10 * np.log10(255.0 ** 2 / rmse8 ** 2)),
            "rmse_linear_mean": float(np.sqrt(np.mean(dl ** 2))), "finite_pixels": int(fin.sum()), "pixels": int(fin.size)}
        print(sc, out["%s_%dx%d_%dspp_d%d" % (sc, w, h, spp, depth)])
    json.dump(out, open(os.path.join(HERE, "noise.json"), "w"), indent=1)
    print("wrote noise.json")


def g_bounce():
    rng = np.random.default_rng(7)
    names = scenes.build("materials").material_order  # m0..m10 in declaration order = driver's index
    n_per = 96
    recs = np.zeros(len(names) * n_per, BOUNCE_IN)
    k = 0
    for mi in range(len(names)):
        for j in range(n_per):
            nrm = rng.normal(size=3)
            nrm /= np.linalg.norm(nrm)
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            front = j % 4 != 3
            if np.dot(d, nrm) > 0:  # unitNormal always faces the ray (HitInfo_SetFaceNormal)
                d = -d
            if j % 8 == 7:
                nrm = nrm * 0.7  # interpolated normals are not unit length
            if j % 16 == 5:  # grazing incidence
                t = np.cross(nrm, rng.normal(size=3))
                t /= np.linalg.norm(t)
                d = t - 1e-3 * nrm / np.linalg.norm(nrm)
                d /= np.linalg.norm(d)
            recs[k] = (mi, d.astype(np.float32), rng.uniform(-10, 10, 3).astype(np.float32), nrm.astype(np.float32),
                       rng.uniform(0, 1, 2).astype(np.float32), int(front), int(rng.integers(1, 2 ** 62)),
                       int(rng.integers(1, 2 ** 62)))
            k += 1
    fi, fo = os.path.join(TMP, "b_in.bin"), os.path.join(TMP, "b_out.bin")
    recs.tofile(fi)
    ref("materials", "bounce", fi, fo)
    save("bounce_materials.npz", inp=recs, out=np.fromfile(fo, BOUNCE_OUT), names=np.array(names))


def g_sky():
    rng = np.random.default_rng(11)
    d = rng.normal(size=(512, 3)).astype("<f4")
    d[:6] = np.eye(3, dtype=np.float32).repeat(2, 0) * np.array([1, -1] * 3, np.float32)[:, None]
    d[6:12] = [[1, 1, 0], [1, -1, 0], [-1, 0, 1], [0, 1, 1], [1, 1, 1], [-1, -1, -1]]
    fi, fo = os.path.join(TMP, "s_in.bin"), os.path.join(TMP, "s_out.bin")
    d.tofile(fi)
    for sc in ["pear"]:
        ref(sc, "sky", fi, fo)
        save("sky_%s.npz" % sc, dirs=d, rgb=np.fromfile(fo, "<f4").reshape(-1, 3))


GROUPS = dict(rng=g_rng, kd=g_kd, primary=g_primary, trace=g_trace, render=g_render, bounce=g_bounce, sky=g_sky, noise=g_noise, sparse=g_sparse)

if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_parity is missing: make -C oracle ref (needs /root/reference)")
    for g in (sys.argv[1:] or list(GROUPS)):
        if g.startswith("render:"):  # render:<substring of the fixture's name>
            g_render(g[7:])
        elif g.startswith("sparse:"):
            g_sparse(g[7:])
        elif g.startswith("primary:"):  # primary:<scene>
            g_primary(g[8:])
        else:
            GROUPS[g]()
