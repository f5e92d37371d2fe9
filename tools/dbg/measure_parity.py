"""Measure (not assert) how close the GPU is to the reference fixtures: numbers the test bounds are then set from."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from raytracer_amd import hip, scenes
from oracle import binding as O
from util import golden, psnr

out = {}
noise = json.load(open("tests/golden/noise.json"))
names = ["render_teapot_96x54_8spp_d8", "render_bunny_96x54_8spp_d8", "render_dragon1_96x54_8spp_d8", "render_pear_96x54_8spp_d16",
         "render_materials_128x72_16spp_d8", "render_dragon1_240x135_64spp_d8", "render_dragon1_192x108_1024spp_d8"]
built = {}
for name in names:
    g = golden(name + ".npz")
    w, h, spp, depth, lo, hi = map(int, g["meta"])
    scene = name.split("_")[1]
    if scene not in built:
        sc = scenes.build(scene); built[scene] = (sc, hip.DeviceScene(sc.desc, 0))
    sc, ds = built[scene]
    bgr, acc, st = ds.render(sc.rtx_camera(), hip.make_params(w, h, spp, depth, seed=(lo, hi)))
    fr, fg = np.isfinite(g["cum"]).all(axis=2), np.isfinite(acc).all(axis=2)
    fin = fr & fg
    d8 = bgr[fin].astype(np.float64) - g["bgr"][fin].astype(np.float64)
    rel = np.abs(acc[fin] - g["cum"][fin]).max(axis=1) / np.maximum(np.abs(g["cum"][fin]).max(axis=1), 1e-3)
    # per-sample view: a pixel's sum differs by at most k samples' worth
    out[name] = dict(psnr=psnr(bgr[fin], g["bgr"][fin]), rmse8=float(np.sqrt(np.mean(d8 ** 2))), nonfinite_ref=int((~fr).sum()), nonfinite_gpu=int((~fg).sum()),
                     nonfinite_differ=int((fr != fg).sum()), frac_1e4=float((rel <= 1e-4).mean()), frac_1e5=float((rel <= 1e-5).mean()), frac_exact=float((rel == 0).mean()),
                     rmse_lin=float(np.sqrt(np.mean(((acc[fin] - g["cum"][fin]).astype(np.float64) / spp) ** 2))),
                     noise=noise.get(name[7:]))
    print(name, out[name], flush=True)

# bounce records
g = golden("bounce_materials.npz")
inp, outp, mnames = g["inp"], g["out"], [str(x) for x in g["names"]]
sc, ds = built["materials"]
b = {}
for mi, name in enumerate(mnames):
    a, want = inp[inp["mat"] == mi], outp[inp["mat"] == mi]
    state = np.array([O.rng_state_after_seed(int(lo), int(hi)) for lo, hi in zip(a["seed_lo"], a["seed_hi"])], np.uint32)
    got = ds.material_bounce(sc.material_index(name), a["dir"], a["pos"], a["n"], a["uv"], a["front"].astype(np.uint8), state)
    ref = O.material_bounce(sc.desc, sc.material_index(name), a["dir"], a["pos"], a["n"], a["uv"], a["front"].astype(np.uint8), state)
    same = got["cont"] == want["cont"].astype(np.uint8)
    rs = (got["rng_state"] == ref["rng_state"]).all(axis=1)
    rec = dict(n=len(a), cont_mismatch=np.nonzero(~same)[0].tolist(), rng_mismatch=np.nonzero(~rs)[0].tolist(), oracle_cont_vs_ref=int((ref["cont"] != want["cont"].astype(np.uint8)).sum()))
    for k, r in [("surface", "surf"), ("emitted", "emit"), ("out_dir", "od")]:
        x, y = got[k], want[r]
        both_fin = np.isfinite(y).all(axis=1) & np.isfinite(x).all(axis=1)
        rec[k + "_nonfinite_differ"] = np.nonzero(np.isfinite(y).all(axis=1) != np.isfinite(x).all(axis=1))[0].tolist()
        m = both_fin & same & ((want["cont"] != 0) if k == "out_dir" else True)
        err = np.abs(x[m] - y[m]) / np.maximum(np.abs(y[m]), 1e-3)
        e = err.max(axis=1) if err.size else np.zeros(0)
        idx = np.nonzero(m)[0]
        rec[k + "_max"] = float(e.max(initial=0)); rec[k + "_gt1e5"] = [(int(idx[i]), float(e[i])) for i in np.nonzero(e > 1e-5)[0]]
    b[name] = rec
    print(name, rec, flush=True)
out["bounce"] = b
json.dump(out, open("gpurun_out/measure_parity.json", "w"), indent=1)
