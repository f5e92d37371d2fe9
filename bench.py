#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one whole frame of the hot path (raygen → [closest-hit → shade/bounce] x depth →
resolve) over synthetic input: the dragon stand-in scene (BASELINE.json's `metric` is quoted on
dragon.obj, which the reference does not ship — see raytracer_amd/assets.py), Disney
glass+clearcoat mixture, light sphere, skybox-material ground plane, 1920x1080, 64 spp, depth 8.
Scene pools are resident in HBM before the timed region; the frame is rendered into a device
f32 accumulation buffer, reduced to rank 0 (RCCL, only for N > 1) and tone-mapped on rank 0.

For N > 1 launch as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
one rank per GPU; the same frame is split by interleaved 8-row bands ("strong" scaling: the total
work is fixed), timed between barriers, MAX over ranks; rank 0 prints ONE JSON line.

The oracle / reference binaries are used only for the `cpu_baseline` leg (rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def alg_bytes(vi, vl, vp, vc, rays):
    """Algorithmic bytes of the closest-hit kernel, SURVEY §8(d)'s per-unit figure: 8 B per inner node and per
    leaf visited, (4 B leaf ref + 36 B vertex positions) per primitive test, 64 B ray record read + 44 B hit
    record write per ray the kernel is handed.  The N_* are the kernel's own counters (culled walk: children whose box the ray
    misses are never fetched; vc only counts them)."""
    return 8 * vi + 8 * vl + 40 * vp + (64 + 44) * rays


def fetched_bytes(vi, vl, vp, vc, rays):
    """What the culled walk actually requests: an inner node is fetched as 32 B (the 8-B kd node + the two
    12-B boxes of its children) — reported beside the roofline line, not used for it."""
    return 32 * vi + 8 * vl + 40 * vp + (64 + 44) * rays


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, cut by the cgroup's CPU quota and by the pool's own limit
    (OMP_NUM_THREADS is set to the box's CPU share on the GPU pool, where os.cpu_count() shows the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        n = min(n, max(1, int(os.environ["OMP_NUM_THREADS"])))
    except Exception:
        pass
    return n


REF_PERF_THREADS = (16, 32, 64, 96, 128, 192, 256)  # oracle/Makefile builds _ref/ref_perf_t<N>: NUM_HYPERTHREADS = N


def cpu_baseline(scene_name, w, h, spp, depth, rays_fn):
    """Time the reference's own multithreaded renderer (oracle/_ref/ref_perf_t<N>: unmodified sources, release flags,
    NUM_HYPERTHREADS = N worker threads through a replacement force-include prelude) on a bounded sample of
    the workload.  N = the variant that fits the CPUs this process may use; when the machine shows more CPUs than that,
    the variant fitting the whole machine is timed as well and the faster of the two is reported."""
    from oracle import binding as O
    from raytracer_amd import assets, scenes
    host, usable = os.cpu_count() or 1, usable_cpus()
    fit = lambda n: max([t for t in REF_PERF_THREADS if t <= max(n, REF_PERF_THREADS[0])])
    variants = sorted({fit(usable), fit(host)})
    out = {"value": None, "unit": "Mrays/s (1280x720 sub-frame of the workload, full spp and depth)", "cores": None, "host_cpus": host, "usable_cpus": usable,
           "kind": "reference", "sample": "%s %dx%d %d spp depth %d (reduced frame of the same scene, full spp)" % (scene_name, w, h, spp, depth)}
    try:
        rays, tried = rays_fn(w, h, spp, depth), []
        for t in variants:
            # 16 threads is the reference's own NUM_HYPERTHREADS (src/common/config.h:23): the stock build with the reference's
            # own prelude (ref_perf) is the baseline there; ref_perf_t<N> (prelude_threads.h) only where N differs
            kind = "perf_t%d" % t if t != 16 and O.ref_binary("perf_t%d" % t) else "perf"
            r = O.ref_run(kind, scenes.write_recipe(scene_name), assets.cache_dir(), "bench", w, h, spp, depth, timeout=900)
            tried.append({"threads": int(r["threads"]), "seconds": r["seconds"], "mrays_s": rays / r["seconds"] / 1e6})
        best = max(tried, key=lambda x: x["mrays_s"])
        out.update(value=best["mrays_s"], cores=best["threads"], seconds=best["seconds"], rays=rays, tried=tried, s_per_full_frame_extrapolated=None)
    except Exception as e:  # binary absent or cannot run on this host: fall back to the CPU port (oracle)
        print("bench.py: cpu_baseline: the reference binary could not be timed (%s); timing the oracle port instead — kind \"port\""
              % str(e)[:200], file=sys.stderr, flush=True)
        from raytracer_amd import hip
        sc = scenes.build(scene_name)
        p = hip.make_params(w, h, max(1, spp // 8), depth)
        t0 = time.time()
        _, _, rays = O.render(sc.desc, sc.rtx_camera(), p, threads=min(usable, 64), want_bgr=False)
        dt = time.time() - t0
        out.update(kind="port", cores=min(usable, 64), value=float(rays.sum()) / dt / 1e6, seconds=dt, rays=int(rays.sum()),
                   sample=out["sample"].replace("%d spp" % spp, "%d spp" % max(1, spp // 8)), note="ref_perf unavailable: %s" % str(e)[:120])
    return out


def kernel_source_hash():
    """sha256 over the sources of the kernels: profiles/traffic.json carries the hash of the build its counters were
    measured on (tools/profile_round.sh), and the counters are attached to a bench line only when it is this build's."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rt_hip.hip", "rtx_device.h", "kd_build.h"):
        h.update(open(os.path.join(REPO, "raytracer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def in_process(args):
    """--in-process N: the frame shared by N devices INSIDE librt_hip.so (rtx_render_multi: one host thread per device,
    peer copies to device 0) — the path a C caller of include/rtx.h gets — timed beside the torch.distributed path.  With
    fewer than N devices visible the handles wrap around the devices there are (a rehearsal, labelled)."""
    import numpy as np
    from raytracer_amd import hip, scenes
    n = args.in_process
    ndev = hip.device_count()
    sc = scenes.build(args.scene)
    cam = sc.rtx_camera()
    dss = [hip.DeviceScene(sc.desc, r % ndev) for r in range(n)]
    p = hip.make_params(args.width, args.height, args.spp, args.depth, seed=(1, 2))
    split = hip.RTX_SPLIT_TILE if args.split == "tile" else hip.RTX_SPLIT_SAMPLE
    for _ in range(args.warmup):
        hip.render_multi(dss, cam, p, split, want_accum=False, want_bgr=False)
    t0 = time.perf_counter()
    sts = [hip.render_multi(dss, cam, p, split, want_accum=False, want_bgr=False)[2] for _ in range(args.steps)]
    dt = time.perf_counter() - t0
    rays = float(sum(s["rays_total"] for st in sts for s in st))
    bgr, acc, _ = hip.render_multi(dss, cam, p, split)
    _, whole, _ = dss[0].render(cam, p, want_bgr=False)
    same = bool(np.array_equal(acc.view(np.uint32), whole.view(np.uint32))) if split == hip.RTX_SPLIT_TILE else bool(np.allclose(acc, whole, rtol=1e-5, atol=1e-5, equal_nan=True))
    out = {"metric": "Mrays/s and s/frame, dragon.obj 1080p 64spp, at 1/2/4/8 MI355X", "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": n,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "s_per_frame": dt / args.steps, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "verified": same,
           "config": {"workload": "%s %dx%d %d spp depth %d, %s split" % (args.scene, args.width, args.height, args.spp, args.depth, args.split),
                      "driver": "rtx_render_multi (in-process, one host thread per device, peer copies to device 0)",
                      "devices_visible": ndev, "rehearsal_on_fewer_devices": ndev < n,
                      "slowest_device_ms": max(s["ms_total"] for s in sts[-1])}}
    print(json.dumps(out), flush=True)
    for ds in dss:
        ds.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="dragon1")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--split", default="tile", choices=["tile", "sample"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="after timing: the reduced frame against a single-rank render of the same frame")
    ap.add_argument("--no-verify", action="store_true", help="N > 1 verifies by default")
    ap.add_argument("--in-process", type=int, default=0, metavar="N", help="share the frame among N devices inside librt_hip.so (rtx_render_multi) instead of one rank per GPU")
    ap.add_argument("--save-bmp", default="")
    args = ap.parse_args()
    if args.in_process > 0:
        return in_process(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched through torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    import torch
    from raytracer_amd import hip, scenes
    from raytracer_amd.distributed import FramePlan, reduce_to_rank0

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU path to benchmark)")
    # RTX_BENCH_REHEARSAL=1: every rank on GPU 0 over gloo — to rehearse the N > 1 path on a one-GPU box
    rehearsal = os.environ.get("RTX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    sc = scenes.build(args.scene)
    ds = hip.DeviceScene(sc.desc, local_rank)
    cam = sc.rtx_camera()
    plan = FramePlan(args.width, args.height, args.spp, args.depth, seed=(1, 2), split=args.split)
    params = plan.rank_params(rank, world)
    accum = torch.zeros(args.height, args.width, 3, dtype=torch.float32, device=dev)
    bgr = torch.zeros(args.height, args.width, 3, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def frame(p):
        accum.zero_()
        st = ds.render_device(cam, p, accum.data_ptr(), stream)
        reduce_to_rank0(accum)
        if rank == 0:
            ds.resolve_device(accum.data_ptr(), args.width, args.height, args.spp, bgr.data_ptr(), stream)
        return st

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # untimed counted pass: kd-tree visit totals of this rank's share (deterministic per frame)
    p_count = plan.rank_params(rank, world)
    p_count.flags |= hip.RTX_RENDER_COUNT_VISITS
    st_count = frame(p_count)
    for _ in range(args.warmup):
        frame(params)
    barrier()
    t0 = time.perf_counter()
    stats = [frame(params) for _ in range(args.steps)]
    barrier()
    dt = time.perf_counter() - t0

    verified = None
    if args.verify or (world > 1 and not args.no_verify):  # N > 1: always, so that a scaling line carries "verified"
        # The frame the ranks just produced together against the whole frame rendered by rank 0 alone.  Tile split: every
        # pixel is written by exactly one rank and the reduce adds zeros to it, so the two buffers are equal BIT FOR BIT;
        # sample split: the same terms in another f32 association (rtol 1e-5).  At N = 1 the check is the partition
        # itself: two band-halves rendered one after the other must reassemble the frame.
        import hashlib
        import numpy as np
        frame(params)
        torch.cuda.synchronize(dev)
        if world == 1:
            whole = accum.clone()
            accum.zero_()
            for r in range(2):
                ds.render_device(cam, plan.rank_params(r, 2), accum.data_ptr(), stream)
            torch.cuda.synchronize(dev)
            parts = accum
        elif rank == 0:
            parts = accum.clone()
            accum.zero_()
            ds.render_device(cam, plan.rank_params(0, 1), accum.data_ptr(), stream)
            torch.cuda.synchronize(dev)
            whole = accum
        if rank == 0:
            a, b = whole.cpu().numpy(), parts.cpu().numpy()
            if args.split == "tile" or world == 1:
                verified = hashlib.sha256(a.tobytes()).hexdigest() == hashlib.sha256(b.tobytes()).hexdigest()
            else:
                verified = bool(np.allclose(a, b, rtol=1e-5, atol=1e-5, equal_nan=True))
        if world > 1:
            dist.barrier()

    rays_rank = float(sum(s["rays_total"] for s in stats))
    vals = torch.tensor([dt, rays_rank, sum(s["ms_trace"] for s in stats), float(sum(s["trace_launches"] for s in stats)),
                         float(alg_bytes(st_count["visits_inner"], st_count["visits_leaf"], st_count["visits_prim"], st_count["visits_culled"],
                                         st_count["rays_total"] - st_count["rays_early_miss"])),
                         float(st_count["rays_total"]), float(st_count["visits_inner"]), float(st_count["visits_leaf"]), float(st_count["visits_prim"]),
                         sum(s["ms_shade"] for s in stats), sum(s["ms_raygen"] for s in stats), sum(s["ms_resolve"] for s in stats),
                         float(st_count["visits_culled"]), sum(s["ms_sort"] for s in stats),
                         float(fetched_bytes(st_count["visits_inner"], st_count["visits_leaf"], st_count["visits_prim"], st_count["visits_culled"],
                                             st_count["rays_total"] - st_count["rays_early_miss"])),
                         float(st_count["rays_early_miss"])],
                        dtype=torch.float64, device=dev)
    tmax = vals[:1].clone()
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
    if rank == 0:
        t = float(tmax[0])
        v = vals.tolist()
        rays_total, ms_trace, launches, bytes_frame, rays_frame = v[1], v[2], v[3], v[4], v[5]
        # roofline of the dominant kernel (closest hit): algorithmic bytes per launch / mean launch time.
        # ms_trace sums HIP-event spans recorded around each launch on the stream it runs on.
        launches_per_frame = launches / args.steps / world
        alg_per_launch = bytes_frame / world / max(launches_per_frame, 1)
        ms_per_launch = ms_trace / max(launches, 1)
        achieved = alg_per_launch / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
        traffic, counters = None, {}
        tj = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tj):  # measured in a separate rocprofv3 --pmc run of this same command (profiles/)
            tr = json.load(open(tj))
            stale = tr.get("kernel_source_hash") != kernel_source_hash()
            if stale:
                print("bench.py: profiles/traffic.json was measured on another build of the kernels (%s, hash %s; this build %s): "
                      "roofline.traffic and roofline.counters left out" % (tr.get("measured_at"), tr.get("kernel_source_hash"), kernel_source_hash()),
                      file=sys.stderr, flush=True)
            if not stale and tr.get("workload") == "%s %dx%d %dspp d%d" % (args.scene, args.width, args.height, args.spp, args.depth) and world == 1:
                traffic = tr.get("hbm_bytes_per_launch")
                counters = {k: tr[k] for k in ("hbm_gbs", "l2_hit_rate", "valu_busy", "valu_lane_utilisation", "valu_insts_per_traced_ray") if tr.get(k) is not None}
                # NOT measured in this run: copied from the rocprofv3 --pmc passes of this same command kept in profiles/
                counters["source"] = "profiles/traffic.json (separate rocprofv3 --pmc runs of this command, tools/profile_round.sh)"
                counters["measured_at"] = tr.get("measured_at", "round 1, kernel v18")
                counters["kernel_source_hash"] = tr.get("kernel_source_hash")
        out = {
            "metric": "Mrays/s and s/frame, dragon.obj 1080p 64spp, at 1/2/4/8 MI355X",
            "value": rays_total / t / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t / args.steps, "s_per_frame": t / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "dragon stand-in (bunny subdivided x%d, %d triangles), Disney glass+clearcoat, %dx%d, %d spp, depth %d, %s split"
                                   % (int(args.scene[6:] or 1) if args.scene.startswith("dragon") else 0, sc.desc.contents.n_tris, args.width,
                                      args.height, args.spp, args.depth, args.split),
                       "scene": args.scene, "triangles": int(sc.desc.contents.n_tris), "kd_nodes": int(sc.desc.contents.n_nodes),
                       "rays_per_frame": rays_frame, "rays_per_camera_path": rays_frame / (args.width * args.height * args.spp),
                       # of rays_per_frame: bounce rays that miss the box of the whole tree and the ground, answered by the shade
                       # pass that creates them (no record in the closest-hit kernel's queue: not in its algorithmic bytes)
                       "rays_answered_in_shade": v[15],
                       # the headline divides ALL Scene_ClosestHit queries by the wall time; these two divide only the rays
                       # handed to the closest-hit kernel, by the wall time and by that kernel's own time
                       "trace_kernel_mrays_s": (rays_frame - v[15]) * args.steps / t / 1e6,
                       "trace_kernel_mrays_s_kernel_time": (rays_frame - v[15]) / max(ms_trace / args.steps / world, 1e-9) / 1e3,
                       "visits_per_ray": {"inner": v[6] / rays_frame, "leaf": v[7] / rays_frame, "prim": v[8] / rays_frame,
                                          "children_culled_unfetched": v[12] / rays_frame},
                       "kernel_ms_per_frame": {"trace": ms_trace / args.steps / world, "shade": v[9] / args.steps / world,
                                               "raygen": v[10] / args.steps / world, "resolve": v[11] / args.steps / world,
                                               "ray_binning": v[13] / args.steps / world}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_trace_flat<false,true,true,false,false>", "alg_bytes_per_launch": alg_per_launch, "ms_per_launch": ms_per_launch,
                         "alg_bytes_per_ray": bytes_frame / rays_frame, "requested_bytes_per_ray_with_child_boxes": v[14] / rays_frame,
                         # rocprofv3 --pmc passes of this same command (profiles/, tools/profile_round.sh): HBM-side GB/s of the
                         # kernel, L2 (TCC) hit rate, share of SIMD cycles issuing vector instructions, active lanes per instruction
                         "counters": counters},
        }
        if verified is not None:
            out["verified"] = verified
        if world == 1 and not args.no_cpu_baseline:
            def gpu_rays(w, h, spp, depth):
                a = torch.zeros(h, w, 3, dtype=torch.float32, device=dev)
                s = ds.render_device(cam, hip.make_params(w, h, spp, depth, seed=(1, 2)), a.data_ptr(), stream)
                return s["rays_total"]
            out["cpu_baseline"] = cpu_baseline(args.scene, 1280, 720, args.spp, args.depth, gpu_rays)
            if out["cpu_baseline"].get("value"):  # the same frame at the CPU's measured rate (the sample is 4/9 of its pixels)
                out["cpu_baseline"]["s_per_full_frame_extrapolated"] = rays_frame / (out["cpu_baseline"]["value"] * 1e6)
        if args.save_bmp:
            from raytracer_amd import host
            import ctypes as C
            img = host.ImageRGB()
            host.lib().ImageRGB_Load_Empty(C.byref(img), args.width, args.height)
            C.memmove(img.pix, bgr.cpu().numpy().ctypes.data, args.width * args.height * 3)
            host.lib().ImageRGB_Save_BMP_Path(C.byref(img), args.save_bmp.encode())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ds.close()


if __name__ == "__main__":
    main()
