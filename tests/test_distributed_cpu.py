"""Multi-process frame partition on CPU: world_size-2 `gloo` group, each rank renders its
share with the ORACLE standing in for the GPU (same rtx_render_params contract), the shares are
sum-reduced to rank 0 exactly as bench.py does over RCCL, and rank 0's frame must equal the
single-process frame: bit-exact for the tile split (disjoint rows), to f32 summation order for the
sample split."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, split, out_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as O
    from raytracer_amd import scenes
    from raytracer_amd.distributed import FramePlan, reduce_to_rank0
    sc = scenes.build("materials")
    plan = FramePlan(64, 38, 4, 4, seed=(5, 6), split=split, band_rows=16)
    cum, _, _ = O.render(sc.desc, sc.rtx_camera(), plan.rank_params(rank, world), threads=2, want_bgr=False)
    if split == "tile":  # a rank must not touch rows it does not own
        mine = set(plan.rank_rows(rank, world))
        other = [y for y in range(plan.height) if y not in mine]
        assert not cum[other].any() and cum[sorted(mine)].any()
    acc = torch.from_numpy(cum)
    reduce_to_rank0(acc)
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split", ["tile", "sample"])
def test_two_rank_partition_reassembles_the_frame(tmp_path, split, built_scenes):
    from oracle import binding as O
    from raytracer_amd.distributed import FramePlan
    out = str(tmp_path / "acc.npy")
    port = 29500 + (os.getpid() % 2000) + (0 if split == "tile" else 1)
    mp.spawn(_worker, args=(2, port, split, out), nprocs=2, join=True)
    got = np.load(out)
    sc = built_scenes("materials")
    plan = FramePlan(64, 38, 4, 4, seed=(5, 6))
    want, _, _ = O.render(sc.desc, sc.rtx_camera(), plan.rank_params(0, 1), threads=4, want_bgr=False)
    if split == "tile":
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    assert np.array_equal(O.resolve(got, 4), O.resolve(want, 4)) or split == "sample"


def test_plan_covers_every_row_and_sample_once():
    from raytracer_amd.distributed import FramePlan
    plan = FramePlan(32, 70, 7, 3, band_rows=16)
    for world in (1, 2, 3, 8):
        rows = sorted(y for r in range(world) for y in plan.rank_rows(r, world))
        assert rows == list(range(70))
    plan.split = "sample"
    for world in (1, 2, 7):
        cover = []
        for r in range(world):
            p = plan.rank_params(r, world)
            cover += list(range(p.sample_begin, p.sample_end or 7))
        assert sorted(cover) == list(range(7))
    with pytest.raises(ValueError):
        plan.rank_params(0, 8)
