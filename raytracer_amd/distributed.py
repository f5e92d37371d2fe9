"""Frame partition across the GPUs of one node — one process per GPU.

Every (pixel, sample) path is independent and carries its own RNG stream keyed by the GLOBAL
(pixel, sample) index, so a frame is sharded with no data-path exchange until the end:

  tile split    rows are grouped in bands of `band_rows`; band b goes to rank b % world
                (interleaved, because the cost concentrates where the mesh projects).  Each rank
                fills its rows of a zero-initialised full-frame f32 accumulation buffer.
  sample split  rank r traces samples [r*spp/world, (r+1)*spp/world) of every pixel.

The one exchange step is a sum-reduce of the f32 accumulation buffer to rank 0
(torch.distributed.reduce — RCCL over xGMI with the nccl backend, gloo in the CPU tests).  With a
tile split the supports are disjoint, so the sum is a gather and the frame is BIT-EXACT the single-GPU
frame.  With a sample split every rank forms the sequential sum of ITS samples and the reduce adds those
partial sums: the same terms as RenderPixel's sequential sum (src/rt/renderer.c:98-107) in another f32
association, equal to rounding (rtol 1e-5 in the tests), not bit for bit.  Rank 0 then applies
avg = sum * (1/spp) and RGB_FromColor.
"""
from dataclasses import dataclass

from .hip import make_params


@dataclass
class FramePlan:
    width: int
    height: int
    spp: int
    max_depth: int
    seed: tuple = (1, 2)
    split: str = "tile"  # "tile" | "sample"
    band_rows: int = 8  # slowest rank at N = 8: 20.2 ms with 8-row bands, 20.8 with 16, 21.6 with 32 (tools/dbg/bandrows.py)
    flags: int = 0
    max_paths_in_flight: int = 0

    def rank_params(self, rank: int, world: int):
        """rtx_render_params of the share of `rank` out of `world` processes."""
        if world <= 1:
            return make_params(self.width, self.height, self.spp, self.max_depth, self.seed, flags=self.flags,
                               max_paths_in_flight=self.max_paths_in_flight)
        if self.split == "tile":
            return make_params(self.width, self.height, self.spp, self.max_depth, self.seed,
                               band=(self.band_rows, rank, world), flags=self.flags,
                               max_paths_in_flight=self.max_paths_in_flight)
        if self.split == "sample":
            if self.spp < world:
                raise ValueError("sample split needs spp >= world size")
            lo, hi = rank * self.spp // world, (rank + 1) * self.spp // world
            return make_params(self.width, self.height, self.spp, self.max_depth, self.seed, sample_range=(lo, hi),
                               flags=self.flags, max_paths_in_flight=self.max_paths_in_flight)
        raise ValueError(self.split)

    def rank_rows(self, rank: int, world: int):
        """image rows owned by `rank` under the tile split"""
        if world <= 1 or self.split != "tile":
            return list(range(self.height))
        return [y for y in range(self.height) if (y // self.band_rows) % world == rank]


def reduce_to_rank0(accum, group=None):
    """Sum-reduce the f32 accumulation tensor to rank 0 (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if accum.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal on one GPU: gloo has no CUDA reduce
            dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM, group=group)
    return accum
