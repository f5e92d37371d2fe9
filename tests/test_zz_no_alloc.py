"""rtx_closest_hit_noalloc (include/rtx.h): the allocation-free variant of rtx_trace_closest, for callers that cannot
call hipMalloc.  Its staging lives in a static device array; the outputs must be rtx_trace_closest's, bit for bit
(same kernels on the same rays) — on two calls in a row, the second smaller, with another scene's call in between
(every scene on a GPU shares that GPU's array), on a side stream, and with the literal walk."""
import numpy as np
import pytest

from raytracer_amd import hip
from util import golden, same_bits

pytestmark = pytest.mark.gpu


def _noalloc(ds, o, d, flags=0, stream=None):
    import torch
    dev = torch.device("cuda", 0)
    n = len(o)
    to = torch.from_numpy(np.ascontiguousarray(o, np.float32)).to(dev)
    td = torch.from_numpy(np.ascontiguousarray(d, np.float32)).to(dev)
    out = dict(prim_id=torch.full((n,), 0x55555555, dtype=torch.int32, device=dev), t=torch.full((n,), -7.0, device=dev),
               position=torch.full((n, 3), -7.0, device=dev), normal=torch.full((n, 3), -7.0, device=dev),
               uv=torch.full((n, 2), -7.0, device=dev), front=torch.full((n,), 9, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize(dev)
    s = stream if stream is not None else torch.cuda.current_stream(dev)
    ds.closest_hit_noalloc(n, to.data_ptr(), td.data_ptr(), out["prim_id"].data_ptr(), out["t"].data_ptr(), out["position"].data_ptr(),
                           out["normal"].data_ptr(), out["uv"].data_ptr(), out["front"].data_ptr(), stream=s.cuda_stream, flags=flags)
    s.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res["prim_id"] = res["prim_id"].view(np.uint32)
    return res


def _same(a, b):
    assert np.array_equal(a["prim_id"], b["prim_id"]) and np.array_equal(a["front"], b["front"])
    for k in ("t", "position", "normal", "uv"):
        assert same_bits(a[k], b[k]), k


def test_noalloc_entry_matches_trace_closest(built_scenes, device_scenes):
    import torch
    g = golden("trace_teapot.npz")
    o, d = g["rays"][:, :3].copy(), g["rays"][:, 3:].copy()  # un-normalised directions: Ray_Make runs on the device
    ds, other = device_scenes("teapot"), device_scenes("materials")
    n1 = len(o)
    n2 = n1 // 3
    want1 = ds.trace_closest(o, d)
    got1 = _noalloc(ds, o, d)
    _same(got1, want1)
    assert (want1["prim_id"] != 0xFFFFFFFF).mean() > 0.03
    # second call, smaller, different rays (the tail of the first batch, reversed): nothing of the first call may show
    o2, d2 = o[::-1][:n2].copy(), d[::-1][:n2].copy()
    _same(_noalloc(ds, o2, d2), ds.trace_closest(o2, d2))
    # another scene on the same GPU uses the same array in between
    gm = golden("trace_materials.npz")
    om, dm = gm["rays"][:, :3].copy(), gm["rays"][:, 3:].copy()
    _same(_noalloc(other, om, dm), other.trace_closest(om, dm))
    # literal walk, unit directions, a side stream
    rec = golden("primary_teapot_B.npz")["rec"]
    fl = hip.RTX_TRACE_UNIT_DIR | hip.RTX_TRACE_LITERAL
    _same(_noalloc(ds, rec["o"], rec["d"], flags=fl, stream=torch.cuda.Stream(torch.device("cuda", 0))), ds.trace_closest(rec["o"], rec["d"], flags=fl))
    _same(_noalloc(ds, o, d), want1)  # and the first batch again


def test_noalloc_entry_limits(device_scenes):
    import torch
    ds = device_scenes("teapot")
    dev = torch.device("cuda", 0)
    n = hip.RTX_NOALLOC_MAX_RAYS
    rng = np.random.default_rng(5)
    o = (rng.random((n, 3), dtype=np.float32) * 40 - 20).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    _same(_noalloc(ds, o, d), ds.trace_closest(o, d))  # exactly the maximum
    buf = torch.zeros(3 * (n + 1), device=dev)
    with pytest.raises(hip.RtxError) as e:
        ds.closest_hit_noalloc(n + 1, buf.data_ptr(), buf.data_ptr())
    assert e.value.code == -5 and "RTX_NOALLOC_MAX_RAYS" in str(e.value)
    with pytest.raises(hip.RtxError):
        ds.closest_hit_noalloc(16, buf.data_ptr(), buf.data_ptr(), flags=hip.RTX_RENDER_COUNT_VISITS)
    ds.closest_hit_noalloc(0, 0, 0)  # empty batch: nothing to do


def test_noalloc_entry_allocates_nothing(device_scenes):
    """The property the entry point exists for: it can be captured into a HIP graph — a hipMalloc, hipFree or
    event creation inside a capture fails or invalidates it — and replaying the graph gives rtx_trace_closest's bits; the
    free device memory is the same before and after plain calls."""
    import torch
    dev = torch.device("cuda", 0)
    ds = device_scenes("teapot")
    g = golden("trace_teapot.npz")
    o, d = g["rays"][:, :3].copy(), g["rays"][:, 3:].copy()
    n = len(o)
    want = ds.trace_closest(o, d)
    to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    out = dict(prim_id=torch.zeros(n, dtype=torch.int32, device=dev), t=torch.zeros(n, device=dev), position=torch.zeros(n, 3, device=dev),
               normal=torch.zeros(n, 3, device=dev), uv=torch.zeros(n, 2, device=dev), front=torch.zeros(n, dtype=torch.uint8, device=dev))

    def call(stream):
        ds.closest_hit_noalloc(n, to.data_ptr(), td.data_ptr(), out["prim_id"].data_ptr(), out["t"].data_ptr(), out["position"].data_ptr(),
                               out["normal"].data_ptr(), out["uv"].data_ptr(), out["front"].data_ptr(), stream=stream.cuda_stream)

    side = torch.cuda.Stream(dev)
    call(side)  # warm-up outside the capture (the first call on a scene asks the runtime for the kernel's occupancy)
    side.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    for _ in range(3):
        call(side)
    side.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free0
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        call(torch.cuda.current_stream(dev))
    for v in out.values():
        v.zero_()
    torch.cuda.synchronize(dev)
    graph.replay()
    torch.cuda.synchronize(dev)
    got = {k: v.cpu().numpy() for k, v in out.items()}
    got["prim_id"] = got["prim_id"].view(np.uint32)
    _same(got, want)
