"""What the fat leaves of a scene hold (CPU only): for the leaves with the most references, how many of the referenced
triangles share one vertex, and how much of the leaf's extent each triangle's own bounds cover per axis.
    python tools/dbg/fat_leaves.py pear"""
import sys
import numpy as np

sys.path.insert(0, ".")
from raytracer_amd import scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pear"
b = scenes.build(name)
d = b.desc.contents if hasattr(b.desc, "contents") else b.desc
n = d.n_nodes
w0 = np.ctypeslib.as_array(d.node_w0, (n,)).copy()
w1 = np.ctypeslib.as_array(d.node_w1, (n,)).copy()
refs = np.ctypeslib.as_array(d.leaf_ref, (d.n_refs,)).copy()
tri = np.ctypeslib.as_array(d.tri_pos, (9 * d.n_tris,)).reshape(3, 3, d.n_tris).transpose(2, 0, 1).copy()  # [c * T + i], c = 3 * vertex + axis
leaf = (w1 >> 30) == 3
cnt = np.where(leaf, w0, 0)
order = np.argsort(-cnt.astype(np.int64))[:12]
print(f"{name}: {d.n_tris} triangles, {int(leaf.sum())} leaves, {int(cnt.sum())} references, "
      f"leaves with > 64 references: {int((cnt > 64).sum())} holding {int(cnt[cnt > 64].sum())} references")
for i in order:
    c, first = int(cnt[i]), int(w1[i] & 0x3FFFFFFF)
    r = refs[first:first + c]
    r = r[(r >> 31) == 0] & 0x3FFFFFFF
    t = tri[r]                      # (c, 3 vertices, xyz)
    verts, counts = np.unique(t.reshape(-1, 3), axis=0, return_counts=True)
    lo, hi = t.min(axis=(0, 1)), t.max(axis=(0, 1))
    ext = np.maximum(hi - lo, 1e-30)
    cover = ((t.max(axis=1) - t.min(axis=1)) / ext)   # per triangle, per axis: share of the union's extent
    print(f"  leaf {i}: {c} refs; the most shared vertex is in {int(counts.max())} of them; union extent {ext.round(3)}; "
          f"a triangle's own extent / the union's, median per axis {np.median(cover, axis=0).round(2)}")
b.close()
