"""Host side (librtx_host.so): OBJ import, SAH kd-tree build, flattener — no GPU.

The kd-tree must be the reference's tree: tests/golden/kd.json holds node/leaf/ref counts and
SHA-256 digests of the node words and leaf references dumped from the real reference's KDTree_New
(oracle/ref_driver.c `kd`), for every parity scene including the 278k-triangle dragon stand-in.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from raytracer_amd import assets, host, scenes
from util import GOLDEN, golden

KD = json.load(open(os.path.join(GOLDEN, "kd.json")))


def _tree_arrays(sc):
    d = sc.desc.contents
    w0 = np.ctypeslib.as_array(d.node_w0, (d.n_nodes,))
    w1 = np.ctypeslib.as_array(d.node_w1, (d.n_nodes,))
    lr = np.ctypeslib.as_array(d.leaf_ref, (d.n_refs,))
    tri_obj = np.ctypeslib.as_array(d.tri_obj, (max(d.n_tris, 1),))
    sph_obj = np.ctypeslib.as_array(d.sph_obj, (max(d.n_spheres, 1),))
    idx = lr & 0x7FFFFFFF
    kd = np.where(lr >> 31, sph_obj[np.minimum(idx, len(sph_obj) - 1)], tri_obj[np.minimum(idx, len(tri_obj) - 1)])
    return np.stack([w0, w1], axis=1).astype(np.uint32), kd.astype(np.uint32), d


@pytest.mark.parametrize("scene", list(KD))
def test_kdtree_is_the_reference_tree(built_scenes, scene):
    want = KD[scene]
    nodes, refs, d = _tree_arrays(built_scenes(scene))
    st = built_scenes(scene).kd_stats()
    assert (st["nodes"], st["leaves"], st["empty_leaves"], st["refs"], st["max_leaf_len"]) == (
        want["nodes"], want["leaves"], want["empty_leaves"], want["refs"], want["max_leaf_len"])
    assert d.root_index == want["root"]
    assert d.n_tris + d.n_spheres == want["kd_objects"] and d.n_planes == want["unbounded"]
    assert hashlib.sha256(nodes.tobytes()).hexdigest() == want["sha256_nodes"]
    assert hashlib.sha256(refs.tobytes()).hexdigest() == want["sha256_refs"]


def test_small_tree_node_for_node(built_scenes):
    g = golden("kd_materials.npz")
    nodes, refs, _ = _tree_arrays(built_scenes("materials"))
    assert np.array_equal(nodes, g["nodes"]) and np.array_equal(refs, g["refs"])


def test_right_child_is_parent_plus_one_and_indices_in_range(built_scenes):
    nodes, _, d = _tree_arrays(built_scenes("teapot"))
    typ, idx = nodes[:, 1] >> 30, nodes[:, 1] & 0x3FFFFFFF
    inner = typ != 3
    assert (idx[inner] < d.n_nodes).all() and (np.nonzero(inner)[0] + 1 < d.n_nodes).all()
    leaf = ~inner
    assert (idx[leaf].astype(np.int64) + nodes[leaf, 0] <= d.n_refs).all()
    assert d.max_depth == int(8.0 + 1.8 * np.log2(d.n_tris + d.n_spheres)) + 1


def _write(tmp_path, text):
    p = tmp_path / "m.obj"
    p.write_text(text)
    return str(p).encode()


def test_obj_import_quads_tris_and_default_normals(tmp_path):
    L = host.lib()
    text = ("v 0 0 0\nv 2 0 0\nv 2 2 0\nv 0 2 0\nvt 0 0\nvt 1 0\nvt 1 1\nvn 0 0 1\n"
            "f 1/1/1 2/2/1 3/3/1 4/1/1\nf 1 2 3\nf 1//1 2//1 3//1\nf 1/2 2/3 3/1\n")
    m = L.Mesh_New()
    assert L.Mesh_Import_OBJ_Path(m, _write(tmp_path, text))
    assert L.Mesh_Triangle_Count(m) == 5  # quad -> (0,1,2),(0,2,3) (mesh.c:165-168) + 3 triangles
    t = L.Mesh_Triangles(m)
    assert (t[1].vtx[1].pos.x, t[1].vtx[1].pos.y) == (2.0, 2.0) and t[1].vtx[2].pos.y == 2.0  # (0,2,3)
    assert t[0].vtx[1].tex.x == 1.0 and t[0].vtx[0].norm.z == 1.0
    # plain triangle: normalised default normal (mesh.c:212-214), uv (0,0)
    assert (t[2].vtx[0].norm.x, t[2].vtx[0].norm.y, t[2].vtx[0].norm.z) == (0.0, 0.0, 1.0)
    assert t[2].vtx[2].tex.x == 0.0
    assert t[3].vtx[0].norm.z == 1.0 and t[4].vtx[0].tex.x == 1.0
    L.Mesh_Delete(m)


def test_obj_import_fast_and_sscanf_paths_agree(tmp_path):
    """The importer parses plain lines directly and leaves everything else to the reference's sscanf chain
    (mesh.c:104-153).  The same mesh written plainly and written with signs, hex floats, trailing text and
    trailing blanks — which only sscanf accepts — must import to the same bits."""
    L = host.lib()
    rng = np.random.default_rng(5)
    v = (rng.normal(size=(60, 3)) * 10.0 ** rng.integers(-6, 6, (60, 1))).astype(np.float32)
    vt, vn = rng.random((40, 2)).astype(np.float32), rng.normal(size=(30, 3)).astype(np.float32)
    faces = [(rng.integers(1, 61, 4), rng.integers(1, 41, 4), rng.integers(1, 31, 4), int(rng.integers(0, 4)), int(rng.integers(3, 5)))
             for _ in range(80)]
    r = lambda x: "%.9g" % x  # noqa: E731  (round-trips f32)

    def odd(x, k):  # the same value in a spelling the fast path declines
        x = float(x)
        return [("+" + r(x)) if x >= 0 else "%.12e" % x, float.hex(x), "%+.12e" % x][k % 3]

    def corner(a, b, c, kind, sign):
        return [f"{sign}{a}/{sign}{b}/{sign}{c}", f"{sign}{a}//{sign}{c}", f"{sign}{a}/{sign}{b}", f"{sign}{a}"][kind]

    plain = "".join("v %s %s %s\n" % tuple(map(r, x)) for x in v) + "".join("vt %s %s\n" % tuple(map(r, x)) for x in vt) + \
        "".join("vn %s %s %s\n" % tuple(map(r, x)) for x in vn)
    fancy = "".join("v %s  %s\t%s trailing\n" % (odd(x[0], i), odd(x[1], i + 1), odd(x[2], i + 2)) for i, x in enumerate(v)) + \
        "".join("vt %s %s\n" % (odd(x[0], i), odd(x[1], i + 3)) for i, x in enumerate(vt)) + \
        "".join("vn   %s %s %s\n" % (odd(x[0], i + 1), odd(x[1], i), odd(x[2], i + 2)) for i, x in enumerate(vn))
    for a, b, c, kind, n in faces:
        plain += "f " + " ".join(corner(a[k], b[k], c[k], kind, "") for k in range(n)) + "\n"
        fancy += "f " + " ".join(corner(a[k], b[k], c[k], kind, "+") + ("x" if k == 1 else "") for k in range(n)) + "\n"
    out = []
    for i, text in enumerate((plain, fancy)):
        pth = tmp_path / ("m%d.obj" % i)
        pth.write_text(text)
        m = L.Mesh_New()
        assert L.Mesh_Import_OBJ_Path(m, str(pth).encode()), host.last_error()
        n = L.Mesh_Triangle_Count(m)
        out.append(np.ctypeslib.as_array(C.cast(L.Mesh_Triangles(m), C.POINTER(C.c_uint32)), (n, 24)).copy())
        L.Mesh_Delete(m)
    assert len(out[0]) == sum(2 if f[4] == 4 else 1 for f in faces) and np.array_equal(out[0], out[1])
    # and the plain spelling really is what Python reads from the same text
    m = L.Mesh_New()
    (tmp_path / "one.obj").write_text("v 1.25 -2.5e-3 .5\nv 1 0 0\nv 0 1. 0\nf 1 2 3\n")
    assert L.Mesh_Import_OBJ_Path(m, str(tmp_path / "one.obj").encode())
    t = L.Mesh_Triangles(m)
    assert (t[0].vtx[0].pos.x, t[0].vtx[0].pos.y, t[0].vtx[0].pos.z) == (1.25, float(np.float32(-2.5e-3)), 0.5)
    L.Mesh_Delete(m)


def test_obj_import_fast_float_path_is_strtof(tmp_path):
    """Numbers of up to 15 digits are converted without strtof (exact double arithmetic, then one rounding to
    f32 that is checked for the double-rounding case).  Same decimal text through sscanf — forced by a '+' on
    the line's first number — must give the same bits, over magnitudes, digit counts and spellings."""
    L = host.lib()
    rng = np.random.default_rng(11)
    n = 6000
    x = rng.normal(size=(n, 2)) * 10.0 ** rng.integers(-32, 32, (n, 2))
    fmts = ["%.9g", "%.17g", "%.15g", "%.6f", "%.3e", "%.12f", "%.1f"]
    txt = [[fmts[(i + j) % len(fmts)] % x[i, j] for j in range(2)] for i in range(n)]
    # midpoints of adjacent floats, written exactly (the double-rounding trap) and one digit short
    f = np.abs(rng.normal(size=400)).astype(np.float32) + np.float32(0.5)
    mid = (f.astype(np.float64) + np.nextafter(f, np.float32(np.inf)).astype(np.float64)) / 2
    txt += [["%.17g" % m, "%.15g" % m] for m in mid]
    plain = "".join("v 1.0 %s %s\n" % (a, b) for a, b in txt) + "f 1 2 3\n"
    forced = "".join("v +1.0 %s %s\n" % (a, b) for a, b in txt) + "f 1 2 3\n"
    got = []
    for i, text in enumerate((plain, forced)):
        # read the vertices back through n-2 fan triangles? simpler: one triangle per vertex triple
        faces = "".join("f %d %d %d\n" % (k + 1, k + 2, k + 3) for k in range(0, len(txt) - 2, 3))
        pth = tmp_path / ("n%d.obj" % i)
        pth.write_text(text.replace("f 1 2 3\n", faces))
        m = L.Mesh_New()
        assert L.Mesh_Import_OBJ_Path(m, str(pth).encode()), host.last_error()
        k = L.Mesh_Triangle_Count(m)
        a = np.ctypeslib.as_array(C.cast(L.Mesh_Triangles(m), C.POINTER(C.c_uint32)), (k, 3, 8)).copy()
        got.append(a[:, :, :3])  # positions only (default normals of huge triangles overflow alike, but keep it simple)
        L.Mesh_Delete(m)
    assert got[0].shape[0] == len(range(0, len(txt) - 2, 3)) and np.array_equal(got[0], got[1])


def _import_bits(L, importer):
    m = L.Mesh_New()
    ok = importer(m)
    n = L.Mesh_Triangle_Count(m)
    a = np.ctypeslib.as_array(C.cast(L.Mesh_Triangles(m), C.POINTER(C.c_uint32)), (n, 24)).copy() if n else np.zeros((0, 24), np.uint32)
    L.Mesh_Delete(m)
    return bool(ok), a, (host.last_error() if not ok else "")


def _import_file(L, path):
    """The sequential importer on a FILE*, as the reference calls it (mesh.c:87)."""
    libc = C.CDLL(None)
    libc.fopen.restype, libc.fopen.argtypes = C.c_void_p, [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]

    def run(m):
        fd = libc.fopen(str(path).encode(), b"r")
        assert fd
        try:
            return L.Mesh_Import_OBJ(m, fd)
        finally:
            libc.fclose(fd)
    return _import_bits(L, run)


def test_obj_import_threaded_equals_sequential(tmp_path, monkeypatch):
    """Mesh_Import_OBJ_Path / _Memory parse the file in memory with several threads; Mesh_Import_OBJ is the
    reference's sequential fgets loop.  Same triangles in the same order, same success and the same message on
    failure — on the assets and on the loader's corner cases: lines longer than the 512-byte buffer (examined as
    several lines), no newline at the end, a last line exactly filling the buffer, a trailing vertex line (pushed
    twice by `while(!feof)`, mesh.c:100), CRLF, empty files, errors early and late in the file."""
    L = host.lib()
    rng = np.random.default_rng(23)
    nv = 4000
    v = rng.normal(size=(nv, 3)).astype(np.float32)
    body = "".join("v %.9g %.9g %.9g\n" % tuple(x) for x in v) + "".join("vn %.9g %.9g %.9g\n" % tuple(x) for x in v[:500]) + \
        "".join("vt %.9g %.9g\n" % tuple(x[:2]) for x in v[:300])
    faces = ""
    for _ in range(6000):
        a, b, c, d = rng.integers(1, nv + 1, 4)
        k = int(rng.integers(0, 4))
        faces += ["f %d %d %d\n" % (a, b, c), "f %d//%d %d//%d %d//%d\n" % (a, a % 500 + 1, b, b % 500 + 1, c, c % 500 + 1),
                  "f %d/%d/%d %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a % 300 + 1, a % 500 + 1, b, b % 300 + 1, b % 500 + 1, c, c % 300 + 1, c % 500 + 1, d, d % 300 + 1, d % 500 + 1),
                  "f %d/%d %d/%d %d/%d\n" % (a, a % 300 + 1, b, b % 300 + 1, c, c % 300 + 1)][k]
    long_comment = "# " + "x" * 700 + " v 1 2 3\n"                      # its tail is a chunk of its own; does not start with v
    long_vertex = "v 1 2 3" + " " * 503 + "\n"                          # 511 bytes before the newline: chunk, then "\n" alone
    long_split = "#" + " " * 510 + "v 7 8 9\n" + "#" + " " * 510 + "f 1 2 3\n"  # a continuation chunk that IS a vertex / a face
    cases = {
        "plain": body + faces,
        "no_final_newline": body + faces + "f 1 2 3",
        "trailing_vertex": body + faces + "v 9 9 9\n",
        "trailing_vertex_no_newline": body + faces + "v 9 9 9",
        "long_lines": body + long_comment + long_vertex + long_split + faces + long_comment,
        "last_line_fills_buffer": body + faces + "v 1 2 3" + " " * 504,
        "crlf": (body + faces).replace("\n", "\r\n"),
        "empty": "",
        "only_newline": "\n",
        "faces_before_vertices": faces[:2000].rsplit("\n", 1)[0] + "\n" + body,
        "error_late": body + faces + "f 1 2 x\n" + faces[:500].rsplit("\n", 1)[0] + "\n",
        "error_early": body + "f 1 2 3 4 5\n" + faces,
        "error_index": body + faces + "f 1 2 %d\n" % (nv + 5),
    }
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("RTX_HOST_THREADS", threads)
        for name, text in cases.items():
            pth = tmp_path / (name + ".obj")
            pth.write_bytes(text.encode())
            want = _import_file(L, pth)
            got = _import_bits(L, lambda m: L.Mesh_Import_OBJ_Path(m, str(pth).encode()))
            mem = _import_bits(L, lambda m: L.Mesh_Import_OBJ_Memory(m, text.encode(), len(text.encode())))
            for g in (got, mem):
                assert g[0] == want[0] and g[2] == want[2], (name, threads, g[2], want[2])
                assert g[1].shape == want[1].shape and np.array_equal(g[1], want[1]), (name, threads)
            assert want[0] == (not name.startswith("error")) or name == "faces_before_vertices", name
    assert _import_file(L, tmp_path / "plain.obj")[1].shape[0] > 6000
    monkeypatch.setenv("RTX_HOST_THREADS", "5")
    for name in ("teapot", "bunny", "pear"):
        want = _import_file(L, assets.obj_path(name))
        got = _import_bits(L, lambda m: L.Mesh_Import_OBJ_Path(m, assets.obj_path(name).encode()))
        assert want[0] and got[0] and np.array_equal(got[1], want[1])


def test_obj_import_quad_default_normal_is_not_normalised(tmp_path):
    L = host.lib()
    m = L.Mesh_New()
    assert L.Mesh_Import_OBJ_Path(m, _write(tmp_path, "v 0 0 0\nv 2 0 0\nv 2 2 0\nv 0 2 0\nf 1 2 3 4\n"))
    t = L.Mesh_Triangles(m)
    assert t[0].vtx[0].norm.z == 4.0  # cross((2,0,0),(2,2,0)) (mesh.c:179-181)
    L.Mesh_Delete(m)


def test_obj_import_error_paths(tmp_path):
    L = host.lib()
    for bad in ["v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3 1 2\n", "v 0 0 0\nf 1 2 3\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 x\n"]:
        m = L.Mesh_New()
        assert not L.Mesh_Import_OBJ_Path(m, _write(tmp_path, bad))
        assert host.last_error()
        L.Mesh_Delete(m)
    m = L.Mesh_New()
    assert not L.Mesh_Import_OBJ_Path(m, b"/nonexistent/file.obj")
    L.Mesh_Delete(m)


def test_packed_meshes_expand_to_reference_triangle_counts():
    L = host.lib()
    for name, tris in [("teapot", 6320), ("bunny", 69451), ("pear", 2848)]:
        m = L.Mesh_New()
        assert L.Mesh_Import_OBJ_Path(m, assets.obj_path(name).encode())
        assert L.Mesh_Triangle_Count(m) == tris
        L.Mesh_Delete(m)


def test_mesh_world_transform_is_scale_then_translate(tmp_path):
    L = host.lib()
    sky = L.Skybox_New_Color(host.Color(0, 0, 0))
    scene = L.Scene_New(sky)
    tex = L.Texture_New()
    L.Texture_Import_Color(tex, host.Color(1, 1, 1))
    mat = L.Material_Diffuse_Make(tex)
    m = L.Mesh_New()
    assert L.Mesh_Import_OBJ_Path(m, _write(tmp_path, "v 1 2 3\nv 2 2 3\nv 1 3 3\nf 1 2 3\n"))
    L.Mesh_Set_Material(m, C.byref(mat))
    L.Mesh_Set_Scale(m, 2.0)
    L.Mesh_Set_Origin(m, host.point3(10, 20, 30))
    L.Mesh_AddToScene(m, scene)
    assert L.Scene_Prepare(scene)
    flat = L.Scene_Flatten(scene)
    d = L.SceneFlat_Desc(flat).contents
    assert d.n_tris == 1 and [d.tri_pos[c] for c in range(3)] == [12.0, 24.0, 36.0]
    assert [d.tri_nrm[c] for c in range(3)] == [0.0, 0.0, 1.0]  # normals untouched by the transform
    L.SceneFlat_Delete(flat)
    L.Mesh_Delete(m)
    L.Scene_Delete(scene)


def test_scene_without_bounded_objects_cannot_be_flattened():
    L = host.lib()
    sky = L.Skybox_New_Color(host.Color(0, 0, 0))
    scene = L.Scene_New(sky)
    tex = L.Texture_New()
    L.Texture_Import_Color(tex, host.Color(1, 1, 1))
    mat = L.Material_Diffuse_Make(tex)
    o = host.Object()
    o.material = C.pointer(mat)
    o.surface = L.Surface_Plane_Make(host.point3(0, 0, 0), host.vec3(0, 0, 1))
    L.Scene_Add_Object(scene, C.byref(o))
    assert L.Scene_Prepare(scene)
    assert not L.Scene_Flatten(scene)  # the reference dereferences a NULL tree here (scene.c:109)
    L.Scene_Delete(scene)


def test_bounding_boxes_are_padded_by_epsilon():
    L = host.lib()
    s = L.Surface_Sphere_Make(host.point3(1, 2, 3), 2.0)
    b = L.Surface_BoundingBox(C.byref(s))
    eps = np.float32(0.0001)
    assert b.min.x == float(np.float32(1) - (np.float32(2) + eps)) and b.max.z == float(np.float32(3) + (np.float32(2) + eps))
    assert L.Surface_Bounded(C.byref(s))
    p = L.Surface_Plane_Make(host.point3(0, 0, 0), host.vec3(0, 0, 1))
    assert not L.Surface_Bounded(C.byref(p))


def test_bmp_roundtrip_and_texture_linearisation(tmp_path):
    L = host.lib()
    rgb = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)  # width 7 -> 3 bytes of row padding
    path = str(tmp_path / "t.bmp")
    assets.write_bmp(path, rgb)
    img = host.ImageRGB()
    assert L.ImageRGB_Load_BMP_Path(C.byref(img), path.encode())
    assert (img.res.width, img.res.height) == (7, 5)
    px = np.ctypeslib.as_array(C.cast(img.pix, C.POINTER(C.c_uint8)), (5, 7, 3))
    assert np.array_equal(px[:, :, ::-1], rgb)  # RGB{b,g,r}; row 0 = top of the picture (image.c:118)
    out = str(tmp_path / "o.bmp")
    assert L.ImageRGB_Save_BMP_Path(C.byref(img), out.encode())
    L.ImageRGB_Unload(C.byref(img))
    tex = L.Texture_New()
    assert L.Texture_Import_BMP_Path(tex, path.encode())
    im = L.Texture_Image(tex).contents
    lin = np.ctypeslib.as_array(C.cast(im.pix, C.POINTER(C.c_float)), (5, 7, 3))
    s = rgb.astype(np.float32) / np.float32(255)
    want = np.where(s < 0.04045, s / np.float32(12.92), ((s + np.float32(0.055)) / np.float32(1.055)) ** np.float32(2.4))
    assert np.allclose(lin, want, rtol=1e-6, atol=1e-7)
    L.Texture_Delete(tex)


def test_camera_block_layout_matches_rtx_camera(built_scenes):
    cam = built_scenes("teapot").rtx_camera()
    assert list(cam.origin) == [20.0, -20.0, 20.0] and cam.lens_radius == 0.0
    w = np.array(list(cam.w))
    assert abs(np.linalg.norm(w) - 1) < 1e-6


def test_obj_import_memory_rejects_null_arguments():
    L = host.lib()
    m = L.Mesh_New()
    assert not L.Mesh_Import_OBJ_Memory(None, b"v 0 0 0\n", 8) and host.last_error()
    assert not L.Mesh_Import_OBJ_Memory(m, None, 8)
    assert L.Mesh_Import_OBJ_Memory(m, None, 0) and L.Mesh_Triangle_Count(m) == 0  # an empty file is an empty mesh
    L.Mesh_Delete(m)
