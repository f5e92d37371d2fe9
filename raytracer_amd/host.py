"""ctypes binding of librtx_host.so — the host-side mirror of the reference's scene API.

Function names, argument order and struct layouts follow the reference headers cited in
include/rt_host.h (src/world/scene.h, src/gfx/mesh.h, src/rt/materials.h, src/rt/surfaces.h,
src/world/camera.h, src/rt/renderer.h), so host code written against the reference reads the same
here.  Nothing in this module traces a ray.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")


class vec2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z)


point3 = vec3
Color = vec3


class RGB(C.Structure):
    _fields_ = [("b", C.c_uint8), ("g", C.c_uint8), ("r", C.c_uint8)]


class _Res(C.Structure):
    _fields_ = [("width", C.c_size_t), ("height", C.c_size_t)]


class ImageRGB(C.Structure):
    _fields_ = [("res", _Res), ("pix", C.POINTER(RGB))]


class ImageColor(C.Structure):
    _fields_ = [("res", _Res), ("pix", C.POINTER(vec3))]


class Skybox(C.Structure):
    _fields_ = [("tex", C.c_void_p * 6)]


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("albedo", C.c_void_p), ("p", C.c_float * 12)]


class BoundingBox(C.Structure):
    _fields_ = [("min", vec3), ("max", vec3)]


class Sphere(C.Structure):
    _fields_ = [("c", vec3), ("r", C.c_float)]


class Vertex(C.Structure):
    _fields_ = [("pos", vec3), ("norm", vec3), ("tex", vec2)]


class Triangle(C.Structure):
    _fields_ = [("vtx", Vertex * 3)]


class Plane(C.Structure):
    _fields_ = [("normal", vec3), ("point", vec3)]


class _SurfaceU(C.Union):
    _fields_ = [("sphere", Sphere), ("triangle", Triangle), ("plane", Plane)]


class Surface(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("type", C.c_int), ("u", _SurfaceU)]


class Object(C.Structure):
    _fields_ = [("material", C.POINTER(Material)), ("surface", Surface)]


class Camera(C.Structure):
    _fields_ = [("origin", vec3), ("bottomLeftCorner", vec3), ("horizontal", vec3), ("vertical", vec3), ("u", vec3),
                ("v", vec3), ("w", vec3), ("lensRadius", C.c_float)]


class KDTreeStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("nodes", "leaves", "empty_leaves", "refs", "max_leaf_len", "max_depth_bound",
                                          "depth_reached", "fnv1a_nodes", "fnv1a_refs")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class rtx_kd_tree(C.Structure):
    _fields_ = [("node_w0", C.POINTER(C.c_uint32)), ("node_w1", C.POINTER(C.c_uint32)), ("n_nodes", C.c_uint64),
                ("leaf_ref", C.POINTER(C.c_uint32)), ("n_refs", C.c_uint64), ("root_index", C.c_uint64),
                ("depth_reached", C.c_uint64), ("ms_device", C.c_double), ("ms_total", C.c_double)]


class rtx_material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("tex", C.c_int32), ("p", C.c_float * 14)]


class rtx_texture_desc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint64)]


class rtx_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("bottom_left", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float)]


_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


class rtx_scene_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("n_nodes", C.c_uint32), ("n_refs", C.c_uint32), ("root_index", C.c_uint32),
        ("n_tris", C.c_uint32), ("n_spheres", C.c_uint32), ("n_planes", C.c_uint32), ("n_materials", C.c_uint32),
        ("n_textures", C.c_uint32), ("max_depth", C.c_uint32),
        ("node_w0", _u32p), ("node_w1", _u32p), ("leaf_ref", _u32p),
        ("tri_pos", _f32p), ("tri_nrm", _f32p), ("tri_uv", _f32p), ("tri_mat", _u8p), ("tri_obj", _u32p),
        ("sph", _f32p), ("sph_mat", _u8p), ("sph_obj", _u32p),
        ("plane", _f32p), ("plane_mat", _u8p),
        ("materials", C.POINTER(rtx_material)), ("textures", C.POINTER(rtx_texture_desc)), ("tex_data", _f32p),
        ("tex_texels", C.c_uint64), ("skybox_tex", C.c_int32 * 6),
    ]


RTX_MAX_DEPTH = 64


class rtx_stats(C.Structure):
    _fields_ = [
        ("rays_total", C.c_uint64), ("rays_at_depth", C.c_uint64 * RTX_MAX_DEPTH), ("camera_paths", C.c_uint64),
        ("visits_inner", C.c_uint64), ("visits_leaf", C.c_uint64), ("visits_prim", C.c_uint64),
        ("hits_shaded", C.c_uint64), ("ms_total", C.c_double), ("ms_trace", C.c_double), ("ms_shade", C.c_double),
        ("ms_raygen", C.c_double), ("ms_resolve", C.c_double), ("trace_launches", C.c_uint32), ("passes", C.c_uint32),
        ("ms_trace_at_depth", C.c_double * RTX_MAX_DEPTH),
        ("visits_culled", C.c_uint64), ("ms_sort", C.c_double), ("wave_stats", C.c_uint64 * 8),
        ("samples_done", C.c_uint32), ("pad_", C.c_uint32), ("rays_early_miss", C.c_uint64),
    ]


class rtx_render_params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
        ("seed_lo", C.c_uint64), ("seed_hi", C.c_uint64), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
        ("band_rows", C.c_uint32), ("band_rank", C.c_uint32), ("band_world", C.c_uint32), ("flags", C.c_uint32),
        ("max_paths_in_flight", C.c_uint64),
    ]


class RenderCtx(C.Structure):
    _fields_ = [("cam", C.POINTER(Camera)), ("scene", C.c_void_p), ("img", C.POINTER(ImageRGB)),
                ("waiter_thread", C.c_void_p), ("waiter_args", C.c_void_p), ("finished", C.c_bool)]


_lib = None


def lib() -> C.CDLL:
    """Load librtx_host.so (built in-tree by __graft_entry__.build / raytracer_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(LIB_DIR, "librtx_host.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -m raytracer_amd.build`")
    L = C.CDLL(path)
    vp = C.c_void_p

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("RtHost_LastError", C.c_char_p)
    sig("ImageRGB_Load_Empty", C.c_bool, C.POINTER(ImageRGB), C.c_size_t, C.c_size_t)
    sig("ImageRGB_Load_BMP_Path", C.c_bool, C.POINTER(ImageRGB), C.c_char_p)
    sig("ImageRGB_Save_BMP_Path", C.c_bool, C.POINTER(ImageRGB), C.c_char_p)
    sig("ImageRGB_Unload", None, C.POINTER(ImageRGB))
    sig("Texture_New", vp)
    sig("Texture_Import_BMP_Path", C.c_bool, vp, C.c_char_p)
    sig("Texture_Import_Color", C.c_bool, vp, Color)
    sig("Texture_Import_Pixels", C.c_bool, vp, C.c_size_t, C.c_size_t, _f32p)
    sig("Texture_Delete", None, vp)
    sig("Texture_Image", C.POINTER(ImageColor), vp)
    sig("Skybox_Import_BMP", C.POINTER(Skybox), C.c_char_p)
    sig("Skybox_New_Color", C.POINTER(Skybox), Color)
    sig("Skybox_Delete", None, C.POINTER(Skybox))
    sig("Material_Diffuse_Make", Material, vp)
    sig("Material_Metal_Make", Material, vp, C.c_float)
    sig("Material_Dielectric_Make", Material, vp, C.c_float)
    sig("Material_DiffuseLight_Make", Material, vp, C.c_float)
    sig("Material_Skybox_Make", Material, C.POINTER(Skybox))
    sig("Material_Disney_Diffuse_Make", Material, vp, C.c_float, C.c_float)
    sig("Material_Disney_Metal_Make", Material, vp, C.c_float, C.c_float)
    sig("Material_Disney_Clearcoat_Make", Material, C.c_float)
    sig("Material_Disney_Glass_Make", Material, vp, C.c_float, C.c_float, C.c_float)
    sig("Material_Disney_Sheen_Make", Material, vp, C.c_float)
    sig("Material_Disney_BSDF_Make", Material, vp, *([C.c_float] * 12))
    sig("Surface_Sphere_Make", Surface, point3, C.c_float)
    sig("Surface_Triangle_Make", Surface, Vertex, Vertex, Vertex)
    sig("Surface_Plane_Make", Surface, point3, vec3)
    sig("Triangle_MakeSimple", Triangle, point3, point3, point3)
    sig("Surface_BoundingBox", BoundingBox, C.POINTER(Surface))
    sig("Surface_Bounded", C.c_bool, C.POINTER(Surface))
    sig("KDTree_New", vp, C.POINTER(Object), C.c_size_t)
    sig("KDTree_Delete", None, vp)
    sig("KDTree_Stats", None, vp, C.POINTER(KDTreeStats))
    sig("Scene_New", vp, C.POINTER(Skybox))
    sig("Scene_Delete", None, vp)
    sig("Scene_Add_Object", C.c_bool, vp, C.POINTER(Object))
    sig("Scene_Prepare", C.c_bool, vp)
    sig("Scene_Prepare_External", C.c_bool, vp, vp, vp, C.c_int, C.POINTER(C.c_double))
    sig("KDTree_New_External", vp, C.POINTER(Object), C.c_size_t, vp, vp, C.c_int, C.POINTER(C.c_double))
    sig("Scene_Object_Count", C.c_size_t, vp)
    sig("Scene_KDTree", vp, vp)
    sig("Mesh_New", vp)
    sig("Mesh_Delete", None, vp)
    sig("Mesh_Import_OBJ_Path", C.c_bool, vp, C.c_char_p)
    sig("Mesh_Import_OBJ", C.c_bool, vp, vp)
    sig("Mesh_Import_OBJ_Memory", C.c_bool, vp, C.c_char_p, C.c_size_t)
    sig("Mesh_AddToScene", None, vp, vp)
    sig("Mesh_Set_Material", None, vp, C.POINTER(Material))
    sig("Mesh_Set_Origin", None, vp, point3)
    sig("Mesh_Set_Scale", None, vp, C.c_float)
    sig("Mesh_Triangle_Count", C.c_size_t, vp)
    sig("Mesh_Triangles", C.POINTER(Triangle), vp)
    sig("Camera_New", C.POINTER(Camera), point3, point3, vec3, C.c_float, C.c_float, C.c_float, C.c_float)
    sig("Camera_Delete", None, C.POINTER(Camera))
    sig("Scene_Flatten", vp, vp)
    sig("SceneFlat_Desc", C.POINTER(rtx_scene_desc), vp)
    sig("SceneFlat_Delete", None, vp)
    sig("SceneFlat_MaterialIndex", C.c_int, vp, C.POINTER(Material))
    sig("Render_New", C.POINTER(RenderCtx), vp, C.POINTER(ImageRGB), C.POINTER(Camera))
    sig("Render_Delete", None, C.POINTER(RenderCtx))
    sig("Render_Start", None, C.POINTER(RenderCtx), C.c_size_t, C.c_size_t)
    sig("Render_Done", C.c_bool, C.POINTER(RenderCtx))
    sig("Render_Set_Seed", None, C.POINTER(RenderCtx), C.c_uint64, C.c_uint64)
    sig("Render_Set_Progressive", None, vp, C.c_uint32)
    sig("Render_Stop", None, vp)
    sig("Render_Samples_Done", C.c_uint32, vp)
    sig("Render_Set_Device", None, C.POINTER(RenderCtx), C.c_int)
    sig("Render_Set_Devices", C.c_bool, C.POINTER(RenderCtx), C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_uint32)
    sig("Render_Status", C.c_int, C.POINTER(RenderCtx), C.POINTER(rtx_stats))
    _lib = L
    return L


def last_error() -> str:
    return lib().RtHost_LastError().decode(errors="replace")
