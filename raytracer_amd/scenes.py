"""Scene recipes: the configs of BASELINE.json as text, and the builder that turns a recipe into a
host `Scene` through the reference-named API of librtx_host.so.

A recipe is a list of directive lines.  The same text is read by oracle/ref_driver.c (which builds
the scene with the REAL reference API) and by `build()` below (which builds it with ours), so both
sides construct the scene by the same sequence of calls, in the order main.c's FillScene does
(/root/reference/src/main.c:60-261: mesh, then light sphere, then ground plane).

directives
    aspect <w/h>
    skybox color r g b | skybox bmp <dir relative to base>
    camera fx fy fz  tx ty tz  ux uy uz  vfov aperture focus(<=0: |from-to|)
    texture <name> color r g b | texture <name> bmp <file relative to base>
    material <name> diffuse <tex> | metal <tex> fuzz | dielectric <tex> ior | light <tex> brightness
                  | skyboxmat | disney_diffuse <tex> roughness subsurface | disney_metal <tex> roughness anistropic
                  | disney_clearcoat gloss | disney_glass <tex> roughness anistropic eta | disney_sheen <tex> tint
                  | disney_bsdf <tex> subsurface specular roughness specular_tint anistropic sheen_tint
                                clearcoat_gloss eta w_sheen w_clearcoat w_specular w_metallic
    mesh <obj relative to base> <material> scale ox oy oz
    sphere cx cy cz r <material>
    plane px py pz nx ny nz <material>
    triangle 9 floats <material>
"""
import ctypes as C
import math
import os

from . import assets, host

# the reference's camera and fixtures (src/main.c:332-338, :230-239, :252-257)
CAMERA = "camera 20 -20 20  0 0 6  0 0 1  40 0 0"
# the same view through a lens: aperture 2 (lensRadius 1) focused in front of the model, so that Camera_GetRay's offset
# branch (camera.c:36-46) and a focus distance other than |from - to| are on the path — no BASELINE config uses them
CAMERA_LENS = "camera 20 -20 20  0 0 6  0 0 1  40 2.0 25"
LIGHT = ["texture white color 1 1 1", "material light light white 5", "sphere 20 20 10 8 light"]


def _rel(path: str) -> str:
    return os.path.relpath(path, assets.cache_dir())


def recipe(name: str) -> list:
    """Named configs.  Mesh paths are relative to assets.cache_dir()."""
    if name in ("teapot", "teapotdof"):  # config 1: teapot, scale 2, origin (0,0,4); "teapotdof": through the lens
        return ["aspect 1.7777778", "skybox color 0.5 0.7 1.0", CAMERA if name == "teapot" else CAMERA_LENS, "texture yellow color 0.8 0.6 0.2",
                "material body diffuse yellow", "mesh %s body 2 0 0 4" % _rel(assets.obj_path("teapot")), *LIGHT,
                "texture grey color 0.5 0.5 0.5", "material ground diffuse grey", "plane 0 0 0 0 0 1 ground"]
    if name == "bunny":  # config 2: bunny Lambert+metal, scale 80, sitting on z=0
        return ["aspect 1.7777778", "skybox color 0.5 0.7 1.0", CAMERA, "texture gold color 0.8 0.6 0.2",
                "material body metal gold 0.1", "mesh %s body 80 1.35 -8.8 4.95" % _rel(assets.obj_path("bunny")), *LIGHT,
                "texture grey color 0.5 0.5 0.5", "material ground diffuse grey", "plane 0 0 0 0 0 1 ground"]
    if name.startswith("dragon"):  # configs 3/5: dragon stand-in, Disney glass+clearcoat mixture
        level = int(name[6:] or 1)
        return ["aspect 1.7777778", "skybox color 0.5 0.7 1.0", CAMERA, "texture yellow color 0.8 0.6 0.2",
                "material body disney_bsdf yellow 0 1 0.2 0 0 0 1 1.52 0 1 1 0",
                "mesh %s body 100 1.68 -11 6.2" % _rel(assets.dragon_standin(level)), *LIGHT,
                "material ground skyboxmat", "plane 0 0 0 0 0 1 ground"]
    if name == "pear":  # config 4: textured pear, Disney glass, cubemap skybox
        return ["aspect 1.7777778", "skybox bmp %s" % _rel(assets.skybox_dir()), CAMERA,
                "texture peartex bmp %s" % _rel(assets.pear_texture()),
                "material body disney_glass peartex 0 0 1.52", "mesh %s body 5 0 0 6" % _rel(assets.obj_path("pear")),
                *LIGHT, "material ground skyboxmat", "plane 0 0 0 0 0 1 ground"]
    if name == "materials":  # every material kind once (spheres of main.c:114-182) + a few triangles
        lines = ["aspect 1.7777778", "skybox bmp %s" % _rel(assets.skybox_dir()), CAMERA,
                 "texture white color 1 1 1", "texture red color 0.9 0.2 0.15", "texture checker bmp %s" % _rel(
                     assets.pear_texture()),
                 "material m0 diffuse red", "material m1 metal white 0.3", "material m2 dielectric white 1.5",
                 "material m3 light white 5", "material m4 skyboxmat", "material m5 disney_diffuse checker 0.4 1.0",
                 "material m6 disney_metal red 0.25 0.3", "material m7 disney_glass white 0.2 0.0 1.54",
                 "material m8 disney_clearcoat 0.3", "material m9 disney_sheen red 1.0",
                 "material m10 disney_bsdf white 0.2 1.0 0.3 1.0 0.2 0.5 1.0 1.52 0.3 0.5 0.5 0.5"]
        for i in range(11):
            if i in (3, 4):
                continue
            lines.append("sphere %g %g 6 4.5 m%d" % (-30 + 10 * (i % 6), 10 if i < 6 else -4, i))
        # two loose triangles through the OBJ path (textured Disney diffuse) + one small one built
        # with Triangle_MakeSimple (un-normalised normal, Lambert: harmless there)
        lines += ["sphere 20 20 10 8 m3", "mesh %s m5 1 0 0 0" % _rel(assets.tiny_mesh()),
                  "triangle 14 -12 0.5  18 -12 0.5  16 -10 4 m0", "plane 0 0 0 0 0 1 m4"]
        return lines
    raise KeyError(name)


def write_recipe(name: str, lines=None) -> str:
    path = os.path.join(assets.cache_dir(), name + ".scene")
    with open(path, "w") as fd:
        fd.write("\n".join(lines if lines is not None else recipe(name)) + "\n")
    return path


class BuiltScene:
    """A host Scene built from a recipe; keeps every ctypes object the C side points at alive."""

    def __init__(self, lines, base_dir=None, kd_device=None):
        """kd_device: None = the host SAH build (KDTree_New); an ordinal = the device build of librt_hip.so
        (rtx_kd_build) through Scene_Prepare_External — the same tree, word for word."""
        L = host.lib()
        base = base_dir or assets.cache_dir()
        self.L = L
        self.textures, self.materials, self.material_order = {}, {}, []
        self.skybox = None
        self.camera = None
        self.aspect = 16.0 / 9.0
        self._flat = None
        for ln in lines:
            t = ln.split()
            if t and t[0] == "aspect":
                self.aspect = float(t[1])
            elif t[:2] == ["skybox", "color"]:
                self.skybox = L.Skybox_New_Color(host.Color(*map(float, t[2:5])))
            elif t[:2] == ["skybox", "bmp"]:
                self.skybox = L.Skybox_Import_BMP(os.path.join(base, t[2]).encode())
                if not self.skybox:
                    raise RuntimeError(host.last_error())
        if not self.skybox:
            self.skybox = L.Skybox_New_Color(host.Color(0.5, 0.7, 1.0))
        self.scene = L.Scene_New(self.skybox)
        f = float
        for ln in lines:
            t = ln.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "camera":
                v = list(map(f, t[1:13]))
                frm, to, up = host.vec3(*v[0:3]), host.vec3(*v[3:6]), host.vec3(*v[6:9])
                focus = v[11]
                if focus <= 0:
                    # vmag(vsub(from,to)) in binary32 (src/main.c:335)
                    import numpy as np
                    d = np.float32(v[0:3]) - np.float32(v[3:6])
                    focus = float(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
                self.camera = L.Camera_New(frm, to, up, self.aspect, v[9], v[10], focus)
            elif t[0] == "texture":
                tex = L.Texture_New()
                if t[2] == "color":
                    ok = L.Texture_Import_Color(tex, host.Color(*map(f, t[3:6])))
                else:
                    ok = L.Texture_Import_BMP_Path(tex, os.path.join(base, t[3]).encode())
                if not ok:
                    raise RuntimeError(host.last_error())
                self.textures[t[1]] = tex
            elif t[0] == "material":
                name, kind, a = t[1], t[2], t[3:]
                T = self.textures
                if kind == "diffuse":
                    m = L.Material_Diffuse_Make(T[a[0]])
                elif kind == "metal":
                    m = L.Material_Metal_Make(T[a[0]], f(a[1]))
                elif kind == "dielectric":
                    m = L.Material_Dielectric_Make(T[a[0]], f(a[1]))
                elif kind == "light":
                    m = L.Material_DiffuseLight_Make(T[a[0]], f(a[1]))
                elif kind == "skyboxmat":
                    m = L.Material_Skybox_Make(self.skybox)
                elif kind == "disney_diffuse":
                    m = L.Material_Disney_Diffuse_Make(T[a[0]], f(a[1]), f(a[2]))
                elif kind == "disney_metal":
                    m = L.Material_Disney_Metal_Make(T[a[0]], f(a[1]), f(a[2]))
                elif kind == "disney_clearcoat":
                    m = L.Material_Disney_Clearcoat_Make(f(a[0]))
                elif kind == "disney_glass":
                    m = L.Material_Disney_Glass_Make(T[a[0]], f(a[1]), f(a[2]), f(a[3]))
                elif kind == "disney_sheen":
                    m = L.Material_Disney_Sheen_Make(T[a[0]], f(a[1]))
                elif kind == "disney_bsdf":
                    m = L.Material_Disney_BSDF_Make(T[a[0]], *map(f, a[1:13]))
                else:
                    raise ValueError(ln)
                self.materials[name] = m  # ctypes struct, address stable while referenced here
                self.material_order.append(name)
            elif t[0] == "mesh":
                mesh = L.Mesh_New()
                if not L.Mesh_Import_OBJ_Path(mesh, os.path.join(base, t[1]).encode()):
                    raise RuntimeError(host.last_error())
                L.Mesh_Set_Material(mesh, C.byref(self.materials[t[2]]))
                L.Mesh_Set_Origin(mesh, host.point3(f(t[4]), f(t[5]), f(t[6])))
                L.Mesh_Set_Scale(mesh, f(t[3]))
                L.Mesh_AddToScene(mesh, self.scene)
                L.Mesh_Delete(mesh)
            elif t[0] == "sphere":
                o = host.Object()
                o.material = C.pointer(self.materials[t[5]])
                o.surface = L.Surface_Sphere_Make(host.point3(f(t[1]), f(t[2]), f(t[3])), f(t[4]))
                L.Scene_Add_Object(self.scene, C.byref(o))
            elif t[0] == "plane":
                o = host.Object()
                o.material = C.pointer(self.materials[t[7]])
                o.surface = L.Surface_Plane_Make(host.point3(f(t[1]), f(t[2]), f(t[3])),
                                                 host.vec3(f(t[4]), f(t[5]), f(t[6])))
                L.Scene_Add_Object(self.scene, C.byref(o))
            elif t[0] == "triangle":
                o = host.Object()
                o.material = C.pointer(self.materials[t[10]])
                o.surface.type = 1
                o.surface.triangle = L.Triangle_MakeSimple(host.point3(*map(f, t[1:4])), host.point3(*map(f, t[4:7])),
                                                           host.point3(*map(f, t[7:10])))
                L.Scene_Add_Object(self.scene, C.byref(o))
        if not self.camera:
            raise ValueError("recipe has no camera")
        self.kd_ms_device = None
        if kd_device is None:
            ok = L.Scene_Prepare(self.scene)
        else:
            from . import hip  # fails loudly when librt_hip.so is missing
            H = hip.lib()
            ms = C.c_double(0.0)
            ok = L.Scene_Prepare_External(self.scene, C.cast(H.rtx_kd_build, C.c_void_p), C.cast(H.rtx_kd_free, C.c_void_p),
                                          int(kd_device), C.byref(ms))
            self.kd_ms_device = ms.value
            if not ok:
                raise RuntimeError("%s / %s" % (host.last_error(), hip.last_error()))
        if not ok:
            raise RuntimeError(host.last_error())

    # -- flattening ---------------------------------------------------------------------------
    def flatten(self):
        if self._flat is None:
            self._flat = self.L.Scene_Flatten(self.scene)
            if not self._flat:
                raise RuntimeError(host.last_error())
        return self._flat

    @property
    def desc(self):
        return self.L.SceneFlat_Desc(self.flatten())

    def material_index(self, name: str) -> int:
        return self.L.SceneFlat_MaterialIndex(self.flatten(), C.byref(self.materials[name]))

    def rtx_camera(self):
        cam = host.rtx_camera()
        C.memmove(C.byref(cam), self.camera, C.sizeof(cam))
        return cam

    def kd_stats(self) -> dict:
        st = host.KDTreeStats()
        self.L.KDTree_Stats(self.L.Scene_KDTree(self.scene), C.byref(st))
        return st.as_dict()

    def close(self):
        L = self.L
        if self._flat:
            L.SceneFlat_Delete(self._flat)
            self._flat = None
        if self.scene:
            L.Scene_Delete(self.scene)
            self.scene = None
        if self.camera:
            L.Camera_Delete(self.camera)
            self.camera = None


def build(name_or_lines, kd_device=None) -> BuiltScene:
    lines = recipe(name_or_lines) if isinstance(name_or_lines, str) else list(name_or_lines)
    return BuiltScene(lines, kd_device=kd_device)
