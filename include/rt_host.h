/*
 * rt_host.h — C ABI of librtx_host.so: the host side that stays on the CPU.
 *
 * BASELINE.json's north_star keeps the reference's scene construction, OBJ/BMP loading and SAH
 * kd-tree build on the host "with their API".  The reference sources cannot travel with this
 * repository, so this library restates that host API from scratch — same function names, argument
 * meaning and struct layouts as the headers cited on each declaration — and adds the one new step
 * the GPU path needs: Scene_Flatten, which turns the built Scene (kd-tree + Object array +
 * materials + textures) into the rtx_scene_desc pools of rtx.h.
 *
 * Differences from the reference, all on error paths: where the reference calls ABORT()
 * (src/common/macros.h:60-64) or invokes undefined behaviour (face with > 4 vertices,
 * out-of-range OBJ index, zero-extent SAH stride), these functions return false/NULL and leave a
 * message in RtHost_LastError().
 *
 * Nothing here runs on the ray path: intersection, traversal and shading live in librt_hip.so.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- math/vec.h:21-98 (only the members the host API passes around) ---- */
typedef struct { float x, y; } vec2;
typedef struct { float x, y, z; } vec3;
typedef vec2 point2;
typedef vec3 point3;
typedef vec3 Color; /* gfx/color.h:25, linear light */

typedef struct { uint8_t b, g, r; } RGB; /* gfx/color.h:19-23 */

/* ---- gfx/image.h:12-29 ---- */
typedef struct { struct { size_t width, height; } res; RGB* pix; } ImageRGB;
typedef struct { struct { size_t width, height; } res; Color* pix; } ImageColor;

bool ImageRGB_Load_Empty(ImageRGB* img, size_t width, size_t height); /* image.c:67  */
bool ImageRGB_Load_BMP(ImageRGB* img, FILE* fd);                      /* image.c:81  */
bool ImageRGB_Save_BMP(ImageRGB* img, FILE* fd);                      /* image.c:179 */
void ImageRGB_Unload(ImageRGB* img);                                  /* image.c:239 */
bool ImageRGB_Load_BMP_Path(ImageRGB* img, const char* path);         /* fopen + Load_BMP */
bool ImageRGB_Save_BMP_Path(ImageRGB* img, const char* path);

/* ---- gfx/texture.h ---- */
typedef struct Texture Texture; /* opaque: {ImageColor* image}, texture.c:5-7 */
Texture* Texture_New(void);                                /* texture.c:9  */
bool     Texture_Import_BMP(Texture* tex, FILE* fd);       /* texture.c:15: sRGB→linear at load */
bool     Texture_Import_BMP_Path(Texture* tex, const char* path);
bool     Texture_Import_Color(Texture* tex, Color color);  /* texture.c:49: 1x1 texture */
bool     Texture_Import_Pixels(Texture* tex, size_t width, size_t height, const float* rgb); /* new: linear f32 */
void     Texture_Delete(Texture* tex);                     /* texture.c:81 */
const ImageColor* Texture_Image(const Texture* tex);

/* ---- world/skybox.h:7-21 ---- */
typedef struct { Texture* tex[6]; /* xpos,xneg,ypos,yneg,zpos,zneg */ } Skybox;
Skybox* Skybox_Import_BMP(const char* folder_path); /* skybox.c:21 */
Skybox* Skybox_New_Color(Color color);              /* new: six 1x1 faces (constant sky) */
void    Skybox_Delete(Skybox* skybox);              /* skybox.c:12; also frees the struct */

/* ---- rt/materials.h:36-90 ---- */
typedef struct {
    uint32_t type; /* RTX_MAT_* = Material_Type */
    union { Texture* albedo; Skybox* skybox; };
    float p[12]; /* same slots as rtx_material.p */
} Material;

Material Material_Diffuse_Make(Texture* tex);                       /* materials.c:16  */
Material Material_Metal_Make(Texture* tex, float fuzz);             /* materials.c:53  */
Material Material_Dielectric_Make(Texture* tex, float refractiveIndex); /* materials.c:88 */
Material Material_DiffuseLight_Make(Texture* tex, float brightness);    /* materials.c:138 */
Material Material_Skybox_Make(Skybox* skybox);                      /* materials.c:169 */
Material Material_Disney_Diffuse_Make(Texture* albedo, float roughness, float subsurface); /* :205 */
Material Material_Disney_Metal_Make(Texture* albedo, float roughness, float anistropic);   /* :326 */
Material Material_Disney_Clearcoat_Make(float gloss);                                      /* :496 */
Material Material_Disney_Glass_Make(Texture* albedo, float roughness, float anistropic, float eta); /* :639 */
Material Material_Disney_Sheen_Make(Texture* albedo, float sheen_tint);                    /* :783 */
Material Material_Disney_BSDF_Make(Texture* albedo, float subsurface, float specular, float roughness,
                                   float specular_tint, float anistropic, float sheen_tint,
                                   float clearcoat_gloss, float eta, float weight_sheen,
                                   float weight_clearcoat, float weight_specular,
                                   float weight_metallic);                                 /* :830 */

/* ---- rt/surfaces.h:9-52 ---- */
typedef struct { point3 min, max; } BoundingBox;
typedef enum { SURFACE_SPHERE, SURFACE_TRIANGLE, SURFACE_PLANE } SurfaceType;
typedef struct { point3 c; float r; } Sphere;
typedef struct { point3 pos; point3 norm; point2 tex; } Vertex;
typedef struct { Vertex vtx[3]; } Triangle;
typedef struct { vec3 normal; point3 point; } Plane;
typedef struct {
    SurfaceType type;
    union { Sphere sphere; Triangle triangle; Plane plane; };
} Surface;

Surface     Surface_Sphere_Make(point3 center, float radius);        /* surfaces.c:12  */
Surface     Surface_Triangle_Make(Vertex v0, Vertex v1, Vertex v2);  /* surfaces.c:104 */
Surface     Surface_Plane_Make(point3 point, vec3 normal);           /* surfaces.c:181 */
Triangle    Triangle_MakeSimple(point3 v0, point3 v1, point3 v2);    /* surfaces.c:91  */
BoundingBox Surface_BoundingBox(Surface* surface);                   /* object.c:26    */
bool        Surface_Bounded(Surface* surface);                       /* object.c:45    */

/* ---- world/object.h:9-18 ---- */
typedef struct { Material* material; Surface surface; } Object;

/* ---- rt/accelerators/kdtree.h ---- */
typedef struct KDTree KDTree;
KDTree* KDTree_New(Object* objs, size_t len); /* kdtree.c:405: SAH build, same node order */
void    KDTree_Delete(KDTree* tree);          /* kdtree.c:446 */

typedef struct {
    uint64_t nodes, leaves, empty_leaves, refs, max_leaf_len, max_depth_bound, depth_reached;
    uint64_t fnv1a_nodes, fnv1a_refs; /* FNV-1a 64 over the flattened node words / leaf refs */
} KDTreeStats;
void KDTree_Stats(const KDTree* tree, KDTreeStats* out);
/* KDTree_New with the recursion (kdtree.c:244-383) done by an external builder of rtx_kd_build's signature
 * (include/rtx.h: the device build of librt_hip.so); boxes, world box and depth bound are computed here
 * as KDTree_New does (kdtree.c:385-408).  The tree is the same, word for word. */
typedef int  (*KDExternalBuild)(int device, const float* box_min_max, uint64_t n, uint64_t max_depth, const float* world_min_max,
                                rtx_kd_tree* out);
typedef void (*KDExternalFree)(rtx_kd_tree* tree);
KDTree* KDTree_New_External(Object* objs, size_t len, KDExternalBuild build, KDExternalFree release, int device, double* ms_device);

/* ---- world/scene.h ---- */
typedef struct Scene Scene;
Scene* Scene_New(Skybox* skybox);                  /* scene.c:24  */
void   Scene_Delete(Scene* scene);                 /* scene.c:60  */
bool   Scene_Add_Object(Scene* scene, Object* obj);/* scene.c:168 (copies *obj) */
bool   Scene_Prepare(Scene* scene);                /* scene.c:145: bounded/unbounded split + KDTree_New */
bool   Scene_Prepare_External(Scene* scene, KDExternalBuild build, KDExternalFree release, int device, double* ms_device);
size_t Scene_Object_Count(const Scene* scene);
const KDTree* Scene_KDTree(const Scene* scene);

/* ---- gfx/mesh.h ---- */
typedef struct Mesh Mesh;
Mesh* Mesh_New(void);                               /* mesh.c:25 */
void  Mesh_Delete(Mesh* mesh);                      /* mesh.c:46 */
bool  Mesh_Import_OBJ(Mesh* mesh, FILE* fd);        /* mesh.c:87 */
/* The same importer over a file in memory / at a path, parsed by several threads (RTX_HOST_THREADS, default the
 * host's cores up to 16; one thread below 1 MB): same triangles in the same order, same quirks (512-byte fgets
 * chunks, the last line examined twice), same errors as Mesh_Import_OBJ.  SURVEY 8(f) rank 3. */
bool  Mesh_Import_OBJ_Memory(Mesh* mesh, const char* data, size_t size);
bool  Mesh_Import_OBJ_Path(Mesh* mesh, const char* path);
void  Mesh_AddToScene(Mesh* mesh, Scene* scene);    /* mesh.c:67; triangles go straight into the scene's
                                                       compact pool, no 112-byte Object per triangle */
void  Mesh_Set_Material(Mesh* mesh, Material* material);
void  Mesh_Set_Origin(Mesh* mesh, point3 origin);
void  Mesh_Set_Scale(Mesh* mesh, float scale);
size_t Mesh_Triangle_Count(const Mesh* mesh);
const Triangle* Mesh_Triangles(const Mesh* mesh);

/* ---- world/camera.h:6-18; same 88-byte layout as rtx_camera ---- */
typedef struct {
    point3 origin, bottomLeftCorner;
    vec3   horizontal, vertical;
    vec3   u, v, w;
    float  lensRadius;
} Camera;
Camera* Camera_New(point3 lookFrom, point3 lookTo, vec3 vup, float aspectRatio, float vertFov,
                   float aperature, float focusDist); /* camera.c:10 */
void    Camera_Delete(Camera* cam);

/* ---- NEW: the flattener ---- */
typedef struct SceneFlat SceneFlat; /* owns the arrays an rtx_scene_desc points into */
SceneFlat*            Scene_Flatten(Scene* scene);
const rtx_scene_desc* SceneFlat_Desc(const SceneFlat* flat);
void                  SceneFlat_Delete(SceneFlat* flat);
int                   SceneFlat_MaterialIndex(const SceneFlat* flat, const Material* material); /* -1: unused */

/* ---- rt/renderer.h:8-22.  Render_Start flattens ctx->scene, uploads it through librt_hip.so
 *      (loaded on first use; failure to load it is an error, there is no CPU renderer here) and
 *      renders into ctx->img->pix on a waiter thread, then flips ctx->finished. ---- */
typedef struct {
    Camera*   cam;
    Scene*    scene;
    ImageRGB* img;
    void*     waiter_thread;
    void*     waiter_args;
    bool      finished;
} RenderCtx;
RenderCtx* Render_New(Scene* scene, ImageRGB* img, Camera* cam);                       /* renderer.c:23  */
void       Render_Delete(RenderCtx* ctx);                                              /* renderer.c:44  */
void       Render_Start(RenderCtx* ctx, size_t samples_per_pixel, size_t max_ray_depth); /* renderer.c:265 */
bool       Render_Done(RenderCtx* ctx);                                                /* renderer.c:271 */
/* extras: seed / device selection for the next Render_Start (defaults: seed (1,2), device 0),
 * the status of the last finished render (RTX_OK or RTX_E_*), and its stats. */
void Render_Set_Seed(RenderCtx* ctx, uint64_t seed_lo, uint64_t seed_hi);
void Render_Set_Device(RenderCtx* ctx, int device);
/* The next Render_Start shares its frame among n devices the way Render_Waiter shares it among NUM_HYPERTHREADS workers
 * (src/rt/renderer.c:186-237): rtx_render_multi with split_mode RTX_SPLIT_TILE (interleaved bands of band_rows rows,
 * 0 = 8; BASELINE config 3) or RTX_SPLIT_SAMPLE (config 5).  false (and no change) for n = 0 or n > 64. */
bool Render_Set_Devices(RenderCtx* ctx, const int* devices, uint32_t n, uint32_t split_mode, uint32_t band_rows);
/* Progressive frame for the next Render_Start (main.c:436-478 polls and blits ctx->img->pix every
 * 200 ms while the workers fill it in): spp_per_pass > 0 makes the image fill in by passes of that many samples
 * (rtx_render_progressive); a frame that runs to the end is byte-identical to the all-at-once frame.
 * Render_Stop (main.c:47-58, the SIGINT export) ends it after the pass in flight: Render_Done turns true and
 * ctx->img->pix holds the average of Render_Samples_Done samples. */
void     Render_Set_Progressive(RenderCtx* ctx, uint32_t spp_per_pass);
void     Render_Stop(RenderCtx* ctx);
uint32_t Render_Samples_Done(RenderCtx* ctx);
int  Render_Status(RenderCtx* ctx, rtx_stats* stats_or_null);

const char* RtHost_LastError(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
