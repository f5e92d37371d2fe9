// oracle/ref_driver.c — headless driver around the UNMODIFIED reference sources.
//
// TEST INFRASTRUCTURE ONLY.  This file is compiled (by oracle/Makefile) together with the
// reference's own .c files where they lie under /root/reference/src; nothing of the reference is
// copied into this repository and the outputs go to oracle/_ref/ (git-ignored).  It exists to
//   (1) pin the CPU restatement in oracle/oracle.cpp and the host kd-tree builder against the real
//       reference (dumps of trees, primary-ray hits, seeded renders, single-bounce vectors), and
//   (2) time the reference's own multithreaded renderer (Render_Start/Render_Done) as the CPU
//       baseline of bench.py.
// The product (librt_hip.so / librtx_host.so) never links, loads or runs it.
//
// The reference's main.c cannot be built here (GLFW/GLEW headers absent); this driver replaces it.
// Four reference translation units are #included rather than linked so that their file-local
// types and functions are reachable: kdtree.c (struct KDTree/KDNode), scene.c (struct Scene),
// texture.c (struct Texture) and renderer.c (RayColor).
//
// Determinism: the stock renderer seeds each worker from rdrand and claims tiles racily
// (renderer.c:143,123-139).  The `render` and `primary` modes below instead call
// Random_Seed(seed_lo ^ (pixel_index*spp + sample), seed_hi) before every camera sample and then
// run the reference's own Camera_GetRay / RayColor / Scene_ClosestHit, so results are reproducible
// and independent of thread scheduling.

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gfx/color.h"
#include "gfx/image.h"
#include "gfx/mesh.h"
#include "math/math.h"
#include "math/random.h"
#include "math/vec.h"
#include "world/camera.h"
#include "world/object.h"

#include "gfx/texture.c"
#include "rt/accelerators/kdtree.c"
#include "world/scene.c"
#include "rt/renderer.c"

// The reference-side binding of librt_hip.so (flatten_ref.c, its own translation unit, sees only the reference's
// HEADERS and re-declares the four private layouts it has to read).  This is the one place where those re-declarations
// and the real definitions are both visible: keep them equal.
#include "flatten_ref.h"
#define SAME_FIELD(Real, Peek, field) \
    static_assert(offsetof(Real, field) == offsetof(Peek, field) && sizeof(((Real*)0)->field) == sizeof(((Peek*)0)->field), #Real "." #field)
static_assert(sizeof(Scene) == sizeof(RtxPeekScene), "struct Scene (scene.c:16-22)");
SAME_FIELD(Scene, RtxPeekScene, skybox);
SAME_FIELD(Scene, RtxPeekScene, objects);
SAME_FIELD(Scene, RtxPeekScene, unboundObjs);
SAME_FIELD(Scene, RtxPeekScene, kdObjects);
SAME_FIELD(Scene, RtxPeekScene, kdTree);
static_assert(sizeof(KDTree) == sizeof(RtxPeekKDTree), "struct KDTree (kdtree.c:128-133)");
SAME_FIELD(KDTree, RtxPeekKDTree, nodes);
SAME_FIELD(KDTree, RtxPeekKDTree, objPtrs);
SAME_FIELD(KDTree, RtxPeekKDTree, worldBox);
SAME_FIELD(KDTree, RtxPeekKDTree, rootIndex);
static_assert(sizeof(Texture) == sizeof(RtxPeekTexture), "struct Texture (texture.c:5-7)");
SAME_FIELD(Texture, RtxPeekTexture, image);
static_assert(sizeof(Vector(Object)) == sizeof(RtxPeekVector) && sizeof(Vector(KDNode)) == sizeof(RtxPeekVector)
                  && sizeof(Vector(ObjectPtr)) == sizeof(RtxPeekVector), "Vector(T) (vector.h:70-77)");
SAME_FIELD(Vector(Object), RtxPeekVector, at);
SAME_FIELD(Vector(Object), RtxPeekVector, length);
SAME_FIELD(Vector(KDNode), RtxPeekVector, length);
SAME_FIELD(Vector(ObjectPtr), RtxPeekVector, at);
static_assert(sizeof(KDNode) == RTX_PEEK_KDNODE_BYTES, "union KDNode (kdtree.c:71-114)");

#define MAX_NAMED 64

typedef struct {
    char     name[64];
    Texture* tex;
} NamedTex;

typedef struct {
    char     name[64];
    Material mat;
} NamedMat;

static NamedTex g_tex[MAX_NAMED];
static size_t   g_ntex;
static NamedMat g_mat[MAX_NAMED];
static size_t   g_nmat;
static Skybox*  g_sky;
static Camera*  g_cam;
static f32      g_aspect = 16.0f / 9.0f;

static Texture* find_tex(const char* name)
{
    for (size_t i = 0; i < g_ntex; i++)
        if (!strcmp(g_tex[i].name, name)) return g_tex[i].tex;
    ABORT("unknown texture %s", name);
}

static Material* find_mat(const char* name)
{
    for (size_t i = 0; i < g_nmat; i++)
        if (!strcmp(g_mat[i].name, name)) return &g_mat[i].mat;
    ABORT("unknown material %s", name);
}

static Skybox* make_color_skybox(Color c)
{
    Skybox* sky = (Skybox*)calloc(1, sizeof(*sky));
    for (size_t i = 0; i < 6; i++) {
        sky->tex[i] = Texture_New();
        Texture_Import_Color(sky->tex[i], c);
    }
    return sky;
}

// Scene recipe file: one directive per line (see raytracer_amd/scenefile.py, which writes them).
static Scene* load_scene_file(const char* path, const char* base_dir)
{
    FILE* fd = fopen(path, "r");
    if (!fd) ABORT("cannot open scene file %s", path);

    Scene* scene = NULL;
    char   line[1024];

    // first pass: skybox + camera aspect (Scene_New needs the skybox)
    while (fgets(line, sizeof line, fd)) {
        char  a[256];
        float r, g, b;
        if (sscanf(line, "skybox color %f %f %f", &r, &g, &b) == 3) {
            g_sky = make_color_skybox((Color){r, g, b});
        } else if (sscanf(line, "skybox bmp %255s", a) == 1) {
            char full[600];
            snprintf(full, sizeof full, "%s/%s", base_dir, a);
            g_sky = Skybox_Import_BMP(full);
            if (!g_sky) ABORT("cannot load skybox %s", full);
        } else if (sscanf(line, "aspect %f", &r) == 1) {
            g_aspect = r;
        }
    }
    if (!g_sky) g_sky = make_color_skybox((Color){0.5f, 0.7f, 1.0f});
    scene = Scene_New(g_sky);
    fseek(fd, 0, SEEK_SET);

    while (fgets(line, sizeof line, fd)) {
        char  kw[64], n1[256], n2[256], n3[256];
        float f[16];
        if (sscanf(line, "%63s", kw) != 1 || kw[0] == '#') continue;

        if (!strcmp(kw, "camera")) {
            // camera fx fy fz tx ty tz ux uy uz vfov aperture focus(<=0: |from-to|)
            if (sscanf(line, "camera %f %f %f %f %f %f %f %f %f %f %f %f", &f[0], &f[1], &f[2], &f[3], &f[4], &f[5],
                       &f[6], &f[7], &f[8], &f[9], &f[10], &f[11])
                != 12)
                ABORT("bad camera line");
            point3 from  = {f[0], f[1], f[2]};
            point3 to    = {f[3], f[4], f[5]};
            vec3   vup   = {f[6], f[7], f[8]};
            f32    focus = f[11] > 0.0f ? f[11] : vmag(vsub(from, to));
            g_cam        = Camera_New(from, to, vup, g_aspect, f[9], f[10], focus);
        } else if (!strcmp(kw, "texture")) {
            if (sscanf(line, "texture %63s color %f %f %f", n1, &f[0], &f[1], &f[2]) == 4) {
                Texture* t = Texture_New();
                Texture_Import_Color(t, (Color){f[0], f[1], f[2]});
                strcpy(g_tex[g_ntex].name, n1);
                g_tex[g_ntex++].tex = t;
            } else if (sscanf(line, "texture %63s bmp %255s", n1, n2) == 2) {
                char full[600];
                snprintf(full, sizeof full, "%s/%s", base_dir, n2);
                FILE* tf = fopen(full, "rb");
                if (!tf) ABORT("cannot open texture %s", full);
                Texture* t = Texture_New();
                Texture_Import_BMP(t, tf);
                fclose(tf);
                strcpy(g_tex[g_ntex].name, n1);
                g_tex[g_ntex++].tex = t;
            } else
                ABORT("bad texture line: %s", line);
        } else if (!strcmp(kw, "material")) {
            Material m;
            if (sscanf(line, "material %63s diffuse %63s", n1, n2) == 2) {
                m = Material_Diffuse_Make(find_tex(n2));
            } else if (sscanf(line, "material %63s metal %63s %f", n1, n2, &f[0]) == 3) {
                m = Material_Metal_Make(find_tex(n2), f[0]);
            } else if (sscanf(line, "material %63s dielectric %63s %f", n1, n2, &f[0]) == 3) {
                m = Material_Dielectric_Make(find_tex(n2), f[0]);
            } else if (sscanf(line, "material %63s light %63s %f", n1, n2, &f[0]) == 3) {
                m = Material_DiffuseLight_Make(find_tex(n2), f[0]);
            } else if (sscanf(line, "material %63s %63s", n1, n3) == 2 && !strcmp(n3, "skyboxmat")) {
                m = Material_Skybox_Make(g_sky);
            } else if (sscanf(line, "material %63s disney_diffuse %63s %f %f", n1, n2, &f[0], &f[1]) == 4) {
                m = Material_Disney_Diffuse_Make(find_tex(n2), f[0], f[1]);
            } else if (sscanf(line, "material %63s disney_metal %63s %f %f", n1, n2, &f[0], &f[1]) == 4) {
                m = Material_Disney_Metal_Make(find_tex(n2), f[0], f[1]);
            } else if (sscanf(line, "material %63s disney_clearcoat %f", n1, &f[0]) == 2) {
                m = Material_Disney_Clearcoat_Make(f[0]);
            } else if (sscanf(line, "material %63s disney_glass %63s %f %f %f", n1, n2, &f[0], &f[1], &f[2]) == 5) {
                m = Material_Disney_Glass_Make(find_tex(n2), f[0], f[1], f[2]);
            } else if (sscanf(line, "material %63s disney_sheen %63s %f", n1, n2, &f[0]) == 3) {
                m = Material_Disney_Sheen_Make(find_tex(n2), f[0]);
            } else if (
                sscanf(line, "material %63s disney_bsdf %63s %f %f %f %f %f %f %f %f %f %f %f %f", n1, n2, &f[0], &f[1],
                       &f[2], &f[3], &f[4], &f[5], &f[6], &f[7], &f[8], &f[9], &f[10], &f[11])
                == 14) {
                // subsurface specular roughness specular_tint anistropic sheen_tint clearcoat_gloss eta
                // weight_sheen weight_clearcoat weight_specular weight_metallic
                m = Material_Disney_BSDF_Make(find_tex(n2), f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], f[9],
                                              f[10], f[11]);
            } else
                ABORT("bad material line: %s", line);
            strcpy(g_mat[g_nmat].name, n1);
            g_mat[g_nmat++].mat = m;
        } else if (!strcmp(kw, "mesh")) {
            // mesh <obj path> <material> scale ox oy oz
            if (sscanf(line, "mesh %255s %63s %f %f %f %f", n1, n2, &f[0], &f[1], &f[2], &f[3]) != 6)
                ABORT("bad mesh line");
            char full[600];
            snprintf(full, sizeof full, "%s/%s", base_dir, n1);
            FILE* mf = fopen(full, "r");
            if (!mf) ABORT("cannot open mesh %s", full);
            Mesh* mesh = Mesh_New();
            Mesh_Import_OBJ(mesh, mf);
            fclose(mf);
            Mesh_Set_Material(mesh, find_mat(n2));
            Mesh_Set_Origin(mesh, (point3){f[1], f[2], f[3]});
            Mesh_Set_Scale(mesh, f[0]);
            Mesh_AddToScene(mesh, scene);
            Mesh_Delete(mesh);
        } else if (!strcmp(kw, "sphere")) {
            if (sscanf(line, "sphere %f %f %f %f %63s", &f[0], &f[1], &f[2], &f[3], n1) != 5) ABORT("bad sphere line");
            Object o = {
                .material = find_mat(n1),
                .surface  = Surface_Sphere_Make((point3){f[0], f[1], f[2]}, f[3]),
            };
            Scene_Add_Object(scene, &o);
        } else if (!strcmp(kw, "plane")) {
            if (sscanf(line, "plane %f %f %f %f %f %f %63s", &f[0], &f[1], &f[2], &f[3], &f[4], &f[5], n1) != 7)
                ABORT("bad plane line");
            Object o = {
                .material = find_mat(n1),
                .surface  = Surface_Plane_Make((point3){f[0], f[1], f[2]}, (vec3){f[3], f[4], f[5]}),
            };
            Scene_Add_Object(scene, &o);
        } else if (!strcmp(kw, "triangle")) {
            // triangle 9 floats material  (Triangle_MakeSimple)
            if (sscanf(line, "triangle %f %f %f %f %f %f %f %f %f %63s", &f[0], &f[1], &f[2], &f[3], &f[4], &f[5], &f[6],
                       &f[7], &f[8], n1)
                != 10)
                ABORT("bad triangle line");
            Object o;
            o.material         = find_mat(n1);
            o.surface.type     = SURFACE_TRIANGLE;
            o.surface.triangle = Triangle_MakeSimple((point3){f[0], f[1], f[2]}, (point3){f[3], f[4], f[5]},
                                                     (point3){f[6], f[7], f[8]});
            Scene_Add_Object(scene, &o);
        }
    }
    fclose(fd);
    if (!g_cam) ABORT("scene file has no camera");
    Scene_Prepare(scene);
    return scene;
}

static f64 now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void put(FILE* fd, const void* p, size_t bytes)
{
    if (fwrite(p, 1, bytes, fd) != bytes) ABORT("short write");
}

// ---------------------------------------------------------------------------------------------
// kd: dump the built tree in the 8-byte node form of rtx_scene_desc
// file: u32 magic 'KDT1', u32 n_nodes, u32 n_refs, u32 root, u32 maxDepth(unused,0),
//       n_nodes x {u32 w0, u32 w1}, n_refs x u32 (index into kdObjects)
// ---------------------------------------------------------------------------------------------
static void mode_kd(Scene* scene, const char* out)
{
    KDTree* t  = scene->kdTree;
    FILE*   fd = fopen(out, "wb");
    if (!fd) ABORT("cannot write %s", out);
    u32 hdr[5] = {0x3154444b, (u32)t->nodes->length, (u32)t->objPtrs->length, (u32)t->rootIndex, 0};
    put(fd, hdr, sizeof hdr);
    size_t leaves = 0, empty = 0, maxleaf = 0;
    for (size_t i = 0; i < t->nodes->length; i++) {
        KDNode* n = &t->nodes->at[i];
        u32     w[2];
        if (n->type == KD_LEAF) {
            w[0] = (u32)n->leaf.len;
            w[1] = (3u << 30) | (u32)n->leaf.objIndex;
            leaves++;
            if (n->leaf.len == 0) empty++;
            if (n->leaf.len > maxleaf) maxleaf = n->leaf.len;
        } else {
            memcpy(&w[0], &n->inode.split, 4);
            w[1] = ((u32)n->type << 30) | (u32)n->inode.leftIndex;
        }
        put(fd, w, 8);
    }
    for (size_t i = 0; i < t->objPtrs->length; i++) {
        u32 idx = (u32)(t->objPtrs->at[i] - scene->kdObjects->at);
        put(fd, &idx, 4);
    }
    fclose(fd);
    printf("{\"nodes\": %zu, \"leaves\": %zu, \"empty_leaves\": %zu, \"refs\": %zu, \"max_leaf_len\": %zu, "
           "\"kd_objects\": %zu, \"unbounded\": %zu}\n",
           t->nodes->length, leaves, empty, t->objPtrs->length, maxleaf, scene->kdObjects->length,
           scene->unboundObjs->length);
}

static u32 object_id(Scene* scene, Object* obj)
{
    if (obj >= scene->kdObjects->at && obj < scene->kdObjects->at + scene->kdObjects->length)
        return (u32)(obj - scene->kdObjects->at);
    return 0x80000000u | (u32)(obj - scene->unboundObjs->at);
}

// One record per ray: origin[3] dir[3] | id t pos[3] n[3] uv[2] front   (raw bits, 17 dwords)
typedef struct {
    f32 o[3], d[3];
    u32 id;
    f32 t, pos[3], n[3], uv[2];
    u32 front;
} HitRec;

static void trace_one(Scene* scene, Ray* ray, HitRec* r)
{
    Object* obj = NULL;
    HitInfo hit;
    memset(&hit, 0, sizeof hit);
    r->o[0] = ray->origin.x, r->o[1] = ray->origin.y, r->o[2] = ray->origin.z;
    r->d[0] = ray->dir.x, r->d[1] = ray->dir.y, r->d[2] = ray->dir.z;
    if (Scene_ClosestHit(scene, ray, &obj, &hit)) {
        r->id     = object_id(scene, obj);
        r->t      = hit.tIntersect;
        r->pos[0] = hit.position.x, r->pos[1] = hit.position.y, r->pos[2] = hit.position.z;
        r->n[0] = hit.unitNormal.x, r->n[1] = hit.unitNormal.y, r->n[2] = hit.unitNormal.z;
        if (obj->surface.type == SURFACE_PLANE) { // Plane_HitAt leaves uv uninitialised (surfaces.c:223-241)
            r->uv[0] = r->uv[1] = 0.0f;
        } else {
            r->uv[0] = hit.uv.x, r->uv[1] = hit.uv.y;
        }
        r->front = hit.frontFace ? 1 : 0;
    } else {
        memset(&r->id, 0, sizeof(HitRec) - offsetof(HitRec, id));
        r->id = 0xFFFFFFFFu;
    }
}

// ---------------------------------------------------------------------------------------------
// primary: camera rays for every pixel (spp samples each), hits dumped as HitRec
//   jitter = 0: pixel centres u=(x+0.5)/(W-1) (ray set A); the RNG is still seeded and the two
//               lens draws of Camera_GetRay still happen
//   jitter = 1: RenderPixel's own jitter (renderer.c:101-102)
//   step: only every step-th pixel in x and y (keeps fixtures small)
// ---------------------------------------------------------------------------------------------
static void mode_primary(Scene* scene, size_t W, size_t H, size_t spp, int jitter, u64 seed_lo, u64 seed_hi, size_t step,
                         const char* out)
{
    FILE* fd = fopen(out, "wb");
    if (!fd) ABORT("cannot write %s", out);
    u64 hits = 0, idsum = 0, n = 0;
    for (size_t yy = 0; yy < H; yy += step) {
        for (size_t xx = 0; xx < W; xx += step) {
            for (size_t s = 0; s < spp; s++) {
                u64 pixel = (u64)yy * W + xx;
                Random_Seed(seed_lo ^ (pixel * spp + s), seed_hi);
                f32 hf, vf;
                if (jitter) {
                    hf = (xx + Random_Unilateral()) / (f32)(W - 1);
                    vf = (yy + Random_Unilateral()) / (f32)(H - 1);
                } else {
                    hf = (xx + 0.5f) / (f32)(W - 1);
                    vf = (yy + 0.5f) / (f32)(H - 1);
                }
                Ray    ray = Camera_GetRay(g_cam, hf, vf);
                HitRec r;
                trace_one(scene, &ray, &r);
                put(fd, &r, sizeof r);
                n++;
                if (r.id != 0xFFFFFFFFu) {
                    hits++;
                    idsum += r.id;
                }
            }
        }
    }
    fclose(fd);
    printf("{\"rays\": %llu, \"hits\": %llu, \"id_sum\": %llu}\n", (unsigned long long)n, (unsigned long long)hits,
           (unsigned long long)idsum);
}

// trace: arbitrary rays (n x 6 f32: origin, direction — Ray_Make is applied) → HitRec
static void mode_trace(Scene* scene, const char* in, const char* out)
{
    FILE* fi = fopen(in, "rb");
    FILE* fo = fopen(out, "wb");
    if (!fi || !fo) ABORT("cannot open %s / %s", in, out);
    f32 v[6];
    u64 n = 0, hits = 0;
    while (fread(v, sizeof v, 1, fi) == 1) {
        Ray    ray = Ray_Make((point3){v[0], v[1], v[2]}, (vec3){v[3], v[4], v[5]});
        HitRec r;
        trace_one(scene, &ray, &r);
        put(fo, &r, sizeof r);
        n++;
        hits += r.id != 0xFFFFFFFFu;
    }
    fclose(fi);
    fclose(fo);
    printf("{\"rays\": %llu, \"hits\": %llu}\n", (unsigned long long)n, (unsigned long long)hits);
}

// ---------------------------------------------------------------------------------------------
// render: seeded, deterministic RenderPixel over the frame (row-parallel; the RNG state is
// thread-local and re-seeded per camera sample so the result does not depend on the thread count)
// file: u32 W, u32 H, u32 spp, u32 depth, W*H x f32[3] cum_color (sum, r,g,b), W*H x u8[3] (b,g,r)
// ---------------------------------------------------------------------------------------------
typedef struct {
    Scene* scene;
    size_t W, H, spp, depth, y0, y1;
    u64    seed_lo, seed_hi;
    f32*   cum;
    u8*    bgr;
    u64    rays;
} RenderJob;

static void* render_rows(void* arg)
{
    RenderJob* j = (RenderJob*)arg;
    for (size_t yy = j->y0; yy < j->y1; yy++) {
        for (size_t xx = 0; xx < j->W; xx++) {
            Color cum = vec(0, 0, 0);
            for (size_t s = 0; s < j->spp; s++) {
                u64 pixel = (u64)yy * j->W + xx;
                Random_Seed(j->seed_lo ^ (pixel * j->spp + s), j->seed_hi);
                f32   hf  = (xx + Random_Unilateral()) / (f32)(j->W - 1);
                f32   vf  = (yy + Random_Unilateral()) / (f32)(j->H - 1);
                Ray   ray = Camera_GetRay(g_cam, hf, vf);
                Color c   = RayColor(j->scene, &ray, j->depth);
                cum       = vadd(cum, c);
            }
            Color  avg          = vdiv(cum, j->spp);
            RGB    px           = RGB_FromColor(avg);
            size_t i            = yy * j->W + xx;
            j->cum[3 * i + 0]   = cum.r;
            j->cum[3 * i + 1]   = cum.g;
            j->cum[3 * i + 2]   = cum.b;
            j->bgr[3 * i + 0]   = px.b;
            j->bgr[3 * i + 1]   = px.g;
            j->bgr[3 * i + 2]   = px.r;
        }
    }
    return NULL;
}

static void mode_render(Scene* scene, size_t W, size_t H, size_t spp, size_t depth, u64 seed_lo, u64 seed_hi,
                        size_t threads, const char* out)
{
    f32* cum = (f32*)calloc(W * H * 3, sizeof(f32));
    u8*  bgr = (u8*)calloc(W * H * 3, 1);
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t th[64];
    RenderJob jobs[64];
    // interleave rows in chunks so threads finish together
    f64 t0 = now_s();
    size_t rows_per = (H + threads - 1) / threads;
    for (size_t i = 0; i < threads; i++) {
        jobs[i]    = (RenderJob){scene, W, H, spp, depth, MIN(i * rows_per, H), MIN((i + 1) * rows_per, H), seed_lo, seed_hi, cum, bgr, 0};
        pthread_attr_t attr;
        pthread_attr_init(&attr);
        pthread_attr_setstacksize(&attr, 8 * 1024 * 1024);
        pthread_create(&th[i], &attr, render_rows, &jobs[i]);
    }
    for (size_t i = 0; i < threads; i++) pthread_join(th[i], NULL);
    f64 t1 = now_s();
    FILE* fd = fopen(out, "wb");
    if (!fd) ABORT("cannot write %s", out);
    u32 hdr[4] = {(u32)W, (u32)H, (u32)spp, (u32)depth};
    put(fd, hdr, sizeof hdr);
    put(fd, cum, W * H * 3 * sizeof(f32));
    put(fd, bgr, W * H * 3);
    fclose(fd);
    printf("{\"width\": %zu, \"height\": %zu, \"spp\": %zu, \"depth\": %zu, \"seconds\": %.6f, \"threads\": %zu}\n", W, H,
           spp, depth, t1 - t0, threads);
}

// ---------------------------------------------------------------------------------------------
// render_pixels: RenderPixel (renderer.c:88-111) for a LIST of pixels of a W x H x spp frame, with the seeding of
// `render` — Random_Seed(seed_lo ^ (pixel * spp + sample), seed_hi) per camera sample — so that pixel p of this mode is
// pixel p of the whole frame, at any frame size (the full-size goldens of BASELINE configs 2-5: a few thousand
// pixels of a 1080p x 64 spp or 4K x 1024 spp frame take seconds where the frame takes hours).
// in : n x u32 pixel index (y * W + x, y = 0 is the bottom row)
// out: u32 W, H, spp, depth, n; n x u32 pixel; n x f32[3] cum (r, g, b); n x u8[3] (b, g, r)
// ---------------------------------------------------------------------------------------------
typedef struct {
    Scene* scene;
    size_t W, H, spp, depth, i0, i1;
    u64    seed_lo, seed_hi;
    u32*   pix;
    f32*   cum;
    u8*    bgr;
} PixelJob;

static void* render_pixel_list(void* arg)
{
    PixelJob* j = (PixelJob*)arg;
    for (size_t i = j->i0; i < j->i1; i++) {
        u64    pixel = j->pix[i];
        size_t xx = pixel % j->W, yy = pixel / j->W;
        Color  cum = vec(0, 0, 0);
        for (size_t s = 0; s < j->spp; s++) {
            Random_Seed(j->seed_lo ^ (pixel * j->spp + s), j->seed_hi);
            f32   hf  = (xx + Random_Unilateral()) / (f32)(j->W - 1);
            f32   vf  = (yy + Random_Unilateral()) / (f32)(j->H - 1);
            Ray   ray = Camera_GetRay(g_cam, hf, vf);
            Color c   = RayColor(j->scene, &ray, j->depth);
            cum       = vadd(cum, c);
        }
        Color avg         = vdiv(cum, j->spp);
        RGB   px          = RGB_FromColor(avg);
        j->cum[3 * i + 0] = cum.r, j->cum[3 * i + 1] = cum.g, j->cum[3 * i + 2] = cum.b;
        j->bgr[3 * i + 0] = px.b, j->bgr[3 * i + 1] = px.g, j->bgr[3 * i + 2] = px.r;
    }
    return NULL;
}

static void mode_render_pixels(Scene* scene, size_t W, size_t H, size_t spp, size_t depth, u64 seed_lo, u64 seed_hi,
                               size_t threads, const char* in, const char* out)
{
    FILE* fi = fopen(in, "rb");
    if (!fi) ABORT("cannot open %s", in);
    fseek(fi, 0, SEEK_END);
    size_t n = (size_t)ftell(fi) / 4;
    fseek(fi, 0, SEEK_SET);
    u32* pix = (u32*)calloc(n ? n : 1, 4);
    if (fread(pix, 4, n, fi) != n) ABORT("short read");
    fclose(fi);
    for (size_t i = 0; i < n; i++)
        if (pix[i] >= W * H) ABORT("pixel %u outside the %zu x %zu frame", pix[i], W, H);
    f32* cum = (f32*)calloc(n * 3 + 1, sizeof(f32));
    u8*  bgr = (u8*)calloc(n * 3 + 1, 1);
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t th[64];
    PixelJob  jobs[64];
    f64       t0  = now_s();
    size_t    per = (n + threads - 1) / threads;
    for (size_t i = 0; i < threads; i++) {
        jobs[i] = (PixelJob){scene, W, H, spp, depth, MIN(i * per, n), MIN((i + 1) * per, n), seed_lo, seed_hi, pix, cum, bgr};
        pthread_attr_t attr;
        pthread_attr_init(&attr);
        pthread_attr_setstacksize(&attr, 8 * 1024 * 1024);
        pthread_create(&th[i], &attr, render_pixel_list, &jobs[i]);
    }
    for (size_t i = 0; i < threads; i++) pthread_join(th[i], NULL);
    f64   t1 = now_s();
    FILE* fd = fopen(out, "wb");
    if (!fd) ABORT("cannot write %s", out);
    u32 hdr[5] = {(u32)W, (u32)H, (u32)spp, (u32)depth, (u32)n};
    put(fd, hdr, sizeof hdr);
    put(fd, pix, n * 4);
    put(fd, cum, n * 3 * sizeof(f32));
    put(fd, bgr, n * 3);
    fclose(fd);
    printf("{\"width\": %zu, \"height\": %zu, \"spp\": %zu, \"depth\": %zu, \"pixels\": %zu, \"seconds\": %.6f, \"threads\": %zu}\n",
           W, H, spp, depth, n, t1 - t0, threads);
}

// ---------------------------------------------------------------------------------------------
// flatten: Rtx_Flatten (flatten_ref.c) over the reference's own Scene; every pool dumped for the byte comparison with
// the product's Scene_Flatten (tests/test_reference_binding.py).
// file: u32 magic 'RFL1', then the desc's 10 counts (n_nodes n_refs root n_tris n_spheres n_planes n_materials n_textures
//       max_depth abi), u64 tex_texels, i32 skybox_tex[6], then the arrays in the order of rtx_scene_desc
// ---------------------------------------------------------------------------------------------
static void mode_flatten(Scene* scene, const char* out)
{
    // the packed-byte decode of flatten_ref.c against the bit-fields of the real union, node by node
    RtxFlat* f = Rtx_Flatten(scene);
    if (!f) ABORT("Rtx_Flatten failed");
    const rtx_scene_desc* d = &f->desc;
    KDTree*               t = scene->kdTree;
    for (size_t i = 0; i < t->nodes->length; i++) {
        KDNode* n  = &t->nodes->at[i];
        u32     w0 = 0, w1;
        if (n->type == KD_LEAF) w0 = (u32)n->leaf.len, w1 = (3u << 30) | (u32)n->leaf.objIndex;
        else memcpy(&w0, &n->inode.split, 4), w1 = ((u32)n->type << 30) | (u32)n->inode.leftIndex;
        if (d->node_w0[i] != w0 || d->node_w1[i] != w1) ABORT("node %zu decoded differently from the bit-fields", i);
    }
    FILE* fd = fopen(out, "wb");
    if (!fd) ABORT("cannot write %s", out);
    u32 hdr[11] = {0x314c4652, d->n_nodes, d->n_refs, d->root_index, d->n_tris, d->n_spheres, d->n_planes, d->n_materials, d->n_textures, d->max_depth, d->abi_version};
    put(fd, hdr, sizeof hdr);
    put(fd, &d->tex_texels, 8);
    put(fd, d->skybox_tex, sizeof d->skybox_tex);
    const size_t T = d->n_tris, S = d->n_spheres, P = d->n_planes;
    put(fd, d->node_w0, 4ull * d->n_nodes), put(fd, d->node_w1, 4ull * d->n_nodes), put(fd, d->leaf_ref, 4ull * d->n_refs);
    put(fd, d->tri_pos, 36 * T), put(fd, d->tri_nrm, 36 * T), put(fd, d->tri_uv, 24 * T), put(fd, d->tri_mat, T), put(fd, d->tri_obj, 4 * T);
    put(fd, d->sph, 16 * S), put(fd, d->sph_mat, S), put(fd, d->sph_obj, 4 * S);
    put(fd, d->plane, 24 * P), put(fd, d->plane_mat, P);
    put(fd, d->materials, sizeof(rtx_material) * d->n_materials), put(fd, d->textures, sizeof(rtx_texture_desc) * d->n_textures);
    put(fd, d->tex_data, 12ull * d->tex_texels);
    fclose(fd);
    printf("{\"nodes\": %u, \"refs\": %u, \"tris\": %u, \"spheres\": %u, \"planes\": %u, \"materials\": %u, \"textures\": %u, \"texels\": %llu}\n",
           d->n_nodes, d->n_refs, d->n_tris, d->n_spheres, d->n_planes, d->n_materials, d->n_textures, (unsigned long long)d->tex_texels);
    Rtx_FlatDelete(f);
}

// render_hip: the reference's own render loop — Render_New, Render_Done polling, Render_Delete, its Scene, Camera and
// ImageRGB — with ONE call changed: Render_Start -> Render_Start_HIP (flatten_ref.c).  Needs librt_hip.so (RTX_HIP_LIB)
// and an MI355X; RTX_SEED_LO / RTX_SEED_HI pin the frame.  file: W*H x u8[3] = ImageRGB.pix
static void mode_render_hip(Scene* scene, size_t W, size_t H, size_t spp, size_t depth, const char* out)
{
    ImageRGB img;
    ImageRGB_Load_Empty(&img, W, H);
    RenderCtx* ctx = Render_New(scene, &img, g_cam);
    f64        t0  = now_s();
    Render_Start_HIP(ctx, spp, depth);
    while (!Render_Done(ctx)) { // main.c:464,477 polls every 200 ms
        struct timespec ts = {0, 2000000};
        nanosleep(&ts, NULL);
    }
    f64 t1 = now_s();
    Thread_Join(ctx->waiter_thread);
    FILE* fd = fopen(out, "wb");
    if (!fd) ABORT("cannot write %s", out);
    put(fd, img.pix, W * H * 3);
    fclose(fd);
    printf("{\"width\": %zu, \"height\": %zu, \"spp\": %zu, \"depth\": %zu, \"seconds\": %.6f}\n", W, H, spp, depth, t1 - t0);
    Render_Delete(ctx);
}

// ---------------------------------------------------------------------------------------------
// bounce: single Material_Bounce calls.
// in : records of {u32 material_index, f32 ray_dir[3], pos[3], normal[3], uv[2], u32 front, u64 seed_lo, seed_hi}
// out: records of {u32 cont, f32 surface[3], emitted[3], out_origin[3], out_dir[3], u32 rng_after[4](next 4 draws' bits)}
// ---------------------------------------------------------------------------------------------
#pragma pack(push, 1)
typedef struct {
    u32 mat;
    f32 dir[3], pos[3], n[3], uv[2];
    u32 front;
    u64 seed_lo, seed_hi;
} BounceIn;

typedef struct {
    u32 cont;
    f32 surf[3], emit[3], oo[3], od[3];
    u32 draws[4];
} BounceOut;
#pragma pack(pop)

static void mode_bounce(const char* in, const char* out)
{
    FILE* fi = fopen(in, "rb");
    FILE* fo = fopen(out, "wb");
    if (!fi || !fo) ABORT("cannot open %s / %s", in, out);
    BounceIn bi;
    u64      n = 0;
    while (fread(&bi, sizeof bi, 1, fi) == 1) {
        if (bi.mat >= g_nmat) ABORT("material index %u out of range", bi.mat);
        Ray ray;
        memset(&ray, 0, sizeof ray);
        ray.dir = (vec3){bi.dir[0], bi.dir[1], bi.dir[2]};
        HitInfo hit;
        hit.position   = (point3){bi.pos[0], bi.pos[1], bi.pos[2]};
        hit.unitNormal = (vec3){bi.n[0], bi.n[1], bi.n[2]};
        hit.uv         = (vec2){bi.uv[0], bi.uv[1]};
        hit.tIntersect = 1.0f;
        hit.frontFace  = bi.front != 0;
        Random_Seed(bi.seed_lo, bi.seed_hi);
        Color surf = COLOR_BLACK, emit = COLOR_BLACK;
        Ray   ro;
        memset(&ro, 0, sizeof ro);
        bool      cont = Material_Bounce(&g_mat[bi.mat].mat, &ray, &hit, &surf, &emit, &ro);
        BounceOut bo;
        memset(&bo, 0, sizeof bo);
        bo.cont    = cont;
        bo.surf[0] = surf.r, bo.surf[1] = surf.g, bo.surf[2] = surf.b;
        bo.emit[0] = emit.r, bo.emit[1] = emit.g, bo.emit[2] = emit.b;
        bo.oo[0] = ro.origin.x, bo.oo[1] = ro.origin.y, bo.oo[2] = ro.origin.z;
        bo.od[0] = ro.dir.x, bo.od[1] = ro.dir.y, bo.od[2] = ro.dir.z;
        for (int k = 0; k < 4; k++) {
            f32 u = Random_Unilateral();
            memcpy(&bo.draws[k], &u, 4);
        }
        put(fo, &bo, sizeof bo);
        n++;
    }
    fclose(fi);
    fclose(fo);
    printf("{\"records\": %llu}\n", (unsigned long long)n);
}

// sky: Skybox_ColorAt for n directions (n x f32[3] in → n x f32[3] out)
static void mode_sky(Scene* scene, const char* in, const char* out)
{
    FILE* fi = fopen(in, "rb");
    FILE* fo = fopen(out, "wb");
    if (!fi || !fo) ABORT("cannot open %s / %s", in, out);
    f32 v[3];
    while (fread(v, sizeof v, 1, fi) == 1) {
        Color c = Scene_Get_SkyColor(scene, (vec3){v[0], v[1], v[2]});
        f32   o[3] = {c.r, c.g, c.b};
        put(fo, o, sizeof o);
    }
    fclose(fi);
    fclose(fo);
}

static void mode_rng(u64 lo, u64 hi, size_t n)
{
    Random_Seed(lo, hi);
    printf("[");
    for (size_t i = 0; i < n; i++) {
        f32 u = Random_Unilateral();
        u32 b;
        memcpy(&b, &u, 4);
        printf("%s%u", i ? ", " : "", b);
    }
    printf("]\n");
}

// bench: the reference's own renderer, untouched (Render_New/Start/Done, NUM_HYPERTHREADS workers,
// rdrand seeding) timed the way main.c:430,466 does.
static void mode_bench(Scene* scene, size_t W, size_t H, size_t spp, size_t depth, const char* bmp_out)
{
    ImageRGB img;
    ImageRGB_Load_Empty(&img, W, H);
    RenderCtx* ctx = Render_New(scene, &img, g_cam);
    f64        t0  = now_s();
    Render_Start(ctx, spp, depth);
    while (!Render_Done(ctx)) {
        struct timespec ts = {0, 2000000};
        nanosleep(&ts, NULL);
    }
    f64 t1 = now_s();
    if (bmp_out && bmp_out[0]) {
        FILE* fd = fopen(bmp_out, "wb+");
        if (fd) {
            ImageRGB_Save_BMP(&img, fd);
            fclose(fd);
        }
    }
    printf("{\"width\": %zu, \"height\": %zu, \"spp\": %zu, \"depth\": %zu, \"seconds\": %.6f, \"threads\": %llu}\n", W, H,
           spp, depth, t1 - t0, (unsigned long long)NUM_HYPERTHREADS);
    Render_Delete(ctx);
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        fprintf(stderr,
                "usage: %s rng lo hi n\n"
                "       %s <scene.scene> <base_dir> kd out.bin\n"
                "       ... primary W H spp jitter seed_lo seed_hi step out.bin\n"
                "       ... trace rays.bin out.bin\n"
                "       ... render W H spp depth seed_lo seed_hi threads out.bin\n"
                "       ... render_pixels W H spp depth seed_lo seed_hi threads pixels.bin out.bin\n"
                "       ... bounce in.bin out.bin\n"
                "       ... sky in.bin out.bin\n"
                "       ... flatten out.bin\n"
                "       ... render_hip W H spp depth out.bin\n"
                "       ... bench W H spp depth [out.bmp]\n",
                argv[0], argv[0]);
        return 2;
    }
    if (!strcmp(argv[1], "rng")) {
        mode_rng(strtoull(argv[2], 0, 0), strtoull(argv[3], 0, 0), strtoull(argv[4], 0, 0));
        return 0;
    }
    if (argc < 4) return 2;
    f64    t0    = now_s();
    Scene* scene = load_scene_file(argv[1], argv[2]);
    f64    t1    = now_s();
    fprintf(stderr, "scene loaded + prepared in %.3f s\n", t1 - t0);
    const char* mode = argv[3];
    char**      a    = argv + 4;
    int         n    = argc - 4;
    if (!strcmp(mode, "kd") && n == 1) {
        mode_kd(scene, a[0]);
    } else if (!strcmp(mode, "primary") && n == 8) {
        mode_primary(scene, strtoull(a[0], 0, 0), strtoull(a[1], 0, 0), strtoull(a[2], 0, 0), atoi(a[3]),
                     strtoull(a[4], 0, 0), strtoull(a[5], 0, 0), strtoull(a[6], 0, 0), a[7]);
    } else if (!strcmp(mode, "trace") && n == 2) {
        mode_trace(scene, a[0], a[1]);
    } else if (!strcmp(mode, "render") && n == 8) {
        mode_render(scene, strtoull(a[0], 0, 0), strtoull(a[1], 0, 0), strtoull(a[2], 0, 0), strtoull(a[3], 0, 0),
                    strtoull(a[4], 0, 0), strtoull(a[5], 0, 0), strtoull(a[6], 0, 0), a[7]);
    } else if (!strcmp(mode, "render_pixels") && n == 9) {
        mode_render_pixels(scene, strtoull(a[0], 0, 0), strtoull(a[1], 0, 0), strtoull(a[2], 0, 0), strtoull(a[3], 0, 0),
                           strtoull(a[4], 0, 0), strtoull(a[5], 0, 0), strtoull(a[6], 0, 0), a[7], a[8]);
    } else if (!strcmp(mode, "bounce") && n == 2) {
        mode_bounce(a[0], a[1]);
    } else if (!strcmp(mode, "sky") && n == 2) {
        mode_sky(scene, a[0], a[1]);
    } else if (!strcmp(mode, "flatten") && n == 1) {
        mode_flatten(scene, a[0]);
    } else if (!strcmp(mode, "render_hip") && n == 5) {
        mode_render_hip(scene, strtoull(a[0], 0, 0), strtoull(a[1], 0, 0), strtoull(a[2], 0, 0), strtoull(a[3], 0, 0), a[4]);
    } else if (!strcmp(mode, "bench") && (n == 4 || n == 5)) {
        mode_bench(scene, strtoull(a[0], 0, 0), strtoull(a[1], 0, 0), strtoull(a[2], 0, 0), strtoull(a[3], 0, 0),
                   n == 5 ? a[4] : "");
    } else {
        fprintf(stderr, "bad mode/arguments\n");
        return 2;
    }
    return 0;
}
