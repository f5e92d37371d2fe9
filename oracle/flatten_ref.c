// oracle/flatten_ref.c — reference-side binding of librt_hip.so (see flatten_ref.h; in the reference tree this is
// src/rt/hip/flatten.c).  Compiled with the reference's own flags against its unmodified headers: C compiled as C++20
// by clang, common.h force-included.  It uses nothing of this repository but include/rtx.h.
//
//   Rtx_Flatten        Scene (opaque: src/world/scene.c:16-22) -> rtx_scene_desc pools
//   Render_Start_HIP   Render_Start's contract (src/rt/renderer.c:239-269) with the frame rendered by librt_hip.so
//
// Index conventions (include/rtx.h): a triangle / sphere keeps the index of its Object in scene->kdObjects as
// tri_obj / sph_obj (what rtx_trace_closest reports as prim_id); planes are the unboundObjs list in list order;
// materials and textures are numbered in the order they are first met (skybox faces, then kdObjects, then planes).
#include "flatten_ref.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "platform/threads.h"

#define RTX_MAX_TEXTURES 1024

typedef struct {
    RtxFlat*              f;
    const RtxPeekScene*   scene;
    const RtxPeekTexture* tex_ptr[RTX_MAX_TEXTURES];
    size_t                n_tex, n_mat, texels;
    int                   ok;
} FlatCtx;

static int32_t intern_texture(FlatCtx* c, const Texture* texture)
{
    const RtxPeekTexture* t = (const RtxPeekTexture*)texture;
    if (!t || !t->image) return -1;
    for (size_t i = 0; i < c->n_tex; i++)
        if (c->tex_ptr[i] == t) return (int32_t)i;
    if (c->n_tex == RTX_MAX_TEXTURES) {
        fprintf(stderr, "Rtx_Flatten: more than %d textures\n", RTX_MAX_TEXTURES);
        c->ok = 0;
        return -1;
    }
    const size_t w = t->image->res.width, h = t->image->res.height;
    RtxFlat*     f = c->f;
    f->textures    = (rtx_texture_desc*)realloc(f->textures, (c->n_tex + 1) * sizeof *f->textures);
    f->tex_data    = (float*)realloc(f->tex_data, (c->texels + w * h) * 3 * sizeof(float));
    if (!f->textures || !f->tex_data) ABORT("Rtx_Flatten: out of memory");
    f->textures[c->n_tex].width  = (uint32_t)w;
    f->textures[c->n_tex].height = (uint32_t)h;
    f->textures[c->n_tex].offset = c->texels;
    memcpy(f->tex_data + 3 * c->texels, t->image->pix, w * h * 3 * sizeof(float)); // ImageColor: f32 r,g,b, row 0 first (image.h:23-29)
    c->texels += w * h;
    c->tex_ptr[c->n_tex] = t;
    return (int32_t)c->n_tex++;
}

static uint8_t intern_material(FlatCtx* c, const Material* m)
{
    RtxFlat* f = c->f;
    if (!m) {
        fprintf(stderr, "Rtx_Flatten: object without material\n");
        c->ok = 0;
        return 0;
    }
    for (size_t i = 0; i < c->n_mat; i++)
        if (f->mat_ptr[i] == m) return (uint8_t)i;
    if (c->n_mat == 256) {
        fprintf(stderr, "Rtx_Flatten: more than 256 distinct materials\n");
        c->ok = 0;
        return 0;
    }
    rtx_material r;
    memset(&r, 0, sizeof r);
    r.type = (uint32_t)m->type; // RTX_MAT_* = Material_Type (materials.h:17-34)
    r.tex  = -1;
    switch (m->type) {
        case MATERIAL_DIFFUSE: r.tex = intern_texture(c, m->diffuse.albedo); break;
        case MATERIAL_METAL: r.tex = intern_texture(c, m->metal.albedo), r.p[0] = m->metal.fuzz; break;
        case MATERIAL_DIELECTRIC: r.tex = intern_texture(c, m->dielectric.albedo), r.p[0] = m->dielectric.refactiveIndex; break;
        case MATERIAL_DIFFUSE_LIGHT: r.tex = intern_texture(c, m->diffuse_light.albedo), r.p[0] = m->diffuse_light.brightness; break;
        case MATERIAL_SKYBOX:
            if (m->skybox.skybox != c->scene->skybox) { // one cube map per scene on the device
                fprintf(stderr, "Rtx_Flatten: Material_Skybox must reference the scene's own skybox\n");
                c->ok = 0;
            }
            break;
        default: { // the Disney kinds share Material_Disney_BSDF (materials.h:59-76); p[] keeps its field order
            const Material_Disney_BSDF* d = &m->disney;
            r.tex                         = intern_texture(c, d->albedo);
            const float p[12] = {d->subsurface, d->specular, d->roughness, d->specular_tint, d->anistropic, d->sheen_tint,
                                 d->clearcoat_gloss, d->eta, d->weights.clearcoat, d->weights.sheen, d->weights.metallic, d->weights.specular};
            memcpy(r.p, p, sizeof p);
        }
    }
    f->materials             = (rtx_material*)realloc(f->materials, (c->n_mat + 1) * sizeof *f->materials);
    if (!f->materials) ABORT("Rtx_Flatten: out of memory");
    f->materials[c->n_mat] = r;
    f->mat_ptr[c->n_mat]   = m;
    return (uint8_t)c->n_mat++;
}

template <typename T> static T* flat_alloc(size_t n) // (the reference compiles its C as C++20)
{
    T* p = (T*)calloc(n ? n : 1, sizeof(T));
    if (!p) ABORT("Rtx_Flatten: out of memory");
    return p;
}

RtxFlat* Rtx_Flatten(const Scene* scene_opaque)
{
    const RtxPeekScene* scene = (const RtxPeekScene*)scene_opaque;
    if (!scene || !scene->kdTree || !scene->kdObjects || scene->kdObjects->length == 0) {
        // the reference dereferences a NULL tree here (scene.c:109,159-163)
        fprintf(stderr, "Rtx_Flatten: the scene has no bounded object / kd-tree (Scene_Prepare first)\n");
        return NULL;
    }
    const RtxPeekKDTree* tree = scene->kdTree;
    const Object*        kd   = (const Object*)scene->kdObjects->at;
    const size_t         nk   = scene->kdObjects->length;
    const Object*        unb  = (const Object*)scene->unboundObjs->at;
    const size_t         P    = scene->unboundObjs->length;
    const size_t         N = tree->nodes->length, R = tree->objPtrs->length;
    if (nk >= (1u << 30) || N >= (1u << 30) || R >= (1u << 30)) {
        fprintf(stderr, "Rtx_Flatten: pool larger than 2^30 entries\n");
        return NULL;
    }

    RtxFlat* f = flat_alloc<RtxFlat>(1);
    FlatCtx  c;
    memset(&c, 0, sizeof c);
    c.f = f, c.scene = scene, c.ok = 1;

    int32_t sky[6];
    for (int i = 0; i < 6; i++) sky[i] = intern_texture(&c, scene->skybox ? scene->skybox->tex[i] : NULL); // xpos,xneg,ypos,yneg,zpos,zneg (skybox.c:30)

    size_t T = 0, S = 0;
    for (size_t i = 0; i < nk; i++) {
        if (kd[i].surface.type == SURFACE_TRIANGLE) T++;
        else if (kd[i].surface.type == SURFACE_SPHERE) S++;
        else {
            fprintf(stderr, "Rtx_Flatten: kdObjects[%zu] is neither a triangle nor a sphere\n", i);
            Rtx_FlatDelete(f);
            return NULL;
        }
    }
    f->tri_pos = flat_alloc<float>(9 * T), f->tri_nrm = flat_alloc<float>(9 * T), f->tri_uv = flat_alloc<float>(6 * T);
    f->tri_mat = flat_alloc<uint8_t>(T), f->tri_obj = flat_alloc<uint32_t>(T);
    f->sph = flat_alloc<float>(4 * S), f->sph_mat = flat_alloc<uint8_t>(S), f->sph_obj = flat_alloc<uint32_t>(S);
    f->plane = flat_alloc<float>(6 * P), f->plane_mat = flat_alloc<uint8_t>(P);
    uint32_t* kd2ref = flat_alloc<uint32_t>(nk); // kdObjects index -> leaf reference (pool bit | index in the pool)

    size_t ti = 0, si = 0;
    for (size_t i = 0; i < nk; i++) {
        const Object* o = &kd[i];
        if (o->surface.type == SURFACE_TRIANGLE) {
            const Triangle* tr = &o->surface.triangle;
            for (int v = 0; v < 3; v++) { // component-major SoA: [c * T + i]
                f->tri_pos[(3 * v + 0) * T + ti] = tr->vtx[v].pos.x, f->tri_pos[(3 * v + 1) * T + ti] = tr->vtx[v].pos.y, f->tri_pos[(3 * v + 2) * T + ti] = tr->vtx[v].pos.z;
                f->tri_nrm[(3 * v + 0) * T + ti] = tr->vtx[v].norm.x, f->tri_nrm[(3 * v + 1) * T + ti] = tr->vtx[v].norm.y, f->tri_nrm[(3 * v + 2) * T + ti] = tr->vtx[v].norm.z;
                f->tri_uv[(2 * v + 0) * T + ti] = tr->vtx[v].tex.x, f->tri_uv[(2 * v + 1) * T + ti] = tr->vtx[v].tex.y;
            }
            f->tri_mat[ti] = intern_material(&c, o->material);
            f->tri_obj[ti] = (uint32_t)i;
            kd2ref[i]      = (uint32_t)ti++;
        } else {
            const Sphere* sp = &o->surface.sphere;
            f->sph[0 * S + si] = sp->c.x, f->sph[1 * S + si] = sp->c.y, f->sph[2 * S + si] = sp->c.z, f->sph[3 * S + si] = sp->r;
            f->sph_mat[si] = intern_material(&c, o->material);
            f->sph_obj[si] = (uint32_t)i;
            kd2ref[i]      = RTX_REF_SPHERE | (uint32_t)si++;
        }
    }
    for (size_t i = 0; i < P; i++) {
        if (unb[i].surface.type != SURFACE_PLANE) {
            fprintf(stderr, "Rtx_Flatten: unboundObjs[%zu] is not a plane\n", i);
            c.ok = 0;
            break;
        }
        const Plane* pl = &unb[i].surface.plane;
        f->plane[0 * P + i] = pl->normal.x, f->plane[1 * P + i] = pl->normal.y, f->plane[2 * P + i] = pl->normal.z;
        f->plane[3 * P + i] = pl->point.x, f->plane[4 * P + i] = pl->point.y, f->plane[5 * P + i] = pl->point.z;
        f->plane_mat[i]     = intern_material(&c, unb[i].material);
    }

    // kd nodes: 10 packed bytes -> two words (see RTX_PEEK_KDNODE_BYTES); leaf lists: Object* -> leaf reference
    f->node_w0 = flat_alloc<uint32_t>(N), f->node_w1 = flat_alloc<uint32_t>(N), f->leaf_ref = flat_alloc<uint32_t>(R);
    const unsigned char* nb = (const unsigned char*)tree->nodes->at;
    for (size_t i = 0; i < N; i++, nb += RTX_PEEK_KDNODE_BYTES) {
        uint64_t hi = 0;
        memcpy(&f->node_w0[i], nb, 4);
        memcpy(&hi, nb + 4, 6);
        const uint64_t index = hi & ((1ull << 46) - 1), type = hi >> 46;
        if (index >= (1u << 30)) ABORT("Rtx_Flatten: kd index beyond 2^30 (kdtree.c:412-413 caps both pools there)");
        f->node_w1[i] = (uint32_t)type << 30 | (uint32_t)index;
    }
    Object* const* ptrs = (Object* const*)tree->objPtrs->at;
    for (size_t i = 0; i < R; i++) {
        const size_t k = (size_t)(ptrs[i] - kd);
        if (k >= nk) ABORT("Rtx_Flatten: leaf entry does not point into kdObjects");
        f->leaf_ref[i] = kd2ref[k];
    }
    free(kd2ref);
    if (!c.ok) {
        Rtx_FlatDelete(f);
        return NULL;
    }

    rtx_scene_desc* d = &f->desc;
    d->abi_version    = RTX_ABI_VERSION;
    d->n_nodes = (uint32_t)N, d->n_refs = (uint32_t)R, d->root_index = (uint32_t)tree->rootIndex;
    d->n_tris = (uint32_t)T, d->n_spheres = (uint32_t)S, d->n_planes = (uint32_t)P;
    d->n_materials = (uint32_t)c.n_mat, d->n_textures = (uint32_t)c.n_tex;
    d->max_depth   = (uint32_t)(size_t)(8.0 + 1.8 * log2(nk)) + 1; // KDTree_New's depth bound (kdtree.c:408) + the root
    d->node_w0 = f->node_w0, d->node_w1 = f->node_w1, d->leaf_ref = f->leaf_ref;
    d->tri_pos = f->tri_pos, d->tri_nrm = f->tri_nrm, d->tri_uv = f->tri_uv, d->tri_mat = f->tri_mat, d->tri_obj = f->tri_obj;
    d->sph = f->sph, d->sph_mat = f->sph_mat, d->sph_obj = f->sph_obj;
    d->plane = f->plane, d->plane_mat = f->plane_mat;
    d->materials = f->materials, d->textures = f->textures, d->tex_data = f->tex_data, d->tex_texels = c.texels;
    for (int i = 0; i < 6; i++) d->skybox_tex[i] = sky[i];
    return f;
}

void Rtx_FlatDelete(RtxFlat* f)
{
    if (!f) return;
    free(f->node_w0), free(f->node_w1), free(f->leaf_ref), free(f->tri_obj), free(f->sph_obj);
    free(f->tri_pos), free(f->tri_nrm), free(f->tri_uv), free(f->sph), free(f->plane), free(f->tex_data);
    free(f->tri_mat), free(f->sph_mat), free(f->plane_mat), free(f->materials), free(f->textures);
    free(f);
}

// ---------------------------------------------------------------------------------------------------------------------
// Render_Start_HIP
// ---------------------------------------------------------------------------------------------------------------------
typedef struct {
    void* handle;
    int (*scene_create)(const rtx_scene_desc*, int, rtx_scene**);
    int (*scene_destroy)(rtx_scene*);
    int (*render)(rtx_scene*, const rtx_camera*, const rtx_render_params*, uint8_t*, float*, rtx_stats*);
    int (*render_multi)(rtx_scene* const*, uint32_t, const rtx_camera*, const rtx_render_params*, uint32_t, uint32_t, uint8_t*, float*, rtx_stats*);
    const char* (*last_error)(void);
} RtxLib;

static RtxLib* rtx_lib(void)
{
    static RtxLib lib;
    if (lib.handle) return &lib;
    const char* path = getenv("RTX_HIP_LIB"); // default: found through the loader's search path
    void*       h    = dlopen(path ? path : "librt_hip.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) ABORT("Render_Start_HIP: %s", dlerror());
    lib.scene_create  = (int (*)(const rtx_scene_desc*, int, rtx_scene**))dlsym(h, "rtx_scene_create");
    lib.scene_destroy = (int (*)(rtx_scene*))dlsym(h, "rtx_scene_destroy");
    lib.render        = (int (*)(rtx_scene*, const rtx_camera*, const rtx_render_params*, uint8_t*, float*, rtx_stats*))dlsym(h, "rtx_render");
    lib.render_multi  = (int (*)(rtx_scene* const*, uint32_t, const rtx_camera*, const rtx_render_params*, uint32_t, uint32_t, uint8_t*, float*, rtx_stats*))dlsym(h, "rtx_render_multi");
    lib.last_error    = (const char* (*)(void))dlsym(h, "rtx_last_error");
    int (*abi)(void)  = (int (*)(void))dlsym(h, "rtx_abi_version");
    if (!lib.scene_create || !lib.scene_destroy || !lib.render || !lib.render_multi || !lib.last_error || !abi) ABORT("Render_Start_HIP: librt_hip.so lacks the rtx_* entry points");
    if (abi() != RTX_ABI_VERSION) ABORT("Render_Start_HIP: librt_hip.so has ABI %d, this binding %d", abi(), RTX_ABI_VERSION);
    lib.handle = h;
    return &lib;
}

typedef struct {
    RenderCtx* ctx;
    size_t     samples_per_pixel, max_ray_depth;
} HipWaiterArg;

static u64 seed_word(const char* env_name) // RTX_SEED_LO / RTX_SEED_HI pin the frame; else rdrand like renderer.c:143
{
    const char* e = getenv(env_name);
    if (e) return strtoull(e, NULL, 0);
    unsigned long long v = 0;
    for (int i = 0; i < 16 && !__builtin_ia32_rdrand64_step(&v); i++) {}
    return v;
}

#define RTX_MAX_DEVICES 64

// RTX_DEVICES="0,1,2,3,4,5,6,7" (default: RTX_DEVICE, default 0): the devices that share the frame — where Render_Waiter
// sizes its worker array by NUM_HYPERTHREADS (renderer.c:208-209)
static uint32_t device_list(int* ids)
{
    const char* list = getenv("RTX_DEVICES");
    uint32_t    n    = 0;
    if (list && *list) {
        for (const char* p = list; *p && n < RTX_MAX_DEVICES;) {
            char* end;
            long  v = strtol(p, &end, 10);
            if (end == p || v < 0) ABORT("RTX_DEVICES: cannot read '%s'", list);
            ids[n++] = (int)v;
            p        = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') ABORT("RTX_DEVICES: cannot read '%s'", list);
        }
    } else {
        const char* one = getenv("RTX_DEVICE");
        ids[n++]        = one ? atoi(one) : 0;
    }
    return n;
}

static void Render_Waiter_HIP(void* arg) // stands where Render_Waiter + Render_Worker stand (renderer.c:141-237)
{
    HipWaiterArg* a   = (HipWaiterArg*)arg;
    RenderCtx*    ctx = a->ctx;
    RtxLib*       lib = rtx_lib();
    RtxFlat*      flat = Rtx_Flatten(ctx->scene);
    if (!flat) ABORT("Render_Start_HIP: the scene cannot be flattened");
    int        ids[RTX_MAX_DEVICES];
    rtx_scene* dev[RTX_MAX_DEVICES];
    uint32_t   n = device_list(ids);
    for (uint32_t i = 0; i < n; i++) // the scene is replicated: every device holds the whole tree (the reference's workers share one)
        if (lib->scene_create(&flat->desc, ids[i], &dev[i]) != RTX_OK) ABORT("rtx_scene_create on device %d: %s", ids[i], lib->last_error()); // the reference aborts on any failure (macros.h:60-64)
    rtx_render_params p;
    memset(&p, 0, sizeof p);
    p.width = (uint32_t)ctx->img->res.width, p.height = (uint32_t)ctx->img->res.height;
    p.spp = (uint32_t)a->samples_per_pixel, p.max_depth = (uint32_t)a->max_ray_depth;
    p.seed_lo = seed_word("RTX_SEED_LO"), p.seed_hi = seed_word("RTX_SEED_HI");
    // ImageRGB.pix is RGB{b,g,r} (color.h:19-23), row 0 = bottom scanline: exactly rtx_render's bgr_out; Camera is rtx_camera
    if (n == 1) {
        if (lib->render(dev[0], (const rtx_camera*)ctx->cam, &p, (uint8_t*)ctx->img->pix, NULL, NULL) != RTX_OK) ABORT("rtx_render: %s", lib->last_error());
    } else {
        // RTX_SPLIT=sample: every device traces its range of the samples of every pixel (BASELINE config 5); default: tile
        // split, interleaved bands of RTX_BAND_ROWS rows (config 3).  One worker per device inside the library, joined there.
        const char* split = getenv("RTX_SPLIT");
        const char* rows  = getenv("RTX_BAND_ROWS");
        if (lib->render_multi(dev, n, (const rtx_camera*)ctx->cam, &p, split && !strcmp(split, "sample") ? RTX_SPLIT_SAMPLE : RTX_SPLIT_TILE,
                              rows ? (uint32_t)atoi(rows) : 0u, (uint8_t*)ctx->img->pix, NULL, NULL)
            != RTX_OK)
            ABORT("rtx_render_multi: %s", lib->last_error());
    }
    for (uint32_t i = 0; i < n; i++) lib->scene_destroy(dev[i]);
    Rtx_FlatDelete(flat);
    ctx->finished = true; // renderer.c:236
}

void Render_Start_HIP(RenderCtx* ctx, size_t samples_per_pixel, size_t max_ray_depth)
{
    ctx->finished = false; // renderer.c:267
    // Start_WaiterThread (renderer.c:239-263) with the other entry point; Render_Delete (:44-52) frees both as before
    HipWaiterArg* arg = (HipWaiterArg*)calloc(1, sizeof *arg);
    if (arg == NULL) ABORT("Failed to create render waiter thread args");
    arg->ctx = ctx, arg->samples_per_pixel = samples_per_pixel, arg->max_ray_depth = max_ray_depth;
    Thread* waiter = Thread_New();
    if (waiter == NULL) ABORT("Failed to create render waiter thread");
    Thread_Set_StackSize(waiter, 8 * 1024 * 1024); // the HIP runtime wants more than the CPU waiter's 64 KiB
    ctx->waiter_thread = waiter;
    ctx->waiter_args   = arg;
    if (!Thread_Spawn(waiter, Render_Waiter_HIP, arg)) ABORT("Failed to start render waiter thread");
}
