/*
 * rtx.h — C ABI of librt_hip.so, the MI355X (gfx950) ray→scene query engine that replaces the
 * hot path of deggua/raytracer:  Scene_ClosestHit (src/world/scene.c:100) → KDTree_HitAt
 * (src/rt/accelerators/kdtree.c:572) → Surface_HitAt (src/world/object.c:7) → Material_Bounce
 * (src/world/object.c:64), driven by RayColor/RenderPixel (src/rt/renderer.c:54-111).
 *
 * The reference has no FFI: the seam is its render loop (Render_Start/Render_Done,
 * src/rt/renderer.h:19-22).  A caller flattens the host scene (kd-tree nodes, leaf index runs,
 * triangles, spheres, planes, materials, textures) into the pools of `rtx_scene_desc`, uploads them
 * once with rtx_scene_create and renders whole frames with rtx_render.  INTEGRATION.md shows the
 * shim a reference maintainer adds next to Render_Start.
 *
 * Conventions: plain C, no C++ types, no exceptions, no abort(); every function returns 0 (RTX_OK)
 * or a negative RTX_E_* code and sets a thread-local message readable through rtx_last_error().
 * All multi-byte values are little-endian; all indices are u32; all floats IEEE binary32.
 * The library fails loudly (RTX_E_NO_DEVICE) when no gfx950 device is visible: there is no CPU path.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 1

enum {
    RTX_OK            = 0,
    RTX_E_INVALID     = -1, /* bad argument / malformed scene description        */
    RTX_E_NO_DEVICE   = -2, /* no HIP device, or the requested ordinal is absent  */
    RTX_E_HIP         = -3, /* a HIP runtime call failed (message has the detail) */
    RTX_E_NOMEM       = -4, /* host or device allocation failed                   */
    RTX_E_LIMIT       = -5, /* a documented size limit was exceeded               */
    RTX_E_EMPTY_SCENE = -6, /* no bounded primitive: the reference dereferences a NULL tree
                               here (src/world/scene.c:109,159-163); we refuse instead      */
};

/* Material kinds: values match Material_Type (src/rt/materials.h:17-34). */
enum {
    RTX_MAT_DIFFUSE          = 0,
    RTX_MAT_METAL            = 1,
    RTX_MAT_DIELECTRIC       = 2,
    RTX_MAT_DIFFUSE_LIGHT    = 3,
    RTX_MAT_SKYBOX           = 4,
    RTX_MAT_DISNEY_DIFFUSE   = 5,
    RTX_MAT_DISNEY_METAL     = 6,
    RTX_MAT_DISNEY_GLASS     = 7,
    RTX_MAT_DISNEY_CLEARCOAT = 8,
    RTX_MAT_DISNEY_SHEEN     = 9,
    RTX_MAT_DISNEY_BSDF      = 10,
};

/* One material, 64 bytes.  Mirrors `Material` (src/rt/materials.h:78-90) with the Texture*
 * replaced by an index into the texture table (-1 = none; clearcoat has none).
 *   simple kinds: p[0] = fuzz (metal) | refractive index (dielectric) | brightness (light)
 *   Disney kinds: p[0..11] = subsurface, specular, roughness, specular_tint, anistropic,
 *                 sheen_tint, clearcoat_gloss, eta, w.clearcoat, w.sheen, w.metallic, w.specular
 *                 (field order of Material_Disney_BSDF, src/rt/materials.h:59-76)            */
typedef struct rtx_material {
    uint32_t type;
    int32_t  tex;
    float    p[14];
} rtx_material;

/* A texture = linear-light f32 RGB image (ImageColor, src/gfx/image.h:23-29), row-major,
 * row 0 first, 3 floats per texel, starting at tex_data[3*offset]. */
typedef struct rtx_texture_desc {
    uint32_t width;
    uint32_t height;
    uint64_t offset; /* in texels */
} rtx_texture_desc;

/* Camera block = `Camera` (src/world/camera.h:6-13), 88 bytes, produced by Camera_New. */
typedef struct rtx_camera {
    float origin[3];
    float bottom_left[3];
    float horizontal[3];
    float vertical[3];
    float u[3], v[3], w[3];
    float lens_radius;
} rtx_camera;

/* leaf_ref encoding: bit 31 = primitive pool (0 triangle, 1 sphere), bits 0-30 = pool index. */
#define RTX_REF_SPHERE 0x80000000u
/* prim_id convention of rtx_trace_closest: index into the reference's kdObjects
 * array for bounded hits, RTX_ID_UNBOUNDED|index into unboundObjs for plane hits, -1 for a miss. */
#define RTX_ID_UNBOUNDED 0x80000000u
#define RTX_ID_MISS      0xFFFFFFFFu

/* Flattened scene.  Arrays named "[c*N + i]" are component-major SoA: component c of item i.
 * The caller keeps ownership; rtx_scene_create copies everything it needs.
 *
 * kd-tree (KDNode, src/rt/accelerators/kdtree.c:71-114, re-packed to 8 bytes):
 *   node_w0[i]  = f32 bits of `split` (internal) or `len` (leaf)
 *   node_w1[i]  = type<<30 | index     type 0/1/2 = split axis X/Y/Z, 3 = leaf
 *                 index = leftIndex (internal; right child is i+1, kdtree.c:153-155,507)
 *                       | first entry in leaf_ref (leaf)
 *   leaf_ref[k] = objPtrs (kdtree.c:130,185-193) with each Object* replaced by pool|index     */
typedef struct rtx_scene_desc {
    uint32_t abi_version; /* RTX_ABI_VERSION */
    uint32_t n_nodes, n_refs, root_index;
    uint32_t n_tris, n_spheres, n_planes;
    uint32_t n_materials, n_textures;
    uint32_t max_depth; /* kd-tree depth bound (kdtree.c:408) + 1: sizes the traversal stack */

    const uint32_t* node_w0;
    const uint32_t* node_w1;
    const uint32_t* leaf_ref;

    /* triangles (Triangle/Vertex, src/rt/surfaces.h:29-37) */
    const float*    tri_pos; /* [c*T+i], c = 3*vertex+axis, 9 components  */
    const float*    tri_nrm; /* [c*T+i], 9 components, NOT normalised      */
    const float*    tri_uv;  /* [c*T+i], c = 2*vertex+{u,v}, 6 components  */
    const uint8_t*  tri_mat; /* material index                             */
    const uint32_t* tri_obj; /* index in the reference's kdObjects array   */

    /* spheres (Sphere, surfaces.h:24-27): [c*S+i], c = cx,cy,cz,r */
    const float*    sph;
    const uint8_t*  sph_mat;
    const uint32_t* sph_obj;

    /* planes (Plane, surfaces.h:39-42), the "unbounded" list of Scene_Prepare
     * (src/world/scene.c:150-156), in list order: [c*P+i], c = nx,ny,nz,px,py,pz */
    const float*   plane;
    const uint8_t* plane_mat;

    const rtx_material*     materials;
    const rtx_texture_desc* textures;
    const float*            tex_data;
    uint64_t                tex_texels; /* total texels in tex_data */
    int32_t                 skybox_tex[6]; /* xpos,xneg,ypos,yneg,zpos,zneg (src/world/skybox.c:30) */
} rtx_scene_desc;

typedef struct rtx_scene rtx_scene; /* opaque; one per (scene, device) */

#define RTX_MAX_DEPTH 64

/* Frame partition for multi-GPU.  Rows are grouped in bands of `band_rows`; this call
 * renders band b iff b % band_world == band_rank.  Samples [sample_begin, sample_end) of every
 * rendered pixel are traced; the RNG stream of a path is keyed by its GLOBAL (pixel, sample):
 *   Random_Seed(seed_lo ^ (pixel_index*spp + sample), seed_hi)     (src/math/random.c:78-85)
 * so the union over ranks equals the single-GPU sample set. */
typedef struct rtx_render_params {
    uint32_t width, height;
    uint32_t spp;       /* samples per pixel of the whole job (argv[1] of the reference) */
    uint32_t max_depth; /* max_ray_depth of Render_Start (argv[2]); counts rays, not bounces */
    uint64_t seed_lo, seed_hi;
    uint32_t sample_begin, sample_end; /* 0,0 = all */
    uint32_t band_rows;                /* 0 = whole frame on this device */
    uint32_t band_rank, band_world;
    uint32_t flags;
    uint64_t max_paths_in_flight; /* 0 = auto (whole assigned frame if it fits the HBM budget) */
} rtx_render_params;

#define RTX_RENDER_COUNT_VISITS 1u /* also count kd-tree visits (slower kernel variant)      */
#define RTX_RENDER_PIXEL_CENTRES 2u /* replace the two jitter draws by 0.5 (config-1 ray set A) */
#define RTX_TRACE_UNIT_DIR 4u /* rtx_trace_closest: dir_xyz already went through Ray_Make; use as is */
#define RTX_TRACE_LITERAL 8u  /* closest hit by the literal kd walk (kdtree.c:453-577 step for step), without
                               * the subtree-box culling; same results, more node visits */

typedef struct rtx_stats {
    uint64_t rays_total;
    uint64_t rays_at_depth[RTX_MAX_DEPTH];
    uint64_t camera_paths;
    uint64_t visits_inner, visits_leaf, visits_prim; /* only with RTX_RENDER_COUNT_VISITS */
    uint64_t hits_shaded;                            /* rays that reached Material_Bounce   */
    double   ms_total;                               /* whole call, device time (events)    */
    double   ms_trace;                               /* sum over closest-hit launches; a queue of <= 2 M rays finishes all its
                                                      * remaining depths in ONE launch that also shades: that
                                                      * launch counts here, at the depth it starts at (no COUNT_VISITS) */
    double   ms_shade;
    double   ms_raygen;
    double   ms_resolve;
    uint32_t trace_launches;
    uint32_t passes; /* slabs */
    double   ms_trace_at_depth[RTX_MAX_DEPTH]; /* closest-hit kernel time per ray depth */
    uint64_t visits_culled; /* nodes fetched and skipped by the box test (COUNT_VISITS, culled walk) */
    double   ms_sort;       /* ray binning between bounce passes */
    /* COUNT_VISITS, rtx_render*: closest-hit wave statistics summed over waves — loop iterations, INNER block
     * runs and the lanes they served, LEAF runs and lanes, POP runs and lanes, FIN runs */
    uint64_t wave_stats[8];
    uint32_t samples_done; /* rtx_render_progressive: samples per pixel in the image when the call returned */
    uint32_t pad_;
    uint64_t rays_early_miss; /* of rays_total: rays that miss the box of the whole kd-tree and every unbounded object, answered
                               * (sky) by the shade pass that created them instead of a lane of the closest-hit kernel */
} rtx_stats;

/* ---- library ---- */
int         rtx_abi_version(void);
const char* rtx_last_error(void);
int         rtx_device_count(int* count);

/* ---- tuning knobs.  The render loop's switches (most exist so that a test can render a frame both ways and compare the
 * bits; none changes a result) live in one process-wide table: OVERLAP, SORT_DEPTH, SORT_MIN, SORT_EVERY, TAIL_MAX,
 * TAIL_SHADE_MIN, JUMP_TABLE, RAYGEN_LEAN, EARLY_MISS, FAST_DIV, BAND_TILES, TRACE_CHUNK, AHEAD, TILE_INTERLEAVE.  The environment
 * (RTX_<NAME>) is read ONCE, at first use, for the defaults; no entry point reads the environment per call.  Set a knob
 * only while no call is in flight; a value outside a knob's range is RTX_E_INVALID (TRACE_CHUNK 0..4096, OVERLAP 0..3,
 * TAIL_SHADE_MIN 1..64, the on/off knobs 0..1), and so is any change while rtx_render_multi runs.  No counterpart in the reference (its switches are compile-time, src/common/config.h). ---- */
int  rtx_tune_set(const char* name, long long value);
int  rtx_tune_get(const char* name, long long* value);
void rtx_tune_reset(void); /* back to the defaults (built-in, or RTX_<NAME> as it was at first use) */

/* ---- scene hand-over: replaces the pointer the render workers read (ctx->scene,
 *      src/rt/renderer.c:151) with device-resident pools ---- */
int rtx_scene_create(const rtx_scene_desc* desc, int device, rtx_scene** out);
/* The same with flags.  Textures are kept on the device in a compact, LOSSLESS form where their values allow it
 * (SURVEY §8f-3): a texture whose channels take at most 256 distinct binary32 values — every imported BMP, whose texels
 * are sRGBToLinear(byte / 255) (src/gfx/texture.c:15-47) — is stored as 3 bytes per texel + a 256-entry palette of the
 * original bit patterns, so Texture_ColorAt / Skybox_ColorAt (src/gfx/texture.c:88-95, src/world/skybox.c:66-134)
 * return exactly the floats of the f32 image at a quarter of the footprint.  RTX_SCENE_F32_TEXTURES keeps every
 * texture as handed over (the tests compare the two). */
#define RTX_SCENE_F32_TEXTURES 1u
/* A skybox face that is a single texel (Texture_Import_Color, src/gfx/texture.c:49-70: the constant skies of the BASELINE
 * configs) is answered without the lookup: Texture_ColorAt returns that texel whatever u and v are, so Skybox_ColorAt's two
 * quotients and three dependent loads buy nothing.  RTX_SCENE_SKY_LOOKUP keeps the full lookup for every face (same floats). */
#define RTX_SCENE_SKY_LOOKUP 2u
int rtx_scene_create_ex(const rtx_scene_desc* desc, int device, uint32_t flags, rtx_scene** out);
int rtx_scene_destroy(rtx_scene* scene);
int rtx_scene_device_bytes(const rtx_scene* scene, uint64_t* bytes);
/* Parity seam for the culled walk (host only, needs no device): the conservative, quantised subtree
 * boxes rtx_scene_create uploads — qbox6: n_nodes x {min x y z, max x y z} as 16-bit steps of the grid
 * lo + q * scale (min.x = 0xFFFF with max.x = 0 meaning "never cull") — and info14 = {centre xyz, origin
 * radius^2, tMax stretch, lo xyz, scale xyz, 0 0 0}, where the device evaluates a plane's ray parameter as
 * fma(float(q), scale, -(o - lo)) * (1/d), so that a CPU checker can replay the very same culling decisions.  No
 * counterpart in the reference (its KDTree.worldBox, kdtree.c:435, is never used by the traversal). */
int rtx_cull_data_build(const rtx_scene_desc* desc, uint16_t* qbox6, float* info14);
/* Second half of that seam (host only).  Cut-off nodes — inner nodes with an empty leaf for a child, 44 % of what KDTree_New
 * (src/rt/accelerators/kdtree.c:405-444) builds — are folded into the step of their non-empty child on the device:
 * flags[n] bit 0 = node n is never fetched (its plane, CheckHitInternalNode :501-570, is evaluated when its non-empty child is
 * entered), bit 1 = node n carries its parent's plane and is entered without a box test of its own (its box is its parent's).
 * n_nodes bytes. */
int rtx_cull_node_flags(const rtx_scene_desc* desc, uint8_t* flags);

/* ---- ray-level parity seam: batched Scene_ClosestHit (src/world/scene.h:12).
 * Host pointers, interleaved xyz.  dir need not be normalised: Ray_Make (src/rt/ray.c:3-16)
 * is applied on the device.  Outputs may be NULL.  `stats` may be NULL. ---- */
int rtx_trace_closest(rtx_scene* scene, size_t n, const float* origin_xyz, const float* dir_xyz,
                      uint32_t* prim_id, float* t, float* position_xyz, float* normal_xyz, float* uv,
                      uint8_t* front_face, rtx_stats* stats, uint32_t flags);

/* Same query on device-resident SoA rays (ox,oy,oz,dx,dy,dz each n floats at rays+c*n), already
 * normalised by a previous Ray_Make; writes prim ref/t/u/v (4 x n dwords, SoA) to `hits`.
 * Asynchronous on `stream` (a hipStream_t).
 * SERIALISE the calls on one rtx_scene: every call that takes a scene handle (this one, rtx_render*, rtx_trace_closest)
 * uses that scene's work counters, workspace and events; two such calls in flight at once — from two threads or on two
 * streams — race.  Distinct rtx_scene handles are independent. */
int rtx_trace_closest_device(rtx_scene* scene, size_t n, const float* d_rays, uint32_t* d_hits,
                             void* stream, uint32_t flags, uint64_t* d_visit_counters /* 4 x u64: inner, leaf, prim, culled */);

/* ---- the same query for callers that CANNOT ALLOCATE.  No counterpart in the reference: it is there for code
 * under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  rtx_trace_closest with DEVICE
 * pointers in the same interleaved layouts (origin / dir / position / normal: xyz; uv: 2 floats; outputs may be
 * NULL), asynchronous on `stream`; it calls no hipMalloc / hipFree or other allocating routine, only kernel
 * launches and one hipMemsetAsync.  The staging rtx_trace_closest allocates per call (the SoA ray queue, the
 * kernel's hit records, its work counters: 40 bytes per ray + 512) lives in a static device array of
 * RTX_NOALLOC_MAX_RAYS rays (10 MB) that every GPU the library runs on holds until the process ends; a batch
 * beyond that is refused with RTX_E_LIMIT.  All calls on one GPU — whatever the scene or the stream — share that
 * GPU's array: calls that overlap on one GPU race; serialise them (one stream, or events).  RTX_TRACE_UNIT_DIR
 * and RTX_TRACE_LITERAL as above; RTX_RENDER_COUNT_VISITS is refused.  The calling thread's current device is the same
 * after the call as before it (every other entry point that takes a scene leaves the scene's device current). ---- */
#define RTX_NOALLOC_MAX_RAYS (1u << 18)
int rtx_closest_hit_noalloc(rtx_scene* scene, size_t n, const float* d_origin_xyz, const float* d_dir_xyz,
                            uint32_t* d_prim_id, float* d_t, float* d_position_xyz, float* d_normal_xyz, float* d_uv,
                            uint8_t* d_front_face, void* stream, uint32_t flags);

/* ---- frame-level seam: replaces Render_Worker/RenderTile/RenderPixel/RayColor
 *      (src/rt/renderer.c:54-178).  Blocking.  bgr_out = ImageRGB.pix (RGB{b,g,r},
 *      src/gfx/color.h:19-23; row 0 = bottom scanline), width*height*3 bytes, host memory, may be
 *      NULL.  accum_out = width*height*3 f32, host memory, per-pixel SUM of the traced samples'
 *      RayColor in r,g,b order (rows not assigned to this call are left untouched), may be NULL. */
int rtx_render(rtx_scene* scene, const rtx_camera* cam, const rtx_render_params* params,
               uint8_t* bgr_out, float* accum_out, rtx_stats* stats);

/* Progressive frame.  The reference's workers write finished tiles into ImageRGB.pix while the
 * preview loop blits it every 200 ms (src/main.c:436-478) and a SIGINT dumps what is there (:47-58).  Here the
 * frame fills in by SAMPLES: passes of `spp_per_pass` samples of every assigned pixel; after each pass bgr_out
 * (host, may be read concurrently by a preview: every byte is always a valid pixel value) holds the image of
 * the samples so far, sum * (1.0f/samples_done), and `progress` (may be NULL) is called — a nonzero return
 * stops the frame there (the partial export), which is not an error.  The accumulation continues the
 * sequential per-pixel sum in sample order across passes (renderer.c:98-107), so a frame that runs to the
 * end is BYTE-IDENTICAL to rtx_render's, whatever spp_per_pass.  Blocking; accum_out as in rtx_render. */
typedef int (*rtx_progress_fn)(void* user, uint32_t samples_done, uint32_t samples_total);
int rtx_render_progressive(rtx_scene* scene, const rtx_camera* cam, const rtx_render_params* params, uint32_t spp_per_pass,
                           uint8_t* bgr_out, float* accum_out, rtx_progress_fn progress, void* user, rtx_stats* stats);

/* ---- one frame over SEVERAL devices: what Render_Waiter does with its NUM_HYPERTHREADS workers (src/rt/renderer.c:186-237:
 *      spawn one worker per hardware thread, let them share the frame, join) with one host thread per DEVICE.  scenes[r] is
 *      the scene uploaded to device r of the job (rtx_scene_create of the same description on n devices; handles on the
 *      same device are allowed and serialise there).  `params` describes the WHOLE frame (band_* and sample_* must be 0).
 *        RTX_SPLIT_TILE:   device r renders the bands b = r (mod n) of `band_rows` rows (0 = 8) — BASELINE config 3; the
 *                          shares are disjoint and are copied into scenes[0]'s buffer device to device: bgr_out / accum_out
 *                          are rtx_render's, bit for bit;
 *        RTX_SPLIT_SAMPLE: device r renders samples [r spp/n, (r+1) spp/n) of every pixel — config 5; the shares are added
 *                          on scenes[0]'s device in rank order (partial sums of renderer.c:106 in another association:
 *                          equal to rtx_render's sums to f32 rounding, reproducible from run to run).
 *      The shares travel by hipMemcpyPeerAsync over xGMI (n-1 peers, n-1 links into the root); no communicator, no
 *      library beyond the HIP runtime.  Blocking.  bgr_out / accum_out: host, as in rtx_render, may be NULL (the call
 *      then only renders and reduces: what a benchmark times).  stats: n entries (one per device) or NULL.
 *      The tuning table is read-only while a call runs (rtx_tune_set fails). ---- */
#define RTX_SPLIT_TILE   0u
#define RTX_SPLIT_SAMPLE 1u
int rtx_render_multi(rtx_scene* const* scenes, uint32_t n_scenes, const rtx_camera* cam, const rtx_render_params* params,
                     uint32_t split_mode, uint32_t band_rows, uint8_t* bgr_out, float* accum_out, rtx_stats* stats);

/* Device-output variant for multi-GPU: d_accum (f32 x 3 x W x H, device) receives the per-pixel
 * sums of the rows/samples assigned to this call, ADDED sample by sample, in sample order, to what the buffer
 * holds (zero-fill it before the first call of a frame; a later call with the next sample range continues the
 * same sequential sum);
 * the caller reduces it across ranks (RCCL) and then calls rtx_resolve_device. Runs on `stream`
 * and synchronises it before returning (stats need the event times). */
int rtx_render_device(rtx_scene* scene, const rtx_camera* cam, const rtx_render_params* params,
                      float* d_accum, void* stream, rtx_stats* stats);

/* RenderPixel's tail (src/rt/renderer.c:109-110): avg = sum * (1.0f/spp); RGB_FromColor
 * (src/gfx/color.c:18-29,49-56) → BGR u8.  d_accum and d_bgr are device pointers. */
int rtx_resolve_device(rtx_scene* scene, const float* d_accum, uint32_t width, uint32_t height,
                       uint32_t spp, uint8_t* d_bgr, void* stream);

/* ---- unit-level parity seams (host pointers; small batches) ---- */

/* RenderPixel jitter + Camera_GetRay + Ray_Make (src/rt/renderer.c:101-104, src/world/camera.c:34-49)
 * for paths (pixel_index[i], sample[i]); outputs origin/dir (interleaved xyz) and the RNG state
 * after the four draws. */
int rtx_camera_rays(rtx_scene* scene, const rtx_camera* cam, const rtx_render_params* params,
                    size_t n, const uint32_t* pixel_index, const uint32_t* sample, float* origin_xyz,
                    float* dir_xyz, uint32_t* rng_state4);

/* Random_Seed(seed_lo, seed_hi) then `n` Random_Unilateral() draws (src/math/random.c:78-105). */
int rtx_rng_unilateral(rtx_scene* scene, uint64_t seed_lo, uint64_t seed_hi, size_t n, float* out);

/* Material_Bounce (src/world/object.c:64-143) for n independent (ray, hit) records against
 * material `mat_index`, each with its own RNG state (in/out).  cont[i] = the bool returned. */
int rtx_material_bounce(rtx_scene* scene, uint32_t mat_index, size_t n, const float* ray_dir_xyz,
                        const float* hit_position_xyz, const float* hit_normal_xyz, const float* hit_uv,
                        const uint8_t* front_face, uint32_t* rng_state4, float* surface_rgb,
                        float* emitted_rgb, float* out_origin_xyz, float* out_dir_xyz, uint8_t* cont);

/* Skybox_ColorAt (src/world/skybox.c:66-134) for n directions. */
int rtx_skybox_color(rtx_scene* scene, size_t n, const float* dir_xyz, float* rgb);

/* ---- SAH kd-tree build on the device: replaces the recursion under KDTree_New
 *      (src/rt/accelerators/kdtree.c:244-444: BuildNode / ComputeSplitSAH / BuildParentNode /
 *      BuildLeafNode) for the boxes CreateKDBBs (:385-403) produced.  Level-synchronous on the device,
 *      every f32 operation of the SAH search is the reference's, and the result is numbered in the
 *      reference's order (parent, RIGHT subtree, LEFT subtree; right child = parent + 1), so the three
 *      arrays are identical to the ones the host build of rt_host.h produces — same node words, same
 *      leaf lists.  box_min_max: n x {min x y z, max x y z} (host); world_min_max: BoxBoundingAll
 *      (:207-230) of them; max_depth: (size_t)(8.0 + 1.8*log2(n)) (:408).  The arrays in `out` are
 *      malloc'ed by the library: release them with rtx_kd_free. ---- */
typedef struct rtx_kd_tree {
    uint32_t* node_w0;       /* split plane (f32 bits) of an inner node | length of a leaf's list     */
    uint32_t* node_w1;       /* tag:2 (axis 0/1/2, 3 = leaf) | idx:30 (left child | first leaf_ref)    */
    uint64_t  n_nodes;
    uint32_t* leaf_ref;      /* indices into the n boxes                                               */
    uint64_t  n_refs;
    uint64_t  root_index;
    uint64_t  depth_reached; /* deepest level that holds a node (root = 0)                            */
    double    ms_device;     /* wall time of the device part (levels), for the record                 */
    double    ms_total;      /* + upload, download and the final numbering pass on the host           */
} rtx_kd_tree;
int  rtx_kd_build(int device, const float* box_min_max, uint64_t n, uint64_t max_depth, const float* world_min_max,
                  rtx_kd_tree* out);
void rtx_kd_free(rtx_kd_tree* tree);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
