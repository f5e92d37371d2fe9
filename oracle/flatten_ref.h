// oracle/flatten_ref.h — the reference-side binding of librt_hip.so, as a maintainer of deggua/raytracer would add it
// (INTEGRATION.md §2 reproduces this file and flatten_ref.c; in the reference tree they are src/rt/hip/flatten.{h,c}).
//
// TEST INFRASTRUCTURE in this repository: compiled by oracle/Makefile against the UNMODIFIED reference headers, linked
// into oracle/_ref/ref_parity, and checked by tests/test_reference_binding.py (pools byte-compared with Scene_Flatten's;
// a frame rendered by the reference's own Render_New / Render_Done loop through Render_Start_HIP).  The product never
// loads it.
//
// The flattener has to read four layouts that no reference header exports.  They are re-declared here under other names
// ("peek" structs); oracle/ref_driver.c, the one translation unit that sees both these and the real definitions (it
// #includes scene.c, kdtree.c and texture.c), holds the static_asserts that keep them equal (sizes and every offset).
#pragma once

#include <stddef.h>
#include <sys/types.h>

#include "rt/renderer.h" // RenderCtx, Scene (opaque), Camera, ImageRGB, Object, Material, Skybox, Texture (opaque)
#include "rtx.h"

// Vector(T), src/ctl/containers/vector.h:70-77
typedef struct RtxPeekVector {
    void*  at;
    size_t length;
    size_t capacity;
} RtxPeekVector;

// struct KDTree, src/rt/accelerators/kdtree.c:128-133
typedef struct RtxPeekKDTree {
    RtxPeekVector* nodes;   // Vector(KDNode): 10-byte packed nodes, kdtree.c:71-123
    RtxPeekVector* objPtrs; // Vector(ObjectPtr): Object* into scene->kdObjects->at
    BoundingBox    worldBox;
    ssize_t        rootIndex;
} RtxPeekKDTree;

// struct Scene, src/world/scene.c:16-22
typedef struct RtxPeekScene {
    Skybox*        skybox;
    RtxPeekVector* objects;
    RtxPeekVector* unboundObjs; // Vector(Object): the planes, scene.c:150-156
    RtxPeekVector* kdObjects;   // Vector(Object): everything the kd-tree holds
    RtxPeekKDTree* kdTree;
} RtxPeekScene;

// struct Texture, src/gfx/texture.c:5-7
typedef struct RtxPeekTexture {
    ImageColor* image;
} RtxPeekTexture;

// union KDNode, kdtree.c:71-114: 10 bytes, little-endian.  bytes 0-3 = split (f32, inner) | len (u32, leaf);
// bytes 4-9 = 48 bits: low 46 = leftIndex | objIndex, top 2 = type (0/1/2 = axis, 3 = leaf)
#define RTX_PEEK_KDNODE_BYTES 10

static_assert(sizeof(Object) == 112, "Object{Material*, Surface} is 112 bytes (src/world/object.h:9-18)");
static_assert(sizeof(Material) == 64, "Material is 64 bytes (src/rt/materials.h:78-90)");
static_assert(sizeof(Camera) == 88 && sizeof(rtx_camera) == 88, "rtx_camera is Camera (src/world/camera.h:6-13)");
static_assert(sizeof(Vertex) == 32 && sizeof(Triangle) == 96, "Vertex{pos,norm,tex}, Triangle{vtx[3]} (src/rt/surfaces.h:29-37)");
static_assert(sizeof(rtx_material) == 64, "rtx_material is 64 bytes");

typedef struct RtxFlat {
    rtx_scene_desc desc; // points into the arrays below
    uint32_t *     node_w0, *node_w1, *leaf_ref, *tri_obj, *sph_obj;
    float *        tri_pos, *tri_nrm, *tri_uv, *sph, *plane, *tex_data;
    uint8_t *      tri_mat, *sph_mat, *plane_mat;
    rtx_material*  materials;
    rtx_texture_desc* textures;
    const Material*   mat_ptr[256]; // materials[i] was made from mat_ptr[i]
} RtxFlat;

// Scene (after Scene_Prepare) -> the pools of rtx_scene_desc.  NULL + message on stderr when the scene cannot be
// expressed (no bounded object, more than 256 materials, a Material_Skybox of another skybox).
RtxFlat* Rtx_Flatten(const Scene* scene);
void     Rtx_FlatDelete(RtxFlat* flat);

// Render_Start (src/rt/renderer.c:265-269) with the frame rendered by librt_hip.so: same asynchronous contract —
// returns at once, a waiter thread fills ctx->img->pix and then sets ctx->finished, Render_Done / Render_Delete unchanged.
void Render_Start_HIP(RenderCtx* ctx, size_t samples_per_pixel, size_t max_ray_depth);
