// rtx_device.h — device-side (gfx950) building blocks of the ray→scene query path.
//
// Everything here is compiled with -ffp-contract=off: the reference's hit decisions depend on each
// binary32 rounding, so every expression below keeps the operation order of the
// reference function it cites and is never contracted into an FMA.  Divisions and square roots
// are the correctly rounded forms (hipcc default).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtx.h"

#define RT_EPSILON 0.0001f // src/common/config.h:15
#define PI32 3.141592653589793238462643383279502884197169399375105820f // src/math/math.h:7
#define RTX_INF (__builtin_inff())
#define RTX_STACK 64 // >= desc.max_depth (checked at rtx_scene_create)
#ifndef TRI_BYTES
#define TRI_BYTES 36 // bytes per hot triangle record, a multiple of 4, 36 used.  Packed beats one 64-byte line per
                     // triangle: 108.0 / 107.2 / 106.6 / 106.6 ms per bench frame for 64 / 48 / 40 / 36 bytes
#endif
// a record is read as float4 [0], [1] and the first float of [2] (some kernels read all of [2]: the pool is padded);
// the address is only 4-byte aligned when TRI_BYTES is no multiple of 16, which global loads allow
#define TRI_PTR(S, r) ((const float4*)((const char*)(S).tri + (uint64_t)TRI_BYTES * (r)))
#ifndef TRI_WAIT
#define TRI_WAIT 0 // 1: take a record's first 16 bytes, wait, read the rest from L1 (see the leaf blocks)
#endif

#ifndef NODE_WORDS
#define NODE_WORDS 10 // dwords per record of the culled kernel's node pool
#endif

#define DEV __device__ __forceinline__

// hit reference produced by the closest-hit kernel
#define HIT_MISS 0xFFFFFFFFu
#define HIT_SPHERE 0x80000000u
#define HIT_PLANE 0xC0000000u
#define HIT_KIND(ref) ((ref) >> 30) // 0,1 triangle; 2 sphere; 3 plane (or miss)

// ---------------------------------------------------------------------------------------------
// device-resident scene (built by rtx_scene_create from rtx_scene_desc)
// ---------------------------------------------------------------------------------------------
// A texture on the device.  F32: 3 floats per texel in tex_data, as handed over.  PAL8 (the compact texture
// format): a texture whose channels take at most 256 distinct binary32 values — every imported BMP does: its texels are
// sRGBToLinear(byte / 255) (src/gfx/texture.c:15-47, src/gfx/color.c:7-16) — is kept as 3 bytes per texel plus that palette.
// The palette holds the original bit patterns, so a lookup returns exactly what the f32 texture would: 4x smaller and lossless.
enum { TEX_F32 = 0, TEX_PAL8 = 1 };
struct TexDev {
    uint32_t width, height;
    uint64_t offset; // F32: first texel in tex_data; PAL8: first byte in tex_bytes
    uint32_t fmt, pal; // PAL8: first entry of the texture's palette in tex_pal
};

struct SceneDev {
    const uint2*    nodes;    // {w0 = split bits | len, w1 = type<<30 | index}
    // culled kernel (k_trace_flat<.., CULL>): NODE_WORDS dwords per node — the node, the conservative boxes of BOTH children
    // (a child the ray cannot reach is never fetched) and, where the node's parent is a cut-off node folded into this
    // node's step, that parent's plane (rt_hip.hip, build_merge_plan):
    //   [0..3] = {w0, w1, L.x, L.y}   [4..7] = {L.z, R.x, R.y, R.z}   [8] = right link : 29 | plane axis : 2 (3 = none) | side : 1
    //   [9] = plane split.   L / R = left (link in w1) / right child; links skip the folded cut-off nodes.
    // a box is three words, one per axis, max << 16 | min, on the 16-bit grid  coordinate = cull_lo + q * cull_scale;
    // min.x = 0xFFFF with max.x = 0: never cull.  root_q = the root's box.
    const uint32_t* nodes40;
    uint32_t        root_q[3];
    float           cull_c[3], cull_ro2; // culling is applied to rays whose origin lies within sqrt(ro2) of c
    float           cull_kt;             // a subtree's tMax is stretched by this factor (>= 1) in the box test
    float           cull_lo[3], cull_scale[3];
    float           bin_inv[3]; // ray binning: 32 cells per axis over the box grid, cell = (o - cull_lo) * bin_inv
    const uint32_t* leaf_ref; // pool bit | pool index
    // hot triangle pool, TRI_BYTES bytes per triangle:
    //   [0] = v0.x v0.y v0.z e1.x   [1] = e1.y e1.z e2.x e2.y   [2] = e2.z 0 0 0
    // e1 = v1-v0, e2 = v2-v0: the two subtractions Triangle_HitAt redoes on every call
    // (src/rt/surfaces.c:137-138), hoisted to upload time; the rounded results are identical.
    const float4* tri;
    // cold shading pool, 64 B per triangle:
    //   [0] = n0.x n0.y n0.z n1.x  [1] = n1.y n1.z n2.x n2.y  [2] = n2.z uv0.u uv0.v uv1.u
    //   [3] = uv1.v uv2.u uv2.v <material index as uint bits>
    const float4*   tri_shade;
    const uint32_t* tri_obj;
    const float4*   sph; // c.x c.y c.z r
    const uint32_t* sph_mat;
    const uint32_t* sph_obj;
    const float*    plane; // 8 floats per plane: n.xyz p.xyz mat(bits) 0
    const rtx_material*     mats;
    const TexDev*           tex;
    const uint8_t*          tex_bytes; // palettised textures: 3 palette indices per texel
    const float*            tex_pal;   // their palettes, 256 entries each
    const float*            tex_data;
    uint32_t                n_nodes, n_refs, n_tris, n_spheres, n_planes, n_materials, n_textures;
    uint32_t                root;         // culled kernel: where a link to the root leads (the root itself unless it is a folded cut-off node)
    uint32_t                root_literal; // literal walk: desc.root_index
    int32_t                 skybox_tex[6];
    // Faces whose texture is a single texel (Texture_Import_Color, src/gfx/texture.c:49-70: the constant skies of BASELINE
    // configs 2, 3 and 5): Texture_ColorAt returns that texel whatever u and v are — x = (size_t)(u * 0) — so the lookup
    // needs neither the two quotients nor the three dependent loads.  sky_1x1: bit f set = face f is such a face and
    // sky_rgb[3 f ..] holds its texel (the very bits the texture holds); bit 6 = all six are, with the same texel.
    uint32_t                sky_1x1;
    float                   sky_rgb[18];
};

// ---------------------------------------------------------------------------------------------
// vec3 in the reference's operation order (src/math/vec.c:152-356)
// ---------------------------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};

DEV f3    mk3(float x, float y, float z) { return f3{x, y, z}; }
DEV f3    vadd(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }                       // :157
DEV f3    vsub(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }                       // :166
DEV f3    vmuls(f3 v, float s) { return f3{s * v.x, s * v.y, s * v.z}; }                         // :175
DEV f3    vmulc(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }                      // :189
DEV f3    vdivs(f3 v, float s) { return vmuls(v, 1.0f / s); }                                    // :198
DEV f3    vdivc(f3 a, f3 b) { return f3{a.x / b.x, a.y / b.y, a.z / b.z}; }                      // :203
DEV float vdot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                         // :212
DEV f3    vcross(f3 a, f3 b)                                                                      // :217
{
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
DEV f3    vlerp(f3 a, f3 b, float t) { return vadd(a, vmuls(vsub(b, a), t)); }                   // :226
DEV float vmag(f3 v) { return sqrtf(vdot(v, v)); }                                               // :236
DEV f3    vnorm(f3 v) { return vdivs(v, vmag(v)); }                                              // :241
DEV f3    vneg(f3 v) { return vmuls(v, -1.0f); } // vmul(v, -1): exact sign flip
DEV float velem(f3 v, uint32_t a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

DEV float minf_(float x, float y) { return x > y ? y : x; } // src/math/math.c:30
DEV float maxf_(float x, float y) { return x > y ? x : y; } // src/math/math.c:39
DEV float clampf_(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); } // math.c:9

DEV f3 vec3_Reflect(f3 v, f3 n) // vec.c:273
{
    float d = vdot(v, n);
    return vsub(v, vmuls(n, 2.0f * d));
}

DEV f3 vec3_Refract(f3 v, f3 n, float ratio) // vec.c:280
{
    float cosTheta = fminf(vdot(vneg(v), n), 1.0f);
    f3    perp     = vmuls(vadd(v, vmuls(n, cosTheta)), ratio);
    f3    para     = vmuls(n, -sqrtf(fabsf(1.0f - vdot(perp, perp))));
    return vadd(perp, para);
}

struct basis3 {
    f3 x, y, z;
};

DEV basis3 vec3_OrthonormalBasis(f3 bz) // vec.c:319 (Frisvad)
{
    f3 bx, by;
    if (bz.z < -0.9999999f) {
        by = mk3(0.0f, -1.0f, 0.0f);
        bx = mk3(-1.0f, 0.0f, 0.0f);
    } else {
        float a = 1.0f / (1.0f + bz.z);
        float b = -bz.x * bz.y * a;
        by      = mk3(-1.0f + bz.x * bz.x * a, -b, bz.x);
        bx      = mk3(-b, -1.0f + bz.y * bz.y * a, bz.y);
    }
    return basis3{bx, by, bz};
}

DEV basis3 vec3_OrthonormalBasis_Inverse(basis3 b) // vec.c:340 (transpose)
{
    return basis3{mk3(b.x.x, b.y.x, b.z.x), mk3(b.x.y, b.y.y, b.z.y), mk3(b.x.z, b.y.z, b.z.z)};
}

DEV f3 vec3_Reorient(f3 n, basis3 b) // vec.c:308: (bx*n.x + by*n.y) + bz*n.z
{
    return vadd(vadd(vmuls(b.x, n.x), vmuls(b.y, n.y)), vmuls(b.z, n.z));
}

// ---------------------------------------------------------------------------------------------
// xoshiro128+ (src/math/random.c:24-105); state lives in registers, one stream per path
// ---------------------------------------------------------------------------------------------
struct Rng {
    uint32_t s0, s1, s2, s3;
};

DEV uint32_t rng_next(Rng& r) // random.c:31
{
    uint32_t result = r.s0 + r.s3;
    uint32_t t      = r.s1 << 9;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = (r.s3 << 11) | (r.s3 >> 21);
    return result;
}

DEV void rng_seed(Rng& r, uint64_t a, uint64_t b) // Random_Seed + jump, random.c:53-85
{
    r.s0 = (uint32_t)a;
    r.s1 = (uint32_t)(a >> 32);
    r.s2 = (uint32_t)b;
    r.s3 = (uint32_t)(b >> 32);
    const uint32_t JUMP[4] = {0x8764000bu, 0xf542d2d3u, 0x6fa035c3u, 0x77f2db5bu};
    uint32_t       j0 = 0, j1 = 0, j2 = 0, j3 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll 4
        for (int b2 = 0; b2 < 32; b2++) {
            if (JUMP[i] & (1u << b2)) {
                j0 ^= r.s0;
                j1 ^= r.s1;
                j2 ^= r.s2;
                j3 ^= r.s3;
            }
            rng_next(r);
        }
    }
    r.s0 = j0;
    r.s1 = j1;
    r.s2 = j2;
    r.s3 = j3;
}

DEV float Random_Unilateral(Rng& r) // random.c:100: [0,1)
{
    return __uint_as_float((rng_next(r) >> 9) | 0x3f800000u) - 1.0f;
}

DEV float Random_Bilateral(Rng& r) // random.c:119-128: min + (max-min)*u with (-1, 1)
{
    return -1.0f + (1.0f - -1.0f) * Random_Unilateral(r);
}

DEV f3 Random_InSphere(Rng& r, float radius) // random.c:140
{
    float e0 = Random_Unilateral(r);
    float e1 = Random_Unilateral(r);
    float e2 = Random_Bilateral(r);
    float rho = radius * e0;
    float phi = 2.0f * PI32 * e1;
    float ct  = e2;
    float st  = sqrtf(1.0f - ct * ct); // POWF(x,2): the pinned build folds powf(x,2) to x*x
    float sp, cp;
    sincosf(phi, &sp, &cp); // (one argument reduction for both)
    return mk3(rho * st * cp, rho * st * sp, rho * ct);
}

DEV f3 Random_OnSphere(Rng& r, float radius) // random.c:159
{
    float e1  = Random_Unilateral(r);
    float e2  = Random_Bilateral(r);
    float rho = radius;
    float phi = 2.0f * PI32 * e1;
    float ct  = e2;
    float st  = sqrtf(1.0f - ct * ct);
    float sp, cp;
    sincosf(phi, &sp, &cp);
    return mk3(rho * st * cp, rho * st * sp, rho * ct);
}

DEV f3 Random_InHemisphere(Rng& r, f3 facing, float radius) // random.c:176 (radius applied twice)
{
    f3 v = Random_InSphere(r, radius);
    if (vdot(v, facing) < 0.0f) v = vmuls(v, -1.0f);
    return vmuls(v, radius);
}

DEV void Random_InDisc(Rng& r, float radius, float& x, float& y) // random.c:200
{
    float e0 = Random_Unilateral(r);
    float e1 = Random_Bilateral(r);
    float rr = radius * e0;
    float ct = e1;
    float st = e0 * sqrtf(1.0f - e1 * e1);
    x        = rr * ct;
    y        = rr * st;
}

// ---------------------------------------------------------------------------------------------
// rays (src/rt/ray.c)
// ---------------------------------------------------------------------------------------------
struct RayD {
    f3 o, d;      // origin, unit direction
    f3 inv, odd;  // cache.invDir, cache.originDivDir (ray.c:12-13): true divisions
};

DEV RayD ray_from_unit(f3 o, f3 d)
{
    RayD r;
    r.o   = o;
    r.d   = d;
    r.inv = vdivc(mk3(1.0f, 1.0f, 1.0f), d);
    r.odd = vdivc(o, d);
    return r;
}

DEV f3 Ray_At(f3 o, f3 d, float t) { return vadd(o, vmuls(d, t)); } // ray.c:18

// ---------------------------------------------------------------------------------------------
// primitive tests (src/rt/surfaces.c).  Only t (and u,v for triangles) is produced here; the full
// HitInfo is rebuilt for the one accepted hit by make_hitinfo().
// ---------------------------------------------------------------------------------------------
DEV bool Triangle_HitAt(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float tMin, float tMax, float& t, float& u, float& v) // :135
{
    f3    h = vcross(d, e2);
    float a = vdot(e1, h);
    if (fabsf(a) < RT_EPSILON) return false;
    f3 s = vsub(o, v0);
    u    = vdot(s, h) / a;
    if (u < 0.0f || u > 1.0f) return false;
    f3 q = vcross(s, e1);
    v    = vdot(d, q) / a;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = vdot(e2, q) / a;
    if (t > tMax || t < tMin) return false;
    return true;
}

// Triangle_HitAt without its early returns: the same operations in the same order, the four rejections of surfaces.c:146-165
// as one predicate (a NaN falls through each comparison the same way).  For the cooperative leaf block, where 50-odd lanes
// test different triangles in step: some lane nearly always gets past every rejection, so the early returns skip nothing
// there — they only cost the scalar instructions and branches that nest the execution masks.
DEV bool Triangle_HitAt_straight(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float tMin, float tMax, float& t, float& u, float& v)
{
    const f3    h = vcross(d, e2);
    const float a = vdot(e1, h);
    const f3    s = vsub(o, v0);
    u             = vdot(s, h) / a;
    const f3 q    = vcross(s, e1);
    v             = vdot(d, q) / a;
    t             = vdot(e2, q) / a;
    return !(fabsf(a) < RT_EPSILON) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && !(t > tMax || t < tMin);
}

// Triangle_HitAt for two triangles A and B against the same ray, component 0 = A, component 1 = B.  Every
// arithmetic operation is the one of the scalar routine above, in the same order, on a 2-vector: gfx950 issues
// v_pk_mul_f32 / v_pk_add_f32 (two IEEE binary32 results per lane and instruction, no fusion with
// -ffp-contract=off), so each component is bit-identical to a scalar call while the closest-hit kernel, which is
// VALU-issue bound, pays one instruction for both.  The three divisions stay scalar (IEEE-correct sequences).
typedef float f2v __attribute__((ext_vector_type(2)));
struct f3x2 {
    f2v x, y, z;
};
DEV f3x2 cross2(f3x2 a, f3x2 b) { return f3x2{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV f2v  dot2(f3x2 a, f3x2 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV f3x2 bcast2(f3 v) { return f3x2{f2v{v.x, v.x}, f2v{v.y, v.y}, f2v{v.z, v.z}}; }

DEV void Triangle_HitAt2(f3x2 v0, f3x2 e1, f3x2 e2, f3 o, f3 d, float tMin, float tMax, bool hit[2], float t[2], float u[2], float v[2])
{
    const f3x2 D = bcast2(d), O = bcast2(o);
    const f3x2 h = cross2(D, e2);
    const f2v  a = dot2(e1, h);
    const f3x2 s = f3x2{O.x - v0.x, O.y - v0.y, O.z - v0.z};
    const f2v  nu = dot2(s, h);
    const f3x2 q  = cross2(s, e1);
    const f2v  nv = dot2(D, q);
    const f2v  nt = dot2(e2, q);
#pragma unroll
    for (int k = 0; k < 2; k++) { // the rejections of surfaces.c:146-165 as one predicate (NaNs fall through the same way)
        const float uu = nu[k] / a[k], vv = nv[k] / a[k], tt = nt[k] / a[k];
        hit[k] = !(fabsf(a[k]) < RT_EPSILON) && !(uu < 0.0f || uu > 1.0f) && !(vv < 0.0f || uu + vv > 1.0f) && !(tt > tMax || tt < tMin);
        t[k] = tt, u[k] = uu, v[k] = vv;
    }
}

// Conservative box tests of the culled walk (not part of the reference).  Does the segment
// {o + t d, 0 <= t <= kt * tMax} miss the box (wx, wy, wz)?  Each word holds one axis, max << 16 | min, on the
// 16-bit grid  x(q) = cull_lo + q * cull_scale.  A plane's ray parameter is  fma(float(q), scale, -(o - lo)) * inv:
// one conversion (the 16-bit half is picked by the conversion's operand selector, no shift or mask), one fused
// multiply-add that decodes the plane AND subtracts the origin, one multiplication.  `inv` is 1/d with infinities
// replaced by +-3e38.  The two-box form tests both children of a node at once with packed math: component 0 = left
// child, 1 = right child (oracle/oracle.cpp replays the same operations).
#define BOX_LO(w) ((float)((w) & 0xFFFFu))
#define BOX_HI(w) ((float)((w) >> 16))
DEV bool box_missed(const SceneDev& S, uint32_t wx, uint32_t wy, uint32_t wz, f3 o, f3 inv, float tMax)
{
    const float nx = -(o.x - S.cull_lo[0]), ny = -(o.y - S.cull_lo[1]), nz = -(o.z - S.cull_lo[2]);
    const float t1x = fmaf(BOX_LO(wx), S.cull_scale[0], nx) * inv.x, t2x = fmaf(BOX_HI(wx), S.cull_scale[0], nx) * inv.x;
    const float t1y = fmaf(BOX_LO(wy), S.cull_scale[1], ny) * inv.y, t2y = fmaf(BOX_HI(wy), S.cull_scale[1], ny) * inv.y;
    const float t1z = fmaf(BOX_LO(wz), S.cull_scale[2], nz) * inv.z, t2z = fmaf(BOX_HI(wz), S.cull_scale[2], nz) * inv.z;
    const float tn  = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fmaxf(fminf(t1z, t2z), 0.0f));
    const float tf  = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fminf(fmaxf(t1z, t2z), tMax * S.cull_kt));
    return wx != 0x0000FFFFu && tn > tf;
}

// The same box as a parameter range: [tn, tf] = the part of {t >= 0, t <= kt * tMax} the ray spends inside the box (empty:
// tn > tf, box_missed's verdict); false for a box marked "never cull".  Same operations as box_missed.  For the ray-range
// tasks of k_trace_long: every operation is monotone in the box's planes, so a box nested in another on the grid (a child
// in its parent) has a range nested in the other's.
DEV bool box_range(const SceneDev& S, uint32_t wx, uint32_t wy, uint32_t wz, f3 o, f3 inv, float tMax, float& tn, float& tf)
{
    const float nx = -(o.x - S.cull_lo[0]), ny = -(o.y - S.cull_lo[1]), nz = -(o.z - S.cull_lo[2]);
    const float t1x = fmaf(BOX_LO(wx), S.cull_scale[0], nx) * inv.x, t2x = fmaf(BOX_HI(wx), S.cull_scale[0], nx) * inv.x;
    const float t1y = fmaf(BOX_LO(wy), S.cull_scale[1], ny) * inv.y, t2y = fmaf(BOX_HI(wy), S.cull_scale[1], ny) * inv.y;
    const float t1z = fmaf(BOX_LO(wz), S.cull_scale[2], nz) * inv.z, t2z = fmaf(BOX_HI(wz), S.cull_scale[2], nz) * inv.z;
    tn = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fmaxf(fminf(t1z, t2z), 0.0f));
    tf = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fminf(fmaxf(t1z, t2z), tMax * S.cull_kt));
    return wx != 0x0000FFFFu;
}

// (The walk's form: `rot` = 16 on an axis where the direction is negative, else 0.  Rotating a word by it puts the plane the
// ray meets first into the low half and the other into the high half — with min <= max on the grid and a monotone
// decode that IS min / max of the two ray parameters, so the twelve v_min / v_max of the slab test become six rotations
// and the outcome is the one box_missed computes.)
struct u3 {
    uint32_t x, y, z;
};
DEV void box_missed2(const SceneDev& S, uint32_t lx, uint32_t ly, uint32_t lz, uint32_t rx, uint32_t ry, uint32_t rz, f3 o, f3 inv, u3 rot, f2v tMax,
                     bool miss[2])
{
    const f2v sx = f2v{S.cull_scale[0], S.cull_scale[0]}, sy = f2v{S.cull_scale[1], S.cull_scale[1]}, sz = f2v{S.cull_scale[2], S.cull_scale[2]};
    const float nx1 = -(o.x - S.cull_lo[0]), ny1 = -(o.y - S.cull_lo[1]), nz1 = -(o.z - S.cull_lo[2]);
    const f2v nx = f2v{nx1, nx1}, ny = f2v{ny1, ny1}, nz = f2v{nz1, nz1};
    const f2v ix = f2v{inv.x, inv.x}, iy = f2v{inv.y, inv.y}, iz = f2v{inv.z, inv.z};
    const uint32_t Lx = __builtin_amdgcn_alignbit(lx, lx, rot.x), Rx = __builtin_amdgcn_alignbit(rx, rx, rot.x);
    const uint32_t Ly = __builtin_amdgcn_alignbit(ly, ly, rot.y), Ry = __builtin_amdgcn_alignbit(ry, ry, rot.y);
    const uint32_t Lz = __builtin_amdgcn_alignbit(lz, lz, rot.z), Rz = __builtin_amdgcn_alignbit(rz, rz, rot.z);
    const f2v t1x = __builtin_elementwise_fma(f2v{BOX_LO(Lx), BOX_LO(Rx)}, sx, nx) * ix, t2x = __builtin_elementwise_fma(f2v{BOX_HI(Lx), BOX_HI(Rx)}, sx, nx) * ix;
    const f2v t1y = __builtin_elementwise_fma(f2v{BOX_LO(Ly), BOX_LO(Ry)}, sy, ny) * iy, t2y = __builtin_elementwise_fma(f2v{BOX_HI(Ly), BOX_HI(Ry)}, sy, ny) * iy;
    const f2v t1z = __builtin_elementwise_fma(f2v{BOX_LO(Lz), BOX_LO(Rz)}, sz, nz) * iz, t2z = __builtin_elementwise_fma(f2v{BOX_HI(Lz), BOX_HI(Rz)}, sz, nz) * iz;
    const f2v tk  = tMax * f2v{S.cull_kt, S.cull_kt};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float tn = fmaxf(fmaxf(t1x[k], t1y[k]), fmaxf(t1z[k], 0.0f));
        const float tf = fminf(fminf(t2x[k], t2y[k]), fminf(t2z[k], tk[k]));
        miss[k]        = (k == 0 ? lx : rx) != 0x0000FFFFu && tn > tf;
    }
}
DEV u3 box_rotations(f3 inv) { return u3{(__float_as_uint(inv.x) >> 27) & 16u, (__float_as_uint(inv.y) >> 27) & 16u, (__float_as_uint(inv.z) >> 27) & 16u}; }

DEV bool Sphere_HitAt(float4 sp, f3 o, f3 d, float tMin, float tMax, float& t) // :55
{
    f3    c     = mk3(sp.x, sp.y, sp.z);
    f3    dist  = vsub(o, c);
    float polyA = vdot(d, d);
    float halfB = vdot(dist, d);
    float polyC = vdot(dist, dist) - sp.w * sp.w;
    float disc  = halfB * halfB - polyA * polyC;
    if (disc < 0.0f) return false;
    float root1 = (-halfB - sqrtf(disc)) / polyA;
    float root2 = (-halfB + sqrtf(disc)) / polyA;
    if (tMin < root1 && root1 < tMax) {
        t = root1;
    } else if (tMin < root2 && root2 < tMax) {
        t = root2;
    } else {
        return false;
    }
    return true;
}

DEV bool Plane_HitAt(f3 n, f3 p, f3 o, f3 d, float tMin, float tMax, float& t) // :223
{
    float num = vdot(n, vsub(p, o));
    float den = vdot(n, d);
    t         = num / den;
    if (t < tMin || t > tMax) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------
// closest hit: KDTree_HitAt (src/rt/accelerators/kdtree.c:453-577) made iterative, then the
// unbounded list (Scene_ClosestHit, src/world/scene.c:100-143).
//
// The reference recursion visits the origin-side child with tMaxNew = minf(tMax, tI) and, only if
// that subtree reports no hit AND tMaxNew == tI, the opposite child with the OUTER tMax
// (kdtree.c:548-568).  The pending "opposite child + outer tMax" pairs are exactly what the stack
// holds; a leaf that reports a hit ends the walk (every caller returns true immediately).
// ---------------------------------------------------------------------------------------------
struct Visits {
    uint32_t inner, leaf, prim;
    uint32_t culled; // nodes fetched and skipped by the box test (culled kernel only)
};

template <bool COUNT>
DEV void closest_hit(const SceneDev& S, const RayD& ray, uint32_t& ref, float& tHit, float& uHit, float& vHit, Visits& vis)
{
    uint2    stk[RTX_STACK];
    int      sp   = 0;
    uint32_t node = S.root_literal;
    float    tMax = RTX_INF;

    ref  = HIT_MISS;
    tHit = RTX_INF;
    uHit = 0.0f;
    vHit = 0.0f;

    for (;;) {
        const uint2    nd   = S.nodes[node];
        const uint32_t type = nd.y >> 30;
        const uint32_t idx  = nd.y & 0x3FFFFFFFu;
        if (type == 3u) { // CheckHitLeafNode, kdtree.c:475
            if (COUNT) vis.leaf++;
            const uint32_t len   = nd.x;
            float          bestT = RTX_INF;
            uint32_t       best  = HIT_MISS;
            float          bu = 0.0f, bv = 0.0f;
            for (uint32_t i = 0; i < len; i++) {
                const uint32_t r = S.leaf_ref[idx + i];
                if (COUNT) vis.prim++;
                float t, u = 0.0f, v = 0.0f;
                bool  hit;
                if (r & RTX_REF_SPHERE) {
                    hit = Sphere_HitAt(S.sph[r & 0x7FFFFFFFu], ray.o, ray.d, RT_EPSILON, tMax, t);
                } else {
                    const float4* tp = TRI_PTR(S, r);
                    const float4  a = tp[0], b = tp[1], c = tp[2];
                    hit = Triangle_HitAt(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), ray.o, ray.d, RT_EPSILON,
                                         tMax, t, u, v);
                }
                if (hit && t < bestT) { // strict: first in list wins ties (kdtree.c:485)
                    bestT = t;
                    best  = r;
                    bu    = u;
                    bv    = v;
                }
            }
            if (best != HIT_MISS) {
                ref  = best;
                tHit = bestT;
                uHit = bu;
                vHit = bv;
                break;
            }
            if (sp == 0) break;
            const uint2 e = stk[--sp];
            node          = e.x;
            tMax          = __uint_as_float(e.y);
            continue;
        }

        // CheckHitInternalNode, kdtree.c:501
        if (COUNT) vis.inner++;
        const float    split = __uint_as_float(nd.x);
        const uint32_t left  = idx;
        const uint32_t right = node + 1;
        const float    dA    = velem(ray.d, type);
        const float    oA    = velem(ray.o, type);

        if (fabsf(dA) < RT_EPSILON) { // :509 parallel to the plane
            node = (oA >= split) ? right : left;
            continue;
        }
        const float tI      = split * velem(ray.inv, type) - velem(ray.odd, type); // :518 (mul, then sub)
        const float tMaxNew = minf_(tMax, tI);
        if (tI < RT_EPSILON) { // :521 plane behind (or at) the origin: one side only, tMax unchanged
            if (oA > split) {
                node = right;
            } else if (oA < split) {
                node = left;
            } else {
                const float rayAt = oA + 100.0f * dA; // Ray_At(ray, 100)[axis], :529
                node              = (rayAt >= split) ? right : left;
            }
            continue;
        }
        const bool     originRight = oA >= split; // :544
        const uint32_t nearN       = originRight ? right : left;
        const uint32_t farN        = originRight ? left : right;
        if (tMaxNew == tI) { // :551/:563 — the far side is reachable; remember it with the OUTER tMax
            stk[sp++] = make_uint2(farN, __float_as_uint(tMax));
        }
        node = nearN;
        tMax = tMaxNew;
    }

    // Scene_ClosestHitInArray over the unbounded objects, tMin = eps, tMax = INF (scene.c:73-98),
    // then the merge of scene.c:122-142: the unbounded hit wins only if strictly closer.
    float    pBest = RTX_INF;
    uint32_t pRef  = HIT_MISS;
    for (uint32_t i = 0; i < S.n_planes; i++) {
        const float* pl = S.plane + 8ull * i;
        float        t;
        if (Plane_HitAt(mk3(pl[0], pl[1], pl[2]), mk3(pl[3], pl[4], pl[5]), ray.o, ray.d, RT_EPSILON, RTX_INF, t) && t < pBest) {
            pBest = t;
            pRef  = HIT_PLANE | i;
        }
    }
    if (pRef != HIT_MISS) {
        if (ref == HIT_MISS || pBest < tHit) {
            ref  = pRef;
            tHit = pBest;
            uHit = 0.0f;
            vHit = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// HitInfo of the accepted hit (src/rt/ray.h:15-21), rebuilt with the reference's arithmetic
// ---------------------------------------------------------------------------------------------
struct HitD {
    f3       pos, n; // position, unitNormal (not unit length for triangles — reproduced, not fixed)
    float    u, v;   // texture coordinates
    bool     front;
    uint32_t mat;
};

DEV void set_face_normal(HitD& h, f3 d, f3 outward) // HitInfo_SetFaceNormal, ray.c:23
{
    h.front = vdot(d, outward) < 0;
    h.n     = h.front ? outward : vmuls(outward, -1.0f);
}

DEV HitD make_hitinfo(const SceneDev& S, f3 o, f3 d, uint32_t ref, float t, float u, float v)
{
    HitD h;
    h.pos = Ray_At(o, d, t);
    const uint32_t kind = HIT_KIND(ref);
    if (kind < 2u) { // triangle, surfaces.c:167-173
        const float4* sp = S.tri_shade + 4ull * ref;
        const float4  a = sp[0], b = sp[1], c = sp[2], e = sp[3];
        const float   coeff = 1.0f - u - v;
        // uv = (tex0*coeff + tex1*u) + tex2*v ; normal likewise (vsum of three = (a+b)+c)
        h.u = (coeff * c.y + u * c.w) + v * e.y;
        h.v = (coeff * c.z + u * e.x) + v * e.z;
        f3 n0 = mk3(a.x, a.y, a.z), n1 = mk3(a.w, b.x, b.y), n2 = mk3(b.z, b.w, c.x);
        f3 outward = vadd(vadd(vmuls(n0, coeff), vmuls(n1, u)), vmuls(n2, v));
        set_face_normal(h, d, outward);
        h.mat = __float_as_uint(e.w);
    } else if (kind == 2u) { // sphere, surfaces.c:79-83
        const uint32_t i  = ref & 0x3FFFFFFFu;
        const float4   sp = S.sph[i];
        f3             outward = vdivs(vsub(h.pos, mk3(sp.x, sp.y, sp.z)), sp.w);
        set_face_normal(h, d, outward);
        // Sphere_MapUV(outwardNormal), surfaces.c:44-53
        float theta = acosf(clampf_(-outward.y, -0.999f, 0.999f));
        float phi   = atan2f(-outward.z, outward.x) + PI32;
        h.u         = phi / (2 * PI32);
        h.v         = theta / PI32;
        h.mat       = S.sph_mat[i];
    } else { // plane, surfaces.c:233-237; uv is uninitialised in the reference — defined as (0,0) here
        const uint32_t i  = ref & 0x3FFFFFFFu;
        const float*   pl = S.plane + 8ull * i;
        set_face_normal(h, d, mk3(pl[0], pl[1], pl[2]));
        h.u   = 0.0f;
        h.v   = 0.0f;
        h.mat = __float_as_uint(pl[6]);
    }
    return h;
}

// prim_id convention of rtx.h
DEV uint32_t hit_prim_id(const SceneDev& S, uint32_t ref)
{
    if (ref == HIT_MISS) return RTX_ID_MISS;
    const uint32_t kind = HIT_KIND(ref);
    if (kind < 2u) return S.tri_obj[ref];
    if (kind == 2u) return S.sph_obj[ref & 0x3FFFFFFFu];
    return RTX_ID_UNBOUNDED | (ref & 0x3FFFFFFFu);
}

// ---------------------------------------------------------------------------------------------
// textures / skybox (src/gfx/texture.c:88-95, src/world/skybox.c:66-134)
// ---------------------------------------------------------------------------------------------
DEV f3 Texture_ColorAt(const SceneDev& S, int32_t tex, float u, float v)
{
    if (tex < 0) return mk3(0.0f, 0.0f, 0.0f);
    const TexDev td = S.tex[tex];
    // (size_t)(st.x * (width-1)); the reference has no clamp (UB outside [0,1]) — clamp here
    float    fx = u * (float)(td.width - 1);
    float    fy = (1.0f - v) * (float)(td.height - 1);
    uint32_t xx = fx > 0.0f ? (uint32_t)fminf(fx, (float)(td.width - 1)) : 0u;
    uint32_t yy = fy > 0.0f ? (uint32_t)fminf(fy, (float)(td.height - 1)) : 0u;
    const uint64_t texel = (uint64_t)yy * td.width + xx;
    if (td.fmt == TEX_PAL8) {
        const uint8_t* b   = S.tex_bytes + td.offset + 3ull * texel;
        const float*   pal = S.tex_pal + td.pal;
        return mk3(pal[b[0]], pal[b[1]], pal[b[2]]);
    }
    const float* p = S.tex_data + 3ull * (td.offset + texel);
    return mk3(p[0], p[1], p[2]);
}

DEV f3 Skybox_ColorAt(const SceneDev& S, f3 dir)
{
    if (S.sky_1x1 & 64u) return mk3(S.sky_rgb[0], S.sky_rgb[1], S.sky_rgb[2]); // one colour all round: any face, any (u, v), NaNs included
    float absX = fabsf(dir.x), absY = fabsf(dir.y), absZ = fabsf(dir.z);
    bool  xp = dir.x > 0, yp = dir.y > 0, zp = dir.z > 0;
    float maxAxis, uc, vc;
    int   index;
    if (xp && absX >= absY && absX >= absZ) {
        maxAxis = absX, uc = dir.y, vc = dir.z, index = 0;
    } else if (!xp && absX >= absY && absX >= absZ) {
        maxAxis = absX, uc = -dir.y, vc = dir.z, index = 1;
    } else if (yp && absY >= absX && absY >= absZ) {
        maxAxis = absY, uc = -dir.x, vc = dir.z, index = 2;
    } else if (!yp && absY >= absX && absY >= absZ) {
        maxAxis = absY, uc = dir.x, vc = dir.z, index = 3;
    } else if (zp && absZ >= absX && absZ >= absY) {
        maxAxis = absZ, uc = -dir.x, vc = -dir.y, index = 4;
    } else {
        maxAxis = absZ, uc = -dir.x, vc = dir.y, index = 5;
    }
    if ((S.sky_1x1 >> index) & 1u) return mk3(S.sky_rgb[3 * index], S.sky_rgb[3 * index + 1], S.sky_rgb[3 * index + 2]);
    float u = 0.5f * (uc / maxAxis + 1.0f);
    float v = 0.5f * (vc / maxAxis + 1.0f);
    return Texture_ColorAt(S, S.skybox_tex[index], u, v);
}

// ---------------------------------------------------------------------------------------------
// materials (src/rt/materials.c).  Each returns the reference's bool ("continue the path") and
// fills surface colour, emitted colour and the un-normalised bounce direction (Ray_Make is applied
// by the caller, as every Bounce function ends with Ray_Make(hit->position, dir)).
// p[] slots: 0 subsurface 1 specular 2 roughness 3 specular_tint 4 anistropic 5 sheen_tint
//            6 clearcoat_gloss 7 eta 8 w.clearcoat 9 w.sheen 10 w.metallic 11 w.specular
// ---------------------------------------------------------------------------------------------
DEV float pow5(float x) { return powf(x, 5.0f); } // POWF(x,5) is a real powf call (math.h:28-32)

DEV f3 diffuse_scatter(Rng& rng, const HitD& h) // materials.c:37-42 / :187-192
{
    f3 target    = vadd(h.pos, Random_InHemisphere(rng, h.n, 1.0f));
    f3 scattered = vsub(target, h.pos);
    // vequ(scattered, 0.0f): equalf(|s|^2, 0) is true only for an exactly zero squared length
    if (vdot(scattered, scattered) == 0.0f) scattered = h.n;
    return scattered;
}

DEV f3 HalfVector(f3 a, f3 b) { return vnorm(vadd(a, b)); } // :217

DEV f3 CosWeightedHemisphere_Sample(Rng& rng, f3 n) // :274
{
    float e0 = Random_Unilateral(rng);
    float e1 = Random_Unilateral(rng);
    float st = sqrtf(1.0f - e0);
    float ct = sqrtf(e0);
    float phi = 2.0f * PI32 * e1;
    float sp, cp;
    sincosf(phi, &sp, &cp);
    f3    local = mk3(cp * st, sp * st, ct);
    return vec3_Reorient(local, vec3_OrthonormalBasis(n));
}

DEV f3 BRDF_Diffuse(f3 base, f3 w_in, f3 w_out, f3 n, float roughness, float subsurface) // :266
{
    // BRDF_Disney_BaseDiffuse :232
    f3    hv    = HalfVector(w_in, w_out);
    float hdo   = vdot(hv, w_out);
    float f_d90 = 0.5f + 2.0f * roughness * (hdo * hdo);
    float fd_in  = 1.0f + (f_d90 - 1.0f) * pow5(1.0f - fabsf(vdot(n, w_in)));
    float fd_out = 1.0f + (f_d90 - 1.0f) * pow5(1.0f - fabsf(vdot(n, w_out)));
    float att    = (1.0f / PI32) * fd_in * fd_out;
    f3    base_diffuse = vmuls(base, att);
    // BRDF_Disney_Subsurface :250
    f3    hv2    = HalfVector(w_in, w_out);
    float hdo2   = vdot(hv2, w_out);
    float f_ss90 = roughness * (hdo2 * hdo2);
    float a_in   = fabsf(vdot(n, w_in));
    float a_out  = fabsf(vdot(n, w_out));
    float fss_in  = 1.0f + (f_ss90 - 1.0f) * pow5(1.0f - fabsf(vdot(n, w_in)));
    float fss_out = 1.0f + (f_ss90 - 1.0f) * pow5(1.0f - fabsf(vdot(n, w_out)));
    float att2    = (1.25f / PI32) * (fss_in * fss_out * (1.0f / (a_in + a_out) - 0.5f) + 0.5f);
    f3    sub     = vmuls(base, att2);
    return vlerp(base_diffuse, sub, subsurface);
}

DEV float Lambda_m(f3 w, float ax, float ay) // :352
{
    float wx = w.x * ax, wy = w.y * ay;
    float num = sqrtf(1.0f + (wx * wx + wy * wy) / (w.z * w.z)) - 1.0f;
    return num / 2.0f;
}

DEV float G_m1(f3 w, float ax, float ay) { return 1.0f / (1.0f + Lambda_m(w, ax, ay)); } // :360

DEV f3 GGXVNDF_Sample(f3 Ve, float ax, float ay, float U1, float U2) // :415
{
    f3    Vh    = vnorm(mk3(ax * Ve.x, ay * Ve.y, Ve.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    f3    T1    = lensq > 0.0f ? vmuls(mk3(-Vh.y, Vh.x, 0.0f), 1.0f / sqrtf(lensq)) : mk3(1.0f, 0.0f, 0.0f);
    f3    T2    = vcross(Vh, T1);
    float r     = sqrtf(U1);
    float phi   = 2.0f * PI32 * U2;
    float sp, cp;
    sincosf(phi, &sp, &cp);
    float t1    = r * cp;
    float t2    = r * sp;
    float s     = 0.5f * (1.0f + Vh.z);
    t2          = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    f3 Nh = vadd(vadd(vmuls(T1, t1), vmuls(T2, t2)), vmuls(Vh, sqrtf(maxf_(0.0f, 1.0f - t1 * t1 - t2 * t2))));
    return vnorm(mk3(ax * Nh.x, ay * Nh.y, maxf_(0.0f, Nh.z)));
}

DEV void disney_alpha(const rtx_material& m, float& ax, float& ay) // :456-459 / :732-735
{
    float aspect = sqrtf(1.0f - 0.9f * m.p[4]);
    float r2     = m.p[2] * m.p[2];
    ax           = maxf_(0.0001f, r2 / aspect);
    ay           = maxf_(0.0001f, r2 * aspect);
}

DEV float Color_Luminance(f3 c) { return vmag(c) / vmag(mk3(1.0f, 1.0f, 1.0f)); } // color.c:105
DEV f3    C_tint(f3 base) { return vdivs(base, Color_Luminance(base)); }           // materials.c:385

DEV bool Disney_Diffuse_Bounce(const SceneDev& S, const rtx_material& m, f3 din, const HitD& h, Rng& rng, f3& surf,
                               f3& emit, f3& dout) // :291
{
    f3 out    = CosWeightedHemisphere_Sample(rng, h.n);
    f3 albedo = Texture_ColorAt(S, m.tex, h.u, h.v);
    f3 w_in   = vmuls(din, -1.0f);
    f3 brdf   = BRDF_Diffuse(albedo, w_in, out, h.n, m.p[2], m.p[0]);
    surf      = vmuls(brdf, PI32);
    emit      = mk3(0.0f, 0.0f, 0.0f);
    dout      = out;
    return true;
}

DEV bool Disney_Sheen_Bounce(const SceneDev& S, const rtx_material& m, f3 din, const HitD& h, Rng& rng, f3& surf, f3& emit,
                             f3& dout) // :805
{
    f3 w_in  = vmuls(din, -1.0f);
    f3 w_out = CosWeightedHemisphere_Sample(rng, h.n);
    f3 mn    = HalfVector(w_in, w_out);
    f3 albedo = Texture_ColorAt(S, m.tex, h.u, h.v);
    f3 csheen = vlerp(mk3(1.0f, 1.0f, 1.0f), C_tint(albedo), m.p[5]); // C_sheen :794
    f3 brdf   = vmuls(csheen, pow5(1.0f - fabsf(vdot(mn, w_out))));
    surf      = vmuls(brdf, PI32);
    emit      = mk3(0.0f, 0.0f, 0.0f);
    dout      = w_out;
    return true;
}

DEV bool Disney_Metal_Bounce(const SceneDev& S, const rtx_material& m, f3 din, const HitD& h, Rng& rng, f3& surf, f3& emit,
                             f3& dout) // :446
{
    float ax, ay;
    disney_alpha(m, ax, ay);
    basis3 n2w = vec3_OrthonormalBasis(h.n);
    basis3 w2n = vec3_OrthonormalBasis_Inverse(n2w);
    f3     w_in = vec3_Reorient(vmuls(din, -1.0f), w2n);
    float  u1 = Random_Unilateral(rng);
    float  u2 = Random_Unilateral(rng);
    f3     mn = GGXVNDF_Sample(w_in, ax, ay, u1, u2);
    f3     w_out = vnorm(vec3_Reflect(vmuls(w_in, -1.0f), mn));
    f3     albedo = Texture_ColorAt(S, m.tex, h.u, h.v);
    // BRDF_Disney_Metal :391
    f3    k_s = vlerp(mk3(1.0f, 1.0f, 1.0f), C_tint(albedo), m.p[3]);     // K_s :380
    float em1 = m.p[7] - 1.0f, ep1 = m.p[7] + 1.0f;
    float r0  = (em1 * em1) / (ep1 * ep1);                                 // R_0 :370
    f3    c0  = vlerp(vmuls(k_s, m.p[1] * r0), albedo, m.p[10]);           // C_0 :375
    float ct  = vdot(w_out, mn);
    f3    f_m = vadd(c0, vmuls(vsub(mk3(1.0f, 1.0f, 1.0f), c0), pow5(1.0f - ct))); // Fresnel_Schlick_Chromatic :339
    float g_m = G_m1(w_out, ax, ay);
    surf      = vmuls(f_m, g_m);
    emit      = mk3(0.0f, 0.0f, 0.0f);
    dout      = vec3_Reorient(w_out, n2w);
    return true;
}

DEV bool Disney_Clearcoat_Bounce(const rtx_material& m, f3 din, const HitD& h, Rng& rng, f3& surf, f3& emit, f3& dout) // :598
{
    float  a_g = (1.0f - m.p[6]) * 0.1f + m.p[6] * 0.001f;
    basis3 n2w = vec3_OrthonormalBasis(h.n);
    basis3 w2n = vec3_OrthonormalBasis_Inverse(n2w);
    f3     w_in = vec3_Reorient(vmuls(din, -1.0f), w2n);
    f3     macro = mk3(0.0f, 0.0f, 1.0f);
    // Disney_Clearcoat_SampleNormal :564
    float u0 = Random_Unilateral(rng);
    float u1 = Random_Unilateral(rng);
    float a2 = a_g * a_g;
    float ce = sqrtf((1.0f - powf(a2, 1.0f - u0)) / (1.0f - a2));
    float az = 2.0f * PI32 * u1;
    float se = sqrtf(1.0f - ce * ce);
    float sa, ca;
    sincosf(az, &sa, &ca);
    f3    mn = mk3(se * ca, se * sa, ce);
    f3    w_out = vnorm(vec3_Reflect(vmuls(w_in, -1.0f), mn));
    // BRDF_Disney_Clearcoat :548
    float r_0 = ((1.5f - 1.0f) * (1.5f - 1.0f)) / ((1.5f + 1) * (1.5f + 1));
    float f_c = r_0 + (1.0f - r_0) * pow5(1.0f - fabsf(vdot(mn, w_out)));
    // Lambda_c :525 (roughness 0.25)
    auto G_c1 = [](f3 w) {
        float r2  = 0.25f * 0.25f;
        float num = sqrtf(1.0f + (r2 * (w.x * w.x + w.y * w.y)) / (w.z * w.z)) - 1.0f;
        return 1.0f / (1.0f + num / 2.0f);
    };
    float g_c  = G_c1(w_in) * G_c1(w_out);
    float brdf = (f_c * 1.0f * g_c) / fabsf(vdot(macro, w_in));
    // Disney_Clearcoat_SampleNormal_InvPDF :586
    float inv_pdf = fabsf(vdot(mn, w_out)) / (1.0f * fabsf(vdot(macro, mn)));
    float c       = brdf * inv_pdf;
    surf          = mk3(c, c, c);
    emit          = mk3(0.0f, 0.0f, 0.0f);
    dout          = vec3_Reorient(w_out, n2w);
    return true;
}

DEV float Fresnel_Achromatic_DotProducts(float di, float dot_out, float eta) // :652
{
    float r_s = (di - eta * dot_out) / (di + eta * dot_out);
    float r_p = (eta * di - dot_out) / (eta * di + dot_out);
    return 0.5f * (r_s * r_s + r_p * r_p);
}

DEV bool Disney_Glass_Bounce(const SceneDev& S, const rtx_material& m, f3 din, const HitD& h, Rng& rng, f3& surf, f3& emit,
                             f3& dout) // :722
{
    float ax, ay;
    disney_alpha(m, ax, ay);
    float  eta = h.front ? 1.0f / m.p[7] : m.p[7];
    basis3 n2w = vec3_OrthonormalBasis(h.n);
    basis3 w2n = vec3_OrthonormalBasis_Inverse(n2w);
    f3     w_in = vec3_Reorient(vmuls(din, -1.0f), w2n);
    float  u0 = Random_Unilateral(rng);
    float  u1 = Random_Unilateral(rng);
    f3     mn = GGXVNDF_Sample(w_in, ax, ay, u0, u1);
    float  mdi = vdot(mn, w_in);
    // Fresnel_Achromatic_IncidentOnly :669
    float f_g;
    {
        float d2 = 1.0f - (eta * eta) * (1.0f - mdi * mdi);
        if (d2 < 0.0f) {
            f_g = 1.0f;
        } else {
            f_g = Fresnel_Achromatic_DotProducts(mdi, sqrtf(d2), eta);
        }
    }
    float u2 = Random_Unilateral(rng);
    f3    w_out;
    bool  reflected;
    if (u2 <= f_g) {
        w_out     = vec3_Reflect(vmuls(w_in, -1.0f), mn);
        reflected = true;
    } else {
        w_out     = vec3_Refract(vmuls(w_in, -1.0f), mn, eta);
        reflected = false;
    }
    w_out     = vnorm(w_out);
    f3 albedo = Texture_ColorAt(S, m.tex, h.u, h.v);
    float g_g = G_m1(w_out, ax, ay);
    if (reflected) { // BRDF_Disney_Glass :696
        surf = vmuls(albedo, g_g);
    } else {
        surf = vmuls(mk3(sqrtf(albedo.x), sqrtf(albedo.y), sqrtf(albedo.z)), g_g);
    }
    emit = mk3(0.0f, 0.0f, 0.0f);
    dout = vec3_Reorient(w_out, n2w);
    return true;
}

DEV bool Material_Bounce(const SceneDev& S, const rtx_material& m, f3 din, const HitD& h, Rng& rng, f3& surf, f3& emit,
                         f3& dout) // src/world/object.c:64
{
    surf = mk3(0.0f, 0.0f, 0.0f);
    emit = mk3(0.0f, 0.0f, 0.0f);
    dout = h.n;
    switch (m.type) {
        case RTX_MAT_DIFFUSE: { // materials.c:27
            dout = diffuse_scatter(rng, h);
            surf = Texture_ColorAt(S, m.tex, h.u, h.v);
            return true;
        }
        case RTX_MAT_METAL: { // :64
            f3 reflected = vec3_Reflect(din, h.n);
            f3 fuzz      = vmuls(Random_OnSphere(rng, 1.0f), m.p[0]);
            surf         = Texture_ColorAt(S, m.tex, h.u, h.v);
            dout         = vadd(reflected, fuzz);
            // the test is on rayOut->dir, i.e. the NORMALISED direction (:83)
            return vdot(vnorm(dout), h.n) > 0.0f;
        }
        case RTX_MAT_DIELECTRIC: { // :106
            float ratio = h.front ? (1.0f / m.p[0]) : m.p[0];
            float ct    = fminf(vdot(vmuls(din, -1.0f), h.n), 1.0f);
            float st    = sqrtf(1.0f - ct * ct);
            bool  cannotRefract = ratio * st > 1.0f;
            bool  reflect       = cannotRefract;
            if (!reflect) { // `||` short-circuit: the draw happens only when refraction is possible
                float r0 = (1.0f - ratio) / (1.0f + ratio);
                r0       = r0 * r0;
                float refl = r0 + (1.0f - r0) * powf(1.0f - ct, 5.0f);
                reflect    = refl > Random_Unilateral(rng);
            }
            dout = reflect ? vec3_Reflect(din, h.n) : vec3_Refract(din, h.n, ratio);
            surf = Texture_ColorAt(S, m.tex, h.u, h.v);
            return true;
        }
        case RTX_MAT_DIFFUSE_LIGHT: { // :149
            emit = vmuls(Texture_ColorAt(S, m.tex, h.u, h.v), m.p[0]);
            return false;
        }
        case RTX_MAT_SKYBOX: { // :179
            dout = diffuse_scatter(rng, h);
            surf = Skybox_ColorAt(S, din);
            return true;
        }
        case RTX_MAT_DISNEY_DIFFUSE: return Disney_Diffuse_Bounce(S, m, din, h, rng, surf, emit, dout);
        case RTX_MAT_DISNEY_METAL: return Disney_Metal_Bounce(S, m, din, h, rng, surf, emit, dout);
        case RTX_MAT_DISNEY_GLASS: return Disney_Glass_Bounce(S, m, din, h, rng, surf, emit, dout);
        case RTX_MAT_DISNEY_CLEARCOAT: return Disney_Clearcoat_Bounce(m, din, h, rng, surf, emit, dout);
        case RTX_MAT_DISNEY_SHEEN: return Disney_Sheen_Bounce(S, m, din, h, rng, surf, emit, dout);
        case RTX_MAT_DISNEY_BSDF: { // :868
            float wd, ws, wm, wc, wg;
            const float w_clear = m.p[8], w_sheen = m.p[9], w_metal = m.p[10], w_spec = m.p[11];
            if (h.front) {
                wd = (1.0f - w_spec) * (1.0f - w_metal);
                ws = (1.0f - w_metal) * w_sheen;
                wm = (1.0f - w_spec * (1.0f - w_metal));
                wc = 0.25f * w_clear;
                wg = (1.0f - w_metal) * w_spec;
            } else {
                wd = 0.0f, ws = 0.0f, wm = 0.0f, wc = 0.0f;
                wg = (1.0f - w_metal) * w_spec;
            }
            float total = wd + ws + wm + wc + wg;
            float sel   = Random_Unilateral(rng) * total;
            float d_hi = wd, s_hi = d_hi + ws, m_hi = s_hi + wm, c_hi = m_hi + wc, g_hi = c_hi + wg;
            if (0.0f <= sel && sel < d_hi) return Disney_Diffuse_Bounce(S, m, din, h, rng, surf, emit, dout);
            if (d_hi <= sel && sel < s_hi) return Disney_Sheen_Bounce(S, m, din, h, rng, surf, emit, dout);
            if (s_hi <= sel && sel < m_hi) return Disney_Metal_Bounce(S, m, din, h, rng, surf, emit, dout);
            if (m_hi <= sel && sel < c_hi) return Disney_Clearcoat_Bounce(m, din, h, rng, surf, emit, dout);
            if (c_hi <= sel && sel < g_hi) return Disney_Glass_Bounce(S, m, din, h, rng, surf, emit, dout);
            return false; // :915-923: black, stop
        }
        default: return false;
    }
}
