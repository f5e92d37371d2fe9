// oracle/oracle.cpp — CPU restatement of the reference's ray→scene query path.
//
// TEST INFRASTRUCTURE ONLY: see oracle.h.  Parity status: PINNED — every function below is checked
// against outputs of the real reference (oracle/_ref/ref_parity → tests/golden/*.npz) by
// tests/test_oracle_pinned.py.
//
// Build: ROCm clang++ -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile) — the compiler and FP
// flags of the pinned reference build, so that libm calls, constant folding (e.g. powf(x, 2) →
// x*x) and rounding are the same on both sides.
//
// Written in the reference's own shape (recursive kd traversal, recursive RayColor, thread-local
// RNG) on purpose: the HIP kernels restructure all of that (explicit stack, wavefront queues,
// forward throughput), and the oracle is what shows the restructuring changed nothing.
// Each function cites the reference lines it follows (paths relative to /root/reference/src).

#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <string.h>

#include <vector>

#define RT_EPSILON (0.0001f) // common/config.h:15
#define PI32 (3.141592653589793238462643383279502884197169399375105820f) // math/math.h:7
#define INF (__builtin_inff())
#define POWF(base, exp) (powf((base), (exp))) // math/math.h:31

typedef float    f32;
typedef uint32_t u32;
typedef uint64_t u64;

// ---- math/vec.h, math/vec.c ----
struct vec2 {
    f32 x, y;
};
struct vec3 {
    f32 x, y, z;
};
typedef vec3 point3;
typedef vec3 Color;
struct basis3 {
    vec3 x, y, z;
};

static inline vec3 vec(f32 x, f32 y, f32 z) { return vec3{x, y, z}; }
static inline vec3 vadd(vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }          // vec.c:157
static inline vec3 vsub(vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }          // vec.c:166
static inline vec3 vmul(vec3 v, f32 s) { return vec3{s * v.x, s * v.y, s * v.z}; }                 // vec.c:175
static inline vec3 vmul(f32 s, vec3 v) { return vmul(v, s); }                                      // vec.c:184
static inline vec3 vmul(vec3 a, vec3 b) { return vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }          // vec.c:189
static inline vec3 vdiv(vec3 v, f32 s) { return vmul(v, 1.0f / s); }                               // vec.c:198
static inline vec3 vdiv(vec3 a, vec3 b) { return vec3{a.x / b.x, a.y / b.y, a.z / b.z}; }          // vec.c:203
static inline f32  vdot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }              // vec.c:212
static inline vec3 vcross(vec3 a, vec3 b)                                                          // vec.c:217
{
    return vec3{(a.y * b.z - a.z * b.y), (a.z * b.x - a.x * b.z), (a.x * b.y - a.y * b.x)};
}
static inline vec3 vlerp(vec3 a, vec3 b, f32 t) { return vadd(a, vmul(vsub(b, a), t)); }           // vec.c:226
static inline f32  vmag(vec3 v) { return sqrtf(vdot(v, v)); }                                      // vec.c:236
static inline vec3 vnorm(vec3 v) { return vdiv(v, vmag(v)); }                                      // vec.c:241
static inline vec3 vsum(vec3 a, vec3 b, vec3 c) { return vadd(vadd(a, b), c); }                    // vec.h:390
static inline vec2 vadd(vec2 a, vec2 b) { return vec2{a.x + b.x, a.y + b.y}; }                     // vec.c:16
static inline vec2 vmul(vec2 v, f32 s) { return vec2{s * v.x, s * v.y}; }                          // vec.c:32
static inline vec2 vsum(vec2 a, vec2 b, vec2 c) { return vadd(vadd(a, b), c); }
static inline f32  elem(vec3 v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }

static inline f32 clampf(f32 x, f32 min, f32 max) { return x < min ? min : (x > max ? max : x); } // math.c:9
static inline f32 minf(f32 x, f32 y) { return x > y ? y : x; }                                     // math.c:30
static inline f32 maxf(f32 x, f32 y) { return x > y ? x : y; }                                     // math.c:39

static bool equalf(f32 a, f32 b) // math.c:48-63
{
    f32 absA = fabsf(a), absB = fabsf(b), diff = fabsf(a - b);
    if (a == b) {
        return true;
    } else if (a == 0.0f || b == 0.0f || (absA + absB < __FLT_MIN__)) {
        return diff < (__FLT_EPSILON__ * __FLT_MIN__);
    } else {
        return (diff / minf((absA + absB), __FLT_MAX__)) < __FLT_EPSILON__;
    }
}
static inline bool vequ(vec3 v, f32 mag) { return equalf(vdot(v, v), mag * mag); } // vec.c:246

static vec3 vec3_Reflect(vec3 v, vec3 n) // vec.c:273
{
    f32  vecDotNormal      = vdot(v, n);
    vec3 angleScaledNormal = vmul(n, 2.0f * vecDotNormal);
    return vsub(v, angleScaledNormal);
}

static vec3 vec3_Refract(vec3 v, vec3 n, f32 refractRatio) // vec.c:280
{
    f32  cosTheta   = fminf(vdot(vmul(v, -1), n), 1.0f);
    vec3 vecOutPerp = vmul(vadd(v, vmul(n, cosTheta)), refractRatio);
    vec3 vecOutPara = vmul(n, -sqrtf(fabsf(1.0f - vdot(vecOutPerp, vecOutPerp))));
    return vadd(vecOutPerp, vecOutPara);
}

static vec3 vec3_Reorient(vec3 n, basis3 b) { return vsum(vmul(b.x, n.x), vmul(b.y, n.y), vmul(b.z, n.z)); } // vec.c:308

static basis3 vec3_OrthonormalBasis(vec3 bz) // vec.c:319
{
    vec3 bx, by;
    if (bz.z < -0.9999999f) {
        by = vec3{0.0f, -1.0f, 0.0f};
        bx = vec3{-1.0f, 0.0f, 0.0f};
    } else {
        f32 a = 1.0f / (1.0f + bz.z);
        f32 b = -bz.x * bz.y * a;
        by    = vec3{-1.0f + bz.x * bz.x * a, -b, bz.x};
        bx    = vec3{-b, -1.0f + bz.y * bz.y * a, bz.y};
    }
    return basis3{bx, by, bz};
}

static basis3 vec3_OrthonormalBasis_Inverse(basis3 o) // vec.c:340
{
    return basis3{vec(o.x.x, o.y.x, o.z.x), vec(o.x.y, o.y.y, o.z.y), vec(o.x.z, o.y.z, o.z.z)};
}

// ---- math/random.c ----
static thread_local u32 s[4];

static inline u32 rotl(u32 x, int k) { return (x << k) | (x >> (32 - k)); }

static u32 next(void) // random.c:31
{
    u32 result = s[0] + s[3];
    u32 t      = s[1] << 9;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 11);
    return result;
}

static void jump(void) // random.c:53
{
    static const u32 JUMP[] = {0x8764000b, 0xf542d2d3, 0x6fa035c3, 0x77f2db5b};
    u32              s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (size_t i = 0; i < 4; i++)
        for (size_t b = 0; b < 32; b++) {
            if (JUMP[i] & UINT32_C(1) << b) {
                s0 ^= s[0];
                s1 ^= s[1];
                s2 ^= s[2];
                s3 ^= s[3];
            }
            next();
        }
    s[0] = s0, s[1] = s1, s[2] = s2, s[3] = s3;
}

static void Random_Seed(u64 s1, u64 s2) // random.c:78
{
    s[0] = (u32)s1;
    s[1] = (u32)(s1 >> 32);
    s[2] = (u32)s2;
    s[3] = (u32)(s2 >> 32);
    jump();
}

static f32 Random_Unilateral(void) // random.c:100
{
    u32 nextVal = next();
    u32 bits    = (nextVal >> 9) | 0x3f800000;
    f32 f;
    memcpy(&f, &bits, 4);
    return f - 1.0f;
}

static f32 Random_InRange(f32 min, f32 max) { return min + (max - min) * Random_Unilateral(); } // random.c:119
static f32 Random_Bilateral(void) { return Random_InRange(-1.0f, 1.0f); }                      // random.c:125

static vec3 Random_InSphere(f32 radius) // random.c:140
{
    f32 epsilon_0 = Random_Unilateral();
    f32 epsilon_1 = Random_Unilateral();
    f32 epsilon_2 = Random_Bilateral();
    f32 rho       = radius * epsilon_0;
    f32 phi       = 2.0f * PI32 * epsilon_1;
    f32 cos_theta = epsilon_2;
    f32 sin_theta = sqrtf(1.0f - POWF(cos_theta, 2));
    vec3 r;
    r.x = rho * sin_theta * cosf(phi);
    r.y = rho * sin_theta * sinf(phi);
    r.z = rho * cos_theta;
    return r;
}

static vec3 Random_OnSphere(f32 radius) // random.c:159
{
    f32 epsilon_1 = Random_Unilateral();
    f32 epsilon_2 = Random_Bilateral();
    f32 rho       = radius;
    f32 phi       = 2.0f * PI32 * epsilon_1;
    f32 cos_theta = epsilon_2;
    f32 sin_theta = sqrtf(1.0f - POWF(cos_theta, 2));
    vec3 r;
    r.x = rho * sin_theta * cosf(phi);
    r.y = rho * sin_theta * sinf(phi);
    r.z = rho * cos_theta;
    return r;
}

static vec3 Random_InHemisphere(vec3 facing, f32 radius) // random.c:176
{
    vec3 in_hemisphere = Random_InSphere(radius);
    if (vdot(in_hemisphere, facing) < 0.0f) {
        in_hemisphere = vmul(-1.0f, in_hemisphere);
    }
    return vmul(radius, in_hemisphere);
}

static vec2 Random_InDisc(f32 radius) // random.c:200
{
    f32 epsilon_0 = Random_Unilateral();
    f32 epsilon_1 = Random_Bilateral();
    f32 r         = radius * epsilon_0;
    f32 cos_theta = epsilon_1;
    f32 sin_theta = epsilon_0 * sqrtf(1.0f - POWF(epsilon_1, 2));
    return vec2{r * cos_theta, r * sin_theta};
}

// ---- rt/ray.h, rt/ray.c ----
struct Ray {
    struct {
        vec3 invDir, originDivDir;
    } cache;
    point3 origin;
    vec3   dir;
};

struct HitInfo {
    point3 position;
    vec3   unitNormal;
    vec2   uv;
    f32    tIntersect;
    bool   frontFace;
};

static Ray Ray_FromUnit(point3 origin, vec3 dir)
{
    Ray r;
    r.origin             = origin;
    r.dir                = dir;
    r.cache.invDir       = vdiv(vec(1.0f, 1.0f, 1.0f), dir);
    r.cache.originDivDir = vdiv(origin, dir);
    return r;
}

static Ray    Ray_Make(point3 origin, vec3 dir) { return Ray_FromUnit(origin, vnorm(dir)); } // ray.c:3
static point3 Ray_At(const Ray* ray, f32 dist) { return vadd(ray->origin, vmul(ray->dir, dist)); } // ray.c:18

static void HitInfo_SetFaceNormal(HitInfo* hit, const Ray* ray, vec3 outwardNormal) // ray.c:23
{
    hit->frontFace  = vdot(ray->dir, outwardNormal) < 0;
    hit->unitNormal = hit->frontFace ? outwardNormal : vmul(outwardNormal, -1.0f);
}

// ---- scene view over the flattened pools ----
// Checker for the GPU's subtree-box culling (NOT part of the reference): when `qbox` is
// set, CheckHitNextNode skips a node whose box the segment [0, kt*tMax] of the ray misses — with the
// device's own boxes (rtx_cull_data_build) and the device's own test, operation for operation.
struct Cull {
    const uint16_t* qbox; // 6 per node: min x y z, max x y z on the 16-bit grid; min.x = 0xFFFF, max.x = 0: never cull
    const uint8_t*  flags; // per node (rtx_cull_node_flags) or NULL; bit 1: the device enters this node without a box test of its
                           // own (its parent is a cut-off node whose plane the device evaluates in this node's step)
    vec3            c;
    f32             ro2, kt;
    vec3            lo, scale; // grid x(q) = lo + q * scale
    // per ray
    bool ok;
    vec3 o, inv; // origin, and 1/dir with infinities replaced by +-3e38
    vec3 nol;    // -(o - lo), one rounding per axis
};

typedef unsigned __int128 u128;
struct Scene {
    const rtx_scene_desc* d;
    u64                   nInner, nLeaf, nPrim; // visit counters (per thread copy)
    Cull*                 cull;
    u64                   nCulled;
    // the place of the walk in the literal visiting order (depth-first, origin side before opposite side): bit (127 - level)
    // of `key` is set where the path took the opposite side of a two-sided inner node.  Bookkeeping for the speculative
    // replay (oracle_trace_speculative) only; the walk itself never reads it unless `speculative` is set.
    u128 key;
    int  level;
    // speculative replay (checker of the GPU library's k_trace_long): a leaf that accepts a hit does NOT end the
    // walk — it is noted if it comes before the best one noted so far in the visiting order, and the walk goes on through
    // every subtree the sequential walk could reach; subtrees that come after the best leaf noted are skipped
    bool    speculative, spec_any;
    u128    spec_key;
    u32     spec_obj;
    HitInfo spec_hit;
};

static void cull_begin_ray(Cull* c, const Ray* ray)
{
    const f32  big = 3.0e38f;
    const vec3 o = ray->origin, inv = ray->cache.invDir;
    c->inv = vec(fabsf(inv.x) > big ? copysignf(big, inv.x) : inv.x, fabsf(inv.y) > big ? copysignf(big, inv.y) : inv.y,
                 fabsf(inv.z) > big ? copysignf(big, inv.z) : inv.z);
    c->o   = o;
    c->nol = vec(-(o.x - c->lo.x), -(o.y - c->lo.y), -(o.z - c->lo.z));
    const f32 rx = o.x - c->c.x, ry = o.y - c->c.y, rz = o.z - c->c.z;
    const f32 r2 = rx * rx + ry * ry + rz * rz;
    const f32 d2 = ray->dir.x * ray->dir.x + ray->dir.y * ray->dir.y + ray->dir.z * ray->dir.z;
    c->ok        = r2 <= c->ro2 && fabsf(d2 - 1.0f) <= 1.0e-3f;
}

// a box plane's ray parameter, as the device computes it: fma(float(q), scale, -(o - lo)) * inv
static inline f32 plane_t(u32 q, f32 scale, f32 nol, f32 inv) { return fmaf((f32)q, scale, nol) * inv; }

// the part [tn, tf] of {0 <= t <= kt * tMax} the ray spends inside a node's box (the device's box_range); false: never cull
static bool cull_range(const Cull* c, u32 node, f32 tMax, f32* tn, f32* tf)
{
    const uint16_t* q = c->qbox + 6ull * node;
    const f32 t1x = plane_t(q[0], c->scale.x, c->nol.x, c->inv.x), t2x = plane_t(q[3], c->scale.x, c->nol.x, c->inv.x);
    const f32 t1y = plane_t(q[1], c->scale.y, c->nol.y, c->inv.y), t2y = plane_t(q[4], c->scale.y, c->nol.y, c->inv.y);
    const f32 t1z = plane_t(q[2], c->scale.z, c->nol.z, c->inv.z), t2z = plane_t(q[5], c->scale.z, c->nol.z, c->inv.z);
    *tn = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fmaxf(fminf(t1z, t2z), 0.0f));
    *tf = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fminf(fmaxf(t1z, t2z), tMax * c->kt));
    return !(q[0] == 0xFFFFu && q[3] == 0u);
}

static bool cull_node(const Cull* c, u32 node, f32 tMax)
{
    f32        tn, tf;
    const bool bounded = cull_range(c, node, tMax, &tn, &tf);
    return c->ok && bounded && tn > tf;
}

struct Prim { // what the reference calls Object*: a pool reference
    u32 ref;  // leaf_ref encoding, or 0xC0000000|i for planes, 0xFFFFFFFF none
};

static inline vec3 tri_pos(const rtx_scene_desc* d, u32 i, int v)
{
    size_t T = d->n_tris;
    return vec(d->tri_pos[(3 * v + 0) * T + i], d->tri_pos[(3 * v + 1) * T + i], d->tri_pos[(3 * v + 2) * T + i]);
}
static inline vec3 tri_nrm(const rtx_scene_desc* d, u32 i, int v)
{
    size_t T = d->n_tris;
    return vec(d->tri_nrm[(3 * v + 0) * T + i], d->tri_nrm[(3 * v + 1) * T + i], d->tri_nrm[(3 * v + 2) * T + i]);
}
static inline vec2 tri_tex(const rtx_scene_desc* d, u32 i, int v)
{
    size_t T = d->n_tris;
    return vec2{d->tri_uv[(2 * v + 0) * T + i], d->tri_uv[(2 * v + 1) * T + i]};
}

// ---- rt/surfaces.c ----
static vec2 Sphere_MapUV(point3 onSphere) // surfaces.c:44
{
    f32 theta = acosf(clampf(-onSphere.y, -0.999f, 0.999f));
    f32 phi   = atan2f(-onSphere.z, onSphere.x) + PI32;
    return vec2{phi / (2 * PI32), theta / PI32};
}

static bool Sphere_HitAt(const rtx_scene_desc* d, u32 i, const Ray* ray, f32 tMin, f32 tMax, HitInfo* hit) // surfaces.c:55
{
    size_t S = d->n_spheres;
    vec3   c = vec(d->sph[0 * S + i], d->sph[1 * S + i], d->sph[2 * S + i]);
    f32    r = d->sph[3 * S + i];

    vec3 dist         = vsub(ray->origin, c);
    f32  polyA        = vdot(ray->dir, ray->dir);
    f32  halfPolyB    = vdot(dist, ray->dir);
    f32  polyC        = vdot(dist, dist) - r * r;
    f32  discriminant = halfPolyB * halfPolyB - polyA * polyC;

    if (discriminant < 0.0f) {
        return false;
    } else {
        f32 root1 = (-halfPolyB - sqrtf(discriminant)) / polyA;
        f32 root2 = (-halfPolyB + sqrtf(discriminant)) / polyA;
        f32 tIntersect;
        if (tMin < root1 && root1 < tMax) {
            tIntersect = root1;
        } else if (tMin < root2 && root2 < tMax) {
            tIntersect = root2;
        } else {
            return false;
        }
        hit->tIntersect    = tIntersect;
        hit->position      = Ray_At(ray, tIntersect);
        vec3 outwardNormal = vdiv(vsub(hit->position, c), r);
        HitInfo_SetFaceNormal(hit, ray, outwardNormal);
        hit->uv = Sphere_MapUV(outwardNormal);
        return true;
    }
}

static bool Triangle_HitAt(const rtx_scene_desc* d, u32 i, const Ray* ray, f32 tMin, f32 tMax, HitInfo* hit) // surfaces.c:135
{
    vec3 p0 = tri_pos(d, i, 0), p1 = tri_pos(d, i, 1), p2 = tri_pos(d, i, 2);
    vec3 edge1 = vsub(p1, p0);
    vec3 edge2 = vsub(p2, p0);

    vec3 h = vcross(ray->dir, edge2);
    f32  a = vdot(edge1, h);
    if (fabsf(a) < RT_EPSILON) {
        return false;
    }
    vec3 sv = vsub(ray->origin, p0);
    f32  u  = vdot(sv, h) / a;
    if (u < 0.0f || u > 1.0f) {
        return false;
    }
    vec3 q = vcross(sv, edge1);
    f32  v = vdot(ray->dir, q) / a;
    if (v < 0.0f || u + v > 1.0f) {
        return false;
    }
    f32 t = vdot(edge2, q) / a;
    if (t > tMax || t < tMin) {
        return false;
    } else {
        hit->position   = Ray_At(ray, t);
        hit->tIntersect = t;
        f32 coeff       = 1.0f - u - v;
        hit->uv = vsum(vmul(tri_tex(d, i, 0), coeff), vmul(tri_tex(d, i, 1), u), vmul(tri_tex(d, i, 2), v));
        vec3 outwardNormal = vsum(vmul(tri_nrm(d, i, 0), coeff), vmul(tri_nrm(d, i, 1), u), vmul(tri_nrm(d, i, 2), v));
        HitInfo_SetFaceNormal(hit, ray, outwardNormal);
        return true;
    }
}

static bool Plane_HitAt(const rtx_scene_desc* d, u32 i, const Ray* ray, f32 t_min, f32 t_max, HitInfo* hit) // surfaces.c:223
{
    size_t P      = d->n_planes;
    vec3   normal = vec(d->plane[0 * P + i], d->plane[1 * P + i], d->plane[2 * P + i]);
    vec3   point  = vec(d->plane[3 * P + i], d->plane[4 * P + i], d->plane[5 * P + i]);

    f32 numerator   = vdot(normal, vsub(point, ray->origin));
    f32 denominator = vdot(normal, ray->dir);
    f32 t_intersect = numerator / denominator;
    if (t_intersect < t_min || t_intersect > t_max) {
        return false;
    } else {
        hit->position   = Ray_At(ray, t_intersect);
        hit->tIntersect = t_intersect;
        HitInfo_SetFaceNormal(hit, ray, normal);
        hit->uv = vec2{0.0f, 0.0f}; // uninitialised in the reference; defined as (0,0) on both sides
        return true;
    }
}

static bool Surface_HitAt(const rtx_scene_desc* d, u32 ref, const Ray* ray, f32 tMin, f32 tMax, HitInfo* hit) // world/object.c:7
{
    if ((ref >> 30) == 3u) return Plane_HitAt(d, ref & 0x3FFFFFFFu, ray, tMin, tMax, hit);
    if (ref & RTX_REF_SPHERE) return Sphere_HitAt(d, ref & 0x7FFFFFFFu, ray, tMin, tMax, hit);
    return Triangle_HitAt(d, ref, ray, tMin, tMax, hit);
}

// ---- rt/accelerators/kdtree.c:453-577 (recursive, as written) ----
static bool CheckHitNextNode(Scene* sc, u32 node, const Ray* ray, Prim* objHit, HitInfo* hit, f32 tMax);

static bool CheckHitLeafNode(Scene* sc, u32 node, const Ray* ray, Prim* objHit, HitInfo* hit, f32 tMax) // :475
{
    const rtx_scene_desc* d = sc->d;
    HitInfo hitClosest;
    memset(&hitClosest, 0, sizeof hitClosest);
    hitClosest.tIntersect = INF;
    u32 objClosest        = 0xFFFFFFFFu;
    u32 len = d->node_w0[node], objIndex = d->node_w1[node] & 0x3FFFFFFFu;
    sc->nLeaf++;
    for (size_t ii = 0; ii < len; ii++) {
        HitInfo hitCur;
        u32     objCur = d->leaf_ref[objIndex + ii];
        sc->nPrim++;
        if (Surface_HitAt(d, objCur, ray, RT_EPSILON, tMax, &hitCur)) {
            if (hitCur.tIntersect < hitClosest.tIntersect) {
                objClosest = objCur;
                hitClosest = hitCur;
            }
        }
    }
    if (objClosest != 0xFFFFFFFFu) {
        if (sc->speculative) { // note it, and walk on as if nothing had been found
            if (!sc->spec_any || sc->key < sc->spec_key) sc->spec_any = true, sc->spec_key = sc->key, sc->spec_obj = objClosest, sc->spec_hit = hitClosest;
            return false;
        }
        *hit        = hitClosest;
        objHit->ref = objClosest;
        return true;
    }
    return false;
}

static bool CheckHitInternalNode(Scene* sc, u32 node, const Ray* ray, Prim* objHit, HitInfo* hit, int axis, f32 tMax) // :501
{
    const rtx_scene_desc* d = sc->d;
    f32 split;
    memcpy(&split, &d->node_w0[node], 4);
    u32 left  = d->node_w1[node] & 0x3FFFFFFFu;
    u32 right = node + 1;
    sc->nInner++;

    if (fabsf(elem(ray->dir, axis)) < RT_EPSILON) {
        if (elem(ray->origin, axis) >= split) {
            return CheckHitNextNode(sc, right, ray, objHit, hit, tMax);
        } else {
            return CheckHitNextNode(sc, left, ray, objHit, hit, tMax);
        }
    }

    f32 tIntersect = split * elem(ray->cache.invDir, axis) - elem(ray->cache.originDivDir, axis);
    f32 tMaxNew    = minf(tMax, tIntersect);

    if (tIntersect < RT_EPSILON) {
        if (elem(ray->origin, axis) > split) {
            return CheckHitNextNode(sc, right, ray, objHit, hit, tMax);
        } else if (elem(ray->origin, axis) < split) {
            return CheckHitNextNode(sc, left, ray, objHit, hit, tMax);
        } else {
            f32 rayAt = elem(Ray_At(ray, 100.0f), axis);
            if (rayAt >= split) {
                return CheckHitNextNode(sc, right, ray, objHit, hit, tMax);
            } else {
                return CheckHitNextNode(sc, left, ray, objHit, hit, tMax);
            }
        }
    } else {
        u32 originSide, oppositeSide;
        if (elem(ray->origin, axis) >= split) {
            originSide   = right;
            oppositeSide = left;
        } else {
            originSide   = left;
            oppositeSide = right;
        }
        const u128 key   = sc->key; // (bookkeeping of the range replay: where the walk is in the visiting order)
        const int  level = sc->level;
        sc->level        = level + 1;
        if (CheckHitNextNode(sc, originSide, ray, objHit, hit, tMaxNew)) {
            return true;
        } else if (tMaxNew == tIntersect) {
            sc->key   = key | ((u128)1 << (127 - (level < 127 ? level : 127)));
            sc->level = level + 1;
            return CheckHitNextNode(sc, oppositeSide, ray, objHit, hit, tMax);
        } else {
            return false;
        }
    }
}

static bool CheckHitNextNode(Scene* sc, u32 node, const Ray* ray, Prim* objHit, HitInfo* hit, f32 tMax) // :458
{
    if (sc->speculative && sc->spec_any && sc->key > sc->spec_key) return false; // this subtree comes after the best leaf noted
    if (sc->cull && !(sc->cull->flags && (sc->cull->flags[node] & 2u)) && cull_node(sc->cull, node, tMax)) { // checker mode only, see struct Cull
        sc->nCulled++;
        return false;
    }
    u32 type = sc->d->node_w1[node] >> 30;
    if (type == 3u) return CheckHitLeafNode(sc, node, ray, objHit, hit, tMax);
    return CheckHitInternalNode(sc, node, ray, objHit, hit, (int)type, tMax);
}

static bool KDTree_HitAt(Scene* sc, const Ray* ray, Prim* objHit, HitInfo* hit) // :572
{
    sc->key = 0, sc->level = 0;
    return CheckHitNextNode(sc, sc->d->root_index, ray, objHit, hit, INF);
}

// ---- world/scene.c:73-143 ----
static bool Scene_ClosestHitInArray(Scene* sc, const Ray* ray, Prim* objHit, HitInfo* hit) // :73 over unboundObjs
{
    u32     closestObjectHit = 0xFFFFFFFFu;
    HitInfo closestHit;
    memset(&closestHit, 0, sizeof closestHit);
    closestHit.tIntersect = INF;
    for (u32 ii = 0; ii < sc->d->n_planes; ii++) {
        HitInfo curHit;
        bool    hitDetected = Surface_HitAt(sc->d, 0xC0000000u | ii, ray, RT_EPSILON, INF, &curHit);
        if (hitDetected && (curHit.tIntersect < closestHit.tIntersect)) {
            closestHit       = curHit;
            closestObjectHit = 0xC0000000u | ii;
        }
    }
    if (closestObjectHit != 0xFFFFFFFFu) {
        *hit        = closestHit;
        objHit->ref = closestObjectHit;
        return true;
    }
    return false;
}

static bool Scene_ClosestHit(Scene* sc, const Ray* ray, Prim* obj_hit, HitInfo* hit) // :100
{
    Prim    obj_hit_bounded = {0xFFFFFFFFu};
    HitInfo hit_info_bounded;
    memset(&hit_info_bounded, 0, sizeof hit_info_bounded);
    hit_info_bounded.tIntersect = INF;
    bool hit_bounded            = KDTree_HitAt(sc, ray, &obj_hit_bounded, &hit_info_bounded);

    Prim    obj_hit_unbounded = {0xFFFFFFFFu};
    HitInfo hit_info_unbounded;
    memset(&hit_info_unbounded, 0, sizeof hit_info_unbounded);
    hit_info_unbounded.tIntersect = INF;
    bool hit_unbounded            = Scene_ClosestHitInArray(sc, ray, &obj_hit_unbounded, &hit_info_unbounded);

    if (hit_bounded && hit_unbounded) {
        if (hit_info_unbounded.tIntersect < hit_info_bounded.tIntersect) {
            *obj_hit = obj_hit_unbounded;
            *hit     = hit_info_unbounded;
        } else {
            *obj_hit = obj_hit_bounded;
            *hit     = hit_info_bounded;
        }
        return true;
    } else if (hit_bounded) {
        *obj_hit = obj_hit_bounded;
        *hit     = hit_info_bounded;
        return true;
    } else if (hit_unbounded) {
        *obj_hit = obj_hit_unbounded;
        *hit     = hit_info_unbounded;
        return true;
    }
    return false;
}

static u32 prim_id(const rtx_scene_desc* d, u32 ref)
{
    if (ref == 0xFFFFFFFFu) return RTX_ID_MISS;
    if ((ref >> 30) == 3u) return RTX_ID_UNBOUNDED | (ref & 0x3FFFFFFFu);
    if (ref & RTX_REF_SPHERE) return d->sph_obj[ref & 0x7FFFFFFFu];
    return d->tri_obj[ref];
}

static u32 prim_material(const rtx_scene_desc* d, u32 ref)
{
    if ((ref >> 30) == 3u) return d->plane_mat[ref & 0x3FFFFFFFu];
    if (ref & RTX_REF_SPHERE) return d->sph_mat[ref & 0x7FFFFFFFu];
    return d->tri_mat[ref];
}

// ---- gfx/texture.c:88, world/skybox.c:66 ----
static Color Texture_ColorAt(const rtx_scene_desc* d, int32_t tex, vec2 st)
{
    if (tex < 0) return vec(0, 0, 0);
    const rtx_texture_desc* t = &d->textures[tex];
    // the reference casts straight to size_t (UB outside the image); both sides clamp instead
    f32    fx = st.x * (t->width - 1);
    f32    fy = (1.0f - st.y) * (t->height - 1);
    size_t xx = fx > 0.0f ? (size_t)fminf(fx, (f32)(t->width - 1)) : 0;
    size_t yy = fy > 0.0f ? (size_t)fminf(fy, (f32)(t->height - 1)) : 0;
    const float* p = d->tex_data + 3 * (t->offset + yy * t->width + xx);
    return vec(p[0], p[1], p[2]);
}

static Color Skybox_ColorAt(const rtx_scene_desc* d, vec3 dir)
{
    f32  absX = fabsf(dir.x), absY = fabsf(dir.y), absZ = fabsf(dir.z);
    bool isXPositive = dir.x > 0 ? true : false;
    bool isYPositive = dir.y > 0 ? true : false;
    bool isZPositive = dir.z > 0 ? true : false;
    f32  maxAxis, uc, vc;
    size_t index = 0;
    if (isXPositive && absX >= absY && absX >= absZ) {
        maxAxis = absX, uc = dir.y, vc = dir.z, index = 0;
    } else if (!isXPositive && absX >= absY && absX >= absZ) {
        maxAxis = absX, uc = -dir.y, vc = dir.z, index = 1;
    } else if (isYPositive && absY >= absX && absY >= absZ) {
        maxAxis = absY, uc = -dir.x, vc = dir.z, index = 2;
    } else if (!isYPositive && absY >= absX && absY >= absZ) {
        maxAxis = absY, uc = dir.x, vc = dir.z, index = 3;
    } else if (isZPositive && absZ >= absX && absZ >= absY) {
        maxAxis = absZ, uc = -dir.x, vc = -dir.y, index = 4;
    } else {
        maxAxis = absZ, uc = -dir.x, vc = dir.y, index = 5;
    }
    vec2 uv = vec2{0.5f * (uc / maxAxis + 1.0f), 0.5f * (vc / maxAxis + 1.0f)};
    return Texture_ColorAt(d, d->skybox_tex[index], uv);
}

// ---- rt/materials.c ----
#define COLOR_WHITE (vec(1.0f, 1.0f, 1.0f))
#define COLOR_BLACK (vec(0.0f, 0.0f, 0.0f))
enum { P_SUBSURFACE, P_SPECULAR, P_ROUGHNESS, P_SPECULAR_TINT, P_ANISTROPIC, P_SHEEN_TINT, P_CLEARCOAT_GLOSS, P_ETA,
       P_W_CLEARCOAT, P_W_SHEEN, P_W_METALLIC, P_W_SPECULAR };

static bool Material_Diffuse_Bounce(const rtx_scene_desc* d, const rtx_material* m, Ray* rayIn, HitInfo* hit, Color* colorSurface,
                                    Color* colorEmitted, Ray* rayOut) // :27
{
    (void)rayIn;
    vec3 target    = vadd(hit->position, Random_InHemisphere(hit->unitNormal, 1.0f));
    vec3 scattered = vsub(target, hit->position);
    if (vequ(scattered, 0.0f)) {
        scattered = hit->unitNormal;
    }
    *colorSurface = Texture_ColorAt(d, m->tex, hit->uv);
    *colorEmitted = COLOR_BLACK;
    *rayOut       = Ray_Make(hit->position, scattered);
    return true;
}

static bool Material_Metal_Bounce(const rtx_scene_desc* d, const rtx_material* m, Ray* rayIn, HitInfo* hit, Color* colorSurface,
                                  Color* colorEmitted, Ray* rayOut) // :64
{
    vec3 reflected = vec3_Reflect(rayIn->dir, hit->unitNormal);
    vec3 fuzz      = vmul(Random_OnSphere(1.0f), m->p[0]);
    *colorSurface  = Texture_ColorAt(d, m->tex, hit->uv);
    *colorEmitted  = COLOR_BLACK;
    *rayOut        = Ray_Make(hit->position, vadd(reflected, fuzz));
    return vdot(rayOut->dir, hit->unitNormal) > 0.0f;
}

static inline f32 reflectance(f32 cosine, f32 refractiveIndex) // :99
{
    f32 r0 = (1.0f - refractiveIndex) / (1.0f + refractiveIndex);
    r0     = r0 * r0;
    return r0 + (1.0f - r0) * powf((1.0f - cosine), 5.0f);
}

static bool Material_Dielectric_Bounce(const rtx_scene_desc* d, const rtx_material* m, Ray* rayIn, HitInfo* hit, Color* colorSurface,
                                       Color* colorEmitted, Ray* rayOut) // :106
{
    f32  refractionRatio = hit->frontFace ? (1.0f / m->p[0]) : m->p[0];
    vec3 normRayDir      = rayIn->dir;
    f32  cosTheta        = fminf(vdot(vmul(normRayDir, -1), hit->unitNormal), 1.0f);
    f32  sinTheta        = sqrtf(1.0f - cosTheta * cosTheta);
    bool cannotRefract   = refractionRatio * sinTheta > 1.0f;
    vec3 bounce;
    if (cannotRefract || reflectance(cosTheta, refractionRatio) > Random_Unilateral()) {
        bounce = vec3_Reflect(normRayDir, hit->unitNormal);
    } else {
        bounce = vec3_Refract(normRayDir, hit->unitNormal, refractionRatio);
    }
    *colorSurface = Texture_ColorAt(d, m->tex, hit->uv);
    *colorEmitted = COLOR_BLACK;
    *rayOut       = Ray_Make(hit->position, bounce);
    return true;
}

static bool Material_DiffuseLight_Bounce(const rtx_scene_desc* d, const rtx_material* m, HitInfo* hit, Color* colorEmitted) // :149
{
    *colorEmitted = vmul(Texture_ColorAt(d, m->tex, hit->uv), m->p[0]); // Color_Brighten, color.c:58
    return false;
}

static bool Material_Skybox_Bounce(const rtx_scene_desc* d, Ray* rayIn, HitInfo* hit, Color* colorSurface, Color* colorEmitted,
                                   Ray* rayOut) // :179
{
    vec3 target    = vadd(hit->position, Random_InHemisphere(hit->unitNormal, 1.0f));
    vec3 scattered = vsub(target, hit->position);
    if (vequ(scattered, 0.0f)) {
        scattered = hit->unitNormal;
    }
    Color skyboxColor = Skybox_ColorAt(d, rayIn->dir);
    *colorSurface     = skyboxColor;
    *colorEmitted     = COLOR_BLACK;
    *rayOut           = Ray_Make(hit->position, scattered);
    return true;
}

static inline vec3 HalfVector(vec3 w_in, vec3 w_out) { return vnorm(vadd(w_in, w_out)); } // :217

static inline f32 F_D90(vec3 w_out, vec3 half_vector, f32 roughness) // :222
{
    return 0.5f + 2.0f * roughness * POWF(vdot(half_vector, w_out), 2);
}
static inline f32 F_D(vec3 w, vec3 shading_normal, f32 f_d90) // :227
{
    return 1.0f + (f_d90 - 1.0f) * POWF(1.0f - fabsf(vdot(shading_normal, w)), 5);
}
static inline Color BRDF_Disney_BaseDiffuse(Color base_color, vec3 w_in, vec3 w_out, vec3 shading_normal, f32 roughness) // :232
{
    vec3 half_vector = HalfVector(w_in, w_out);
    f32  f_d90       = F_D90(w_out, half_vector, roughness);
    f32  attenuation = (1.0f / PI32) * F_D(w_in, shading_normal, f_d90) * F_D(w_out, shading_normal, f_d90);
    return vmul(base_color, attenuation);
}
static inline f32 F_SS90(vec3 w_out, vec3 half_vector, f32 roughness) { return roughness * POWF(vdot(half_vector, w_out), 2); } // :240
static inline f32 F_SS(vec3 w, vec3 shading_normal, f32 f_ss90) // :245
{
    return 1.0f + (f_ss90 - 1.0f) * POWF(1.0f - fabsf(vdot(shading_normal, w)), 5);
}
static inline Color BRDF_Disney_Subsurface(Color base_color, vec3 w_in, vec3 w_out, vec3 shading_normal, f32 roughness) // :250
{
    vec3 half_vector = HalfVector(w_in, w_out);
    f32  f_ss90      = F_SS90(w_out, half_vector, roughness);
    f32  abs_n_w_in  = fabsf(vdot(shading_normal, w_in));
    f32  abs_n_w_out = fabsf(vdot(shading_normal, w_out));
    f32  f_ss_w_in   = F_SS(w_in, shading_normal, f_ss90);
    f32  f_ss_w_out  = F_SS(w_out, shading_normal, f_ss90);
    f32  attenuation = (1.25f / PI32) * (f_ss_w_in * f_ss_w_out * (1.0f / (abs_n_w_in + abs_n_w_out) - 0.5f) + 0.5f);
    return vmul(base_color, attenuation);
}
static inline Color BRDF_Diffuse(Color base_color, vec3 w_in, vec3 w_out, vec3 shading_normal, f32 roughness, f32 subsurface) // :266
{
    Color brdf_base_diffuse = BRDF_Disney_BaseDiffuse(base_color, w_in, w_out, shading_normal, roughness);
    Color brdf_subsurface   = BRDF_Disney_Subsurface(base_color, w_in, w_out, shading_normal, roughness);
    return vlerp(brdf_base_diffuse, brdf_subsurface, subsurface);
}

static inline vec3 CosWeightedHemisphere_Sample(vec3 unit_normal) // :274
{
    f32  epsilon_0 = Random_Unilateral();
    f32  epsilon_1 = Random_Unilateral();
    f32  sin_theta = sqrtf(1.0f - epsilon_0);
    f32  cos_theta = sqrtf(epsilon_0);
    f32  phi       = 2.0f * PI32 * epsilon_1;
    vec3 xyz_normal_space = {cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta};
    basis3 normal_to_world = vec3_OrthonormalBasis(unit_normal);
    return vec3_Reorient(xyz_normal_space, normal_to_world);
}

static bool Material_Disney_Diffuse_Bounce(const rtx_scene_desc* d, const rtx_material* mat, Ray* ray_in, HitInfo* hit, Color* surface_color,
                                           Color* emitted_color, Ray* ray_out) // :291
{
    vec3  ray_dir_out    = CosWeightedHemisphere_Sample(hit->unitNormal);
    vec3  shading_normal = hit->unitNormal;
    Color albedo         = Texture_ColorAt(d, mat->tex, hit->uv);
    vec3  w_in           = vmul(-1.0f, ray_in->dir);
    vec3  w_out          = ray_dir_out;
    Color brdf_diffuse   = BRDF_Diffuse(albedo, w_in, w_out, shading_normal, mat->p[P_ROUGHNESS], mat->p[P_SUBSURFACE]);
    f32   inv_pdf        = PI32;
    brdf_diffuse         = vmul(brdf_diffuse, inv_pdf);
    *surface_color       = brdf_diffuse;
    *emitted_color       = COLOR_BLACK;
    *ray_out             = Ray_Make(hit->position, ray_dir_out);
    return true;
}

static inline Color Fresnel_Schlick_Chromatic(Color r_0, f32 cos_theta) // :339
{
    return vadd(r_0, vmul(vsub(vec(1.0f, 1.0f, 1.0f), r_0), POWF(1.0f - cos_theta, 5)));
}
static inline f32 Lambda_m(vec3 w, f32 a_x, f32 a_y) // :352
{
    f32 numerator   = sqrtf(1.0f + (POWF(w.x * a_x, 2) + POWF(w.y * a_y, 2)) / POWF(w.z, 2)) - 1.0f;
    f32 denominator = 2.0f;
    return numerator / denominator;
}
static inline f32   G_m1(vec3 w, f32 a_x, f32 a_y) { return 1.0f / (1.0f + Lambda_m(w, a_x, a_y)); }            // :360
static inline f32   R_0(f32 eta) { return POWF(eta - 1.0f, 2) / POWF(eta + 1.0f, 2); }                          // :370
static inline f32   Color_Luminance(Color c) { return vmag(c) / vmag(COLOR_WHITE); }                            // color.c:105
static inline Color C_tint(Color base_color) { return vdiv(base_color, Color_Luminance(base_color)); }          // :385
static inline Color K_s(Color c_tint, f32 specular_tint) { return vlerp(COLOR_WHITE, c_tint, specular_tint); }  // :380
static inline Color C_0(Color base_color, Color k_s, f32 metallic, f32 specular, f32 eta)                       // :375
{
    return vlerp(vmul(k_s, specular * R_0(eta)), base_color, metallic);
}
static inline Color BRDF_Disney_Metal(Color base_color, vec3 w_out, vec3 w_micronormal, f32 metallic, f32 specular, f32 specular_tint,
                                      f32 a_x, f32 a_y, f32 eta) // :391
{
    Color k_s = K_s(C_tint(base_color), specular_tint);
    Color c_0 = C_0(base_color, k_s, metallic, specular, eta);
    Color f_m = Fresnel_Schlick_Chromatic(c_0, vdot(w_out, w_micronormal));
    f32   g_m = G_m1(w_out, a_x, a_y);
    return vmul(f_m, g_m);
}

static inline vec3 GGXVNDF_Sample(vec3 Ve, f32 alpha_x, f32 alpha_y, f32 U1, f32 U2) // :415
{
    vec3 Vh    = vnorm(vec(alpha_x * Ve.x, alpha_y * Ve.y, Ve.z));
    f32  lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    vec3 T1    = lensq > 0.0f ? vmul(vec(-Vh.y, Vh.x, 0.0f), (1.0f / sqrtf(lensq))) : vec(1.0f, 0.0f, 0.0f);
    vec3 T2    = vcross(Vh, T1);
    f32  r     = sqrtf(U1);
    f32  phi   = 2.0f * PI32 * U2;
    f32  t1    = r * cosf(phi);
    f32  t2    = r * sinf(phi);
    f32  sS    = 0.5f * (1.0f + Vh.z);
    t2         = (1.0f - sS) * sqrtf(1.0f - t1 * t1) + sS * t2;
    vec3 Nh    = vsum(vmul(t1, T1), vmul(t2, T2), vmul(sqrtf(maxf(0.0f, 1.0f - t1 * t1 - t2 * t2)), Vh));
    vec3 Ne    = vnorm(vec(alpha_x * Nh.x, alpha_y * Nh.y, maxf(0.0f, Nh.z)));
    return Ne;
}

static bool Material_Disney_Metal_Bounce(const rtx_scene_desc* d, const rtx_material* mat, Ray* ray_in, HitInfo* hit, Color* surface_color,
                                         Color* emitted_color, Ray* ray_out) // :446
{
    f32 aspect = sqrtf(1.0f - 0.9f * mat->p[P_ANISTROPIC]);
    f32 a_min  = 0.0001f;
    f32 a_x    = maxf(a_min, POWF(mat->p[P_ROUGHNESS], 2) / aspect);
    f32 a_y    = maxf(a_min, POWF(mat->p[P_ROUGHNESS], 2) * aspect);

    basis3 normal_to_world = vec3_OrthonormalBasis(hit->unitNormal);
    basis3 world_to_normal = vec3_OrthonormalBasis_Inverse(normal_to_world);
    vec3   ray_in_dir      = vmul(-1.0f, ray_in->dir);
    vec3   w_in            = vec3_Reorient(ray_in_dir, world_to_normal);

    f32  u1            = Random_Unilateral(); // Disney_Metal_SampleNormal :437
    f32  u2            = Random_Unilateral();
    vec3 w_micronormal = GGXVNDF_Sample(w_in, a_x, a_y, u1, u2);
    vec3 w_out         = vnorm(vec3_Reflect(vmul(-1.0f, w_in), w_micronormal));

    Color albedo = Texture_ColorAt(d, mat->tex, hit->uv);
    Color brdf   = BRDF_Disney_Metal(albedo, w_out, w_micronormal, mat->p[P_W_METALLIC], mat->p[P_SPECULAR], mat->p[P_SPECULAR_TINT],
                                     a_x, a_y, mat->p[P_ETA]);
    vec3 ray_out_dir = vec3_Reorient(w_out, normal_to_world);
    *surface_color   = brdf;
    *emitted_color   = COLOR_BLACK;
    *ray_out         = Ray_Make(hit->position, ray_out_dir);
    return true;
}

static inline f32 Fresnel_Schlick_Achromatic(vec3 w_out, vec3 w_micronormal) // :506
{
    f32 eta = 1.5f;
    f32 r_0 = POWF(eta - 1.0f, 2) / POWF(eta + 1, 2);
    return r_0 + (1.0f - r_0) * POWF(1.0f - fabsf(vdot(w_micronormal, w_out)), 5);
}
static inline f32 Lambda_c(vec3 w) // :525
{
    f32 roughness_2 = 0.25f * 0.25f;
    f32 w_l_x_2     = POWF(w.x, 2);
    f32 w_l_y_2     = POWF(w.y, 2);
    f32 w_l_z_2     = POWF(w.z, 2);
    f32 numerator   = sqrtf(1.0f + (roughness_2 * (w_l_x_2 + w_l_y_2)) / w_l_z_2) - 1.0f;
    f32 denominator = 2.0f;
    return numerator / denominator;
}
static inline f32 G_c1(vec3 w) { return 1.0f / (1.0f + Lambda_c(w)); }                       // :538
static inline f32 G_c2(vec3 w_in, vec3 w_out) { return G_c1(w_in) * G_c1(w_out); }           // :543
static inline f32 BRDF_Disney_Clearcoat(vec3 w_in, vec3 w_out, vec3 w_micronormal, vec3 w_macronormal) // :548
{
    f32 f_c         = Fresnel_Schlick_Achromatic(w_out, w_micronormal);
    f32 d_c         = 1.0f;
    f32 g_c         = G_c2(w_in, w_out);
    f32 numerator   = f_c * d_c * g_c;
    f32 denominator = fabsf(vdot(w_macronormal, w_in));
    return numerator / denominator;
}
static inline vec3 Disney_Clearcoat_SampleNormal(f32 a_g) // :564
{
    f32 u_0        = Random_Unilateral();
    f32 u_1        = Random_Unilateral();
    f32 a_g_2      = POWF(a_g, 2);
    f32 cos_h_elev = sqrtf((1.0f - powf(a_g_2, 1.0f - u_0)) / (1.0f - a_g_2));
    f32 h_azi      = 2.0f * PI32 * u_1;
    f32 sin_h_elev = sqrtf(1.0f - POWF(cos_h_elev, 2));
    f32 cos_h_azi  = cosf(h_azi);
    f32 sin_h_azi  = sinf(h_azi);
    return vec3{sin_h_elev * cos_h_azi, sin_h_elev * sin_h_azi, cos_h_elev};
}
static inline f32 Disney_Clearcoat_SampleNormal_InvPDF(vec3 w_out, vec3 w_micronormal, vec3 w_macronormal) // :586
{
    f32 d_c         = 1.0f;
    f32 numerator   = d_c * fabsf(vdot(w_macronormal, w_micronormal));
    f32 denominator = fabsf(vdot(w_micronormal, w_out));
    return denominator / numerator;
}

static bool Material_Disney_Clearcoat_Bounce(const rtx_material* mat, Ray* ray_in, HitInfo* hit, Color* surface_color, Color* emitted_color,
                                             Ray* ray_out) // :598
{
    f32    a_g             = (1.0f - mat->p[P_CLEARCOAT_GLOSS]) * 0.1f + mat->p[P_CLEARCOAT_GLOSS] * 0.001f;
    basis3 normal_to_world = vec3_OrthonormalBasis(hit->unitNormal);
    basis3 world_to_normal = vec3_OrthonormalBasis_Inverse(normal_to_world);
    vec3   ray_in_dir      = vmul(-1.0f, ray_in->dir);
    vec3   w_in            = vec3_Reorient(ray_in_dir, world_to_normal);
    vec3   w_macronormal   = vec(0.0f, 0.0f, 1.0f);
    vec3   w_micronormal   = Disney_Clearcoat_SampleNormal(a_g);
    vec3   w_out           = vnorm(vec3_Reflect(vmul(-1.0f, w_in), w_micronormal));
    f32    brdf            = BRDF_Disney_Clearcoat(w_in, w_out, w_micronormal, w_macronormal);
    f32    inv_pdf         = Disney_Clearcoat_SampleNormal_InvPDF(w_out, w_micronormal, w_macronormal);
    Color  brdf_color      = vec(brdf * inv_pdf, brdf * inv_pdf, brdf * inv_pdf);
    vec3   ray_out_dir     = vec3_Reorient(w_out, normal_to_world);
    *surface_color         = brdf_color;
    *emitted_color         = COLOR_BLACK;
    *ray_out               = Ray_Make(hit->position, ray_out_dir);
    return true;
}

static inline f32 Fresnel_Achromatic_DotProducts(f32 micronormal_dot_in, f32 micronormal_dot_out, f32 eta) // :652
{
    f32 r_s = (micronormal_dot_in - eta * micronormal_dot_out) / (micronormal_dot_in + eta * micronormal_dot_out);
    f32 r_p = (eta * micronormal_dot_in - micronormal_dot_out) / (eta * micronormal_dot_in + micronormal_dot_out);
    return 0.5f * (POWF(r_s, 2) + POWF(r_p, 2));
}
static inline f32 Fresnel_Achromatic_IncidentOnly(f32 micronormal_dot_in, f32 eta) // :669
{
    f32 micronormal_dot_out_2 = 1.0f - POWF(eta, 2) * (1.0f - POWF(micronormal_dot_in, 2));
    if (micronormal_dot_out_2 < 0.0f) {
        return 1.0f;
    }
    f32 micronormal_dot_out = sqrtf(micronormal_dot_out_2);
    return Fresnel_Achromatic_DotProducts(micronormal_dot_in, micronormal_dot_out, eta);
}
static inline Color BRDF_Disney_Glass(Color base_color, vec3 w_out, f32 a_x, f32 a_y, bool reflected) // :696
{
    if (reflected) {
        f32 g_g = G_m1(w_out, a_x, a_y);
        return vmul(base_color, g_g);
    } else {
        Color sqrt_color = vec(sqrtf(base_color.x), sqrtf(base_color.y), sqrtf(base_color.z));
        f32   g_g        = G_m1(w_out, a_x, a_y);
        return vmul(sqrt_color, g_g);
    }
}

static bool Material_Disney_Glass_Bounce(const rtx_scene_desc* d, const rtx_material* mat, Ray* ray_in, HitInfo* hit, Color* surface_color,
                                         Color* emitted_color, Ray* ray_out) // :722
{
    f32 aspect = sqrtf(1.0f - 0.9f * mat->p[P_ANISTROPIC]);
    f32 a_min  = 0.0001f;
    f32 a_x    = maxf(a_min, POWF(mat->p[P_ROUGHNESS], 2) / aspect);
    f32 a_y    = maxf(a_min, POWF(mat->p[P_ROUGHNESS], 2) * aspect);
    f32 eta    = hit->frontFace ? 1.0f / mat->p[P_ETA] : mat->p[P_ETA];

    basis3 normal_to_world = vec3_OrthonormalBasis(hit->unitNormal);
    basis3 world_to_normal = vec3_OrthonormalBasis_Inverse(normal_to_world);
    vec3   w_in            = vec3_Reorient(vmul(-1.0f, ray_in->dir), world_to_normal);

    f32  u0            = Random_Unilateral(); // Disney_Glass_SampleNormal :712
    f32  u1            = Random_Unilateral();
    vec3 w_micronormal = GGXVNDF_Sample(w_in, a_x, a_y, u0, u1);

    vec3 w_out;
    bool reflected;
    {
        f32 micronormal_dot_in = vdot(w_micronormal, w_in);
        f32 f_g                = Fresnel_Achromatic_IncidentOnly(micronormal_dot_in, eta);
        f32 u2                 = Random_Unilateral();
        if (u2 <= f_g) {
            w_out     = vec3_Reflect(vmul(-1.0f, w_in), w_micronormal);
            reflected = true;
        } else {
            w_out     = vec3_Refract(vmul(-1.0f, w_in), w_micronormal, eta);
            reflected = false;
        }
    }
    w_out            = vnorm(w_out);
    Color albedo     = Texture_ColorAt(d, mat->tex, hit->uv);
    Color brdf_color = BRDF_Disney_Glass(albedo, w_out, a_x, a_y, reflected);
    vec3  ray_out_dir = vec3_Reorient(w_out, normal_to_world);
    *surface_color    = brdf_color;
    *emitted_color    = COLOR_BLACK;
    *ray_out          = Ray_Make(hit->position, ray_out_dir);
    return true;
}

static inline Color C_sheen(Color base_color, f32 sheen_tint) { return vlerp(COLOR_WHITE, C_tint(base_color), sheen_tint); } // :794
static inline Color BRDF_Disney_Sheen(Color base_color, vec3 w_out, vec3 w_micronormal, f32 sheen_tint) // :800
{
    return vmul(C_sheen(base_color, sheen_tint), POWF(1.0f - fabsf(vdot(w_micronormal, w_out)), 5));
}

static bool Material_Disney_Sheen_Bounce(const rtx_scene_desc* d, const rtx_material* mat, Ray* ray_in, HitInfo* hit, Color* surface_color,
                                         Color* emitted_color, Ray* ray_out) // :805
{
    vec3  w_in          = vmul(-1.0, ray_in->dir);
    vec3  w_out         = CosWeightedHemisphere_Sample(hit->unitNormal);
    vec3  w_micronormal = HalfVector(w_in, w_out);
    Color albedo        = Texture_ColorAt(d, mat->tex, hit->uv);
    Color brdf_color    = BRDF_Disney_Sheen(albedo, w_out, w_micronormal, mat->p[P_SHEEN_TINT]);
    f32   inv_pdf       = PI32;
    *surface_color      = vmul(brdf_color, inv_pdf);
    *emitted_color      = COLOR_BLACK;
    *ray_out            = Ray_Make(hit->position, w_out);
    return true;
}

static bool Material_Disney_BSDF_Bounce(const rtx_scene_desc* d, const rtx_material* mat, Ray* ray_in, HitInfo* hit, Color* surface_color,
                                        Color* emitted_color, Ray* ray_out) // :868
{
    f32 weight_diffuse, weight_sheen, weight_metal, weight_clearcoat, weight_glass;
    f32 w_specular = mat->p[P_W_SPECULAR], w_metallic = mat->p[P_W_METALLIC], w_sheen = mat->p[P_W_SHEEN], w_clearcoat = mat->p[P_W_CLEARCOAT];
    if (hit->frontFace) {
        weight_diffuse   = (1.0f - w_specular) * (1.0f - w_metallic);
        weight_sheen     = (1.0f - w_metallic) * w_sheen;
        weight_metal     = (1.0f - w_specular * (1.0f - w_metallic));
        weight_clearcoat = 0.25f * w_clearcoat;
        weight_glass     = (1.0f - w_metallic) * w_specular;
    } else {
        weight_diffuse   = 0.0f;
        weight_sheen     = 0.0f;
        weight_metal     = 0.0f;
        weight_clearcoat = 0.0f;
        weight_glass     = (1.0f - w_metallic) * w_specular;
    }
    f32 total_weight = weight_diffuse + weight_sheen + weight_metal + weight_clearcoat + weight_glass;
    f32 sel_interval = Random_Unilateral() * total_weight;

    struct RngInterval {
        f32 lower, upper;
    };
    RngInterval diffuse   = {0.0f, weight_diffuse};
    RngInterval sheen     = {diffuse.upper, diffuse.upper + weight_sheen};
    RngInterval metal     = {sheen.upper, sheen.upper + weight_metal};
    RngInterval clearcoat = {metal.upper, metal.upper + weight_clearcoat};
    RngInterval glass     = {clearcoat.upper, clearcoat.upper + weight_glass};

    if (diffuse.lower <= sel_interval && sel_interval < diffuse.upper) {
        return Material_Disney_Diffuse_Bounce(d, mat, ray_in, hit, surface_color, emitted_color, ray_out);
    } else if (sheen.lower <= sel_interval && sel_interval < sheen.upper) {
        return Material_Disney_Sheen_Bounce(d, mat, ray_in, hit, surface_color, emitted_color, ray_out);
    } else if (metal.lower <= sel_interval && sel_interval < metal.upper) {
        return Material_Disney_Metal_Bounce(d, mat, ray_in, hit, surface_color, emitted_color, ray_out);
    } else if (clearcoat.lower <= sel_interval && sel_interval < clearcoat.upper) {
        return Material_Disney_Clearcoat_Bounce(mat, ray_in, hit, surface_color, emitted_color, ray_out);
    } else if (glass.lower <= sel_interval && sel_interval < glass.upper) {
        return Material_Disney_Glass_Bounce(d, mat, ray_in, hit, surface_color, emitted_color, ray_out);
    } else {
        *surface_color = COLOR_BLACK;
        *emitted_color = COLOR_BLACK;
        return false;
    }
}

static bool Material_Bounce(const rtx_scene_desc* d, const rtx_material* m, Ray* ray_in, HitInfo* hit, Color* surface_color,
                            Color* emitted_color, Ray* ray_out) // world/object.c:64
{
    switch (m->type) {
        case RTX_MAT_DIFFUSE: return Material_Diffuse_Bounce(d, m, ray_in, hit, surface_color, emitted_color, ray_out);
        case RTX_MAT_METAL: return Material_Metal_Bounce(d, m, ray_in, hit, surface_color, emitted_color, ray_out);
        case RTX_MAT_DIELECTRIC: return Material_Dielectric_Bounce(d, m, ray_in, hit, surface_color, emitted_color, ray_out);
        case RTX_MAT_DIFFUSE_LIGHT: return Material_DiffuseLight_Bounce(d, m, hit, emitted_color);
        case RTX_MAT_SKYBOX: return Material_Skybox_Bounce(d, ray_in, hit, surface_color, emitted_color, ray_out);
        case RTX_MAT_DISNEY_DIFFUSE: return Material_Disney_Diffuse_Bounce(d, m, ray_in, hit, surface_color, emitted_color, ray_out);
        case RTX_MAT_DISNEY_METAL: return Material_Disney_Metal_Bounce(d, m, ray_in, hit, surface_color, emitted_color, ray_out);
        case RTX_MAT_DISNEY_GLASS: return Material_Disney_Glass_Bounce(d, m, ray_in, hit, surface_color, emitted_color, ray_out);
        case RTX_MAT_DISNEY_CLEARCOAT: return Material_Disney_Clearcoat_Bounce(m, ray_in, hit, surface_color, emitted_color, ray_out);
        case RTX_MAT_DISNEY_SHEEN: return Material_Disney_Sheen_Bounce(d, m, ray_in, hit, surface_color, emitted_color, ray_out);
        case RTX_MAT_DISNEY_BSDF: return Material_Disney_BSDF_Bounce(d, m, ray_in, hit, surface_color, emitted_color, ray_out);
    }
    return false;
}

// ---- world/camera.c:34 ----
static Ray Camera_GetRay(const rtx_camera* cam, f32 u, f32 v)
{
    vec2 rd     = Random_InDisc(cam->lens_radius);
    vec3 offset = vec3{u * rd.x, v * rd.y, 0};
    vec3 origin = vec(cam->origin[0], cam->origin[1], cam->origin[2]);
    vec3 blc    = vec(cam->bottom_left[0], cam->bottom_left[1], cam->bottom_left[2]);
    vec3 hor    = vec(cam->horizontal[0], cam->horizontal[1], cam->horizontal[2]);
    vec3 ver    = vec(cam->vertical[0], cam->vertical[1], cam->vertical[2]);

    point3 rayOrigin = vadd(origin, offset);
    vec3   rayDir    = vsub(vadd(blc, vadd(vmul(hor, u), vmul(ver, v))), vadd(origin, offset));
    return Ray_Make(rayOrigin, rayDir);
}

// ---- rt/renderer.c:54 ----
struct RenderState {
    Scene sc;
    u64   rays_at_depth[RTX_MAX_DEPTH + 1];
    size_t max_depth;
};

static Color RayColor(RenderState* rs, Ray* ray, size_t depth)
{
    if (depth == 0) {
        return COLOR_BLACK;
    }
    rs->rays_at_depth[rs->max_depth - depth]++;
    Prim    objHit = {0xFFFFFFFFu};
    HitInfo hit;
    if (Scene_ClosestHit(&rs->sc, ray, &objHit, &hit)) {
        Ray   bouncedRay;
        Color surfaceColor = COLOR_BLACK;
        Color emittedColor = COLOR_BLACK;
        const rtx_material* m = &rs->sc.d->materials[prim_material(rs->sc.d, objHit.ref)];
        if (Material_Bounce(rs->sc.d, m, ray, &hit, &surfaceColor, &emittedColor, &bouncedRay)) {
            // Color_BrightenBy(emitted, Color_Tint(surface, RayColor(...)))  (color.c:63,82)
            return vadd(emittedColor, vmul(surfaceColor, RayColor(rs, &bouncedRay, depth - 1)));
        } else {
            return emittedColor;
        }
    }
    return Skybox_ColorAt(rs->sc.d, ray->dir); // Scene_Get_SkyColor, scene.c:173
}

// ---- gfx/color.c:18,49 ----
static inline uint8_t LinearTosRGB(f32 channel)
{
    f32 linear = clampf(channel, 0.0f, 1.0f);
    if (linear < 0.0031308f) {
        f32 srgb = linear * 12.92f;
        return (uint8_t)(255.0f * srgb);
    } else {
        f32 srgb = 1.055f * powf(linear, 1.0f / 2.4f) - 0.055f;
        return (uint8_t)(255.0f * srgb);
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int oracle_trace(const rtx_scene_desc* desc, size_t n, const float* o, const float* dxyz, int unit_dir, uint32_t* pid, float* t,
                            float* pos, float* nrm, float* uv, uint8_t* front, uint64_t* visits3, uint32_t* steps_per_ray)
{
    return oracle_trace_culled(desc, NULL, NULL, NULL, n, o, dxyz, unit_dir, pid, t, pos, nrm, uv, front, visits3, steps_per_ray);
}

extern "C" int oracle_trace_culled(const rtx_scene_desc* desc, const uint16_t* qbox6, const float* info14, const uint8_t* node_flags, size_t n, const float* o,
                                   const float* dxyz, int unit_dir, uint32_t* pid, float* t, float* pos, float* nrm, float* uv,
                                   uint8_t* front, uint64_t* visits3, uint32_t* steps_per_ray)
{
    Scene sc;
    memset(&sc, 0, sizeof sc);
    sc.d = desc;
    Cull  cull;
    if (qbox6) {
        cull.qbox   = qbox6;
        cull.flags  = node_flags;
        cull.c      = vec(info14[0], info14[1], info14[2]);
        cull.ro2    = info14[3];
        cull.kt     = info14[4];
        cull.lo     = vec(info14[5], info14[6], info14[7]);
        cull.scale  = vec(info14[8], info14[9], info14[10]);
        sc.cull     = &cull;
    }
    for (size_t i = 0; i < n; i++) {
        const u64 before = sc.nInner + sc.nLeaf + sc.nPrim + sc.nCulled;
        vec3 org = vec(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        vec3 dir = vec(dxyz[3 * i], dxyz[3 * i + 1], dxyz[3 * i + 2]);
        Ray  ray = unit_dir ? Ray_FromUnit(org, dir) : Ray_Make(org, dir);
        if (sc.cull) cull_begin_ray(&cull, &ray);
        Prim    obj = {0xFFFFFFFFu};
        HitInfo hit;
        memset(&hit, 0, sizeof hit);
        bool h = Scene_ClosestHit(&sc, &ray, &obj, &hit);
        if (!h) memset(&hit, 0, sizeof hit);
        if (pid) pid[i] = h ? prim_id(desc, obj.ref) : RTX_ID_MISS;
        if (t) t[i] = hit.tIntersect;
        if (pos) pos[3 * i] = hit.position.x, pos[3 * i + 1] = hit.position.y, pos[3 * i + 2] = hit.position.z;
        if (nrm) nrm[3 * i] = hit.unitNormal.x, nrm[3 * i + 1] = hit.unitNormal.y, nrm[3 * i + 2] = hit.unitNormal.z;
        if (uv) uv[2 * i] = hit.uv.x, uv[2 * i + 1] = hit.uv.y;
        if (front) front[i] = h && hit.frontFace ? 1 : 0;
        if (steps_per_ray) steps_per_ray[i] = (u32)(sc.nInner + sc.nLeaf + sc.nPrim + sc.nCulled - before);
    }
    if (visits3) {
        visits3[0] = sc.nInner, visits3[1] = sc.nLeaf, visits3[2] = sc.nPrim;
        if (qbox6) visits3[3] = sc.nCulled; // 4 counters in checker mode
    }
    return 0;
}

// The speculative replay (checker of the GPU library's k_trace_long; not part of the reference).  k_trace_long
// walks the pending far subtrees of a long walk ahead of time, on other lanes, and takes — of all leaves that accept a hit —
// the one that comes first in the literal visiting order.  Here the (culled) walk goes through EVERY subtree the sequential
// walk could reach, whatever an earlier leaf found (a superset of what any schedule of the kernel visits; subtrees that come
// after the best leaf noted so far are skipped, as the kernel drops them), and the answer is the accepting leaf with the
// smallest key.  Must equal oracle_trace on every ray.  steps (may be NULL): node and primitive visits per ray.
extern "C" int oracle_trace_speculative(const rtx_scene_desc* desc, const uint16_t* qbox6, const float* info14, const uint8_t* node_flags, size_t n,
                                        const float* o, const float* dxyz, int unit_dir, uint32_t* pid, float* t, float* pos, float* nrm, float* uv,
                                        uint8_t* front, uint32_t* steps)
{
    Scene sc;
    memset(&sc, 0, sizeof sc);
    sc.d = desc;
    Cull cull;
    if (qbox6) {
        cull.qbox  = qbox6;
        cull.flags = node_flags;
        cull.c     = vec(info14[0], info14[1], info14[2]);
        cull.ro2   = info14[3];
        cull.kt    = info14[4];
        cull.lo    = vec(info14[5], info14[6], info14[7]);
        cull.scale = vec(info14[8], info14[9], info14[10]);
        sc.cull    = &cull;
    }
    for (size_t i = 0; i < n; i++) {
        vec3 org = vec(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        vec3 dir = vec(dxyz[3 * i], dxyz[3 * i + 1], dxyz[3 * i + 2]);
        Ray  ray = unit_dir ? Ray_FromUnit(org, dir) : Ray_Make(org, dir);
        if (sc.cull) cull_begin_ray(&cull, &ray);
        const u64 before = sc.nInner + sc.nLeaf + sc.nPrim + sc.nCulled;
        sc.speculative = true, sc.spec_any = false;
        Prim    dummy = {0xFFFFFFFFu};
        HitInfo dummy_hit;
        memset(&dummy_hit, 0, sizeof dummy_hit);
        (void)KDTree_HitAt(&sc, &ray, &dummy, &dummy_hit); // reports nothing: every accepting leaf is only noted
        sc.speculative = false;
        // Scene_ClosestHit's merge with the unbounded list (scene.c:100-143)
        Prim    obj_u = {0xFFFFFFFFu};
        HitInfo hit_u;
        memset(&hit_u, 0, sizeof hit_u);
        hit_u.tIntersect = INF;
        const bool hu = Scene_ClosestHitInArray(&sc, &ray, &obj_u, &hit_u);
        Prim       obj = {sc.spec_obj};
        HitInfo    hit = sc.spec_hit;
        bool       h   = sc.spec_any;
        if (hu && (!h || hit_u.tIntersect < hit.tIntersect)) obj = obj_u, hit = hit_u, h = true;
        if (!h) memset(&hit, 0, sizeof hit);
        if (pid) pid[i] = h ? prim_id(desc, obj.ref) : RTX_ID_MISS;
        if (t) t[i] = hit.tIntersect;
        if (pos) pos[3 * i] = hit.position.x, pos[3 * i + 1] = hit.position.y, pos[3 * i + 2] = hit.position.z;
        if (nrm) nrm[3 * i] = hit.unitNormal.x, nrm[3 * i + 1] = hit.unitNormal.y, nrm[3 * i + 2] = hit.unitNormal.z;
        if (uv) uv[2 * i] = hit.uv.x, uv[2 * i + 1] = hit.uv.y;
        if (front) front[i] = h && hit.frontFace ? 1 : 0;
        if (steps) steps[i] = (u32)(sc.nInner + sc.nLeaf + sc.nPrim + sc.nCulled - before);
    }
    return 0;
}

struct RenderJob {
    const rtx_scene_desc*    desc;
    const rtx_camera*        cam;
    const rtx_render_params* p;
    float*                   cum;
    uint8_t*                 bgr;
    int                      tid, nthreads;
    u64                      rays_at_depth[RTX_MAX_DEPTH + 1];
    const uint32_t*          pixels; // NULL: the frame's rows
    size_t                   n_pixels;
};

static bool row_assigned(const rtx_render_params* p, uint32_t y)
{
    if (p->band_rows == 0 || p->band_world <= 1) return true;
    return (y / p->band_rows) % p->band_world == p->band_rank;
}

// RenderPixel's sample loop (renderer.c:98-108) for samples [sb, se) of pixel (xx, yy)
static Color render_pixel_sum(RenderState* rs, const rtx_camera* cam, const rtx_render_params* p, size_t xx, size_t yy, uint32_t sb, uint32_t se)
{
    const size_t image_width = p->width, image_height = p->height;
    Color        cum_color   = vec(0, 0, 0);
    for (size_t samples = sb; samples < se; samples++) {
        u64 pixel = (u64)yy * image_width + xx;
        Random_Seed(p->seed_lo ^ (pixel * p->spp + samples), p->seed_hi);
        f32 horizontal_fraction, vertical_fraction;
        if (p->flags & RTX_RENDER_PIXEL_CENTRES) {
            horizontal_fraction = (xx + 0.5f) / (f32)(image_width - 1);
            vertical_fraction   = (yy + 0.5f) / (f32)(image_height - 1);
        } else {
            horizontal_fraction = (xx + Random_Unilateral()) / (f32)(image_width - 1);
            vertical_fraction   = (yy + Random_Unilateral()) / (f32)(image_height - 1);
        }
        Ray   ray       = Camera_GetRay(cam, horizontal_fraction, vertical_fraction);
        Color ray_color = RayColor(rs, &ray, p->max_depth);
        cum_color       = vadd(cum_color, ray_color);
    }
    return cum_color;
}

static void* render_rows(void* arg)
{
    RenderJob*               j = (RenderJob*)arg;
    const rtx_render_params* p = j->p;
    RenderState rs;
    memset(&rs, 0, sizeof rs);
    rs.sc.d      = j->desc;
    rs.max_depth = p->max_depth;
    uint32_t sb = p->sample_begin, se = p->sample_end;
    if (sb == 0 && se == 0) se = p->spp;
    const size_t image_width = p->width, image_height = p->height;
    if (j->pixels) { // a list of pixels of the frame instead of its rows: entry i -> cum[3 i]
        for (size_t i = j->tid; i < j->n_pixels; i += j->nthreads) {
            Color c = render_pixel_sum(&rs, j->cam, p, j->pixels[i] % image_width, j->pixels[i] / image_width, sb, se);
            j->cum[3 * i + 0] = c.x, j->cum[3 * i + 1] = c.y, j->cum[3 * i + 2] = c.z;
        }
        memcpy(j->rays_at_depth, rs.rays_at_depth, sizeof rs.rays_at_depth);
        return NULL;
    }
    for (size_t yy = j->tid; yy < image_height; yy += j->nthreads) {
        if (!row_assigned(p, (uint32_t)yy)) continue;
        for (size_t xx = 0; xx < image_width; xx++) { // RenderPixel, renderer.c:88
            Color  cum_color  = render_pixel_sum(&rs, j->cam, p, xx, yy, sb, se);
            size_t i          = yy * image_width + xx;
            j->cum[3 * i + 0] = cum_color.x, j->cum[3 * i + 1] = cum_color.y, j->cum[3 * i + 2] = cum_color.z;
            if (j->bgr) {
                Color avg_color   = vdiv(cum_color, (f32)p->spp);
                j->bgr[3 * i + 0] = LinearTosRGB(avg_color.z);
                j->bgr[3 * i + 1] = LinearTosRGB(avg_color.y);
                j->bgr[3 * i + 2] = LinearTosRGB(avg_color.x);
            }
        }
    }
    memcpy(j->rays_at_depth, rs.rays_at_depth, sizeof rs.rays_at_depth);
    return NULL;
}

static int run_render_jobs(const rtx_scene_desc* desc, const rtx_camera* cam, const rtx_render_params* p, float* cum_out, uint8_t* bgr_out,
                           uint64_t* rays_at_depth, int n_threads, const uint32_t* pixels, size_t n_pixels)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    std::vector<RenderJob> jobs(n_threads);
    std::vector<pthread_t> th(n_threads);
    for (int i = 0; i < n_threads; i++) {
        jobs[i] = RenderJob{desc, cam, p, cum_out, bgr_out, i, n_threads, {0}, pixels, n_pixels};
        pthread_attr_t attr;
        pthread_attr_init(&attr);
        pthread_attr_setstacksize(&attr, 16 << 20); // the kd recursion + RayColor recursion are deep
        if (pthread_create(&th[i], &attr, render_rows, &jobs[i]) != 0) return -2;
    }
    if (rays_at_depth) memset(rays_at_depth, 0, RTX_MAX_DEPTH * sizeof(uint64_t));
    for (int i = 0; i < n_threads; i++) {
        pthread_join(th[i], NULL);
        if (rays_at_depth)
            for (int k = 0; k < RTX_MAX_DEPTH; k++) rays_at_depth[k] += jobs[i].rays_at_depth[k];
    }
    return 0;
}

extern "C" int oracle_render(const rtx_scene_desc* desc, const rtx_camera* cam, const rtx_render_params* p, float* cum_out, uint8_t* bgr_out,
                             uint64_t* rays_at_depth, int n_threads)
{
    if (!desc || !cam || !p || !cum_out || p->max_depth > RTX_MAX_DEPTH) return -1;
    return run_render_jobs(desc, cam, p, cum_out, bgr_out, rays_at_depth, n_threads, NULL, 0);
}

// RenderPixel for a list of pixels (index y * W + x) of the frame params describes: cum_out[3 i] = the f32 sum of pixel i's samples
extern "C" int oracle_render_pixels(const rtx_scene_desc* desc, const rtx_camera* cam, const rtx_render_params* p, size_t n, const uint32_t* pixel_index,
                                    float* cum_out, uint64_t* rays_at_depth, int n_threads)
{
    if (!desc || !cam || !p || !cum_out || !pixel_index || p->max_depth > RTX_MAX_DEPTH) return -1;
    for (size_t i = 0; i < n; i++)
        if (pixel_index[i] >= (u64)p->width * p->height) return -1;
    if (n == 0) return 0;
    return run_render_jobs(desc, cam, p, cum_out, NULL, rays_at_depth, n_threads, pixel_index, n);
}

extern "C" int oracle_camera_rays(const rtx_camera* cam, const rtx_render_params* p, size_t n, const uint32_t* pixel_index, const uint32_t* sample,
                                  float* o, float* d, uint32_t* state)
{
    for (size_t i = 0; i < n; i++) {
        size_t xx = pixel_index[i] % p->width, yy = pixel_index[i] / p->width;
        Random_Seed(p->seed_lo ^ ((u64)pixel_index[i] * p->spp + sample[i]), p->seed_hi);
        f32 hf, vf;
        if (p->flags & RTX_RENDER_PIXEL_CENTRES) {
            hf = (xx + 0.5f) / (f32)(p->width - 1);
            vf = (yy + 0.5f) / (f32)(p->height - 1);
        } else {
            hf = (xx + Random_Unilateral()) / (f32)(p->width - 1);
            vf = (yy + Random_Unilateral()) / (f32)(p->height - 1);
        }
        Ray ray = Camera_GetRay(cam, hf, vf);
        o[3 * i] = ray.origin.x, o[3 * i + 1] = ray.origin.y, o[3 * i + 2] = ray.origin.z;
        d[3 * i] = ray.dir.x, d[3 * i + 1] = ray.dir.y, d[3 * i + 2] = ray.dir.z;
        if (state) memcpy(state + 4 * i, s, 16);
    }
    return 0;
}

extern "C" int oracle_rng_unilateral(uint64_t lo, uint64_t hi, size_t n, float* out)
{
    Random_Seed(lo, hi);
    for (size_t i = 0; i < n; i++) out[i] = Random_Unilateral();
    return 0;
}

extern "C" int oracle_material_bounce(const rtx_scene_desc* desc, uint32_t mat_index, size_t n, const float* dir, const float* pos,
                                      const float* nrm, const float* uv, const uint8_t* front, uint32_t* state, float* surf, float* emit,
                                      float* oo, float* od, uint8_t* cont)
{
    if (mat_index >= desc->n_materials) return -1;
    for (size_t i = 0; i < n; i++) {
        Ray ray;
        memset(&ray, 0, sizeof ray);
        ray.dir = vec(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        HitInfo hit;
        hit.position   = vec(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        hit.unitNormal = vec(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
        hit.uv         = vec2{uv[2 * i], uv[2 * i + 1]};
        hit.tIntersect = 1.0f;
        hit.frontFace  = front[i] != 0;
        memcpy(s, state + 4 * i, 16);
        Color sc = COLOR_BLACK, ec = COLOR_BLACK;
        Ray   ro;
        memset(&ro, 0, sizeof ro);
        bool c = Material_Bounce(desc, &desc->materials[mat_index], &ray, &hit, &sc, &ec, &ro);
        surf[3 * i] = sc.x, surf[3 * i + 1] = sc.y, surf[3 * i + 2] = sc.z;
        emit[3 * i] = ec.x, emit[3 * i + 1] = ec.y, emit[3 * i + 2] = ec.z;
        oo[3 * i] = ro.origin.x, oo[3 * i + 1] = ro.origin.y, oo[3 * i + 2] = ro.origin.z;
        od[3 * i] = ro.dir.x, od[3 * i + 1] = ro.dir.y, od[3 * i + 2] = ro.dir.z;
        cont[i]   = c ? 1 : 0;
        memcpy(state + 4 * i, s, 16);
    }
    return 0;
}

extern "C" int oracle_skybox_color(const rtx_scene_desc* desc, size_t n, const float* dir, float* rgb)
{
    for (size_t i = 0; i < n; i++) {
        Color c = Skybox_ColorAt(desc, vec(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
    }
    return 0;
}

extern "C" int oracle_resolve(const float* cum, size_t n_pixels, uint32_t spp, uint8_t* bgr)
{
    for (size_t i = 0; i < n_pixels; i++) {
        Color avg = vdiv(vec(cum[3 * i], cum[3 * i + 1], cum[3 * i + 2]), (f32)spp);
        bgr[3 * i + 0] = LinearTosRGB(avg.z);
        bgr[3 * i + 1] = LinearTosRGB(avg.y);
        bgr[3 * i + 2] = LinearTosRGB(avg.x);
    }
    return 0;
}
