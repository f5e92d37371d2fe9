// librt_hip.so — the C ABI of include/rtx.h on top of hand-written gfx950 kernels.
//
// Wavefront path tracer: the frame is cut into slabs of camera paths that live
// in HBM as SoA queues; every bounce is  closest-hit kernel → shade/bounce kernel, the latter
// compacting the surviving paths into the other queue with a wave ballot + one atomic per block,
// so the next closest-hit pass again reads dense, coalesced ray records.  Each path writes its
// final radiance exactly once (the reference's RayColor recursion only ever adds emission at the
// end of a path: src/rt/renderer.c:54-76) into a per-path result slot; a resolve kernel then sums
// the spp results of a pixel sequentially in sample order, like RenderPixel (renderer.c:98-110),
// which keeps the frame deterministic — no float atomics anywhere.
//
// There is no CPU fallback in this library: without a HIP device every entry point fails with
// RTX_E_NO_DEVICE.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "rtx.h"
#include "rtx_device.h"

#define RTX_MAX_LANES 4 // slabs of a frame in flight at once (rtx_render_device)
#define LONG_CAP (1u << 19) // entries of a long list (k_trace_long): a resident grid holds at most 8192 waves x 64 walks

// ray binning between bounce passes (see k_bin_count): cells per axis = 2^BIN_CELL_BITS, times 8 direction octants
#ifndef BIN_CELL_BITS
#define BIN_CELL_BITS 6 // 64^3 cells x 8 octants = 2^21 bins (32^3: +0.75 ms per bench frame; spread3 takes at most 6 bits)
#endif
#ifndef BIN_DIR_EXTRA
#define BIN_DIR_EXTRA 0 // 2: the direction's dominant axis joins the octant in the key (experiment)
#endif
#define BIN_BITS (3 * BIN_CELL_BITS + 3 + BIN_DIR_EXTRA)

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHECK(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(RTX_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Tuning knobs.  Every switch the render loop has — most exist so that a test can render a frame both ways and compare
// the bits — lives in ONE process-wide table.  The environment (RTX_<NAME>) is read once, when the table is first used,
// and only sets the defaults; after that a call's behaviour depends on its arguments and on rtx_tune_set, never on the
// environment of the moment.  Change a knob only while no call is in flight.
// ---------------------------------------------------------------------------------------------
#ifndef LONG_AFTER_DEFAULT
#define LONG_AFTER_DEFAULT (-1) // off until it has been measured
#endif
#ifndef TAIL_MAX_DEFAULT
#define TAIL_MAX_DEFAULT (2ll << 20) // sweep on a rank's share of the bench frame at N = 8 and 4 (tools/dbg/share8.py): longer queues are better off with a pass per depth
#endif
enum { K_OVERLAP, K_SORT_DEPTH, K_SORT_MIN, K_SORT_EVERY, K_TAIL_MAX, K_TAIL_SHADE_MIN, K_JUMP_TABLE, K_RAYGEN_LEAN, K_EARLY_MISS, K_FAST_DIV,
       K_BAND_TILES, K_TRACE_CHUNK, K_AHEAD, K_TILE_INTERLEAVE, K_LONG_AFTER, K_LONG_ALL, K_COUNT };
struct Knob {
    const char* name;
    long long   dflt, value;
    long long   lo, hi; // accepted range (rtx_tune_set refuses the rest: several knobs reach kernels as 32-bit launch arguments)
};
static Knob g_knobs[K_COUNT] = {
    {"OVERLAP", 0, 0, 0, 3},                  // 1..3: the assigned pixels go through the bounce loop as 2..4 parts on as many streams
    {"SORT_DEPTH", 2, 0, 0, RTX_MAX_DEPTH},   // bin the queues that feed depth >= this; 0 = never
    {"SORT_MIN", 1500000, 0, 0, 1ll << 40},   // ... if they hold at least this many rays
    {"SORT_EVERY", 1, 0, 1, 1ll << 20},       // ... every k-th eligible depth
    {"TAIL_MAX", TAIL_MAX_DEFAULT, 0, 0, 1ll << 40}, // longest queue handed to the tail launch; 0 = never
    {"TAIL_SHADE_MIN", 16, 0, 1, 64},         // lanes that wait before the tail launch shades in place
    {"JUMP_TABLE", 1, 0, 0, 1},               // 0: the literal 128-step seed jump in raygen
    {"RAYGEN_LEAN", 1, 0, 0, 1},              // 0: raygen stores whole path records
    {"EARLY_MISS", 1, 0, 0, 1},               // 0: every bounce ray goes through the closest-hit kernel
    {"FAST_DIV", 1, 0, 0, 1},                 // 0: plain divisions in the index decode
    {"BAND_TILES", 1, 0, 0, 1},               // 0: 32 x 32 path tiles under the band split too
    {"TRACE_CHUNK", 0, 0, 0, 4096},           // rays per work grab of the closest-hit kernel (16 at least); 0 = by queue length
    {"AHEAD", 1, 0, 0, 1},                    // 0: a depth's closest-hit pass is launched only after the host has read the queue's length
    {"TILE_INTERLEAVE", 4096, 0, 0, 1ll << 40}, // camera pass: the queue's eight parts (one per XCD) are runs of this many rays dealt out in turn; 0: its eighths
    {"LONG_AFTER", LONG_AFTER_DEFAULT, 0, -1, 1ll << 20}, // walks still in flight this many iterations after the queue has run dry go to k_trace_long; -1 = never
    {"LONG_ALL", 0, 0, 0, 1},                 // 1: EVERY ray goes through k_trace_long instead of the persistent kernel (the tests' mode)
};
static std::atomic<int> g_multi_in_flight{0}; // rtx_render_multi calls running: the table is read-only meanwhile
static std::once_flag g_knobs_once;
static void knobs_init()
{
    std::call_once(g_knobs_once, [] {
        for (Knob& k : g_knobs) {
            char env[64];
            snprintf(env, sizeof env, "RTX_%s", k.name);
            const char* e = getenv(env);
            if (e) { // the environment at load time sets the default (out of range: the built-in default stands, loudly)
                const long long v = atoll(e);
                if (v >= k.lo && v <= k.hi) k.dflt = v;
                else fprintf(stderr, "librt_hip: %s=%s is outside [%lld, %lld]; ignored\n", env, e, k.lo, k.hi);
            }
            k.value = k.dflt;
        }
    });
}
static long long knob(int k)
{
    knobs_init();
    return g_knobs[k].value;
}
static_assert(sizeof g_knobs / sizeof g_knobs[0] == K_COUNT, "one table entry per knob, in the order of the enum");
static Knob* find_knob(const char* name)
{
    knobs_init();
    if (name)
        for (Knob& k : g_knobs)
            if (!strcmp(k.name, name)) return &k;
    return nullptr;
}
extern "C" int rtx_tune_set(const char* name, long long value)
{
    Knob* k = find_knob(name);
    if (!k) return fail(RTX_E_INVALID, "unknown tuning knob '%s'", name ? name : "(null)");
    if (value < k->lo || value > k->hi) return fail(RTX_E_INVALID, "tuning knob %s = %lld is outside [%lld, %lld]", k->name, value, k->lo, k->hi);
    if (g_multi_in_flight.load() != 0) return fail(RTX_E_INVALID, "tuning knobs are read-only while rtx_render_multi runs");
    k->value = value;
    return RTX_OK;
}
extern "C" int rtx_tune_get(const char* name, long long* value)
{
    Knob* k = find_knob(name);
    if (!k || !value) return fail(RTX_E_INVALID, "unknown tuning knob '%s'", name ? name : "(null)");
    *value = k->value;
    return RTX_OK;
}
extern "C" void rtx_tune_reset(void)
{
    knobs_init();
    if (g_multi_in_flight.load() != 0) return; // read-only while rtx_render_multi runs
    for (Knob& k : g_knobs) k.value = k.dflt;
}

extern "C" int         rtx_abi_version(void) { return RTX_ABI_VERSION; }
extern "C" const char* rtx_last_error(void) { return g_err; }

extern "C" int rtx_device_count(int* count)
{
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        if (count) *count = 0;
        return fail(RTX_E_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    }
    if (count) *count = n;
    return RTX_OK;
}

// ---------------------------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------------------------
struct rtx_scene {
    int                device;
    SceneDev           dev;
    std::vector<void*> allocs;
    uint64_t           bytes;
    // per-scene scratch reused across calls
    uint64_t* d_counters; // (second lane: + 80) [0..3] visits (inner, leaf, prim, culled), [4] shaded hits, [6] queue count, [16..79] work counters of the persistent kernels
    hipEvent_t ev[2 * 1024];
    int        n_ev;
    // render workspace (path queues, hit records, per-path results): grown on demand, kept between
    // frames so that a render loop does not pay hipMalloc/hipFree of gigabytes per frame
    void*    ws;
    uint64_t ws_bytes;
    // second lane of rtx_render_device (two slabs in flight on two streams, see there)
    const uint4*        d_jump_tab; // Random_Seed's jump as 8 x 256 lookups (FrameGeom::jump_tab)
    hipStream_t         aux_stream[RTX_MAX_LANES - 1];
    hipEvent_t          ev_fork, ev_join[RTX_MAX_LANES - 1];
    unsigned long long* h_alive; // pinned: the surviving-path counts of the lanes
    bool     cull_useful; // the scene's boxes are tight enough for the culled kernel to pay (build_cull_data)
    uint32_t* d_long;     // RTX_MAX_LANES long lists of LONG_CAP entries (k_trace_long), one per lane of the render loop
    uint32_t  tree_depth; // desc.max_depth
    void*    multi_buf[2];   // rtx_render_multi: this scene's share of the frame (f32 x 3 x W x H) and, on the root, the staging of the peers' shares
    uint64_t multi_bytes[2];
    hipStream_t multi_stream;
    unsigned char* noalloc_scratch; // this device's copy of the static scratch array of rtx_closest_hit_noalloc (hipGetSymbolAddress, at create)
    unsigned persist_blocks[10]; // resident grid of the persistent kernels: [count + 2 * (variant - 2)], [8 + variant - 4]: tail (0 = not queried yet)
};

static int workspace(rtx_scene* sc, uint64_t bytes, char** out)
{
    if (sc->ws_bytes < bytes) {
        if (sc->ws) (void)hipFree(sc->ws);
        sc->ws       = nullptr;
        sc->ws_bytes = 0;
        hipError_t e = hipMalloc(&sc->ws, bytes);
        if (e != hipSuccess) return fail(RTX_E_NOMEM, "cannot allocate a %llu-byte render workspace: %s", (unsigned long long)bytes, hipGetErrorString(e));
        sc->ws_bytes = bytes;
    }
    *out = (char*)sc->ws;
    return RTX_OK;
}

template <typename T>
static int upload(rtx_scene* sc, const T* host, size_t count, const T** out)
{
    *out = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    void*  p     = nullptr;
    HIPCHECK(hipMalloc(&p, bytes));
    sc->allocs.push_back(p);
    sc->bytes += bytes;
    if (count) HIPCHECK(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return RTX_OK;
}

// Random_Seed (random.c:78-85) on the host: the state after the jump, for the lookup table of the device
static void host_jump(uint64_t a, uint64_t b, uint32_t out[4])
{
    uint32_t s[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    const uint32_t JUMP[4] = {0x8764000bu, 0xf542d2d3u, 0x6fa035c3u, 0x77f2db5bu};
    uint32_t       j[4]    = {0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int bit = 0; bit < 32; bit++) {
            if (JUMP[i] & (1u << bit))
                for (int k = 0; k < 4; k++) j[k] ^= s[k];
            const uint32_t t = s[1] << 9; // the generator's state update (random.c:31-45)
            s[2] ^= s[0], s[3] ^= s[1], s[1] ^= s[2], s[0] ^= s[3], s[2] ^= t;
            s[3] = (s[3] << 11) | (s[3] >> 21);
        }
    memcpy(out, j, sizeof j);
}

extern "C" int rtx_scene_destroy(rtx_scene* sc)
{
    if (!sc) return RTX_OK;
    (void)hipSetDevice(sc->device);
    for (void* p : sc->allocs) (void)hipFree(p);
    if (sc->ws) (void)hipFree(sc->ws);
    for (void* b : sc->multi_buf)
        if (b) (void)hipFree(b);
    if (sc->multi_stream) (void)hipStreamDestroy(sc->multi_stream);
    for (int i = 0; i < sc->n_ev; i++) (void)hipEventDestroy(sc->ev[i]);
    for (int i = 0; i < RTX_MAX_LANES - 1; i++) {
        if (sc->aux_stream[i]) (void)hipStreamDestroy(sc->aux_stream[i]);
        if (sc->ev_join[i]) (void)hipEventDestroy(sc->ev_join[i]);
    }
    if (sc->ev_fork) (void)hipEventDestroy(sc->ev_fork);
    if (sc->h_alive) (void)hipHostFree(sc->h_alive);
    delete sc;
    return RTX_OK;
}

extern "C" int rtx_scene_device_bytes(const rtx_scene* sc, uint64_t* bytes)
{
    if (!sc || !bytes) return fail(RTX_E_INVALID, "null argument");
    *bytes = sc->bytes;
    return RTX_OK;
}

#ifndef TAIL_WAVES
#define TAIL_WAVES 4 // waves per SIMD the tail kernel is compiled for (128 VGPRs: the shading code spills 26 values)
#endif
#define EARLY_COUNTERS (8 * RTX_MAX_DEPTH) // after the TAIL_COUNTERS: [RTX_MAX_DEPTH * k + d]: rays of depth d answered in the shade pass before
#define TAIL_COUNTERS 160 // [80 + 8 k]: paths shaded by k_shade, spread over 8 counters; u64 after the lanes' counters: rays started per depth by the tail kernel [RTX_MAX_DEPTH], paths it shaded
// body(lo, hi) over [0, n) in contiguous ranges on the host's cores (RTX_HOST_THREADS, default up to 16): the per-node
// and per-triangle passes of rtx_scene_create.  Results do not depend on the split.
template <class F> static void parallel_ranges(size_t n, F&& body)
{
    unsigned T = std::thread::hardware_concurrency();
    if (T == 0) T = 1;
    if (T > 16) T = 16;
    if (const char* e = getenv("RTX_HOST_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) T = (unsigned)v;
    }
    if (n < (1u << 15)) T = 1;
    if (T == 1) {
        body((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned k = 1; k < T; k++) th.emplace_back([&body, n, k, T] { body(n * k / T, n * (k + 1) / T); });
    body((size_t)0, n / T);
    for (auto& t : th) t.join();
}

struct HostTimer { // RTX_HOST_TIMING=1: phases of rtx_scene_create on stderr
    bool   on = getenv("RTX_HOST_TIMING") != nullptr;
    double t0 = now();
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
    void lap(const char* what) { if (on) { const double t = now(); fprintf(stderr, "rtx_scene_create: %-28s %.3f s\n", what, t - t0); t0 = t; } }
};
static int validate_desc(const rtx_scene_desc* d)
{
    if (!d) return fail(RTX_E_INVALID, "null scene description");
    if (d->abi_version != RTX_ABI_VERSION) return fail(RTX_E_INVALID, "scene description ABI %u, library %d", d->abi_version, RTX_ABI_VERSION);
    if (d->n_nodes == 0 || d->n_tris + d->n_spheres == 0)
        return fail(RTX_E_EMPTY_SCENE, "scene has no bounded primitive / kd-tree (the reference dereferences NULL here, scene.c:109)");
    if (d->n_nodes >= (1u << 30) || d->n_refs >= (1u << 30) || d->n_tris >= (1u << 30) || d->n_spheres >= (1u << 30) || d->n_planes >= (1u << 30))
        return fail(RTX_E_LIMIT, "pool larger than 2^30 entries");
    if (d->max_depth == 0 || d->max_depth > RTX_STACK) return fail(RTX_E_LIMIT, "kd-tree depth bound %u exceeds the traversal stack (%d)", d->max_depth, RTX_STACK);
    if (d->root_index >= d->n_nodes) return fail(RTX_E_INVALID, "root index out of range");
    if (!d->node_w0 || !d->node_w1 || (d->n_refs && !d->leaf_ref)) return fail(RTX_E_INVALID, "null kd-tree pool");
    if (d->n_tris && (!d->tri_pos || !d->tri_nrm || !d->tri_uv || !d->tri_mat || !d->tri_obj)) return fail(RTX_E_INVALID, "null triangle pool");
    if (d->n_spheres && (!d->sph || !d->sph_mat || !d->sph_obj)) return fail(RTX_E_INVALID, "null sphere pool");
    if (d->n_planes && (!d->plane || !d->plane_mat)) return fail(RTX_E_INVALID, "null plane pool");
    if (d->n_materials == 0 || !d->materials) return fail(RTX_E_INVALID, "no materials");
    if (d->n_textures && (!d->textures || !d->tex_data)) return fail(RTX_E_INVALID, "null texture pool");
    // every index the kernels will follow is checked here, once, on the host: a bad index on the
    // device is a GPU fault
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        uint32_t type = d->node_w1[i] >> 30, idx = d->node_w1[i] & 0x3FFFFFFFu;
        if (type == 3u) {
            if ((uint64_t)idx + d->node_w0[i] > d->n_refs) return fail(RTX_E_INVALID, "leaf %u runs past the reference pool", i);
        } else {
            if (idx >= d->n_nodes || i + 1 >= d->n_nodes) return fail(RTX_E_INVALID, "inner node %u has a child out of range", i);
        }
    }
    // ... and the SHAPE the walks rely on: what hangs below the root must be a tree — a node reached twice is a cycle (the
    // device walk and build_cull_data's walk would never end) or a shared subtree — no deeper than the traversal stack:
    // a walk keeps at most one pending far child per inner node on its path, and the per-lane stack has
    // min(max_depth, RTX_STACK) entries.  desc.max_depth is a promise; the depth is measured here.
    {
        std::vector<uint8_t>  seen(d->n_nodes, 0);
        std::vector<uint64_t> st; // node | level << 32
        st.push_back(d->root_index);
        uint32_t deepest = 0;
        while (!st.empty()) {
            const uint32_t n = (uint32_t)st.back(), level = (uint32_t)(st.back() >> 32);
            st.pop_back();
            if (seen[n]) return fail(RTX_E_INVALID, "kd node %u is reached twice from the root (a cycle or a shared subtree): not a tree", n);
            seen[n] = 1;
            deepest = std::max(deepest, level);
            if (level + 1 > std::min<uint32_t>(d->max_depth, RTX_STACK))
                return fail(RTX_E_LIMIT, "kd-tree is deeper than %u levels (desc.max_depth %u, traversal stack %d) at node %u",
                            std::min<uint32_t>(d->max_depth, RTX_STACK), d->max_depth, RTX_STACK, n);
            if ((d->node_w1[n] >> 30) != 3u) {
                st.push_back((uint64_t)(level + 1) << 32 | (n + 1));
                st.push_back((uint64_t)(level + 1) << 32 | (d->node_w1[n] & 0x3FFFFFFFu));
            }
        }
    }
    for (uint32_t i = 0; i < d->n_refs; i++) {
        uint32_t r = d->leaf_ref[i];
        if ((r & RTX_REF_SPHERE) ? ((r & 0x7FFFFFFFu) >= d->n_spheres) : (r >= d->n_tris)) return fail(RTX_E_INVALID, "leaf reference %u out of range", i);
    }
    for (uint32_t i = 0; i < d->n_tris; i++)
        if (d->tri_mat[i] >= d->n_materials) return fail(RTX_E_INVALID, "triangle %u: material out of range", i);
    for (uint32_t i = 0; i < d->n_spheres; i++)
        if (d->sph_mat[i] >= d->n_materials) return fail(RTX_E_INVALID, "sphere %u: material out of range", i);
    for (uint32_t i = 0; i < d->n_planes; i++)
        if (d->plane_mat[i] >= d->n_materials) return fail(RTX_E_INVALID, "plane %u: material out of range", i);
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rtx_texture_desc& t = d->textures[i];
        if (t.width == 0 || t.height == 0 || t.offset + (uint64_t)t.width * t.height > d->tex_texels) return fail(RTX_E_INVALID, "texture %u out of range", i);
    }
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rtx_material& m = d->materials[i];
        if (m.type > RTX_MAT_DISNEY_BSDF) return fail(RTX_E_INVALID, "material %u: unknown type %u", i, m.type);
        if (m.tex >= (int32_t)d->n_textures) return fail(RTX_E_INVALID, "material %u: texture out of range", i);
        bool needs_tex = m.type != RTX_MAT_SKYBOX && m.type != RTX_MAT_DISNEY_CLEARCOAT;
        if (needs_tex && m.tex < 0) return fail(RTX_E_INVALID, "material %u needs a texture", i);
    }
    for (int i = 0; i < 6; i++)
        if (d->skybox_tex[i] < 0 || d->skybox_tex[i] >= (int32_t)d->n_textures) return fail(RTX_E_INVALID, "skybox face %d has no texture", i);
    return RTX_OK;
}

// ---------------------------------------------------------------------------------------------
// Conservative subtree boxes for the culled closest-hit kernel.
//
// The kernel may skip a kd node when the segment {o + t d, 0 <= t <= kt * tMax} misses the node's box.
// That is safe when the box contains, for every primitive referenced below the node, every point of a
// ray at which the REFERENCE's binary32 intersection routine could still report a hit, plus the slack of
// the box test's own arithmetic.  With eps = 2^-24, rays with |d| = 1 (+-1e-3) whose origin lies within
// D of every vertex, E = max(|e1|, |e2|) and the determinant cut-off |a| >= alpha = 1e-4
// (Triangle_HitAt, surfaces.c:135-177), standard forward error analysis of
//     h = d x e2, a = e1.h, s = o - v0, u = (s.h)/a, q = s x e1, v = (d.q)/a, t = (e2.q)/a
// (|delta s| <= eps |s|; |delta h|, per component <= 2 eps |d||e2|, so <= 3.5 eps E; a 3-term dot product adds
// 3 eps |x||y|) gives |a - a*| <= 6.5 eps E^2, |u - u*|, |v - v*| <= (7.5 eps D E + |u*| 6.5 eps E^2)/alpha + eps and
// |t - t*| <= (7.5 eps D E^2 + |t*| 6.5 eps E^2)/alpha + eps |t|   (x* = exact value).  So an accepted hit
// (u, v >= 0, u + v <= 1, 1e-4 <= t <= tMax) has its exact plane point v0 + u* e1 + v* e2 inside the
// triangle's bounds widened per axis by 4 beta ext_axis, at an exact ray parameter in
// [-tau, tMax (1 + kappa') + tau]; a point within tau of the tested segment lies in the box widened by
// tau.  Spheres (Sphere_HitAt, surfaces.c:55-87): the discriminant is off by at most
// Delta = 24 eps (D^2 + r^2), so a reported root is within 2 sqrt(Delta) + 5 eps D of an exact point of
// the sphere or of the ray's closest approach, which is then within sqrt(Delta) of the sphere.
// All bounds carry a further factor 1.25; a triangle whose kappa = 6.5 eps E^2 / alpha reaches 1/4 (edges over
// ~8 units) gets an unbounded box, i.e. is never culled.  The pads grow with D E / alpha: with big triangles
// (beta ~ 1 and more) the boxes stop culling anything, and `useful` tells the launcher to walk literally.  The box test itself computes (b - o) * (1/d) in binary32:
// its rounding equals a displacement of the box plane by <= 3 eps |b - o| <= 6 eps D, covered by 32 eps D.
// ---------------------------------------------------------------------------------------------
struct CullData {
    std::vector<uint16_t> q16; // 6 per node: min x y z, max x y z on the grid; min.x = 0xFFFF, max.x = 0: never cull
    float                 c[3], ro2, kt;
    float                 lo[3], scale[3]; // grid: x = lo + q * scale, both binary32 values taken exactly
    bool                  useful;                    // median triangle pad below its own size
};

static float round_down(double v) { float f = (float)v; return (double)f > v ? nextafterf(f, -INFINITY) : f; }
static float round_up(double v) { float f = (float)v; return (double)f < v ? nextafterf(f, INFINITY) : f; }

static void build_cull_data(const rtx_scene_desc* d, CullData& out)
{
    const double EPS = 5.9604644775390625e-8, ALPHA = 1e-4;
    const size_t T = d->n_tris, Sn = d->n_spheres, N = d->n_nodes;
    // scene bounds -> centre, radius; rays are culled only while their origin is within RO of the centre
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int c = 0; c < 9; c++) { // min / max are exact: any order of the comparisons gives the same bounds
        const float* col = d->tri_pos + (size_t)c * T;
        float        mn = INFINITY, mx = -INFINITY;
        for (size_t i = 0; i < T; i++) mn = col[i] < mn ? col[i] : mn, mx = col[i] > mx ? col[i] : mx;
        if (T) lo[c % 3] = std::min(lo[c % 3], (double)mn), hi[c % 3] = std::max(hi[c % 3], (double)mx);
    }
    for (size_t i = 0; i < Sn; i++)
        for (int a = 0; a < 3; a++) {
            double x = d->sph[a * Sn + i], r = fabs((double)d->sph[3 * Sn + i]);
            lo[a] = std::min(lo[a], x - r), hi[a] = std::max(hi[a], x + r);
        }
    double c[3], R = 0.0;
    for (int a = 0; a < 3; a++) c[a] = 0.5 * (lo[a] + hi[a]), R += 0.25 * (hi[a] - lo[a]) * (hi[a] - lo[a]);
    R               = sqrt(R) + 1e-6;
    double ro_factor = 3.0;
    if (const char* e = getenv("RTX_CULL_RO")) ro_factor = std::max(0.1, atof(e)); // experiments: culling radius in scene radii
    const double RO = ro_factor * R, D = 1.01 * (RO + R), SAFETY = 1.25;
    std::vector<float> betas(T);
    for (int a = 0; a < 3; a++) out.c[a] = (float)c[a];
    out.ro2 = round_down(RO * RO * 0.999);

    std::vector<double> plo(3 * (T + Sn)), phi(3 * (T + Sn)); // padded primitive bounds
    double              kappa_scene = 0.0;
    std::mutex          kmx;
    parallel_ranges(T, [&](size_t t_lo, size_t t_hi) {
    double kappa_part = 0.0;
    for (size_t i = t_lo; i < t_hi; i++) {
        float p[9];
        for (int k = 0; k < 9; k++) p[k] = d->tri_pos[k * T + i];
        const float  e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
        const double E     = std::max(sqrt((double)e1[0] * e1[0] + (double)e1[1] * e1[1] + (double)e1[2] * e1[2]),
                                      sqrt((double)e2[0] * e2[0] + (double)e2[1] * e2[1] + (double)e2[2] * e2[2]));
        const double kappa = 6.5 * EPS * E * E / ALPHA;
        const bool   finite = std::isfinite(E) && kappa < 0.25;
        const double beta  = finite ? SAFETY * ((7.5 * EPS * D * E / ALPHA + kappa + EPS) / (1.0 - kappa)) : 0.0;
        const double tau   = finite ? SAFETY * (7.5 * EPS * D * E * E / ALPHA) / (1.0 - kappa) : 0.0;
        if (finite) kappa_part = std::max(kappa_part, SAFETY * (kappa + 2.0 * EPS) / (1.0 - kappa));
        betas[i] = finite ? (float)beta : INFINITY;
        for (int a = 0; a < 3; a++) {
            // bounds over v0, v1, v2 and v0 + e1, v0 + e2 as the kernel's records hold them
            double x0 = p[a], x1 = p[3 + a], x2 = p[6 + a], y1 = (double)p[a] + e1[a], y2 = (double)p[a] + e2[a];
            double mn = std::min({x0, x1, x2, y1, y2}), mx = std::max({x0, x1, x2, y1, y2});
            double pad = 4.0 * beta * (mx - mn) + tau + 32.0 * EPS * D;
            plo[3 * i + a] = finite ? mn - pad : -INFINITY;
            phi[3 * i + a] = finite ? mx + pad : INFINITY;
        }
    }
    std::lock_guard<std::mutex> g(kmx);
    kappa_scene = std::max(kappa_scene, kappa_part);
    });
    for (size_t i = 0; i < Sn; i++) {
        const double r = fabs((double)d->sph[3 * Sn + i]), Delta = 24.0 * EPS * (D * D + r * r);
        const double pad = SAFETY * (3.0 * sqrt(Delta) + 5.0 * EPS * D) + 32.0 * EPS * D;
        for (int a = 0; a < 3; a++) {
            plo[3 * (T + i) + a] = (double)d->sph[a * Sn + i] - r - pad;
            phi[3 * (T + i) + a] = (double)d->sph[a * Sn + i] + r + pad;
        }
    }
    out.kt = round_up(1.0 + kappa_scene + 64.0 * EPS);
    out.useful = true;
    if (!betas.empty()) {
        std::nth_element(betas.begin(), betas.begin() + betas.size() / 2, betas.end());
        out.useful = betas[betas.size() / 2] <= 1.0f;
    }

    // subtree boxes, children before parents (explicit stack: the tree can be 60 levels deep but is not balanced)
    std::unique_ptr<double[]> blo_mem(new double[3 * N]), bhi_mem(new double[3 * N]); // filled below, node by node
    double* const             blo = blo_mem.get();
    double* const             bhi = bhi_mem.get();
    // leaves first, on all cores (the random reads of the primitives' bounds are most of this pass); an empty leaf
    // and, for now, every inner node get the empty box
    parallel_ranges(N, [&](size_t n_lo, size_t n_hi) {
        for (size_t n = n_lo; n < n_hi; n++) {
            for (int a = 0; a < 3; a++) blo[3 * n + a] = INFINITY, bhi[3 * n + a] = -INFINITY;
            if ((d->node_w1[n] >> 30) != 3u) continue;
            const uint32_t len = d->node_w0[n], first = d->node_w1[n] & 0x3FFFFFFFu;
            for (uint32_t j = 0; j < len; j++) {
                const uint32_t r = d->leaf_ref[first + j];
                const size_t   p = (r & RTX_REF_SPHERE) ? T + (r & 0x7FFFFFFFu) : r;
                for (int a = 0; a < 3; a++) blo[3 * n + a] = std::min(blo[3 * n + a], plo[3 * p + a]), bhi[3 * n + a] = std::max(bhi[3 * n + a], phi[3 * p + a]);
            }
        }
    });
    std::vector<uint32_t> order, st;
    // KDTree_New numbers a node before its subtrees (right child = node + 1, left child further on: kdtree.c:137-171),
    // so the plain index order, backwards, has the children before the parents; any other numbering gets the
    // order from a walk
    bool preorder = true;
    for (size_t n = 0; n < N && preorder; n++)
        if ((d->node_w1[n] >> 30) != 3u && (d->node_w1[n] & 0x3FFFFFFFu) <= n) preorder = false;
    if (!preorder) {
        order.reserve(N);
        st.push_back(d->root_index);
        while (!st.empty()) {
            uint32_t n = st.back();
            st.pop_back();
            order.push_back(n);
            if ((d->node_w1[n] >> 30) != 3u) st.push_back(n + 1), st.push_back(d->node_w1[n] & 0x3FFFFFFFu);
        }
    }
    for (size_t k = preorder ? N : order.size(); k-- > 0;) {
        const uint32_t n = preorder ? (uint32_t)k : order[k];
        if ((d->node_w1[n] >> 30) != 3u) { // inner nodes, children before parents
            const uint32_t l = d->node_w1[n] & 0x3FFFFFFFu, r = n + 1;
            for (int a = 0; a < 3; a++) blo[3 * n + a] = std::min(blo[3 * l + a], blo[3 * r + a]), bhi[3 * n + a] = std::max(bhi[3 * l + a], bhi[3 * r + a]);
        }
    }
    // Quantise to 16 bits per axis on a grid over all finite boxes (+3 steps of margin on every side).  The grid is
    // x(q) = lo + q * scale with lo and scale the binary32 values the kernel holds, taken exactly (the quantisation below
    // runs in double against those very values).  The kernel evaluates a plane's ray parameter as
    // fma(float(q), scale, -fl(o - lo)) * inv: against the exact (x(q) - o) * inv that is off by eps |o - lo| (the rounded
    // origin offset) + eps |x(q) - o| (the fma) + eps |t| (the product), a displacement of the plane by at most
    // eps (|o - lo| + 2 |x - o|) <= 5.2 eps D — inside the 32 eps D every primitive's pad carries for the box test.
    // Box minima round down and maxima round up, plus one further step outward.  A node with an unbounded box is
    // marked "never cull".
    double glo[3] = {INFINITY, INFINITY, INFINITY}, ghi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t k = 0; k < 3 * (T + Sn); k++)
        if (std::isfinite(plo[k]) && std::isfinite(phi[k])) glo[k % 3] = std::min(glo[k % 3], plo[k]), ghi[k % 3] = std::max(ghi[k % 3], phi[k]);
    const double LEVELS = 65535.0;
    double       scale[3];
    for (int a = 0; a < 3; a++) {
        if (!(glo[a] <= ghi[a])) glo[a] = 0.0, ghi[a] = 1.0; // no finite box at all
        const double ext = std::max(ghi[a] - glo[a], 1e-6);
        out.scale[a]     = (float)(ext / (LEVELS - 8.0));
        scale[a]         = (double)out.scale[a];
        out.lo[a]        = (float)(glo[a] - 4.0 * scale[a]);
        glo[a]           = (double)out.lo[a]; // the origin the kernel decodes against
    }
    out.q16.assign(6 * N, 0);
    parallel_ranges(N, [&](size_t n_lo, size_t n_hi) {
    for (size_t n = n_lo; n < n_hi; n++) {
        uint16_t* q      = &out.q16[6 * n];
        bool      nocull = false;
        for (int a = 0; a < 3; a++) {
            const double l = blo[3 * n + a], h = bhi[3 * n + a];
            if (!(l <= h)) continue; // node never reached from the root, or an empty leaf (handled by its len == 0)
            if (!std::isfinite(l) || !std::isfinite(h)) {
                nocull = true;
                continue;
            }
            const double ql = floor((l - glo[a]) / scale[a]) - 1.0, qh = ceil((h - glo[a]) / scale[a]) + 1.0;
            q[a]     = (uint16_t)std::min<double>(std::max(ql, 0.0), LEVELS);
            q[3 + a] = (uint16_t)std::min<double>(std::max(qh, 0.0), LEVELS);
        }
        if (nocull) q[0] = 0xFFFF, q[3] = 0;
    }
    });
}

// a box as the kernel reads it: {x, y, z}, max << 16 | min
static void pack_box(const uint16_t* q, uint32_t w[3])
{
    for (int a = 0; a < 3; a++) w[a] = (uint32_t)q[3 + a] << 16 | q[a];
}

// ---------------------------------------------------------------------------------------------
// Cut-off nodes folded into the step of their only real child.
//
// 44 % of the inner nodes KDTree_New builds cut empty space off: one child is an empty leaf.  Half of all node steps of a
// walk are spent on them, and such a step can only do two things — clip tMax, or end the subtree's walk.  The culled
// kernel therefore evaluates the plane of a cut-off node P (CheckHitInternalNode, kdtree.c:501-570, with one child known
// to report nothing) at the head of the step of P's non-empty child X, and links P's parent straight to X: P is never
// fetched and costs no step of its own.  One plane per record: X may carry P only if P carries none itself (a chain of k
// cut-offs above a node keeps every other one).  Results are those of stepping through P — same comparisons, same
// arithmetic; the only difference is that X is entered without a box test of its own (its box is P's box, which P's
// parent tested), which the CPU replay follows through the flags (rtx_cull_node_flags).
// ---------------------------------------------------------------------------------------------
enum { NODE_ABSORBED = 1, NODE_CARRIES_PARENT = 2 };
struct MergePlan {
    std::vector<uint8_t>  flags;  // per node: NODE_ABSORBED | NODE_CARRIES_PARENT
    std::vector<uint32_t> parent; // of a node that carries its parent's plane
};

static bool is_empty_leaf(const rtx_scene_desc* d, uint32_t n) { return (d->node_w1[n] >> 30) == 3u && d->node_w0[n] == 0u; }

static void build_merge_plan(const rtx_scene_desc* d, MergePlan& mp) // the tree's shape has been validated
{
    const size_t N = d->n_nodes;
    mp.flags.assign(N, 0);
    mp.parent.assign(N, 0xFFFFFFFFu);
    std::vector<uint32_t> st;
    st.push_back(d->root_index);
    while (!st.empty()) { // parents before children
        const uint32_t n = st.back();
        st.pop_back();
        if ((d->node_w1[n] >> 30) == 3u) continue;
        const uint32_t l = d->node_w1[n] & 0x3FFFFFFFu, r = n + 1;
        const bool     el = is_empty_leaf(d, l), er = is_empty_leaf(d, r);
        if (el != er && !(mp.flags[n] & NODE_CARRIES_PARENT)) { // a cut-off node that carries no plane itself: its real child takes it over
            const uint32_t x = el ? r : l;
            mp.flags[n] |= NODE_ABSORBED;
            mp.flags[x] |= NODE_CARRIES_PARENT;
            mp.parent[x] = n;
        }
        st.push_back(r), st.push_back(l);
    }
}

static uint32_t merge_resolve(const rtx_scene_desc* d, const MergePlan& mp, uint32_t n) // the node a link to n leads to
{
    if (!(mp.flags[n] & NODE_ABSORBED)) return n;
    const uint32_t l = d->node_w1[n] & 0x3FFFFFFFu;
    return is_empty_leaf(d, l) ? n + 1 : l;
}

extern "C" int rtx_cull_node_flags(const rtx_scene_desc* d, uint8_t* flags)
{
    if (!flags) return fail(RTX_E_INVALID, "null argument");
    int rc = validate_desc(d);
    if (rc != RTX_OK) return rc;
    MergePlan mp;
    build_merge_plan(d, mp);
    memcpy(flags, mp.flags.data(), mp.flags.size());
    return RTX_OK;
}

extern "C" int rtx_cull_data_build(const rtx_scene_desc* d, uint16_t* qbox6, float* info14)
{
    if (!qbox6 || !info14) return fail(RTX_E_INVALID, "null argument");
    int rc = validate_desc(d);
    if (rc != RTX_OK) return rc;
    CullData cd;
    build_cull_data(d, cd);
    memcpy(qbox6, cd.q16.data(), cd.q16.size() * sizeof(uint16_t));
    for (int a = 0; a < 3; a++) info14[a] = cd.c[a], info14[5 + a] = cd.lo[a], info14[8 + a] = cd.scale[a], info14[11 + a] = 0.0f;
    info14[3] = cd.ro2;
    info14[4] = cd.kt;
    return RTX_OK;
}

static int noalloc_scratch_address(rtx_scene* sc);

extern "C" int rtx_scene_create(const rtx_scene_desc* d, int device, rtx_scene** out) { return rtx_scene_create_ex(d, device, 0u, out); }

extern "C" int rtx_scene_create_ex(const rtx_scene_desc* d, int device, uint32_t flags, rtx_scene** out)
{
    if (!out) return fail(RTX_E_INVALID, "null output");
    if (flags & ~(RTX_SCENE_F32_TEXTURES | RTX_SCENE_SKY_LOOKUP)) return fail(RTX_E_INVALID, "unknown scene flags 0x%x", flags);
    const bool compact_textures = !(flags & RTX_SCENE_F32_TEXTURES);
    *out = nullptr;
    int n = 0, rc;
    if ((rc = rtx_device_count(&n)) != RTX_OK) return rc;
    if (device < 0 || device >= n) return fail(RTX_E_NO_DEVICE, "device %d not present (%d visible)", device, n);
    HostTimer tm;
    if ((rc = validate_desc(d)) != RTX_OK) return rc;
    tm.lap("validate");
    HIPCHECK(hipSetDevice(device));

    rtx_scene* sc = new rtx_scene();
    sc->device    = device;
    sc->bytes     = 0;
    sc->n_ev      = 0;
    sc->ws        = nullptr;
    sc->ws_bytes  = 0;
    sc->ev_fork = nullptr, sc->h_alive = nullptr;
    sc->noalloc_scratch = nullptr;
    for (int i = 0; i < RTX_MAX_LANES - 1; i++) sc->aux_stream[i] = nullptr, sc->ev_join[i] = nullptr;
    memset(sc->persist_blocks, 0, sizeof sc->persist_blocks);
    memset(&sc->dev, 0, sizeof sc->dev);
    SceneDev& S = sc->dev;

#define UP(call)                      \
    if ((rc = (call)) != RTX_OK) {    \
        rtx_scene_destroy(sc);        \
        return rc;                    \
    }
    {
        std::vector<uint2> nodes(d->n_nodes);
        for (uint32_t i = 0; i < d->n_nodes; i++) nodes[i] = make_uint2(d->node_w0[i], d->node_w1[i]);
        UP(upload(sc, nodes.data(), nodes.size(), &S.nodes));
    }
    tm.lap("nodes");
    {
        CullData cd;
        build_cull_data(d, cd);
        tm.lap("culling boxes");
        {
            // 40-byte records: {w0, w1 (left link resolved), L.x, L.y | L.z, R.x, R.y, R.z | right link : 29, carried plane's
            // axis : 2 (3 = none), side : 1 (1: this node is the carried parent's right child), carried plane's split}
            MergePlan mp;
            build_merge_plan(d, mp);
            if (d->n_nodes >= (1u << 29)) {
                rtx_scene_destroy(sc);
                return fail(RTX_E_LIMIT, "more than 2^29 kd nodes");
            }
            std::vector<uint32_t> n40(NODE_WORDS * (size_t)d->n_nodes + 4, 0u); // + 16 bytes: the last record is read as 16 + 16 + 8
            parallel_ranges(d->n_nodes, [&](size_t i_lo, size_t i_hi) {
            for (size_t i = i_lo; i < i_hi; i++) {
                uint32_t l[3] = {0, 0, 0}, r[3] = {0, 0, 0}, w1 = d->node_w1[i], right = 0;
                if ((w1 >> 30) != 3u) { // indices checked by validate_desc
                    const uint32_t li = w1 & 0x3FFFFFFFu;
                    pack_box(&cd.q16[6 * (size_t)li], l);
                    pack_box(&cd.q16[6 * ((size_t)i + 1)], r);
                    w1    = (w1 & 0xC0000000u) | merge_resolve(d, mp, li);
                    right = merge_resolve(d, mp, (uint32_t)i + 1);
                }
                uint32_t axis = 3u, side = 0u, split = 0u;
                if (mp.flags[i] & NODE_CARRIES_PARENT) {
                    const uint32_t pn = mp.parent[i];
                    axis = d->node_w1[pn] >> 30, side = (i == (size_t)pn + 1) ? 1u : 0u, split = d->node_w0[pn];
                }
                uint32_t* w = &n40[NODE_WORDS * i];
                w[0] = d->node_w0[i], w[1] = w1, w[2] = l[0], w[3] = l[1], w[4] = l[2], w[5] = r[0], w[6] = r[1], w[7] = r[2];
                w[8] = right | axis << 29 | side << 31, w[9] = split;
            }
            });
            UP(upload(sc, n40.data(), n40.size(), &S.nodes40));
            S.root = merge_resolve(d, mp, d->root_index);
            pack_box(&cd.q16[6 * (size_t)d->root_index], S.root_q);
        }
        for (int a = 0; a < 3; a++) S.cull_c[a] = cd.c[a], S.cull_lo[a] = cd.lo[a], S.cull_scale[a] = cd.scale[a];
        S.cull_ro2 = cd.ro2;
        S.cull_kt  = cd.kt;
        for (int a = 0; a < 3; a++) S.bin_inv[a] = (float)(1 << BIN_CELL_BITS) / (cd.scale[a] * 65535.0f); // cells over the box grid
        sc->cull_useful = cd.useful;
    }
    tm.lap("32-byte nodes");
    {
        // Device-side triangle order = order of first appearance in the leaf reference stream (the
        // tree's depth-first leaf order).  Triangles of one leaf, and of leaves that are neighbours
        // in space, then share cache lines; rocprof showed the bounce passes bound by L2-miss traffic
        // (one 128-B line per 48-B triangle fetched at random).  Only addresses change: tri_obj keeps
        // the reference's kdObjects index of every triangle.
        const size_t          T = d->n_tris;
        std::vector<uint32_t> new_of_old(T, 0xFFFFFFFFu), old_of_new;
        old_of_new.reserve(T);
        for (uint32_t k = 0; k < d->n_refs; k++) {
            uint32_t r = d->leaf_ref[k];
            if (!(r & RTX_REF_SPHERE) && new_of_old[r] == 0xFFFFFFFFu) {
                new_of_old[r] = (uint32_t)old_of_new.size();
                old_of_new.push_back(r);
            }
        }
        for (uint32_t i = 0; i < T; i++)
            if (new_of_old[i] == 0xFFFFFFFFu) {
                new_of_old[i] = (uint32_t)old_of_new.size();
                old_of_new.push_back(i);
            }
        std::vector<uint32_t> refs(d->n_refs);
        for (uint32_t k = 0; k < d->n_refs; k++) {
            uint32_t r = d->leaf_ref[k];
            refs[k]    = (r & RTX_REF_SPHERE) ? r : new_of_old[r];
        }
        UP(upload(sc, refs.data(), refs.size(), &S.leaf_ref));

        std::vector<float>    hot((size_t)(TRI_BYTES / 4) * T + 4, 0.0f); // + 16 bytes: the last record is read as three float4
        std::vector<float4>   cold(4 * T);
        std::vector<uint32_t> obj(T);
        parallel_ranges(T, [&](size_t j_lo, size_t j_hi) {
        for (size_t j = j_lo; j < j_hi; j++) {
            const size_t i = old_of_new[j];
            float        p[9], nn[9], uv[6];
            for (int c = 0; c < 9; c++) p[c] = d->tri_pos[c * T + i], nn[c] = d->tri_nrm[c * T + i];
            for (int c = 0; c < 6; c++) uv[c] = d->tri_uv[c * T + i];
            // edge1 = v1 - v0, edge2 = v2 - v0 (surfaces.c:137-138), one binary32 subtraction each
            float e1x = p[3] - p[0], e1y = p[4] - p[1], e1z = p[5] - p[2];
            float e2x = p[6] - p[0], e2y = p[7] - p[1], e2z = p[8] - p[2];
            float* h = &hot[(size_t)(TRI_BYTES / 4) * j];
            h[0] = p[0], h[1] = p[1], h[2] = p[2], h[3] = e1x, h[4] = e1y, h[5] = e1z, h[6] = e2x, h[7] = e2y, h[8] = e2z;
            uint32_t mat    = d->tri_mat[i];
            float    matf;
            memcpy(&matf, &mat, 4);
            cold[4 * j + 0] = make_float4(nn[0], nn[1], nn[2], nn[3]);
            cold[4 * j + 1] = make_float4(nn[4], nn[5], nn[6], nn[7]);
            cold[4 * j + 2] = make_float4(nn[8], uv[0], uv[1], uv[2]);
            cold[4 * j + 3] = make_float4(uv[3], uv[4], uv[5], matf);
            obj[j]          = d->tri_obj[i];
        }
        });
        {
            const float* dtri = nullptr;
            UP(upload(sc, hot.data(), hot.size(), &dtri));
            S.tri = (const float4*)dtri;
        }
        UP(upload(sc, cold.data(), cold.size(), &S.tri_shade));
        UP(upload(sc, obj.data(), T, &S.tri_obj));
    }
    tm.lap("triangle records");
    {
        const size_t          Sn = d->n_spheres;
        std::vector<float4>   sp(Sn);
        std::vector<uint32_t> sm(Sn);
        for (size_t i = 0; i < Sn; i++) {
            sp[i] = make_float4(d->sph[0 * Sn + i], d->sph[1 * Sn + i], d->sph[2 * Sn + i], d->sph[3 * Sn + i]);
            sm[i] = d->sph_mat[i];
        }
        UP(upload(sc, sp.data(), Sn, &S.sph));
        UP(upload(sc, sm.data(), Sn, &S.sph_mat));
        UP(upload(sc, d->sph_obj, Sn, &S.sph_obj));
    }
    {
        const size_t       P = d->n_planes;
        std::vector<float> pl(8 * P);
        for (size_t i = 0; i < P; i++) {
            for (int c = 0; c < 6; c++) pl[8 * i + c] = d->plane[c * P + i];
            uint32_t mat = d->plane_mat[i];
            memcpy(&pl[8 * i + 6], &mat, 4);
            pl[8 * i + 7] = 0.0f;
        }
        UP(upload(sc, pl.data(), pl.size(), &S.plane));
    }
    UP(upload(sc, d->materials, d->n_materials, &S.mats));
    {
        // Textures: PAL8 where a texture's values allow it (see TexDev), else the floats as they are.  Only what is kept
        // as f32 goes into the device's float pool.
        std::vector<TexDev>  td(d->n_textures);
        std::vector<float>   f32pool, pal;
        std::vector<uint8_t> bytes;
        for (uint32_t k = 0; k < d->n_textures; k++) {
            const rtx_texture_desc& t = d->textures[k];
            const uint64_t texels = (uint64_t)t.width * t.height;
            const uint32_t* src = (const uint32_t*)(d->tex_data + 3 * t.offset); // bit patterns: -0, NaN payloads stay what they are
            td[k].width = t.width, td[k].height = t.height, td[k].fmt = TEX_F32, td[k].pal = 0;
            std::vector<uint32_t> values; // distinct bit patterns, sorted
            bool paletted = compact_textures && texels >= 64;
            if (paletted) {
                uint32_t last = ~src[0];
                for (uint64_t i = 0; i < 3 * texels && paletted; i++) {
                    const uint32_t v = src[i];
                    if (v == last) continue; // runs of equal values are the common case
                    last = v;
                    auto it = std::lower_bound(values.begin(), values.end(), v);
                    if (it == values.end() || *it != v) {
                        if (values.size() == 256) paletted = false;
                        else values.insert(it, v);
                    }
                }
            }
            if (paletted) {
                td[k].fmt = TEX_PAL8, td[k].pal = (uint32_t)pal.size(), td[k].offset = bytes.size();
                pal.resize(pal.size() + 256, 0.0f);
                memcpy(&pal[td[k].pal], values.data(), values.size() * 4);
                const size_t b0 = bytes.size();
                bytes.resize(b0 + 3 * texels);
                parallel_ranges(3 * texels, [&](size_t lo, size_t hi) {
                    for (size_t i = lo; i < hi; i++) bytes[b0 + i] = (uint8_t)(std::lower_bound(values.begin(), values.end(), src[i]) - values.begin());
                });
            } else {
                td[k].offset = f32pool.size() / 3;
                f32pool.insert(f32pool.end(), d->tex_data + 3 * t.offset, d->tex_data + 3 * (t.offset + texels));
            }
        }
        bytes.resize(bytes.size() + 4, 0); // (a lookup reads single bytes; padding keeps the pool non-empty)
        UP(upload(sc, td.data(), td.size(), &S.tex));
        UP(upload(sc, f32pool.data(), f32pool.size(), &S.tex_data));
        UP(upload(sc, bytes.data(), bytes.size(), &S.tex_bytes));
        UP(upload(sc, pal.data(), pal.size(), &S.tex_pal));
    }
    {
        std::vector<uint4> tab(8 * 256);
        for (int k = 0; k < 8; k++)
            for (uint32_t v = 0; v < 256; v++) {
                uint32_t j[4];
                host_jump((uint64_t)v << (8 * k), 0, j);
                tab[k * 256 + v] = make_uint4(j[0], j[1], j[2], j[3]);
            }
        UP(upload(sc, tab.data(), tab.size(), &sc->d_jump_tab));
    }
    {
        const uint64_t* c = nullptr;
        std::vector<uint64_t> zeros(RTX_MAX_LANES * (16 + 64) + TAIL_COUNTERS + EARLY_COUNTERS, 0); // [80 * l ..]: queue count and work counters of lane l; then the tail kernel's
        UP(upload(sc, zeros.data(), zeros.size(), &c));
        sc->d_counters = (uint64_t*)c;
    }
    {
        void* p = nullptr;
        const size_t bytes = (size_t)RTX_MAX_LANES * LONG_CAP * sizeof(uint32_t);
        if (hipMalloc(&p, bytes) != hipSuccess) {
            rtx_scene_destroy(sc);
            return fail(RTX_E_NOMEM, "cannot allocate the long-walk lists");
        }
        sc->allocs.push_back(p);
        sc->bytes += bytes;
        sc->d_long     = (uint32_t*)p;
        sc->tree_depth = d->max_depth;
    }
    tm.lap("the rest");
    S.n_nodes = d->n_nodes, S.n_refs = d->n_refs, S.n_tris = d->n_tris, S.n_spheres = d->n_spheres;
    S.n_planes = d->n_planes, S.n_materials = d->n_materials, S.n_textures = d->n_textures, S.root_literal = d->root_index;
    for (int i = 0; i < 6; i++) S.skybox_tex[i] = d->skybox_tex[i];
    S.sky_1x1 = 0u;
    for (int i = 0; i < 6; i++) { // single-texel faces: the lookup's answer is known here (see SceneDev)
        const rtx_texture_desc& td = d->textures[d->skybox_tex[i]];
        if (td.width == 1 && td.height == 1) {
            S.sky_1x1 |= 1u << i;
            for (int c = 0; c < 3; c++) S.sky_rgb[3 * i + c] = d->tex_data[3 * td.offset + c];
        } else {
            for (int c = 0; c < 3; c++) S.sky_rgb[3 * i + c] = 0.0f;
        }
    }
    if (S.sky_1x1 == 63u) {
        bool same = true;
        for (int i = 1; i < 6; i++) same = same && memcmp(&S.sky_rgb[3 * i], &S.sky_rgb[0], 12) == 0;
        if (same) S.sky_1x1 |= 64u;
    }
    if (flags & RTX_SCENE_SKY_LOOKUP) S.sky_1x1 = 0u; // the full lookup for every face (the tests compare the two)
    UP(noalloc_scratch_address(sc)); // needs cull_useful (set above) to know which kernels this scene launches
#undef UP
    *out = sc;
    return RTX_OK;
}

// ---------------------------------------------------------------------------------------------
// path queues: SoA, `cap` dwords per component
// ---------------------------------------------------------------------------------------------
enum { Q_OX, Q_OY, Q_OZ, Q_DX, Q_DY, Q_DZ, Q_TR, Q_TG, Q_TB, Q_S0, Q_S1, Q_S2, Q_S3, Q_PID, Q_COMPONENTS };

struct Queue { // component c of path i at base[c * cs + i * es]
    float*   base;
    uint64_t cs, es; // SoA: cs = capacity, es = 1;  AoS (16-dword = 64-byte records, after binning): cs = 1, es = 16
    DEV float&    f(int c, uint64_t i) const { return base[(uint64_t)c * cs + i * es]; }
    DEV uint32_t& u(int c, uint64_t i) const { return ((uint32_t*)base)[(uint64_t)c * cs + i * es]; }
};
#define Q_RECORD 16 // dwords per path in an AoS queue (14 used)

struct Hits { // SoA: ref, t, u, v
    uint32_t* base;
    uint64_t  cap;
};

// Division of a 32-bit number by a divisor fixed for the launch: the generator steps per path and pixel index are
// decoded with ten divisions, ~35 vector instructions each when the divisor is only known at run time (k_raygen spent
// most of its time in them).  libdivide's branch-free u32 scheme: exact for every numerator.
struct FastDiv {
    uint32_t m, s; // s bit 31: divisor 1
};
static FastDiv fastdiv_gen(uint32_t d)
{
    FastDiv r;
    if (d <= 1) {
        r.m = 0, r.s = 0x80000000u;
        return r;
    }
    const uint32_t fl = 31u - (uint32_t)__builtin_clz(d);
    if ((d & (d - 1)) == 0) {
        r.m = 0, r.s = fl - 1;
        return r;
    }
    const uint64_t num = (uint64_t)(1ull << fl) << 32;
    uint64_t       m   = num / d;
    const uint64_t rem = num % d;
    m += m;
    if (rem + rem >= d) m += 1;
    r.m = (uint32_t)(m + 1), r.s = fl;
    return r;
}
DEV uint32_t fdiv(uint32_t n, FastDiv f)
{
    const uint32_t q = __umulhi(f.m, n);
    const uint32_t t = ((n - q) >> 1) + q;
    return (f.s & 0x80000000u) ? n : t >> (f.s & 31u);
}

struct FrameGeom {
    uint32_t width, height, spp;
    uint32_t band_rows, band_rank, band_world; // band_rows == 0: every row
    uint32_t sample_begin, n_samples;
    uint64_t seed_lo, seed_hi;
    uint32_t pixel_centres;
    uint32_t tile, slab_rows; // tile != 0: the slab is slab_rows whole rows and its paths are in tile-major pixel order
    uint32_t tile_h;          // tiles are tile pixels wide and tile_h rows high
    // Random_Seed's jump (random.c:53-76) is linear over GF(2) in the seed words: jump(a, b) = XOR over the bytes of a of
    // jump_tab[byte position][byte value], XOR jump_b = jump(0, b).  8 lookups instead of 128 generator steps.
    const uint4* jump_tab; // 8 x 256 entries, or NULL: do the 128 steps
    uint32_t     jump_b[4];
    // fast != 0: every index below fits 32 bits and the divisions go through these (geom_finish fills them)
    uint32_t fast, h_edge, w_edge;
    FastDiv  fd_ns, fd_w, fd_band, fd_trow, fd_th[2], fd_tw[2]; // n_samples, width, band_rows, width * tile_h, tile * {tile_h, h_edge}, {tile, w_edge}
};

// call after the last change to a FrameGeom's sizes / tiling
static void geom_finish(FrameGeom& g)
{
    g.fast = 0;
    if (knob(K_FAST_DIV) == 0) return; // the tests compare the two
    if ((uint64_t)g.width * g.height >= (1ull << 32) || g.n_samples == 0 || g.width == 0) return;
    g.fd_ns   = fastdiv_gen(g.n_samples);
    g.fd_w    = fastdiv_gen(g.width);
    g.fd_band = fastdiv_gen(g.band_rows ? g.band_rows : 1);
    if (g.tile) {
        if ((uint64_t)g.width * g.tile_h >= (1ull << 32) || (uint64_t)g.tile * g.tile_h >= (1ull << 32)) return;
        g.h_edge   = g.slab_rows % g.tile_h ? g.slab_rows % g.tile_h : g.tile_h;
        g.w_edge   = g.width % g.tile ? g.width % g.tile : g.tile;
        g.fd_trow  = fastdiv_gen(g.width * g.tile_h);
        g.fd_th[0] = fastdiv_gen(g.tile * g.tile_h), g.fd_th[1] = fastdiv_gen(g.tile * g.h_edge);
        g.fd_tw[0] = fastdiv_gen(g.tile), g.fd_tw[1] = fastdiv_gen(g.w_edge);
    }
    g.fast = 1;
}

// path index of a slab -> (pixel position in the slab, sample); pixel index of this rank's list -> (x, row of the list)
DEV void split_path(const FrameGeom& g, uint64_t i, uint64_t& px, uint32_t& s)
{
    if (g.fast) {
        const uint32_t q = fdiv((uint32_t)i, g.fd_ns);
        px = q, s = (uint32_t)i - q * g.n_samples;
    } else {
        px = i / g.n_samples, s = (uint32_t)(i % g.n_samples);
    }
}
DEV void split_pixel(const FrameGeom& g, uint64_t pl, uint32_t& x, uint32_t& row)
{
    if (g.fast) {
        row = fdiv((uint32_t)pl, g.fd_w), x = (uint32_t)pl - row * g.width;
    } else {
        x = (uint32_t)(pl % g.width), row = (uint32_t)(pl / g.width);
    }
}

DEV void rng_seed_tab(Rng& r, uint64_t a, const FrameGeom& g)
{
    uint4 j = make_uint4(g.jump_b[0], g.jump_b[1], g.jump_b[2], g.jump_b[3]);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint4 t = g.jump_tab[k * 256 + (uint32_t)((a >> (8 * k)) & 255u)];
        j.x ^= t.x, j.y ^= t.y, j.z ^= t.z, j.w ^= t.w;
    }
    r.s0 = j.x, r.s1 = j.y, r.s2 = j.z, r.s3 = j.w;
}

// Path order inside a slab of R whole rows: pixels tile by tile (TxT, row-major inside a tile), so that a
// run of consecutive paths — what one XCD's waves hold at any time — covers a compact image region and
// its rays share kd nodes and triangles in that XCD's L2.  j = position in that order → row-major index.
DEV uint64_t slab_pixel(const FrameGeom& g, uint64_t j)
{
    if (g.tile == 0) return j;
    const uint32_t W = g.width, R = g.slab_rows, T = g.tile, TH = g.tile_h;
    if (g.fast) {
        const uint32_t jj  = (uint32_t)j;
        const uint32_t tr  = fdiv(jj, g.fd_trow);
        const uint32_t rem = jj - tr * (W * TH);
        const uint32_t h   = min(TH, R - tr * TH); // TH, or h_edge in the last row of tiles
        const FastDiv  fh  = h != TH ? g.fd_th[1] : g.fd_th[0];
        const uint32_t tc  = fdiv(rem, fh);
        const uint32_t r2  = rem - tc * T * h;
        const uint32_t w   = min(T, W - tc * T); // T, or w_edge in the last column
        const FastDiv  fw  = w != T ? g.fd_tw[1] : g.fd_tw[0];
        const uint32_t rr  = fdiv(r2, fw);
        return (uint64_t)(tr * TH + rr) * W + tc * T + (r2 - rr * w);
    }
    const uint32_t tr = (uint32_t)(j / ((uint64_t)W * TH));
    const uint32_t rem = (uint32_t)(j - (uint64_t)tr * W * TH);
    const uint32_t h   = min(TH, R - tr * TH);
    const uint32_t tc  = rem / (T * h);
    const uint32_t r2  = rem - tc * T * h;
    const uint32_t w   = min(T, W - tc * T);
    return (uint64_t)(tr * TH + r2 / w) * W + tc * T + r2 % w;
}

// k-th row assigned to this rank → image row
DEV uint32_t assigned_row(const FrameGeom& g, uint32_t k)
{
    if (g.band_rows == 0) return k;
    uint32_t j, within;
    if (g.fast) j = fdiv(k, g.fd_band), within = k - j * g.band_rows;
    else j = k / g.band_rows, within = k % g.band_rows;
    return (g.band_rank + j * g.band_world) * g.band_rows + within;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------

// RenderPixel's sample set-up (renderer.c:100-104) + Camera_GetRay (camera.c:34-49) + Ray_Make
DEV void camera_sample(const rtx_camera& cam, const FrameGeom& g, uint32_t xx, uint32_t yy, uint32_t s, Rng& rng, f3& o, f3& d)
{
    uint64_t pixel = (uint64_t)yy * g.width + xx;
    if (g.jump_tab) rng_seed_tab(rng, g.seed_lo ^ (pixel * g.spp + s), g);
    else rng_seed(rng, g.seed_lo ^ (pixel * g.spp + s), g.seed_hi);
    float hf, vf;
    if (g.pixel_centres) {
        hf = ((float)xx + 0.5f) / (float)(g.width - 1);
        vf = ((float)yy + 0.5f) / (float)(g.height - 1);
    } else {
        hf = ((float)xx + Random_Unilateral(rng)) / (float)(g.width - 1);
        vf = ((float)yy + Random_Unilateral(rng)) / (float)(g.height - 1);
    }
    float rdx, rdy;
    Random_InDisc(rng, cam.lens_radius, rdx, rdy);
    f3 offset = mk3(hf * rdx, vf * rdy, 0);
    f3 corg   = mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
    f3 blc    = mk3(cam.bottom_left[0], cam.bottom_left[1], cam.bottom_left[2]);
    f3 hor    = mk3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    f3 ver    = mk3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    o         = vadd(corg, offset);
    f3 dir    = vsub(vadd(blc, vadd(vmuls(hor, hf), vmuls(ver, vf))), vadd(corg, offset));
    d         = vnorm(dir); // Ray_Make, ray.c:6
}

// the RNG state camera_sample leaves behind for path i of a slab (what k_raygen would have stored): the first shade
// pass recomputes it instead of reading it back — 32 bytes less per camera path written and 32 read
DEV Rng camera_rng(const FrameGeom& g, uint64_t first_pixel, uint64_t i)
{
    uint64_t px;
    uint32_t s, xx, row;
    split_path(g, i, px, s);
    s += g.sample_begin;
    split_pixel(g, first_pixel + slab_pixel(g, px), xx, row);
    const uint32_t yy = assigned_row(g, row);
    const uint64_t pixel = (uint64_t)yy * g.width + xx;
    Rng            rng;
    if (g.jump_tab) rng_seed_tab(rng, g.seed_lo ^ (pixel * g.spp + s), g);
    else rng_seed(rng, g.seed_lo ^ (pixel * g.spp + s), g.seed_hi);
    if (!g.pixel_centres) (void)Random_Unilateral(rng), (void)Random_Unilateral(rng); // the jitter draws
    float rx, ry;
    Random_InDisc(rng, 0.0f, rx, ry); // the two lens draws (taken whatever the radius, camera.c:36)
    return rng;
}

// lean: only origin and direction are stored; throughput (1), path id (i) and RNG state are rebuilt by the first shade pass
__global__ __launch_bounds__(256) void k_raygen(rtx_camera cam, FrameGeom g, uint64_t first_pixel, uint64_t n, Queue q, int lean)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t px;
    uint32_t s, xx, row;
    split_path(g, i, px, s);
    s += g.sample_begin;
    split_pixel(g, first_pixel + slab_pixel(g, px), xx, row); // first_pixel + ...: index in this rank's pixel list
    uint32_t yy = assigned_row(g, row);
    Rng      rng;
    f3       o, d;
    camera_sample(cam, g, xx, yy, s, rng, o, d);
    q.f(Q_OX, i) = o.x, q.f(Q_OY, i) = o.y, q.f(Q_OZ, i) = o.z;
    q.f(Q_DX, i) = d.x, q.f(Q_DY, i) = d.y, q.f(Q_DZ, i) = d.z;
    if (lean) return;
    q.f(Q_TR, i) = 1.0f, q.f(Q_TG, i) = 1.0f, q.f(Q_TB, i) = 1.0f;
    q.u(Q_S0, i) = rng.s0, q.u(Q_S1, i) = rng.s1, q.u(Q_S2, i) = rng.s2, q.u(Q_S3, i) = rng.s3;
    q.u(Q_PID, i) = (uint32_t)i;
}

DEV void add_visits(uint64_t* counters, const Visits& v)
{
    // wave reduction, one atomic per wave and counter
    uint32_t a = v.inner, b = v.leaf, c = v.prim, e = v.culled;
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
        c += __shfl_down(c, off);
        e += __shfl_down(e, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long*)&counters[0], (unsigned long long)a);
        atomicAdd((unsigned long long*)&counters[1], (unsigned long long)b);
        atomicAdd((unsigned long long*)&counters[2], (unsigned long long)c);
        atomicAdd((unsigned long long*)&counters[3], (unsigned long long)e);
    }
}

// closest hit for a dense queue of unit-direction rays (6 SoA components at rays + c*stride)
template <bool COUNT>
__global__ __launch_bounds__(64) void k_trace(SceneDev S, const float* __restrict__ rays, uint64_t stride, uint64_t n,
                                              uint32_t* __restrict__ hits, uint64_t hstride, uint64_t* counters)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    Visits   vis = {0, 0, 0};
    if (i < n) {
        const uint64_t cs = stride ? stride : 1, rb = i * (stride ? 1 : Q_RECORD); // stride 0: AoS records
        f3       o = mk3(rays[0 * cs + rb], rays[1 * cs + rb], rays[2 * cs + rb]);
        f3       d = mk3(rays[3 * cs + rb], rays[4 * cs + rb], rays[5 * cs + rb]);
        RayD     r = ray_from_unit(o, d);
        uint32_t ref;
        float    t, u, v;
        closest_hit<COUNT>(S, r, ref, t, u, v, vis);
        hits[0 * hstride + i] = ref;
        hits[1 * hstride + i] = __float_as_uint(t);
        hits[2 * hstride + i] = __float_as_uint(u);
        hits[3 * hstride + i] = __float_as_uint(v);
    }
    if (COUNT) add_visits(counters, vis);
}

// ---------------------------------------------------------------------------------------------
// Persistent-wave closest hit with lane refill.
//
// Why: on incoherent bounce rays the one-ray-per-thread kernel above keeps a whole wave resident
// until its slowest lane finishes; rocprof showed ~8 of 64 lanes active per VALU instruction there
// (a ray that threads through the dense part of the tree without hitting anything walks thousands
// of cells, its 63 neighbours a few dozen).  Here a fixed grid of waves stays resident; every wave
// owns a private chunk [next,end) of the ray queue (one atomic per chunk) and, whenever at least
// REFILL_MIN of its lanes are idle, hands fresh rays to exactly those lanes (ballot + prefix
// popcount).  Lanes therefore run different rays at different stages of the same traversal state
// machine:
//     phase A  descend inner nodes until this lane stands on a non-empty leaf (kdtree.c:501-570)
//     phase B  test the leaf's primitives one per iteration (kdtree.c:475-499)
//     then     hit → finish; miss → pop (far child, OUTER tMax) or finish as a miss
// The per-ray visiting order, comparisons and arithmetic are exactly those of closest_hit().
// Every wave leaves its loop once the work counter is past n and its own lanes are idle.
// ---------------------------------------------------------------------------------------------
#define REFILL_MIN 16

DEV uint64_t wave_bcast_u64(uint64_t v)
{
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// TAIL variant of the closest-hit kernel (k_trace_flat<..., true>): the rest of the bounce loop in ONE launch.  A lane
// that has found its ray's closest hit does not hand the hit to a shade pass: it waits (ST_SHADE) until enough lanes of
// its wave wait with it, Material_Bounce runs for them in place, and a path that goes on starts its next walk at once in
// the same lane; the lane asks the queue for a new path only when its path has ended.  The path state that only shading
// needs (throughput, RNG state, path id) stays in the path's queue record and is updated there.  Used for queues too
// short to fill the chip (a rank's share of a frame split over several GPUs, the deep passes of small frames), where a
// pass per depth costs more in launch ramps and drains than in work.
struct TailArgs {
    float*              result;   // per-path result slots, component stride rcap
    uint64_t            rcap;
    uint32_t            depth0;   // ray depth of the queue's rays
    uint32_t            max_depth;
    unsigned long long* counts;   // [d]: rays traced at depth d beyond depth0's own; [RTX_MAX_DEPTH]: paths shaded
    uint32_t            shade_min; // lanes that must wait before the shade block runs
    uint32_t            il_len;   // != 0: parts are not eighths of the queue but rows of il_len rays dealt out in turn (see the grab loop)
    const unsigned long long* n_dev; // not null: the queue's length is read here (the pass was launched before the host knew it; `n` is an upper bound)
    uint32_t*           long_list;  // not null: walks still in flight `long_after` iterations after the queue has run dry are handed
    uint32_t            long_after; //           to k_trace_long (their ray indices go here, work[LONG_COUNT] counts them)
};
// slots of the persistent kernels' `work` array (64 u64, cleared per launch; [8 p] = rays handed out of part p)
enum { LONG_DRY = 1, LONG_COUNT = 2, LONG_CURSOR = 3 };


template <bool COUNT>
__global__ __launch_bounds__(256) void k_trace_persist(SceneDev S, const float* __restrict__ rays, uint64_t stride, uint64_t n,
                                                       uint32_t* __restrict__ hits, uint64_t hstride, unsigned long long* work,
                                                       uint32_t chunk, uint64_t* counters, uint32_t /*tune*/, TailArgs /*unused*/)
{
    const uint32_t lane    = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    uint64_t next = 0, end = 0; // wave-uniform: this wave's private slice of the queue
    bool     exhausted = false; // wave-uniform: the global counter ran past n
    bool     active    = false; // this lane holds a ray

    f3       o = mk3(0, 0, 0), d = mk3(0, 0, 1), inv = mk3(0, 0, 0), odd = mk3(0, 0, 0);
    uint64_t ridx = 0;
    uint32_t node = 0;
    float    tMax = RTX_INF;
    int      sp   = 0;
    uint2    stk[RTX_STACK];
    bool     inLeaf = false, finish = false;
    uint32_t leafPos = 0, leafEnd = 0;
    float    bestT = RTX_INF, bu = 0.0f, bv = 0.0f;
    uint32_t best = HIT_MISS;
    Visits   vis  = {0, 0, 0};

    for (;;) {
        // ---------------- refill idle lanes ----------------
        const uint64_t idle  = __ballot(!active);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (!exhausted && nidle >= REFILL_MIN) {
            if (next >= end) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(work, (unsigned long long)chunk);
                base = wave_bcast_u64(base);
                if (base >= n) {
                    exhausted = true;
                } else {
                    next = base;
                    end  = base + chunk < n ? base + chunk : n;
                }
            }
            if (!exhausted) {
                const uint32_t avail = (uint32_t)(end - next);
                const uint32_t take  = nidle < avail ? nidle : avail;
                const uint32_t rank  = (uint32_t)__popcll(idle & lt_mask);
                if (!active && rank < take) {
                    ridx = next + rank;
                    const uint64_t cs = stride ? stride : 1, rb = (uint64_t)ridx * (stride ? 1 : Q_RECORD); // stride 0: AoS records
                    o    = mk3(rays[0 * cs + rb], rays[1 * cs + rb], rays[2 * cs + rb]);
                    d    = mk3(rays[3 * cs + rb], rays[4 * cs + rb], rays[5 * cs + rb]);
                    inv  = vdivc(mk3(1.0f, 1.0f, 1.0f), d); // Ray_Make's cached quotients (ray.c:12-13)
                    odd  = vdivc(o, d);
                    // keep the six quotients in registers: without this the optimiser sinks the
                    // divisions behind the per-node axis select and redoes one per node visit
                    asm volatile("" : "+v"(inv.x), "+v"(inv.y), "+v"(inv.z), "+v"(odd.x), "+v"(odd.y), "+v"(odd.z));
                    node   = S.root_literal;
                    tMax   = RTX_INF;
                    sp     = 0;
                    inLeaf = false;
                    active = true;
                }
                next += take;
            }
        }
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;
        }
        const uint32_t stop_idle = exhausted ? 64u : (uint32_t)REFILL_MIN;

        // ---------------- traverse until enough lanes have finished ----------------
        do {
            // phase A: inner nodes (and empty leaves) until this lane stands on a leaf with primitives
            while (active && !inLeaf) {
                const uint2    nd   = S.nodes[node];
                const uint32_t type = nd.y >> 30;
                const uint32_t idx  = nd.y & 0x3FFFFFFFu;
                if (type == 3u) {
                    if (COUNT) vis.leaf++;
                    if (nd.x != 0u) {
                        inLeaf  = true;
                        leafPos = idx;
                        leafEnd = idx + nd.x;
                        bestT   = RTX_INF;
                        best    = HIT_MISS;
                    } else if (sp > 0) { // empty leaf: no hit, resume the pending far child
                        const uint2 e = stk[--sp];
                        node          = e.x;
                        tMax          = __uint_as_float(e.y);
                    } else { // tree exhausted: no bounded hit
                        active = false;
                        finish = true;
                        best   = HIT_MISS;
                    }
                } else {
                    if (COUNT) vis.inner++;
                    const float    split = __uint_as_float(nd.x);
                    const uint32_t right = node + 1;
                    const bool     ax0 = type == 0u, ax1 = type == 1u;
                    const float    dA   = ax0 ? d.x : (ax1 ? d.y : d.z);
                    const float    oA   = ax0 ? o.x : (ax1 ? o.y : o.z);
                    const float    invA = ax0 ? inv.x : (ax1 ? inv.y : inv.z);
                    const float    oddA = ax0 ? odd.x : (ax1 ? odd.y : odd.z);
                    const bool     par    = fabsf(dA) < RT_EPSILON;               // kdtree.c:509
                    const float    tI     = split * invA - oddA;                  // :518
                    const float    tMaxN  = minf_(tMax, tI);                      // :519
                    const bool     behind = tI < RT_EPSILON;                      // :521
                    const bool     both   = !par && !behind;                      // :537
                    // side of the origin: >= for the parallel and the two-sided case (:511,:544); for the
                    // "plane behind" case > / < and, on equality, the sign of Ray_At(100) (:523-535)
                    const bool ge      = oA >= split;
                    const bool tie     = oA == split;
                    const bool tieSide = (oA + 100.0f * dA) >= split;
                    const bool goRight = (!par && behind && tie) ? tieSide : ge;
                    const uint32_t nearN = goRight ? right : idx;
                    if (both && tMaxN == tI) { // :551/:563 far child stays pending with the OUTER tMax
                        stk[sp++] = make_uint2(goRight ? idx : right, __float_as_uint(tMax));
                    }
                    node = nearN;
                    tMax = both ? tMaxN : tMax;
                }
            }
            // phase B: one primitive per iteration
            while (active && inLeaf) {
                const uint32_t r = S.leaf_ref[leafPos];
                if (COUNT) vis.prim++;
                float t, u = 0.0f, v = 0.0f;
                bool  hit;
                if (r & RTX_REF_SPHERE) {
                    hit = Sphere_HitAt(S.sph[r & 0x7FFFFFFFu], o, d, RT_EPSILON, tMax, t);
                } else {
                    const float4* tp = TRI_PTR(S, r);
                    const float4  a = tp[0], b = tp[1], c = tp[2];
                    hit = Triangle_HitAt(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), o, d, RT_EPSILON, tMax, t, u, v);
                }
                if (hit && t < bestT) { // strict: first in list wins ties (kdtree.c:485)
                    bestT = t;
                    best  = r;
                    bu    = u;
                    bv    = v;
                }
                if (++leafPos == leafEnd) {
                    inLeaf = false;
                    if (best != HIT_MISS) { // a leaf that reports a hit ends the walk
                        active = false;
                        finish = true;
                    } else if (sp > 0) {
                        const uint2 e = stk[--sp];
                        node          = e.x;
                        tMax          = __uint_as_float(e.y);
                    } else {
                        active = false;
                        finish = true;
                    }
                }
            }
            if (finish) {
                // Scene_ClosestHitInArray over the unbounded list, then the merge of scene.c:122-142
                uint32_t ref = best;
                float    tH = bestT, uH = bu, vH = bv;
                float    pBest = RTX_INF;
                uint32_t pRef  = HIT_MISS;
                for (uint32_t i = 0; i < S.n_planes; i++) {
                    const float* pl = S.plane + 8ull * i;
                    float        tp2;
                    if (Plane_HitAt(mk3(pl[0], pl[1], pl[2]), mk3(pl[3], pl[4], pl[5]), o, d, RT_EPSILON, RTX_INF, tp2) && tp2 < pBest) {
                        pBest = tp2;
                        pRef  = HIT_PLANE | i;
                    }
                }
                if (ref == HIT_MISS) tH = RTX_INF, uH = 0.0f, vH = 0.0f;
                if (pRef != HIT_MISS && (ref == HIT_MISS || pBest < tH)) ref = pRef, tH = pBest, uH = 0.0f, vH = 0.0f;
                hits[0 * hstride + ridx] = ref;
                hits[1 * hstride + ridx] = __float_as_uint(tH);
                hits[2 * hstride + ridx] = __float_as_uint(uH);
                hits[3 * hstride + ridx] = __float_as_uint(vH);
                finish = false;
            }
        } while ((uint32_t)__popcll(__ballot(!active)) < stop_idle);
    }
    if (COUNT) add_visits(counters, vis);
}

// ---------------------------------------------------------------------------------------------
// Flat state-machine variant of the persistent kernel.  rocprof showed k_trace_persist to be
// instruction-issue bound with as many scalar (exec-mask bookkeeping) as vector instructions: the
// nested while loops with loop-carried booleans make the structurizer emit ~25 SALU ops per step.
// Here every lane carries ONE state word and each iteration of the single loop runs at most four
// flat, independently skippable blocks:
//     INNER  one kd node (inner step, or arrival at a leaf)        kdtree.c:458-473,501-570
//     LEAF   one primitive of the current leaf                      kdtree.c:475-499
//     POP    resume the pending far child, or run out of tree       (the `return false` chain)
//     FIN    merge with the unbounded list, write the hit record    scene.c:100-143
// The inner step is written with selects only (its one memory side effect, the push, is a
// predicated store), and the leaf keeps the NEXT leaf reference in flight while testing the
// current primitive.
// ---------------------------------------------------------------------------------------------
// (DONE: walk finished, hit record not written yet.  "Free for a new ray" — DONE or IDLE — is one comparison: state >= ST_DONE)
enum { ST_INNER = 0, ST_LEAF = 1, ST_SHADE = 2, ST_DONE = 3, ST_IDLE = 4 };
// One wave per block: the waves share nothing, and a block's LDS and wave slots come free the moment ITS wave
// has finished its last ray — with 4-wave blocks the slowest of four decided, which kept the next kernel's
// blocks (the other lane of rtx_render_device, or simply the shade pass) out of the CU during the tail.
#ifndef TRACE_BLOCK
#define TRACE_BLOCK 64
#endif
// Block thresholds once a wave's queue has run dry.  Until then a block waits for enough lanes because every instruction it
// issues competes with five other waves' and new rays keep the lanes full; afterwards the chip empties, an instruction costs
// nothing, and a lane that waits for three others to reach a leaf only lengthens the launch's last millisecond (same-box A/B,
// three rounds: whole frame 79.95 -> 79.50 ms, a rank's share at N = 8 15.82 -> 15.39, at N = 64 6.14 -> 5.87; the tail
// launch's shade threshold lowered the same way changed nothing).
#ifndef DRAIN_MIN
#define DRAIN_MIN 1u
#endif
#ifndef LONG_POLL
#define LONG_POLL 128u // iterations without a refill after which a wave asks whether the launch is draining (k_trace_flat<.., LONG>)
#endif
#ifndef LDS_SLOTS
#define LDS_SLOTS 8 // power of two; 8 slots = 4 KB per wave (16 slots: fewer blocks per CU, 14 % slower)
#endif

// start of part p (0..8) of an n-ray queue: eighths, rounded up to whole 64-ray groups
DEV uint64_t part_begin(uint64_t n, uint32_t p)
{
    const uint64_t b = ((n * p / 8) + 63) & ~63ull;
    return p >= 8 || b > n ? n : b;
}

// which of the 8 XCDs this wave runs on (performance hint only, any value 0..7 is correct)
DEV uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}


// ---- wave-wide scans through DPP (all 64 lanes must be active) ---------------------------------------------
#define DPP_ADD(v, ctrl, rmask) ((v) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xF, false))
DEV uint32_t wave_scan_add(uint32_t v) // inclusive prefix sum over the lanes
{
    v = DPP_ADD(v, 0x111, 0xF); // row_shr:1
    v = DPP_ADD(v, 0x112, 0xF); // row_shr:2
    v = DPP_ADD(v, 0x114, 0xF); // row_shr:4
    v = DPP_ADD(v, 0x118, 0xF); // row_shr:8
    v = DPP_ADD(v, 0x142, 0xA); // row_bcast:15 into rows 1 and 3
    v = DPP_ADD(v, 0x143, 0xC); // row_bcast:31 into rows 2 and 3
    return v;
}
#define DPP_MAX(v, ctrl, rmask)                                                          \
    {                                                                                    \
        const uint32_t w_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xF, false); \
        v                 = w_ > v ? w_ : v;                                             \
    }
DEV uint32_t wave_scan_max(uint32_t v) // inclusive prefix maximum over the lanes
{
    DPP_MAX(v, 0x111, 0xF);
    DPP_MAX(v, 0x112, 0xF);
    DPP_MAX(v, 0x114, 0xF);
    DPP_MAX(v, 0x118, 0xF);
    DPP_MAX(v, 0x142, 0xA);
    DPP_MAX(v, 0x143, 0xC);
    return v;
}
DEV float lane_read(float v, uint32_t src) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)__float_as_uint(v))); }
DEV uint32_t lane_read(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v); }

template <bool COUNT, bool CULL, bool COOP, bool TAIL = false, bool LONG = false>
// (waves_per_eu(6): the walk needs 88 VGPRs as the compiler likes it, 5 waves per SIMD; held to 80 it spills two
// values that only the refill touches and runs 6 waves per SIMD — 95.0 -> 91.8 ms; 7 waves cost 8 spills and lose)
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(TAIL ? TAIL_WAVES : 6))) void k_trace_flat(SceneDev S, const float* __restrict__ rays, uint64_t stride, uint64_t n_bound,
                                                    uint32_t* __restrict__ hits, uint64_t hstride, unsigned long long* work,
                                                    uint32_t chunk_host, uint64_t* counters, uint32_t tune, TailArgs ta)
{
    // The queue's length: the launch argument, or — for a pass the host queued behind the shade pass that fills the queue,
    // without waiting for its count — what that shade pass left on the device (wave-uniform scalar loads)
    uint64_t n     = n_bound;
    uint32_t chunk = chunk_host;
    if (ta.n_dev) {
        n = *ta.n_dev;
        if ((tune >> 26) & 1u) chunk = n >= (64ull << 20) ? 160u : 64u; // launch_trace's choice, made here
    }
    // interleaved parts: rows of `il` rays (a multiple of the grab); 0 = the parts are eighths
    const uint32_t il = ta.il_len / chunk * chunk;
    // Block scheduling (wave-uniform): the INNER and LEAF blocks cost ~60 and ~90 vector instructions whatever
    // the number of lanes that need them, and the kernel is VALU-issue bound.  A block therefore runs only when
    // enough lanes wait for it (or nothing else can run); lanes in the other state simply keep their place.
    uint32_t       leaf_min = tune & 0xFFu, inner_min = (tune >> 8) & 0xFFu; // (1 and 1 once the queue has run dry: nothing is gained by waiting then)
    const uint32_t refill_min = (tune >> 16) & 0xFFu;
    // the root's box is tested when a ray is loaded — unless the caller knows the rays start on the scene's surfaces
    // (bounce passes), where that test never fails and the children's tests one step later do its work
    const bool     root_test = !TAIL && !((tune >> 25) & 1u);
    const uint32_t lane    = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    uint64_t next = 0, end = 0; // wave-uniform
    bool     exhausted = false; // wave-uniform
    uint32_t refill_at = refill_min; // wave-uniform: free lanes at which the wave stops to refill (64 once the queue has run dry: then to leave)
    uint32_t part      = xcc_id(), parts_left = 8; // wave-uniform
    int      state     = ST_IDLE;
    // Long walks (LONG).  A launch used to end on its longest walk — one ray skimming the model through
    // hundreds of leaves while the chip idles.  Here a wave that knows the queue has run dry everywhere gives its walks
    // `long_after` more iterations and then hands the ones still in flight to k_trace_long, which walks each of them with
    // all 64 lanes of a wave.  The loop pays one countdown for it: `patience` is wound up whenever the wave loads rays; a wave
    // that has not loaded any for LONG_POLL iterations (it is full of long walks, and never asks the queue) looks at the
    // flag the first wave to find the queue dry has raised.  (wave-uniform)
    uint32_t patience = LONG_POLL;

    f3       o = mk3(0, 0, 0), d = mk3(0, 0, 1), inv = mk3(0, 0, 0), odd = mk3(0, 0, 0);
    u3       rot = {0, 0, 0}; // CULL: 16 on the axes where the direction is negative (box_missed2)
    uint32_t ridx = 0; // < 2^32: launch_trace checks
    uint32_t node = 0;
    float    tMax = RTX_INF;
    // Traversal stack: the top LDS_SLOTS entries of every lane live in LDS ([slot][lane], 8-byte
    // entries, conflict-free); older entries spill one at a time to the per-lane scratch array and
    // come back one at a time.  With the whole stack in scratch the 6144 resident waves kept ~30 MB
    // of stack lines alive and evicted the kd-tree from the 8 x 4 MB L2s (rocprof: TCC hit rate 50 %).
    __shared__ uint2 s_stk[(TRACE_BLOCK / 64) * LDS_SLOTS * 64];
    uint2* const     lstk = s_stk + (threadIdx.x >> 6) * (LDS_SLOTS * 64) + lane;
    // COOP: per wave, 64 owner slots {closest key (t bits << 32 | slot), winner's {ref, u, v}} and the slot -> owner map
    __shared__ unsigned long long s_key[COOP ? TRACE_BLOCK : 1];
    __shared__ uint4              s_win[COOP ? TRACE_BLOCK : 1];
    __shared__ uint32_t           s_own[COOP ? TRACE_BLOCK : 1];
    const uint32_t wv = COOP ? (threadIdx.x >> 6) * 64 : 0; // this wave's 64 slots (plain LDS accesses in program order:
#define WKEY(i) s_key[wv + (i)]                                // the wave's LDS instructions execute in order, LDS_ORDER
#define WWIN(i) s_win[wv + (i)]                                // keeps the compiler from moving them)
#define WOWN(i) s_own[wv + (i)]
#define LDS_ORDER() asm volatile("" ::: "memory")
    __shared__ uint32_t s_tail[TAIL ? RTX_MAX_DEPTH + 64 : 1]; // TAIL: rays started per depth, [RTX_MAX_DEPTH]: paths shaded
    uint32_t            depth = 0;                             // TAIL: ray depth of this lane's path
    float* const        qw    = const_cast<float*>(rays);      // TAIL: the queue records are updated in place
    if (TAIL) {
        s_tail[lane] = 0u, s_tail[64 + lane] = 0u;
        LDS_ORDER();
    }
    int      sp   = 0;   // stack height
    int      base = 0;   // entries [base, sp) are in LDS, [0, base) in scratch
    uint2    stk[RTX_STACK];
    uint32_t leafPos = 0, leafEnd = 0, curRef = 0, curRef2 = 0; // curRef(2): references leafPos, leafPos + 1, prefetched
    float    bestT = RTX_INF, bu = 0.0f, bv = 0.0f;
    uint32_t best = HIT_MISS;
    float    pBest = RTX_INF;  // closest hit in the unbounded (plane) list, found when the ray is loaded
    uint32_t pRef  = HIT_MISS;
    bool     cullOK = false; // this lane's ray meets the preconditions of the box bounds
    Visits   vis  = {0, 0, 0, 0};
    uint32_t ws[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // COUNT: iterations, INNER runs / lanes, LEAF runs / lanes, POP runs / lanes, FIN runs

    // FIN: Scene_ClosestHit's merge with the unbounded list (scene.c:73-98, 122-142) and the hit record.  The unbounded
    // (plane) list does not depend on the walk; it is tested here, where the records of all the walks a wave has finished
    // are written at once, instead of when a ray is loaded — its result would sit in two registers for the whole walk.
    auto write_hit = [&](uint32_t ref, float tH, float uH, float vH) {
        if (ref == HIT_MISS) tH = RTX_INF, uH = 0.0f, vH = 0.0f;
        if (!TAIL) {
            pBest = RTX_INF;
            pRef  = HIT_MISS;
            for (uint32_t i = 0; i < S.n_planes; i++) {
                const float* pl = S.plane + 8ull * i;
                float        tp2;
                if (Plane_HitAt(mk3(pl[0], pl[1], pl[2]), mk3(pl[3], pl[4], pl[5]), o, d, RT_EPSILON, RTX_INF, tp2) && tp2 < pBest) {
                    pBest = tp2;
                    pRef  = HIT_PLANE | i;
                }
            }
        }
        if (pRef != HIT_MISS && (ref == HIT_MISS || pBest < tH)) ref = pRef, tH = pBest, uH = 0.0f, vH = 0.0f;
        hits[0 * hstride + ridx] = ref;
        hits[1 * hstride + ridx] = __float_as_uint(tH);
        hits[2 * hstride + ridx] = __float_as_uint(uH);
        hits[3 * hstride + ridx] = __float_as_uint(vH);
    };

    // COOP: a walk's closest hit so far is {bestT (register), WWIN(lane) = {ref, u, v} (LDS, written by the lane that made
    // the winning test)}; the three values come into registers only when the record is written
    auto take_best = [&]() {
        best = HIT_MISS, bu = 0.0f, bv = 0.0f;
        if (bestT < RTX_INF) {
            const uint4 w = WWIN(lane);
            best = w.x, bu = __uint_as_float(w.y), bv = __uint_as_float(w.z);
        }
    };

    for (;;) {
        // (a lane whose walk has finished keeps its hit — DONE — until the wave next refills or ends: the record's four
        // stores and the merge with the unbounded list then run for many lanes at once instead of for one or two)
        // (The loop's scheduling is scalar code, and at this kernel's issue rate a scalar instruction costs about half a vector
        // one — 40 of them per iteration, added for the measurement, cost 5.7 ms per frame —, so the common path through here is
        // one ballot, one count and one comparison: refill_at is refill_min, or 64 once the queue has run dry.)
        if (LONG && --patience == 0u) {
            if (exhausted) {
                // hand over: the ray indices of the walks in flight go to the long list, one atomic per wave
                const bool     walking = state == ST_INNER || state == ST_LEAF;
                const uint64_t wm      = __builtin_amdgcn_ballot_w64(walking);
                if (wm) {
                    unsigned long long at = 0;
                    if (lane == 0) at = atomicAdd(work + LONG_COUNT, (unsigned long long)__popcll(wm));
                    at = wave_bcast_u64(at);
                    if (walking) {
                        ta.long_list[(uint32_t)at + (uint32_t)__popcll(wm & lt_mask)] = ridx;
                        state = ST_IDLE;
                    }
                }
                patience = 0x7FFFFFFFu; // (all lanes are free now: the loop's head writes the finished walks' records and leaves)
            } else {
                const unsigned long long dry = __hip_atomic_load(work + LONG_DRY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                patience = LONG_POLL;
                if (__builtin_amdgcn_readfirstlane((uint32_t)dry) != 0u) {
                    // (rays of this wave's last grab that it has not loaded yet go first: any free lane takes one, and the
                    // grab loop then finds the parts empty by itself)
                    if (next >= end) exhausted = true, refill_at = 64u, leaf_min = DRAIN_MIN, inner_min = DRAIN_MIN, patience = ta.long_after + 1u, parts_left = 0;
                    else refill_at = 1u;
                }
            }
        }
        const uint64_t idle  = __builtin_amdgcn_ballot_w64(state >= ST_DONE);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (nidle >= refill_at) {
            if (!TAIL && state == ST_DONE) {
                if (COOP) take_best();
                write_hit(best, bestT, bu, bv);
                state = ST_IDLE;
            }
            if (exhausted) break; // (all 64 lanes are free and there is nothing left to load)
            if (next >= end) {
                // The queue is cut into 8 contiguous parts, one per XCD (work[8 * part] = rays handed out of that
                // part): the waves of an XCD drain their own part front to back, so the rays resident on an XCD
                // are neighbours in path order (= in the image, see slab_pixel) and share that XCD's L2.  A wave
                // whose part is empty moves on to the next one; after 8 empty parts it is done.
                // (Grabs of at most 128 rays: with 1024-ray grabs a launch ended with some waves a whole grab behind the
                // others — 104 -> 95 ms per frame; smaller ones cost the primary pass more in atomics than they balance.
                // Shrinking the grabs as a part drains was no better than the fixed size.)
                //
                // Interleaved parts (il != 0: the camera pass, see render_device_impl): the queue is cut into runs of il rays
                // and run k * 8 + p belongs to part p.
                while (parts_left > 0) {
                    unsigned long long base = 0;
                    const uint64_t     grab = chunk;
                    if (lane == 0) base = atomicAdd(work + 8 * part, (unsigned long long)chunk);
                    base = wave_bcast_u64(base);
                    if (il) {
                        // (il is a multiple of the grab, so a grab never crosses a row's end; offsets past 2^32 are past the queue)
                        const uint32_t v = (uint32_t)min(base, (unsigned long long)0xFFFFFFFFu), r = v / il;
                        const uint64_t at = ((uint64_t)r * 8u + part) * il + (v - r * il);
                        if (at < n) {
                            next = at;
                            end  = at + grab < n ? at + grab : n;
                            break;
                        }
                    } else {
                        const uint64_t pb = part_begin(n, part), pe = part_begin(n, part + 1);
                        base += pb;
                        if (base < pe) {
                            next = base;
                            end  = base + grab < pe ? base + grab : pe;
                            break;
                        }
                    }
                    part = (part + 1) & 7u;
                    parts_left--;
                }
                if (parts_left == 0) {
                    exhausted = true, refill_at = 64u, leaf_min = DRAIN_MIN, inner_min = DRAIN_MIN;
                    if (LONG) {
                        patience = ta.long_after + 1u;
                        if (lane == 0) __hip_atomic_store(work + LONG_DRY, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (!exhausted) {
                if (LONG) patience = LONG_POLL;
                const uint32_t avail = (uint32_t)(end - next);
                const uint32_t take  = nidle < avail ? nidle : avail;
                const uint32_t rank  = (uint32_t)__popcll(idle & lt_mask);
                if (state == ST_IDLE && rank < take) {
                    ridx = (uint32_t)(next + rank);
                    const uint64_t cs = stride ? stride : 1, rb = (uint64_t)ridx * (stride ? 1 : Q_RECORD); // stride 0: AoS records
                    o    = mk3(rays[0 * cs + rb], rays[1 * cs + rb], rays[2 * cs + rb]);
                    d    = mk3(rays[3 * cs + rb], rays[4 * cs + rb], rays[5 * cs + rb]);
                    inv  = vdivc(mk3(1.0f, 1.0f, 1.0f), d); // Ray_Make's cached quotients (ray.c:12-13)
                    odd  = vdivc(o, d);
                    if (CULL) {
                        // 1/0 = inf would turn (b - o) * inv into NaN when b == o.  The literal walk never uses inv
                        // on an axis with |d| < 1e-4 (kdtree.c:509-516), so the box test's finite stand-in can
                        // live in the same register.
                        const float big = 3.0e38f;
                        inv.x = fabsf(inv.x) > big ? copysignf(big, inv.x) : inv.x;
                        inv.y = fabsf(inv.y) > big ? copysignf(big, inv.y) : inv.y;
                        inv.z = fabsf(inv.z) > big ? copysignf(big, inv.z) : inv.z;
                        const float rx = o.x - S.cull_c[0], ry = o.y - S.cull_c[1], rz = o.z - S.cull_c[2];
                        const float r2 = rx * rx + ry * ry + rz * rz, d2 = d.x * d.x + d.y * d.y + d.z * d.z;
                        cullOK         = r2 <= S.cull_ro2 && fabsf(d2 - 1.0f) <= 1.0e-3f; // false for any NaN / inf component
                    }
                    asm volatile("" : "+v"(inv.x), "+v"(inv.y), "+v"(inv.z), "+v"(odd.x), "+v"(odd.y), "+v"(odd.z));
                    if (CULL) {
                        rot = box_rotations(inv);
                        asm volatile("" : "+v"(rot.x), "+v"(rot.y), "+v"(rot.z));
                    }
                    // TAIL: Scene_ClosestHitInArray (scene.c:73-98) when the ray is loaded (the other kernels: in write_hit)
                    pBest = RTX_INF;
                    pRef  = HIT_MISS;
                    for (uint32_t i = 0; TAIL && i < S.n_planes; i++) {
                        const float* pl = S.plane + 8ull * i;
                        float        tp2;
                        if (Plane_HitAt(mk3(pl[0], pl[1], pl[2]), mk3(pl[3], pl[4], pl[5]), o, d, RT_EPSILON, RTX_INF, tp2) && tp2 < pBest) {
                            pBest = tp2;
                            pRef  = HIT_PLANE | i;
                        }
                    }
                    node  = CULL ? S.root : S.root_literal;
                    tMax  = RTX_INF;
                    sp    = 0;
                    base  = 0;
                    state = ST_INNER;
                    if (TAIL) depth = ta.depth0;
                    if (CULL && root_test && cullOK && box_missed(S, S.root_q[0], S.root_q[1], S.root_q[2], o, inv, RTX_INF)) {
                        // the ray misses the box of everything in the tree: finished here, the lane stays free
                        if (COUNT) vis.culled++;
                        write_hit(HIT_MISS, RTX_INF, 0.0f, 0.0f);
                        state = ST_IDLE;
                    }
                }
                next += take;
            }
            // (no `continue`: the lanes just loaded take their first step in this pass through the loop)
        }

        bool           pop = false, fin = false;
        const uint32_t nI = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(state == ST_INNER));
        const uint32_t nS = TAIL ? (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(state == ST_SHADE)) : 0u;
        const uint32_t nL = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(state == ST_LEAF));
        const bool     doLeaf  = TAIL ? (nL >= leaf_min || (nI == 0u && nL != 0u)) : (nL >= leaf_min || nI == 0u);
        const bool     doInner = nI >= inner_min || !doLeaf;
        // TAIL: shading runs for a batch of waiting lanes, or when nothing else in the wave can move
        const bool     doShade = TAIL && nS != 0u && (nS >= ta.shade_min || (nI == 0u && nL == 0u));
        if (COUNT) { // wave statistics (wave-uniform): block executions and the lanes they served
            ws[0]++;
            if (doInner && nI) ws[1]++, ws[2] += nI;
            if (doLeaf && nL) ws[3]++, ws[4] += nL;
        }
        const bool ranInner = doInner && state == ST_INNER;
        if (ranInner) {
            uint2    nd;
            uint4    qa = make_uint4(0, 0, 0, 0), qb = qa; // CULL: {w0, w1, L.x, L.y}, {L.z, R.x, R.y, R.z}: the children's boxes
            uint2    qc = make_uint2(0, 0);                // CULL: {right link | carried plane's axis and side, its split}
            bool     cut = false; // CULL: the cut-off node folded into this step ends the walk of its subtree
            if (CULL) {
                const uint32_t* rec = S.nodes40 + (uint64_t)NODE_WORDS * node;
                qa = *(const uint4*)rec, qb = *(const uint4*)(rec + 4), qc = *(const uint2*)(rec + 8);
                // the whole record requested before anything waits for its first part (else the compiler sinks the boxes'
                // loads below the leaf test: two trips to the cache per step)
                asm volatile("" : "+v"(qa.x), "+v"(qa.y), "+v"(qa.z), "+v"(qa.w), "+v"(qb.x), "+v"(qb.y), "+v"(qb.z), "+v"(qb.w), "+v"(qc.x), "+v"(qc.y));
                nd = make_uint2(qa.x, qa.y);
                const uint32_t cax = (qc.x >> 29) & 3u;
                if (cax != 3u) {
                    // This node's parent P is a cut-off node — one child an empty leaf — that has no record of its own:
                    // CheckHitInternalNode for P (kdtree.c:501-570), with the empty child's visit (no hit, :475-499) left out.
                    if (COUNT) vis.inner++;
                    const float split = __uint_as_float(qc.y);
                    const bool  cx0 = cax == 0u, cx1 = cax == 1u;
                    const float dA   = cx0 ? d.x : (cx1 ? d.y : d.z);
                    const float oA   = cx0 ? o.x : (cx1 ? o.y : o.z);
                    const float invA = cx0 ? inv.x : (cx1 ? inv.y : inv.z);
                    const float oddA = cx0 ? odd.x : (cx1 ? odd.y : odd.z);
                    const bool  par    = fabsf(dA) < RT_EPSILON; // :509
                    const float tI     = split * invA - oddA;    // :518
                    const float tMaxN  = minf_(tMax, tI);        // :519
                    const bool  behind = tI < RT_EPSILON;        // :521
                    const bool  both   = !par && !behind;        // :537
                    const bool  ge      = oA >= split;
                    const bool  tieSide = (oA + 100.0f * dA) >= split;
                    const bool  goRight = (!par && behind && oA == split) ? tieSide : ge;
                    const bool  here    = goRight == ((qc.x >> 31) != 0u); // the origin's side of P's plane is this node's side
                    // here: this node is P's near child, walked with min(tMax, tI) (:546-549, :558-561); P's far child is empty.
                    // else: the near child is the empty leaf; this node is walked — with the outer tMax — only if the plane
                    // lies within the range (:551, :563), else P reports no hit
                    cut  = !here && !(both && tMaxN == tI);
                    tMax = (here && both) ? tMaxN : tMax;
                }
            } else {
                nd = S.nodes[node];
            }
            const uint32_t type = nd.y >> 30;
            const uint32_t idx  = nd.y & 0x3FFFFFFFu;
            if (CULL && cut) {
                pop = true;
            } else if (type == 3u) {
                if (COUNT) vis.leaf++;
                if (nd.x != 0u) {
                    state   = ST_LEAF;
                    leafPos = idx;
                    leafEnd = idx + nd.x;
                    bestT   = RTX_INF;
                    best    = HIT_MISS;
                    if (!COOP) {
                        curRef = S.leaf_ref[idx];
                        if (nd.x > 1u) curRef2 = S.leaf_ref[idx + 1];
                    }
                } else {
                    pop = true; // empty leaf: no hit here
                }
            } else {
                if (COUNT) vis.inner++;
                const float    split = __uint_as_float(nd.x);
                const uint32_t right = CULL ? (qc.x & 0x1FFFFFFFu) : node + 1; // (CULL: links skip the folded cut-off nodes)
                const bool     ax0 = type == 0u, ax1 = type == 1u;
                const float    dA   = ax0 ? d.x : (ax1 ? d.y : d.z);
                const float    oA   = ax0 ? o.x : (ax1 ? o.y : o.z);
                const float    invA = ax0 ? inv.x : (ax1 ? inv.y : inv.z);
                const float    oddA = ax0 ? odd.x : (ax1 ? odd.y : odd.z);
                const bool     par    = fabsf(dA) < RT_EPSILON; // kdtree.c:509
                const float    tI     = split * invA - oddA;    // :518
                const float    tMaxN  = minf_(tMax, tI);        // :519
                const bool     behind = tI < RT_EPSILON;        // :521
                const bool     both   = !par && !behind;        // :537
                const bool     ge      = oA >= split;           // :511, :544
                const bool     tieSide = (oA + 100.0f * dA) >= split; // Ray_At(ray,100)[axis], :529-535
                const bool     goRight = (!par && behind && oA == split) ? tieSide : ge;
                const bool     farPending = both && tMaxN == tI; // :551/:563 far child, with the OUTER tMax
                const float    tNear      = both ? tMaxN : tMax;
                const uint32_t nearNode = goRight ? right : idx, farNode = goRight ? idx : right;
                bool           missNear = false, missFar = false;
                if (CULL) {
                    // both children's boxes against the segments they would be walked with; a child the ray cannot
                    // reach is neither fetched nor pushed
                    bool miss[2];
                    box_missed2(S, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w, o, inv, rot, goRight ? f2v{tMax, tNear} : f2v{tNear, tMax}, miss);
                    // (xor / and of lane masks are scalar instructions; `goRight ? miss[1] : miss[0]` made the compiler turn
                    // both flags into 0/1 registers and select those: 8 vector instructions per step)
                    const bool differ = miss[0] != miss[1], swap = goRight && differ;
                    missNear = cullOK && (miss[0] != swap);
                    missFar  = cullOK && (miss[1] != swap);
                    if (COUNT) vis.culled += (missNear ? 1u : 0u) + (farPending && missFar ? 1u : 0u);
                }
                const bool pushFar = farPending && !missFar;
                if (!missNear) {
                    if (pushFar) {
                        if (sp - base == LDS_SLOTS) { // window full: the oldest resident entry goes to scratch
                            stk[base] = lstk[(base & (LDS_SLOTS - 1)) * 64];
                            base++;
                        }
                        lstk[(sp & (LDS_SLOTS - 1)) * 64] = make_uint2(farNode, __float_as_uint(tMax));
                        sp++;
                    }
                    node = nearNode;
                    tMax = tNear;
                } else if (pushFar) { // near side unreachable: straight to the far side (outer tMax stays)
                    node = farNode;
                } else {
                    pop = true;
                }
            }
        }
        if (COOP && doLeaf) {
            // Cooperative LEAF (all 64 lanes, wave-uniform): the pending primitive tests of every lane that sits in
            // a leaf are laid end to end and dealt out one per lane, whoever owns the ray.  CheckHitLeafNode
            // (kdtree.c:480-498) tests every primitive of a leaf against the same tMax and keeps the strictly
            // smaller t, first in the list on ties: the minimum over (t, position in the list), which is what the
            // 64-bit ds_min below computes (t > 0, so its bit pattern orders like the number).
            const bool     isL   = state == ST_LEAF;
            const uint32_t cnt   = isL ? leafEnd - leafPos : 0u;
            const uint32_t incl  = wave_scan_add(cnt), excl = incl - cnt;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            WOWN(lane) = 0u;
            // a test takes an owner's slot only with a t strictly below the owner's closest so far (an earlier step's hit
            // wins a tie: it is earlier in the list); among this step's tests the smaller (t, position) wins
            WKEY(lane) = (unsigned long long)__float_as_uint(bestT) << 32;
            LDS_ORDER();
            if (cnt != 0u && excl < 64u) WOWN(excl) = excl << 6 | lane; // head of this owner's run of slots
            LDS_ORDER();
            const uint32_t pk    = wave_scan_max(WOWN(lane));
            const uint32_t owner = pk & 63u, k = lane - (pk >> 6);
            const bool     valid = lane < total;
            const f3       so = mk3(lane_read(o.x, owner), lane_read(o.y, owner), lane_read(o.z, owner));
            const f3       sd = mk3(lane_read(d.x, owner), lane_read(d.y, owner), lane_read(d.z, owner));
            const float    sT = lane_read(tMax, owner);
            const uint32_t sP = lane_read(leafPos, owner);
            bool               hitS = false;
            unsigned long long keyS = 0;
            uint32_t           refS = 0;
            float              uS = 0.0f, vS = 0.0f;
            if (valid) {
                refS = S.leaf_ref[sP + k];
                float tS;
                if (refS & RTX_REF_SPHERE) {
                    hitS = Sphere_HitAt(S.sph[refS & 0x7FFFFFFFu], so, sd, RT_EPSILON, sT, tS);
                } else {
                    const float4* tp = TRI_PTR(S, refS);
                    float4        a  = tp[0];
#if TRI_WAIT
                    // (Round 1: a vector-L1 miss is not merged with a miss pending on the same line, so taking the first 16 bytes,
                    // waiting, and reading the rest from L1 saved two of three L2 requests — worth 9 % while the passes were
                    // bound by the rate of L2 misses.  Round 2, v24: the kernel is bound by issue and by its own waits, and the
                    // second wait costs 3.4 ms per frame: 82.1 -> 78.7 ms with the three loads issued together.)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)::"memory");
#endif
                    const float4 b = tp[1];
                    const float  c = *(const float*)(tp + 2);
                    hitS = Triangle_HitAt_straight(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c), so, sd, RT_EPSILON, sT, tS, uS, vS);
                }
                if (hitS) {
                    keyS = (unsigned long long)__float_as_uint(tS) << 32 | (lane + 1u);
                    atomicMin(&WKEY(owner), keyS);
                }
            }
            LDS_ORDER();
            if (hitS && WKEY(owner) == keyS) WWIN(owner) = make_uint4(refS, __float_as_uint(uS), __float_as_uint(vS), 0u);
            LDS_ORDER();
            if (cnt != 0u && excl < 64u) {
                const uint32_t used = cnt < 64u - excl ? cnt : 64u - excl;
                if (COUNT) vis.prim += used;
                bestT = __uint_as_float((uint32_t)(WKEY(lane) >> 32)); // (the winner's {ref, u, v} stay in WWIN(lane))
                leafPos += used;
                if (leafPos == leafEnd) {
                    if (bestT < RTX_INF) fin = true; // a leaf that reports a hit ends the walk
                    else pop = true;
                }
            }
        } else if (!COOP && doLeaf && state == ST_LEAF && !ranInner) {
            // Two primitives of the leaf per step (when two are left and both are triangles): the pair goes through
            // the packed test, one vector instruction for both (Triangle_HitAt2); results are taken in list order.
            const uint32_t r0 = curRef, r1 = curRef2;
            const bool     sph  = (r0 & RTX_REF_SPHERE) != 0u;
            const bool     pair = leafPos + 1 < leafEnd && !sph && !(r1 & RTX_REF_SPHERE);
            const uint32_t step = pair ? 2u : 1u;
            if (leafPos + step < leafEnd) curRef = S.leaf_ref[leafPos + step]; // next references in flight
            if (leafPos + step + 1 < leafEnd) curRef2 = S.leaf_ref[leafPos + step + 1];
            if (COUNT) vis.prim += step;
            if (sph) {
                float t;
                if (Sphere_HitAt(S.sph[r0 & 0x7FFFFFFFu], o, d, RT_EPSILON, tMax, t) && t < bestT) bestT = t, best = r0, bu = 0.0f, bv = 0.0f;
            } else {
                const float4* tpA = TRI_PTR(S, r0);
                const float4* tpB = TRI_PTR(S, (pair ? r1 : r0));
                float4        aA = tpA[0], aB = tpB[0];
#if TRI_WAIT
                // One L2 request per triangle instead of three: a vector-L1 miss is not merged with a miss already
                // pending on the same line (probe: tools/dbg/line.hip).  So take the first 16 bytes, wait, and read
                // the rest from L1.
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(aA.x), "+v"(aA.y), "+v"(aA.z), "+v"(aA.w), "+v"(aB.x), "+v"(aB.y), "+v"(aB.z), "+v"(aB.w)::"memory");
#endif
                const float4 bA = tpA[1], bB = tpB[1];
                const float  cA = *(const float*)(tpA + 2), cB = *(const float*)(tpB + 2); // 36 bytes of a record are used
                bool         hit[2];
                float        t[2], u[2], v[2];
                Triangle_HitAt2(f3x2{f2v{aA.x, aB.x}, f2v{aA.y, aB.y}, f2v{aA.z, aB.z}}, f3x2{f2v{aA.w, aB.w}, f2v{bA.x, bB.x}, f2v{bA.y, bB.y}},
                                f3x2{f2v{bA.z, bB.z}, f2v{bA.w, bB.w}, f2v{cA, cB}}, o, d, RT_EPSILON, tMax, hit, t, u, v);
                if (hit[0] && t[0] < bestT) bestT = t[0], best = r0, bu = u[0], bv = v[0]; // strict: first in list wins ties (kdtree.c:485)
                if (pair && hit[1] && t[1] < bestT) bestT = t[1], best = r1, bu = u[1], bv = v[1];
            }
            leafPos += step;
            if (leafPos == leafEnd) {
                if (best != HIT_MISS) fin = true; // a leaf that reports a hit ends the walk
                else pop = true;
            }
        }
        if (COUNT) {
            const uint32_t np = (uint32_t)__popcll(__ballot(pop)), nf = (uint32_t)__popcll(__ballot(fin));
            if (np) ws[5]++, ws[6] += np;
            if (nf) ws[7]++;
        }
        if (pop) {
            if (sp > 0) {
                sp--;
                if (sp < base) { // window empty (rare): the entry comes back from scratch into its LDS slot first, so that
                    lstk[(sp & (LDS_SLOTS - 1)) * 64] = stk[sp]; // the common path is one LDS read, not a generic-address load
                    base = sp;                                   // with its address selects (a wave's LDS accesses are in order)
                }
                const uint2 e = lstk[(sp & (LDS_SLOTS - 1)) * 64];
                node          = e.x;
                tMax          = __uint_as_float(e.y);
                state         = ST_INNER;
            } else { // tree exhausted
                best  = HIT_MISS;
                bestT = RTX_INF;
                fin   = true;
            }
        }
        if (fin) {
            if (TAIL) { // Scene_ClosestHit's merge with the unbounded list (as in write_hit); the hit stays in the lane
                if (COOP) take_best();
                if (best == HIT_MISS) bestT = RTX_INF, bu = 0.0f, bv = 0.0f;
                if (pRef != HIT_MISS && (best == HIT_MISS || pBest < bestT)) best = pRef, bestT = pBest, bu = 0.0f, bv = 0.0f;
                state = ST_SHADE;
            } else {
                state = ST_DONE;
            }
        }
        if (TAIL && doShade && state == ST_SHADE) {
            // k_shade's body for the lanes that wait: RayColor's miss / bounce / stop cases (renderer.c:54-76)
            const uint64_t cs = stride ? stride : 1, rb = (uint64_t)ridx * (stride ? 1 : Q_RECORD);
            f3             T   = mk3(qw[Q_TR * cs + rb], qw[Q_TG * cs + rb], qw[Q_TB * cs + rb]);
            const uint32_t pid = __float_as_uint(qw[Q_PID * cs + rb]);
            bool           goOn = false;
            f3             L;
            if (best == HIT_MISS) {
                L = vmulc(T, Skybox_ColorAt(S, d));
            } else {
                Rng  rng = Rng{__float_as_uint(qw[Q_S0 * cs + rb]), __float_as_uint(qw[Q_S1 * cs + rb]), __float_as_uint(qw[Q_S2 * cs + rb]),
                               __float_as_uint(qw[Q_S3 * cs + rb])};
                HitD h   = make_hitinfo(S, o, d, best, bestT, bu, bv);
                f3   surf, emit, dout;
                const bool cont = Material_Bounce(S, S.mats[h.mat], d, h, rng, surf, emit, dout);
                atomicAdd(&s_tail[RTX_MAX_DEPTH], 1u);
                if (cont) {
                    T = vmulc(T, surf);
                    if (depth + 1u >= ta.max_depth) {
                        L = vmulc(T, mk3(0.0f, 0.0f, 0.0f)); // depth exhausted: surface * black, NaNs included (see k_shade)
                    } else {
                        goOn = true;
                        o    = h.pos;
                        d    = vnorm(dout); // Ray_Make
                        qw[Q_TR * cs + rb] = T.x, qw[Q_TG * cs + rb] = T.y, qw[Q_TB * cs + rb] = T.z;
                        qw[Q_S0 * cs + rb] = __uint_as_float(rng.s0), qw[Q_S1 * cs + rb] = __uint_as_float(rng.s1);
                        qw[Q_S2 * cs + rb] = __uint_as_float(rng.s2), qw[Q_S3 * cs + rb] = __uint_as_float(rng.s3);
                    }
                } else {
                    L = vmulc(T, emit);
                }
            }
            if (goOn) {
                // the next ray of the path, in the same lane: what the refill does for a ray it loads
                depth++;
                atomicAdd(&s_tail[depth], 1u);
                inv = vdivc(mk3(1.0f, 1.0f, 1.0f), d);
                odd = vdivc(o, d);
                if (CULL) {
                    const float big = 3.0e38f;
                    inv.x = fabsf(inv.x) > big ? copysignf(big, inv.x) : inv.x;
                    inv.y = fabsf(inv.y) > big ? copysignf(big, inv.y) : inv.y;
                    inv.z = fabsf(inv.z) > big ? copysignf(big, inv.z) : inv.z;
                    const float rx = o.x - S.cull_c[0], ry = o.y - S.cull_c[1], rz = o.z - S.cull_c[2];
                    const float r2 = rx * rx + ry * ry + rz * rz, d2 = d.x * d.x + d.y * d.y + d.z * d.z;
                    cullOK         = r2 <= S.cull_ro2 && fabsf(d2 - 1.0f) <= 1.0e-3f;
                }
                asm volatile("" : "+v"(inv.x), "+v"(inv.y), "+v"(inv.z), "+v"(odd.x), "+v"(odd.y), "+v"(odd.z));
                if (CULL) {
                    rot = box_rotations(inv);
                    asm volatile("" : "+v"(rot.x), "+v"(rot.y), "+v"(rot.z));
                }
                pBest = RTX_INF;
                pRef  = HIT_MISS;
                for (uint32_t i = 0; i < S.n_planes; i++) {
                    const float* pl = S.plane + 8ull * i;
                    float        tp2;
                    if (Plane_HitAt(mk3(pl[0], pl[1], pl[2]), mk3(pl[3], pl[4], pl[5]), o, d, RT_EPSILON, RTX_INF, tp2) && tp2 < pBest) {
                        pBest = tp2;
                        pRef  = HIT_PLANE | i;
                    }
                }
                node  = CULL ? S.root : S.root_literal;
                tMax  = RTX_INF;
                sp    = 0;
                base  = 0;
                state = ST_INNER;
            } else {
                ta.result[0 * ta.rcap + pid] = L.x, ta.result[1 * ta.rcap + pid] = L.y, ta.result[2 * ta.rcap + pid] = L.z;
                state = ST_IDLE;
            }
        }
    }
    if (TAIL) { // one wave per block: its LDS counters go out once
        LDS_ORDER();
        const uint32_t c0 = s_tail[lane], c1 = lane == 0 ? s_tail[RTX_MAX_DEPTH] : 0u;
        if (c0) atomicAdd(ta.counts + lane, (unsigned long long)c0);
        if (c1) atomicAdd(ta.counts + RTX_MAX_DEPTH, (unsigned long long)c1);
    }
    if (COUNT) {
        add_visits(counters, vis);
        if (lane == 0)
            for (int i = 0; i < 8; i++) atomicAdd((unsigned long long*)&counters[8 + i], (unsigned long long)ws[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// k_trace_long — one long walk by the 64 lanes of a wave.
//
// A ray that skims the model visits hundreds of leaves one after the other; at ~1 us per dependent step that is the
// last millisecond of every closest-hit launch, with the chip empty.  The walk (KDTree_HitAt, kdtree.c:453-577, with the
// culling of DESIGN.md §5) is a depth-first search: near child, then — if the near subtree reported nothing and the plane
// lies within the range — the far child with the OUTER tMax; the first leaf that accepts a hit ends it.  The tMax a node is
// entered with depends on the path to it alone, so what a subtree reports does not depend on its elder siblings: the
// pending far subtrees can be walked AHEAD of time.  Here a lane that is free takes a pending far subtree off a busy lane
// (the youngest one: the next in the order) and walks it speculatively; every node carries its place in the literal
// visiting order as a key — bit (63 - level) = 1 where the path took the far child of a two-sided inner step —, a leaf that
// accepts a hit reports with its key, the smallest key is the walk's answer (one 64-bit LDS minimum), and whatever stands
// behind the smallest key reported so far is dropped.  The set of nodes walked is a superset of the sequential walk's and
// every one of them is entered with the same tMax, so the leaf with the smallest key that accepts a hit is the first leaf
// the sequential walk finds, with the same hit: bit for bit the same record.
// Checked: oracle_trace_speculative walks ALL reachable subtrees on the CPU and takes the smallest key (tests/test_culling_cpu.py:
// equal to the literal oracle on the golden rays, the awkward rays and the soups), and the GPU tests send EVERY ray of the
// golden sets, the 1080p checksums, the soups and whole frames through this kernel (knob LONG_ALL): the literal kernel's bits.
// One ray per wave.  A tree deeper than the key has room for is walked by one lane alone (no key needed).
// ---------------------------------------------------------------------------------------------
#define LONG_LEVEL_MAX 57 // key bits 63 .. 6 hold the path, bits 5 .. 0 the level
__global__ __launch_bounds__(64) void k_trace_long(SceneDev S, const float* __restrict__ rays, uint64_t stride, uint64_t n_bound, const unsigned long long* n_dev,
                                                   uint32_t* __restrict__ hits, uint64_t hstride, const uint32_t* __restrict__ list,
                                                   unsigned long long* work, uint32_t tree_depth)
{
    __shared__ unsigned long long s_best;
    __shared__ uint4              s_res;
    __shared__ uint4              s_mail[64];
    const uint32_t lane    = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // rays to trace: the long list's entries, or (list == nullptr: every ray of the queue, the tests' mode) 0 .. n - 1
    uint64_t count = list ? __hip_atomic_load(work + LONG_COUNT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (n_dev ? *n_dev : n_bound);
    if (list && count > LONG_CAP) count = LONG_CAP;
    const bool solo = tree_depth > LONG_LEVEL_MAX;
    uint4      stk[RTX_STACK]; // this lane's pending far subtrees: {node, tMax, key lo, key hi} (the level rides in the key's low bits)

    for (;;) {
        unsigned long long gi = 0;
        if (lane == 0) gi = atomicAdd(work + LONG_CURSOR, 1ull);
        gi = wave_bcast_u64(gi);
        if (gi >= count) break;
        const uint32_t ridx = list ? (uint32_t)__builtin_amdgcn_readfirstlane((int)list[gi]) : (uint32_t)gi;
        const uint64_t cs = stride ? stride : 1, rb = (uint64_t)ridx * (stride ? 1 : Q_RECORD);
        const f3       o = mk3(rays[0 * cs + rb], rays[1 * cs + rb], rays[2 * cs + rb]);
        const f3       d = mk3(rays[3 * cs + rb], rays[4 * cs + rb], rays[5 * cs + rb]);
        f3       inv = vdivc(mk3(1.0f, 1.0f, 1.0f), d); // Ray_Make's cached quotients (ray.c:12-13), as k_trace_flat's refill has them
        const f3 odd = vdivc(o, d);
        const float big = 3.0e38f;
        inv.x = fabsf(inv.x) > big ? copysignf(big, inv.x) : inv.x;
        inv.y = fabsf(inv.y) > big ? copysignf(big, inv.y) : inv.y;
        inv.z = fabsf(inv.z) > big ? copysignf(big, inv.z) : inv.z;
        const float rx = o.x - S.cull_c[0], ry = o.y - S.cull_c[1], rz = o.z - S.cull_c[2];
        const float r2 = rx * rx + ry * ry + rz * rz, d2 = d.x * d.x + d.y * d.y + d.z * d.z;
        const bool  cullOK = r2 <= S.cull_ro2 && fabsf(d2 - 1.0f) <= 1.0e-3f;
        if (lane == 0) s_best = ~0ull;
        LDS_ORDER();

        bool               have = lane == 0; // this lane stands on a node
        uint32_t           node = S.root, level = 0;
        unsigned long long key  = 0;
        float              tMax = RTX_INF;
        int                sp   = 0;
        for (;;) {
            bool  produced = false, pop = false;
            uint4 far      = make_uint4(0, 0, 0, 0);
            LDS_ORDER();
            if (have && !solo && (key | level) > s_best) have = false, sp = 0; // this node and all this lane holds come later in the order
            if (have) {
                const uint32_t* rec = S.nodes40 + (uint64_t)NODE_WORDS * node;
                const uint4     qa = *(const uint4*)rec, qb = *(const uint4*)(rec + 4);
                const uint2     qc = *(const uint2*)(rec + 8);
                const uint32_t  cax = (qc.x >> 29) & 3u;
                if (cax != 3u) { // the folded cut-off node above this one: k_trace_flat's code, CheckHitInternalNode kdtree.c:501-570
                    const float split = __uint_as_float(qc.y);
                    const bool  cx0 = cax == 0u, cx1 = cax == 1u;
                    const float dA   = cx0 ? d.x : (cx1 ? d.y : d.z);
                    const float oA   = cx0 ? o.x : (cx1 ? o.y : o.z);
                    const float invA = cx0 ? inv.x : (cx1 ? inv.y : inv.z);
                    const float oddA = cx0 ? odd.x : (cx1 ? odd.y : odd.z);
                    const bool  par    = fabsf(dA) < RT_EPSILON;
                    const float tI     = split * invA - oddA;
                    const float tMaxN  = minf_(tMax, tI);
                    const bool  behind = tI < RT_EPSILON;
                    const bool  both   = !par && !behind;
                    const bool  ge      = oA >= split;
                    const bool  tieSide = (oA + 100.0f * dA) >= split;
                    const bool  goRight = (!par && behind && oA == split) ? tieSide : ge;
                    const bool  here    = goRight == ((qc.x >> 31) != 0u);
                    pop  = !here && !(both && tMaxN == tI);
                    tMax = (here && both) ? tMaxN : tMax;
                }
                const uint32_t type = qa.y >> 30, idx = qa.y & 0x3FFFFFFFu;
                if (pop) {
                } else if (type == 3u) { // CheckHitLeafNode, kdtree.c:475-499
                    float    bestT = RTX_INF, bu = 0.0f, bv = 0.0f;
                    uint32_t best  = HIT_MISS;
                    for (uint32_t i = 0; i < qa.x; i++) {
                        const uint32_t r = S.leaf_ref[idx + i];
                        float          t, u = 0.0f, v = 0.0f;
                        bool           hit;
                        if (r & RTX_REF_SPHERE) {
                            hit = Sphere_HitAt(S.sph[r & 0x7FFFFFFFu], o, d, RT_EPSILON, tMax, t);
                        } else {
                            const float4* tp = TRI_PTR(S, r);
                            const float4  ta = tp[0], tb = tp[1];
                            const float   tc = *(const float*)(tp + 2);
                            hit = Triangle_HitAt(mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc), o, d, RT_EPSILON, tMax, t, u, v);
                        }
                        if (hit && t < bestT) bestT = t, best = r, bu = u, bv = v; // strict: first in the list wins ties (:485)
                    }
                    if (best != HIT_MISS) { // a leaf that reports a hit: the smallest key is the walk's answer; what this lane still holds comes later
                        const unsigned long long full = key | level;
                        atomicMin(&s_best, full);
                        LDS_ORDER();
                        if (s_best == full) s_res = make_uint4(best, __float_as_uint(bestT), __float_as_uint(bu), __float_as_uint(bv));
                        have = false, sp = 0;
                    } else {
                        pop = true;
                    }
                } else { // CheckHitInternalNode, kdtree.c:501-570, as in k_trace_flat
                    const float    split = __uint_as_float(qa.x);
                    const uint32_t right = qc.x & 0x1FFFFFFFu;
                    const bool     ax0 = type == 0u, ax1 = type == 1u;
                    const float    dA   = ax0 ? d.x : (ax1 ? d.y : d.z);
                    const float    oA   = ax0 ? o.x : (ax1 ? o.y : o.z);
                    const float    invA = ax0 ? inv.x : (ax1 ? inv.y : inv.z);
                    const float    oddA = ax0 ? odd.x : (ax1 ? odd.y : odd.z);
                    const bool     par    = fabsf(dA) < RT_EPSILON;
                    const float    tI     = split * invA - oddA;
                    const float    tMaxN  = minf_(tMax, tI);
                    const bool     behind = tI < RT_EPSILON;
                    const bool     both   = !par && !behind;
                    const bool     ge      = oA >= split;
                    const bool     tieSide = (oA + 100.0f * dA) >= split;
                    const bool     goRight = (!par && behind && oA == split) ? tieSide : ge;
                    const bool     farPending = both && tMaxN == tI;
                    const float    tNear      = both ? tMaxN : tMax;
                    const uint32_t nearNode = goRight ? right : idx, farNode = goRight ? idx : right;
                    // the children's boxes against the segments they would be walked with
                    const bool missL = cullOK && box_missed(S, qa.z, qa.w, qb.x, o, inv, goRight ? tMax : tNear);
                    const bool missR = cullOK && box_missed(S, qb.y, qb.z, qb.w, o, inv, goRight ? tNear : tMax);
                    const bool missNear = goRight ? missR : missL, missFar = goRight ? missL : missR;
                    const bool pushFar  = farPending && !missFar;
                    const unsigned long long keyFar = key | (1ull << (63u - (level < 63u ? level : 63u)));
                    if (!missNear) {
                        if (pushFar) {
                            const unsigned long long kf = keyFar | (level + 1u);
                            far      = make_uint4(farNode, __float_as_uint(tMax), (uint32_t)kf, (uint32_t)(kf >> 32));
                            produced = true;
                        }
                        node = nearNode, tMax = tNear, level++;
                    } else if (pushFar) {
                        node = farNode, key = keyFar, level++;
                    } else {
                        pop = true;
                    }
                }
            }
            if (have && pop) { // this lane's next node: its own youngest pending subtree
                if (sp > 0) {
                    const uint4 e = stk[--sp];
                    node = e.x, tMax = __uint_as_float(e.y);
                    const unsigned long long kf = (unsigned long long)e.w << 32 | e.z;
                    key = kf & ~63ull, level = (uint32_t)(kf & 63ull);
                } else {
                    have = false;
                }
            }
            // Pending subtrees change hands: a lane with nothing to do takes the youngest pending subtree — the one just
            // produced, else the top of the stack — of a lane that has one (lanes matched by rank through an LDS mailbox).
            const uint64_t hungry = __builtin_amdgcn_ballot_w64(!have);
            const uint64_t donors = solo ? 0ull : __builtin_amdgcn_ballot_w64(have && (produced || sp > 0));
            const uint32_t nH = (uint32_t)__popcll(hungry), nD = (uint32_t)__popcll(donors);
            const uint32_t m  = nH < nD ? nH : nD;
            const uint32_t dRank = (uint32_t)__popcll(donors & lt_mask), hRank = (uint32_t)__popcll(hungry & lt_mask);
            if (((donors >> lane) & 1ull) && dRank < m) {
                if (produced) s_mail[dRank] = far, produced = false;
                else s_mail[dRank] = stk[--sp];
            }
            if (produced) stk[sp++] = far;
            LDS_ORDER();
            if (((hungry >> lane) & 1ull) && hRank < m) {
                const uint4 e = s_mail[hRank];
                node = e.x, tMax = __uint_as_float(e.y);
                const unsigned long long kf = (unsigned long long)e.w << 32 | e.z;
                key = kf & ~63ull, level = (uint32_t)(kf & 63ull);
                have = true, sp = 0;
            }
            LDS_ORDER();
            if (__builtin_amdgcn_ballot_w64(have) == 0ull) break;
        }
        LDS_ORDER();
        if (lane == 0) { // FIN: Scene_ClosestHit's merge with the unbounded list (scene.c:73-98, 122-142), as k_trace_flat's write_hit
            uint32_t ref = HIT_MISS;
            float    tH = RTX_INF, uH = 0.0f, vH = 0.0f;
            if (s_best != ~0ull) {
                const uint4 w = s_res;
                ref = w.x, tH = __uint_as_float(w.y), uH = __uint_as_float(w.z), vH = __uint_as_float(w.w);
            }
            float    pBest = RTX_INF;
            uint32_t pRef  = HIT_MISS;
            for (uint32_t i = 0; i < S.n_planes; i++) {
                const float* pl = S.plane + 8ull * i;
                float        tp2;
                if (Plane_HitAt(mk3(pl[0], pl[1], pl[2]), mk3(pl[3], pl[4], pl[5]), o, d, RT_EPSILON, RTX_INF, tp2) && tp2 < pBest) {
                    pBest = tp2;
                    pRef  = HIT_PLANE | i;
                }
            }
            if (pRef != HIT_MISS && (ref == HIT_MISS || pBest < tH)) ref = pRef, tH = pBest, uH = 0.0f, vH = 0.0f;
            hits[0 * hstride + ridx] = ref;
            hits[1 * hstride + ridx] = __float_as_uint(tH);
            hits[2 * hstride + ridx] = __float_as_uint(uH);
            hits[3 * hstride + ridx] = __float_as_uint(vH);
        }
        LDS_ORDER();
    }
}

// Material_Bounce for every ray of the queue + compaction of the survivors into `qout`
// (Large blocks: the two counters are bumped once per block.  With 256-thread blocks and a
// per-wave atomic the 2.6 M same-address atomics of a 1080p/64 spp pass, ~12 ns each, WERE the kernel's
// run time — rocprof: 6 % VALU busy, 84 % of wave time waiting.)
#ifndef SHADE_BLOCK
#define SHADE_BLOCK 512 // (round 2, same box: 1024 threads 12.47 ms of shade per frame, 512: 12.28, 384: 13.8, 256: 14.9)
#endif
#ifndef SHADE_WAVES
#define SHADE_WAVES 8 // waves per SIMD the shade pass is compiled for (64 registers)
#endif
DEV uint32_t ray_bin(const SceneDev& S, f3 o, f3 d);
DEV uint32_t wave_group_add(uint32_t* counters, uint32_t key, bool active, uint32_t& base);
// (waves_per_eu(8): at most 64 VGPRs, 32 waves per CU; 6 waves per SIMD with 80 registers: +0.5 ms, 4 with 128: +4.4 ms)
__global__ __launch_bounds__(SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu(SHADE_WAVES))) void k_shade(SceneDev S, Queue qin, Hits hits, uint64_t n, Queue qout,
                                                       unsigned long long* out_count, unsigned long long* shaded_count,
                                                       float* __restrict__ result, uint64_t rcap, int last,
                                                       uint32_t* __restrict__ bin_cnt, uint2* __restrict__ bin_keys, FrameGeom g,
                                                       uint64_t first_pixel, int first, unsigned long long* early_count)
{
    __shared__ uint32_t s_wave_cnt[SHADE_BLOCK / 64], s_wave_shaded[SHADE_BLOCK / 64], s_wave_early[SHADE_BLOCK / 64];
    __shared__ uint64_t s_base;
    uint64_t  i     = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool      alive = false;
    f3        o2 = mk3(0, 0, 0), d2 = mk3(0, 0, 1), T = mk3(0, 0, 0);
    Rng       rng = {0, 0, 0, 0};
    uint32_t  pid = 0;
    bool      shaded = false, early = false;
    if (i < n) {
        f3         o, d;
        const bool aos = qin.es == Q_RECORD; // binned queue: a 64-byte record per path, read as four 16-byte quarters
        uint4      q2 = make_uint4(0, 0, 0, 0), q3 = q2;
        if (aos) {
            const uint4* rec = (const uint4*)((const uint32_t*)qin.base + i * Q_RECORD);
            const uint4  q0 = rec[0], q1 = rec[1];
            q2 = rec[2], q3 = rec[3];
            o   = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
            d   = mk3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
            T   = mk3(__uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x));
            pid = q3.y;
        } else {
            o = mk3(qin.f(Q_OX, i), qin.f(Q_OY, i), qin.f(Q_OZ, i));
            d = mk3(qin.f(Q_DX, i), qin.f(Q_DY, i), qin.f(Q_DZ, i));
            // first: the queue comes from a lean k_raygen
            T    = first ? mk3(1.0f, 1.0f, 1.0f) : mk3(qin.f(Q_TR, i), qin.f(Q_TG, i), qin.f(Q_TB, i));
            pid  = first ? (uint32_t)i : qin.u(Q_PID, i);
        }
        uint32_t ref = hits.base[0 * hits.cap + i];
        if (ref == HIT_MISS) { // Scene_Get_SkyColor(ray->dir), renderer.c:75
            f3 sky = Skybox_ColorAt(S, d);
            f3 L   = vmulc(T, sky);
            result[0 * rcap + pid] = L.x, result[1 * rcap + pid] = L.y, result[2 * rcap + pid] = L.z;
        } else {
            float t = __uint_as_float(hits.base[1 * hits.cap + i]);
            float u = __uint_as_float(hits.base[2 * hits.cap + i]);
            float v = __uint_as_float(hits.base[3 * hits.cap + i]);
            rng     = aos ? Rng{q2.y, q2.z, q2.w, q3.x} : first ? camera_rng(g, first_pixel, i) : Rng{qin.u(Q_S0, i), qin.u(Q_S1, i), qin.u(Q_S2, i), qin.u(Q_S3, i)};
            HitD h  = make_hitinfo(S, o, d, ref, t, u, v);
            f3   surf, emit, dout;
            bool cont = Material_Bounce(S, S.mats[h.mat], d, h, rng, surf, emit, dout);
            shaded    = true;
            if (cont) {
                // emitted is black for every continuing material, so E + S*L(next) == S*L(next)
                T     = vmulc(T, surf);
                o2    = h.pos;
                d2    = vnorm(dout); // Ray_Make
                alive = !last;
                if (alive && early_count) {
                    // The next ray's closest-hit query, answered here when it is trivial: a ray that meets the culling
                    // preconditions and misses the box of everything in the tree (the test the closest-hit kernel makes
                    // when it loads a ray) and no unbounded object either has no hit — RayColor adds the sky seen along it
                    // (renderer.c:75) and the path ends, without a queue record, a lane of the closest-hit pass and a
                    // miss in the next shade pass.  Most rays scattered by the ground leave this way.  Counted as a ray of
                    // the next depth all the same: it is one Scene_ClosestHit answered, only earlier.
                    f3 inv2 = vdivc(mk3(1.0f, 1.0f, 1.0f), d2);
                    const float big = 3.0e38f;
                    inv2.x = fabsf(inv2.x) > big ? copysignf(big, inv2.x) : inv2.x;
                    inv2.y = fabsf(inv2.y) > big ? copysignf(big, inv2.y) : inv2.y;
                    inv2.z = fabsf(inv2.z) > big ? copysignf(big, inv2.z) : inv2.z;
                    const float rx = o2.x - S.cull_c[0], ry = o2.y - S.cull_c[1], rz = o2.z - S.cull_c[2];
                    const float r2 = rx * rx + ry * ry + rz * rz, dd = d2.x * d2.x + d2.y * d2.y + d2.z * d2.z;
                    bool none = r2 <= S.cull_ro2 && fabsf(dd - 1.0f) <= 1.0e-3f;
                    if (none && !box_missed(S, S.root_q[0], S.root_q[1], S.root_q[2], o2, inv2, RTX_INF)) {
                        // inside the root's box, but perhaps outside both of its children's (the first split of these
                        // scenes has the model on one side and the light on the other): the closest-hit kernel's first
                        // step would then find nothing to descend into and nothing to push, whatever tMax it tests with
                        const uint32_t* rec = S.nodes40 + (uint64_t)NODE_WORDS * S.root; // (a cut-off plane carried by the root only ever removes rays)
                        const uint4     qa = *(const uint4*)rec, qb = *(const uint4*)(rec + 4);
                        none = (qa.y >> 30) != 3u && box_missed(S, qa.z, qa.w, qb.x, o2, inv2, RTX_INF) && box_missed(S, qb.y, qb.z, qb.w, o2, inv2, RTX_INF);
                    }
                    if (none) {
                        bool plane = false;
                        for (uint32_t k = 0; k < S.n_planes; k++) {
                            const float* pl = S.plane + 8ull * k;
                            float        tp;
                            plane = plane || Plane_HitAt(mk3(pl[0], pl[1], pl[2]), mk3(pl[3], pl[4], pl[5]), o2, d2, RT_EPSILON, RTX_INF, tp);
                        }
                        if (!plane) {
                            const f3 L = vmulc(T, Skybox_ColorAt(S, d2));
                            result[0 * rcap + pid] = L.x, result[1 * rcap + pid] = L.y, result[2 * rcap + pid] = L.z;
                            alive = false;
                            early = true;
                        }
                    }
                }
                if (last) {
                    // depth exhausted: RayColor(depth 0) is black (renderer.c:56-58) and the path returns
                    // surface * black — zero, or NaN where a lobe produced an inf/NaN surface colour; the
                    // reference's pixels carry those NaNs, so they are reproduced rather than dropped
                    f3 L = vmulc(T, mk3(0.0f, 0.0f, 0.0f));
                    result[0 * rcap + pid] = L.x, result[1 * rcap + pid] = L.y, result[2 * rcap + pid] = L.z;
                }
            } else {
                f3 L = vmulc(T, emit);
                result[0 * rcap + pid] = L.x, result[1 * rcap + pid] = L.y, result[2 * rcap + pid] = L.z;
            }
        }
    }
    // ---- compaction: ballot within the wave, one atomic per block ----
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(alive);
    const uint32_t wcnt = __popcll(mask);
    const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    const unsigned long long smask = __ballot(shaded), emask = __ballot(early);
    if (lane == 0) s_wave_cnt[wave] = wcnt, s_wave_shaded[wave] = (uint32_t)__popcll(smask), s_wave_early[wave] = (uint32_t)__popcll(emask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0, sh = 0, ea = 0;
        for (uint32_t w = 0; w < SHADE_BLOCK / 64; w++) total += s_wave_cnt[w], sh += s_wave_shaded[w], ea += s_wave_early[w];
        if (ea) atomicAdd(early_count + RTX_MAX_DEPTH * (blockIdx.x & 7u), (unsigned long long)ea); // 8 counters per depth
        s_base = total ? atomicAdd(out_count, (unsigned long long)total) : 0ull;
        if (sh) atomicAdd(shaded_count + 8 * (blockIdx.x & 7u), (unsigned long long)sh); // statistics only: 8 counters, 64 bytes apart
    }
    __syncthreads();
    if (alive) {
        uint64_t pos = s_base + rank;
        for (uint32_t w = 0; w < wave; w++) pos += s_wave_cnt[w];
        qout.f(Q_OX, pos) = o2.x, qout.f(Q_OY, pos) = o2.y, qout.f(Q_OZ, pos) = o2.z;
        qout.f(Q_DX, pos) = d2.x, qout.f(Q_DY, pos) = d2.y, qout.f(Q_DZ, pos) = d2.z;
        qout.f(Q_TR, pos) = T.x, qout.f(Q_TG, pos) = T.y, qout.f(Q_TB, pos) = T.z;
        qout.u(Q_S0, pos) = rng.s0, qout.u(Q_S1, pos) = rng.s1, qout.u(Q_S2, pos) = rng.s2, qout.u(Q_S3, pos) = rng.s3;
        qout.u(Q_PID, pos) = pid;
    }
    if (bin_cnt) { // first half of the binning pass (k_bin_scan / k_bin_scatter follow): key and histogram
        // The histogram's atomic already hands every ray a slot of its own inside its bin (old count + rank in the wave's
        // group): kept beside the key, so that the scatter needs no second round of atomics.
        const uint32_t key = alive ? ray_bin(S, o2, d2) : 0u;
        uint32_t       base;
        const uint32_t grank = wave_group_add(bin_cnt, key, alive, base); // one atomic per wave and key
        if (alive) {
            uint64_t pos = s_base + rank;
            for (uint32_t w = 0; w < wave; w++) pos += s_wave_cnt[w];
            bin_keys[pos] = make_uint2(key, base + grank);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ray binning between bounce passes.  The closest-hit passes of depth >= 2 are bound by the rate of L2
// misses (rocprof: ~51 G misses/s against a measured ~58 G/s ceiling for 64-byte gathers that miss L2,
// tools/dbg/gather.hip): the rays of a wave start all over the model.  A counting sort of the queue by
// (origin cell on a 32^3 grid in Morton order, direction octant) puts rays that will walk the same part of
// the tree next to each other, i.e. on the same XCD at the same time (see part_begin).  Order inside a bin
// is arbitrary; results do not depend on queue order (per-path RNG state and result slot).
// ---------------------------------------------------------------------------------------------

DEV uint32_t spread3(uint32_t v) // up to 6 bits -> every third bit
{
    return (v & 1u) | (v & 2u) << 2 | (v & 4u) << 4 | (v & 8u) << 6 | (v & 16u) << 8 | (v & 32u) << 10;
}
DEV uint32_t ray_bin(const SceneDev& S, f3 o, f3 d)
{
    uint32_t c[3];
    const float oo[3] = {o.x, o.y, o.z};
    for (int a = 0; a < 3; a++) {
        float f = (oo[a] - S.cull_lo[a]) * S.bin_inv[a]; // NaN / out of range: clamped below
        f       = fminf(fmaxf(f, 0.0f), (float)((1 << BIN_CELL_BITS) - 1));
        c[a]    = (uint32_t)f;
    }
    const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
#if BIN_DIR_EXTRA == 2
    const float    ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const uint32_t dom = (ax >= ay && ax >= az) ? 0u : (ay >= az ? 1u : 2u);
    return ((spread3(c[0]) | spread3(c[1]) << 1 | spread3(c[2]) << 2) << 3 | oct) << 2 | dom;
#else
    return (spread3(c[0]) | spread3(c[1]) << 1 | spread3(c[2]) << 2) << 3 | oct;
#endif
}

// One atomic per wave and distinct key: neighbours in the queue mostly share a bin, and same-address atomics
// from every lane would serialise (the first version of this pass cost more than the traversal it helped).
// Returns the lane's rank among the wave's lanes with the same key and, in `base`, what the leader's atomic
// returned (valid in every lane of the group).
DEV uint32_t wave_group_add(uint32_t* counters, uint32_t key, bool active, uint32_t& base)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t       rank = 0, my_leader = lane, gsize = 0;
    // group the lanes by key with ballots only; the atomics of all group leaders then go out in ONE instruction
    // (a loop with one atomic round trip per distinct key made the pass latency-bound)
    unsigned long long todo = __ballot(active);
    while (todo) {
        const uint32_t           leader = (uint32_t)__ffsll((long long)todo) - 1u;
        const uint32_t           k      = __shfl(key, (int)leader);
        const unsigned long long grp    = __ballot(active && key == k) & todo;
        if (active && key == k) {
            my_leader = leader;
            rank      = (uint32_t)__popcll(grp & ((1ull << lane) - 1ull));
            gsize     = (uint32_t)__popcll(grp);
        }
        todo &= ~grp;
    }
    uint32_t b = 0;
    if (active && lane == my_leader) b = atomicAdd(&counters[key], gsize);
    base = __shfl(b, (int)my_leader);
    return rank;
}

// Exclusive prefix sums of the 2^BIN_BITS bin counts, in two short launches (one 1024-thread block walking all of them
// took 100 us, on the critical path of every binned pass): each block scans its own 1024 counts in place and
// leaves their total in chunk[block]; one more block turns the chunk totals into chunk bases, which the scatter adds.
#define BIN_SCAN_SHIFT (BIN_BITS > 20 ? BIN_BITS - 20 : 0) // counts per thread of k_bin_scan = 1 << BIN_SCAN_SHIFT: at most 1024 chunks
#define BIN_CHUNK_SHIFT (10 + BIN_SCAN_SHIFT)
#define BIN_CHUNKS ((1u << BIN_BITS) >> BIN_CHUNK_SHIFT)
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* __restrict__ counts, uint32_t* __restrict__ chunk)
{
    __shared__ uint32_t s_w[16];
    constexpr uint32_t  PER = 1u << BIN_SCAN_SHIFT;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i   = (blockIdx.x * 1024u + tid) * PER;
    uint32_t       v[PER], c = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) v[k] = counts[i + k], c += v[k];
    const uint32_t inc = wave_scan_add(c);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < 16; w++) {
        if (w < wave) before += s_w[w];
        total += s_w[w];
    }
    uint32_t run = before + inc - c;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) counts[i + k] = run, run += v[k];
    if (tid == 0) chunk[blockIdx.x] = total;
}
__global__ __launch_bounds__(BIN_CHUNKS) void k_bin_scan_chunks(uint32_t* __restrict__ chunk)
{
    __shared__ uint32_t s_w[BIN_CHUNKS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t c   = chunk[tid];
    const uint32_t inc = wave_scan_add(c);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += s_w[w];
    chunk[tid] = before + inc - c;
}

__global__ __launch_bounds__(256) void k_bin_scatter(Queue qin, uint64_t n, const unsigned long long* n_dev, const uint32_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ chunk, const uint2* __restrict__ keys, Queue qout)
{
    if (n_dev) n = *n_dev; // launched before the host knew the queue's length: `n` was an upper bound for the grid
    // Each lane gathers its own path's 14 dwords (coalesced: the source is SoA); the records then go through LDS so
    // that FOUR lanes store the four 16-byte quarters of ONE record in the same instruction — a store instruction
    // covers 16 whole 64-byte lines instead of a quarter of 64 different ones.
    __shared__ uint4    s_rec[256 * 4 + 256 / 16]; // [thread][quarter], one uint4 of padding per 16 threads
    __shared__ uint64_t s_pos[256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wbase = tid & ~63u;
    // (tiles of 256 paths, block-uniform loop: a grid sized for an upper bound of n costs nothing when n turns out small)
    for (uint64_t tile = blockIdx.x; tile * 256 < n; tile += gridDim.x) {
    const uint64_t i   = tile * 256 + tid;
    uint64_t       pos = ~0ull;
    if (i < n) {
        const uint2 ks = keys[i]; // bin, slot inside the bin
        pos            = (uint64_t)offsets[ks.x] + chunk[ks.x >> BIN_CHUNK_SHIFT] + ks.y;
        uint32_t w[Q_RECORD];
#pragma unroll
        for (int c = 0; c < Q_COMPONENTS; c++) w[c] = qin.u(c, i);
        w[14] = w[15] = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) s_rec[tid * 4 + (tid >> 4) + k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    }
    s_pos[tid] = pos;
    __syncthreads();
    // the destination is an AoS queue: one 64-byte record = one line per path
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t r = wbase + 16u * k + (lane >> 2), q = lane & 3u; // record (thread that gathered it), quarter
        const uint64_t pr = s_pos[r];
        if (pr != ~0ull) ((uint4*)((uint32_t*)qout.base + pr * Q_RECORD))[q] = s_rec[r * 4 + (r >> 4) + q];
    }
    __syncthreads();
    }
}

// RenderPixel's accumulation (renderer.c:98-107): sequential f32 sum over the samples of a pixel, in sample
// order.  A pixel's samples are contiguous in the result slab, so a thread that walked its own pixel would
// read with a 4*spp-byte stride between lanes; instead the block stages 256 pixels x 16 samples at a time
// through LDS (64-byte runs per pixel on the global side, stride-17 rows on the LDS side: conflict-free).
#define RESOLVE_CH 16
__global__ __launch_bounds__(256) void k_resolve_slab(const float* __restrict__ result, uint64_t rcap, FrameGeom g,
                                                       uint64_t first_pixel, uint64_t n_pixels, float* accum)
{
    __shared__ float s_tile[256 * (RESOLVE_CH + 1)];
    const uint64_t p0 = (uint64_t)blockIdx.x * 256;
    const uint64_t p  = p0 + threadIdx.x;
    const uint32_t ns = g.n_samples;
    // the sum continues from what the frame buffer holds (zero before the frame's first sample range): a frame
    // rendered in several sample ranges, in order, is the same sequential sum as a frame rendered at once
    uint64_t o = 0;
    float    cum[3] = {0.0f, 0.0f, 0.0f};
    if (p < n_pixels) {
        uint32_t xx, row;
        split_pixel(g, first_pixel + slab_pixel(g, p), xx, row);
        const uint32_t yy = assigned_row(g, row);
        o                 = ((uint64_t)yy * g.width + xx) * 3;
        cum[0] = accum[o + 0], cum[1] = accum[o + 1], cum[2] = accum[o + 2];
    }
    for (int c = 0; c < 3; c++) {
        for (uint32_t k = 0; k < ns; k += RESOLVE_CH) {
            const uint32_t len = min((uint32_t)RESOLVE_CH, ns - k);
            __syncthreads();
            for (uint32_t e = threadIdx.x; e < 256u * RESOLVE_CH; e += 256u) {
                const uint32_t j = e / RESOLVE_CH, sIdx = e % RESOLVE_CH;
                if (sIdx < len && p0 + j < n_pixels) s_tile[j * (RESOLVE_CH + 1) + sIdx] = result[c * rcap + (p0 + j) * ns + k + sIdx];
            }
            __syncthreads();
            if (p < n_pixels)
                for (uint32_t sIdx = 0; sIdx < len; sIdx++) cum[c] = cum[c] + s_tile[threadIdx.x * (RESOLVE_CH + 1) + sIdx];
        }
    }
    if (p >= n_pixels) return;
    accum[o + 0] = cum[0];
    accum[o + 1] = cum[1];
    accum[o + 2] = cum[2];
}

DEV uint8_t LinearTosRGB(float channel) // src/gfx/color.c:18-29
{
    float linear = clampf_(channel, 0.0f, 1.0f);
    if (linear < 0.0031308f) {
        float srgb = linear * 12.92f;
        return (uint8_t)(255.0f * srgb);
    } else {
        float srgb = 1.055f * powf(linear, 1.0f / 2.4f) - 0.055f;
        return (uint8_t)(255.0f * srgb);
    }
}

__global__ __launch_bounds__(256) void k_tonemap(const float* __restrict__ accum, uint64_t n_pixels, float inv_spp, uint8_t* __restrict__ bgr)
{
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    // avg = vdiv(cum, spp) = cum * (1.0f/spp) (renderer.c:109, vec.c:198); RGB{b,g,r} (color.h:19-23)
    bgr[3 * p + 0] = LinearTosRGB(inv_spp * accum[3 * p + 2]);
    bgr[3 * p + 1] = LinearTosRGB(inv_spp * accum[3 * p + 1]);
    bgr[3 * p + 2] = LinearTosRGB(inv_spp * accum[3 * p + 0]);
}

// ---- parity-seam kernels (small batches) ----
// rtx_trace_closest = k_prep_rays → the SAME closest-hit kernel the renderer launches → k_expand_hits
__global__ void k_prep_rays(const float* __restrict__ o_xyz, const float* __restrict__ d_xyz, uint64_t n, float* __restrict__ rays, int unit_dir)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 d = mk3(d_xyz[3 * i], d_xyz[3 * i + 1], d_xyz[3 * i + 2]);
    if (!unit_dir) d = vnorm(d); // Ray_Make
    rays[0 * n + i] = o_xyz[3 * i], rays[1 * n + i] = o_xyz[3 * i + 1], rays[2 * n + i] = o_xyz[3 * i + 2];
    rays[3 * n + i] = d.x, rays[4 * n + i] = d.y, rays[5 * n + i] = d.z;
}

__global__ void k_expand_hits(SceneDev S, const float* __restrict__ rays, const uint32_t* __restrict__ hits, uint64_t n, uint32_t* prim_id,
                              float* t_out, float* pos, float* nrm, float* uv, uint8_t* front)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3       o   = mk3(rays[0 * n + i], rays[1 * n + i], rays[2 * n + i]);
    const f3       d   = mk3(rays[3 * n + i], rays[4 * n + i], rays[5 * n + i]);
    const uint32_t ref = hits[0 * n + i];
    float          t = __uint_as_float(hits[1 * n + i]), u = __uint_as_float(hits[2 * n + i]), v = __uint_as_float(hits[3 * n + i]);
    if (prim_id) prim_id[i] = hit_prim_id(S, ref);
    HitD h;
    if (ref != HIT_MISS) {
        h = make_hitinfo(S, o, d, ref, t, u, v);
    } else {
        h.pos = mk3(0, 0, 0), h.n = mk3(0, 0, 0), h.u = 0, h.v = 0, h.front = false, h.mat = 0;
        t = 0.0f;
    }
    if (t_out) t_out[i] = t;
    if (pos) pos[3 * i] = h.pos.x, pos[3 * i + 1] = h.pos.y, pos[3 * i + 2] = h.pos.z;
    if (nrm) nrm[3 * i] = h.n.x, nrm[3 * i + 1] = h.n.y, nrm[3 * i + 2] = h.n.z;
    if (uv) uv[2 * i] = h.u, uv[2 * i + 1] = h.v;
    if (front) front[i] = h.front ? 1 : 0;
}

__global__ void k_camera_rays(rtx_camera cam, FrameGeom g, uint64_t n, const uint32_t* pixel_index, const uint32_t* sample, float* o_xyz,
                              float* d_xyz, uint32_t* state)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t xx = pixel_index[i] % g.width, yy = pixel_index[i] / g.width;
    Rng      rng;
    f3       o, d;
    camera_sample(cam, g, xx, yy, sample[i], rng, o, d);
    o_xyz[3 * i] = o.x, o_xyz[3 * i + 1] = o.y, o_xyz[3 * i + 2] = o.z;
    d_xyz[3 * i] = d.x, d_xyz[3 * i + 1] = d.y, d_xyz[3 * i + 2] = d.z;
    if (state) state[4 * i] = rng.s0, state[4 * i + 1] = rng.s1, state[4 * i + 2] = rng.s2, state[4 * i + 3] = rng.s3;
}

__global__ void k_rng(uint64_t lo, uint64_t hi, uint64_t n, float* out)
{
    if (blockIdx.x || threadIdx.x) return;
    Rng rng;
    rng_seed(rng, lo, hi);
    for (uint64_t i = 0; i < n; i++) out[i] = Random_Unilateral(rng);
}

__global__ void k_bounce(SceneDev S, uint32_t mat, uint64_t n, const float* dir, const float* pos, const float* nrm, const float* uv,
                         const uint8_t* front, uint32_t* state, float* surf, float* emit, float* oo, float* od, uint8_t* cont)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    HitD h;
    h.pos   = mk3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
    h.n     = mk3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    h.u     = uv[2 * i];
    h.v     = uv[2 * i + 1];
    h.front = front[i] != 0;
    h.mat   = mat;
    Rng rng = {state[4 * i], state[4 * i + 1], state[4 * i + 2], state[4 * i + 3]};
    f3  s, e, dout;
    bool c = Material_Bounce(S, S.mats[mat], mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), h, rng, s, e, dout);
    f3  dn = vnorm(dout); // Ray_Make(hit->position, dir)
    surf[3 * i] = s.x, surf[3 * i + 1] = s.y, surf[3 * i + 2] = s.z;
    emit[3 * i] = e.x, emit[3 * i + 1] = e.y, emit[3 * i + 2] = e.z;
    oo[3 * i] = h.pos.x, oo[3 * i + 1] = h.pos.y, oo[3 * i + 2] = h.pos.z;
    od[3 * i] = dn.x, od[3 * i + 1] = dn.y, od[3 * i + 2] = dn.z;
    cont[i]   = c ? 1 : 0;
    state[4 * i] = rng.s0, state[4 * i + 1] = rng.s1, state[4 * i + 2] = rng.s2, state[4 * i + 3] = rng.s3;
}

__global__ void k_sky(SceneDev S, uint64_t n, const float* dir, float* rgb)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 c = Skybox_ColorAt(S, mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
    rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
}

// ---------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------
static inline unsigned blocks_for(uint64_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

struct DevBuf { // RAII device buffer for the host-pointer entry points
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T> T* as() { return (T*)p; }
};

static bool trace_variant_forced() // RTX_TRACE_VARIANT given: the scene's own choice between culled and literal is overridden
{
    static const bool f = getenv("RTX_TRACE_VARIANT") != nullptr;
    return f;
}
static int trace_variant()
{
    // 0: one ray per thread, 1: persistent (nested loops), 2: persistent flat state machine,
    // 3: 2 + subtree-box culling, 4: 3 + cooperative leaves (default), 5: 2 + cooperative leaves.
    // 0..2 and 5 walk the tree literally.  (Read once; initialisation of a local static is thread-safe.)
    static const int v = [] {
        const char* e = getenv("RTX_TRACE_VARIANT");
        const int   x = e ? atoi(e) : 4;
        return (x < 0 || x > 5) ? 4 : x;
    }();
    return v;
}

// true when launch_trace would run one of the kernels that have a TAIL variant (the cooperative ones)
static bool tail_capable(const rtx_scene* sc, bool literal)
{
    const int  tv      = trace_variant();
    const bool no_cull = literal || (!trace_variant_forced() && !sc->cull_useful);
    const int  variant = !no_cull ? tv : tv == 3 ? 2 : tv == 4 ? 5 : tv;
    return variant >= 4;
}

// persistent waves: a grid that is fully resident (blocks per CU from the occupancy query; blocks
// that would not fit simply start later and find the queue empty — nothing waits on another block)
typedef void (*PersistKernel)(SceneDev, const float*, uint64_t, uint64_t, uint32_t*, uint64_t, unsigned long long*, uint32_t, uint64_t*, uint32_t,
                              TailArgs);
struct TraceKernel {
    int           variant;
    PersistKernel kern; // null: variant 0, one ray per thread
    int           slot, bs;
};
// default: the culled walk where its boxes help, else the literal one; RTX_TRACE_VARIANT overrides
static TraceKernel pick_trace_kernel(const rtx_scene* sc, bool count, bool literal, bool tail)
{
    const int  tv      = trace_variant();
    const bool no_cull = literal || (!trace_variant_forced() && !sc->cull_useful);
    TraceKernel k;
    k.variant = !no_cull ? tv : tv == 3 ? 2 : tv == 4 ? 5 : tv;
    const int variant = k.variant;
    k.kern = variant == 0   ? nullptr
             : variant == 1 ? (count ? k_trace_persist<true> : k_trace_persist<false>)
             : variant == 2 ? (count ? k_trace_flat<true, false, false> : k_trace_flat<false, false, false>)
             : variant == 3 ? (count ? k_trace_flat<true, true, false> : k_trace_flat<false, true, false>)
             : variant == 4 ? (count ? k_trace_flat<true, true, true> : k_trace_flat<false, true, true>)
                            : (count ? k_trace_flat<true, false, true> : k_trace_flat<false, false, true>);
    if (tail && variant >= 4) k.kern = variant == 4 ? k_trace_flat<false, true, true, true> : k_trace_flat<false, false, true, true>;
    k.slot = tail ? 8 + (variant - 4) : (count ? 1 : 0) + 2 * (variant <= 2 ? 0 : variant - 2);
    k.bs   = variant == 1 ? 256 : TRACE_BLOCK;
    return k;
}
// resident grid of a persistent kernel on the scene's device: queried once per kernel and scene, then cached
static int resident_blocks(rtx_scene* sc, const TraceKernel& k, unsigned* out)
{
    if (sc->persist_blocks[k.slot] == 0) {
        int per_cu = 0, cus = 0;
        HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.kern, k.bs, 0));
        HIPCHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, sc->device));
        const int wpb = k.bs / 64; // RTX_TRACE_BLOCKS_PER_CU counts groups of 4 waves (one per SIMD), whatever the block size
        if (per_cu < 1) per_cu = 1;
        if (per_cu * wpb > 32) per_cu = 32 / wpb;
        if (const char* e = getenv("RTX_TRACE_BLOCKS_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e) * 4 / wpb)); // experiments
        sc->persist_blocks[k.slot] = (unsigned)(per_cu * cus);
    }
    *out = sc->persist_blocks[k.slot];
    return RTX_OK;
}

static int launch_trace(rtx_scene* sc, const float* d_rays, uint64_t stride, uint64_t n, uint32_t* d_hits, uint64_t hstride, hipStream_t st,
                        bool count, uint64_t* d_counters, bool literal, unsigned long long* work = nullptr, bool bounce_rays = false,
                        const TailArgs* tail = nullptr, const unsigned long long* n_dev = nullptr, uint64_t il_len = 0)
{
    // n_dev: the queue's length is still on the device; n is an upper bound (flat kernels only: render_device_impl checks)
    if (n == 0) return RTX_OK;
    if (n > 0xFFFFFF00ull) return fail(RTX_E_LIMIT, "too many rays for one launch");
    const TraceKernel tk      = pick_trace_kernel(sc, count, literal, tail != nullptr);
    const int         variant = tk.variant;
    if (variant == 0) {
        unsigned nb = blocks_for(n, 64);
        if (count) {
            hipLaunchKernelGGL(k_trace<true>, dim3(nb), dim3(64), 0, st, sc->dev, d_rays, stride, n, d_hits, hstride, d_counters);
        } else {
            hipLaunchKernelGGL(k_trace<false>, dim3(nb), dim3(64), 0, st, sc->dev, d_rays, stride, n, d_hits, hstride, d_counters);
        }
        HIPCHECK(hipGetLastError());
        return RTX_OK;
    }
    if (tail && (variant < 4 || count)) return fail(RTX_E_INVALID, "no tail variant of this closest-hit kernel");
    static const uint32_t tune = [] { // RTX_TRACE_TUNE = leaf_min,inner_min,refill_min (lanes); read once
        unsigned    a = 16, b = 8, c = REFILL_MIN;
        if (trace_variant() >= 4) a = 4; // cooperative leaves fill their 64 slots from few lanes
        const char* e = getenv("RTX_TRACE_TUNE");
        if (e) sscanf(e, "%u,%u,%u", &a, &b, &c);
        a = std::min(std::max(a, 1u), 64u), b = std::min(std::max(b, 1u), 64u), c = std::min(std::max(c, 1u), 64u);
        return a | b << 8 | c << 16;
    }();
    const int bs       = tk.bs;
    unsigned  resident = 0;
    int       rc       = resident_blocks(sc, tk, &resident);
    if (rc != RTX_OK) return rc;
    if (!work) work = (unsigned long long*)&sc->d_counters[16]; // 8 part counters, 64 B apart
    HIPCHECK(hipMemsetAsync(work, 0, 64 * sizeof(unsigned long long), st));
    // grab size: 160 rays on the big coherent queues (fewer atomics, neighbouring rays stay on one wave), one wave's worth
    // on the smaller ones, where balance at the end of the launch is worth more (sweep: tools/dbg, RTX_TRACE_CHUNK)
    uint64_t chunk = n >= (64ull << 20) ? 160 : 64;
    if (knob(K_TRACE_CHUNK) > 0) chunk = (uint64_t)std::max(16ll, knob(K_TRACE_CHUNK)); // experiments
    unsigned nb    = (unsigned)std::min<uint64_t>(resident, (n + bs - 1) / bs);
    TailArgs ta;
    memset(&ta, 0, sizeof ta);
    if (tail) ta = *tail;
    if (n_dev && variant < 2) return fail(RTX_E_INVALID, "this closest-hit kernel takes the queue's length from the host");
    ta.n_dev  = n_dev;
    ta.il_len = variant >= 2 && il_len < (1ull << 31) ? (uint32_t)il_len : 0u;
    // Long walks (k_trace_long): only the culled cooperative kernel hands walks over, never under COUNT_VISITS (the visit
    // counts are the plain walk's), and only where `work` is a lane's counter block of this scene (the allocation-free
    // entry point brings its own and has no list).
    uint32_t* long_list = nullptr;
    {
        const ptrdiff_t off = (const uint64_t*)work - sc->d_counters;
        if (variant == 4 && !tail && !count && off >= 16 && (off - 16) % 80 == 0 && (off - 16) / 80 < RTX_MAX_LANES && (uint64_t)nb * bs <= LONG_CAP)
            long_list = sc->d_long + (size_t)((off - 16) / 80) * LONG_CAP;
    }
    static const unsigned long_blocks = getenv("RTX_LONG_BLOCKS") ? (unsigned)atoi(getenv("RTX_LONG_BLOCKS")) : 2048u; // experiments
    if (long_list && knob(K_LONG_ALL) != 0) { // the tests' mode: every ray of the queue through k_trace_long
        hipLaunchKernelGGL(k_trace_long, dim3(long_blocks), dim3(64), 0, st, sc->dev, d_rays, stride, n, n_dev, d_hits, hstride, (const uint32_t*)nullptr, work,
                           sc->tree_depth);
        HIPCHECK(hipGetLastError());
        return RTX_OK;
    }
    const long long long_after = knob(K_LONG_AFTER);
    PersistKernel kern = tk.kern;
    if (long_list && long_after >= 0) {
        ta.long_list = long_list, ta.long_after = (uint32_t)long_after;
        kern = k_trace_flat<false, true, true, false, true>; // (variant 4, no COUNT, no TAIL: checked above)
    }
    static const bool debug = getenv("RTX_LONG_DEBUG") != nullptr; // experiments: rays handed over, the two kernels' times, per launch
    hipEvent_t        em = nullptr;
    if (debug) {
        HIPCHECK(hipEventCreate(&em));
        HIPCHECK(hipEventRecord(em, st));
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(bs), 0, st, sc->dev, d_rays, stride, n, d_hits, hstride, work, (uint32_t)chunk, d_counters,
                       tune | (bounce_rays && !count ? 1u << 25 : 0u) | (n_dev && knob(K_TRACE_CHUNK) <= 0 ? 1u << 26 : 0u), ta);
    HIPCHECK(hipGetLastError());
    if (debug && !ta.long_list) {
        hipEvent_t e0 = nullptr;
        HIPCHECK(hipEventCreate(&e0));
        HIPCHECK(hipEventRecord(e0, st));
        HIPCHECK(hipEventSynchronize(e0));
        float ms_main = 0;
        HIPCHECK(hipEventElapsedTime(&ms_main, em, e0));
        fprintf(stderr, "[long] queue %llu rays: main kernel %.3f ms (no hand-over)\n", (unsigned long long)n, ms_main);
    }
    if (ta.long_list) {
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        if (debug) {
            HIPCHECK(hipEventCreate(&e0));
            HIPCHECK(hipEventCreate(&e1));
            HIPCHECK(hipEventRecord(e0, st));
        }
        hipLaunchKernelGGL(k_trace_long, dim3(long_blocks), dim3(64), 0, st, sc->dev, d_rays, stride, n, n_dev, d_hits, hstride, (const uint32_t*)long_list, work,
                           sc->tree_depth);
        HIPCHECK(hipGetLastError());
        if (debug) {
            HIPCHECK(hipEventRecord(e1, st));
            HIPCHECK(hipEventSynchronize(e1));
            unsigned long long c = 0;
            float              ms = 0;
            HIPCHECK(hipMemcpy(&c, work + LONG_COUNT, 8, hipMemcpyDeviceToHost));
            float ms_main = 0;
            HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
            HIPCHECK(hipEventElapsedTime(&ms_main, em, e0));
            fprintf(stderr, "[long] queue %llu rays: main kernel %.3f ms, %llu handed over, k_trace_long %.3f ms\n", (unsigned long long)n, ms_main, c, ms);
            (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
            (void)e2;
        }
    }
    return RTX_OK;
}

extern "C" int rtx_trace_closest_device(rtx_scene* sc, size_t n, const float* d_rays, uint32_t* d_hits, void* stream, uint32_t flags,
                                        uint64_t* d_visit_counters)
{
    if (!sc || (n && (!d_rays || !d_hits))) return fail(RTX_E_INVALID, "null argument");
    bool count = (flags & RTX_RENDER_COUNT_VISITS) != 0;
    if (count && !d_visit_counters) return fail(RTX_E_INVALID, "COUNT_VISITS needs a counter buffer (4 x u64)");
    HIPCHECK(hipSetDevice(sc->device));
    return launch_trace(sc, d_rays, n, n, d_hits, n, (hipStream_t)stream, count, d_visit_counters, (flags & RTX_TRACE_LITERAL) != 0);
}

extern "C" int rtx_trace_closest(rtx_scene* sc, size_t n, const float* origin_xyz, const float* dir_xyz, uint32_t* prim_id, float* t,
                                 float* position_xyz, float* normal_xyz, float* uv, uint8_t* front_face, rtx_stats* stats, uint32_t flags)
{
    if (!sc || (n && (!origin_xyz || !dir_xyz))) return fail(RTX_E_INVALID, "null argument");
    if (stats) memset(stats, 0, sizeof *stats);
    HIPCHECK(hipSetDevice(sc->device));
    bool         count = (flags & RTX_RENDER_COUNT_VISITS) != 0;
    const int    unit_dir = (flags & RTX_TRACE_UNIT_DIR) ? 1 : 0;
    const size_t CH    = 1u << 22;
    const bool   literal = (flags & RTX_TRACE_LITERAL) != 0;
    DevBuf       bo, bd, bid, bt, bp, bn, buv, bf, brays, bhits;
    size_t       m = std::min(n, CH);
    HIPCHECK(bo.alloc(m * 12));
    HIPCHECK(bd.alloc(m * 12));
    HIPCHECK(bid.alloc(m * 4));
    HIPCHECK(bt.alloc(m * 4));
    HIPCHECK(bp.alloc(m * 12));
    HIPCHECK(bn.alloc(m * 12));
    HIPCHECK(buv.alloc(m * 8));
    HIPCHECK(bf.alloc(m));
    HIPCHECK(brays.alloc(m * 24));
    HIPCHECK(bhits.alloc(m * 16));
    HIPCHECK(hipMemset(sc->d_counters, 0, 16 * sizeof(uint64_t)));
    struct EventPair { // destroyed on every return path
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } evp;
    HIPCHECK(hipEventCreate(&evp.a));
    HIPCHECK(hipEventCreate(&evp.b));
    const hipEvent_t e0 = evp.a, e1 = evp.b;
    double ms_total = 0.0;
    for (size_t off = 0; off < n; off += CH) {
        size_t k = std::min(CH, n - off);
        HIPCHECK(hipMemcpy(bo.p, origin_xyz + 3 * off, k * 12, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(bd.p, dir_xyz + 3 * off, k * 12, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_prep_rays, dim3(blocks_for(k, 256)), dim3(256), 0, 0, bo.as<float>(), bd.as<float>(), (uint64_t)k, brays.as<float>(), unit_dir);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipEventRecord(e0, 0));
        int rc = launch_trace(sc, brays.as<float>(), k, k, bhits.as<uint32_t>(), k, 0, count, sc->d_counters, literal);
        if (rc != RTX_OK) return rc;
        HIPCHECK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(k_expand_hits, dim3(blocks_for(k, 256)), dim3(256), 0, 0, sc->dev, brays.as<float>(), bhits.as<uint32_t>(), (uint64_t)k,
                           bid.as<uint32_t>(), bt.as<float>(), bp.as<float>(), bn.as<float>(), buv.as<float>(), bf.as<uint8_t>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms;
        if (prim_id) HIPCHECK(hipMemcpy(prim_id + off, bid.p, k * 4, hipMemcpyDeviceToHost));
        if (t) HIPCHECK(hipMemcpy(t + off, bt.p, k * 4, hipMemcpyDeviceToHost));
        if (position_xyz) HIPCHECK(hipMemcpy(position_xyz + 3 * off, bp.p, k * 12, hipMemcpyDeviceToHost));
        if (normal_xyz) HIPCHECK(hipMemcpy(normal_xyz + 3 * off, bn.p, k * 12, hipMemcpyDeviceToHost));
        if (uv) HIPCHECK(hipMemcpy(uv + 2 * off, buv.p, k * 8, hipMemcpyDeviceToHost));
        if (front_face) HIPCHECK(hipMemcpy(front_face + off, bf.p, k, hipMemcpyDeviceToHost));
    }
    if (stats) {
        stats->rays_total       = n;
        stats->rays_at_depth[0] = n;
        stats->ms_trace = stats->ms_total = ms_total;
        stats->trace_launches             = (uint32_t)((n + CH - 1) / CH);
        if (count) {
            uint64_t c[4];
            HIPCHECK(hipMemcpy(c, sc->d_counters, sizeof c, hipMemcpyDeviceToHost));
            stats->visits_inner = c[0], stats->visits_leaf = c[1], stats->visits_prim = c[2], stats->visits_culled = c[3];
        }
    }
    return RTX_OK;
}

// ---------------------------------------------------------------------------------------------
// rtx_closest_hit_noalloc — rtx_trace_closest for callers that cannot allocate (rtx.h).  What rtx_trace_closest keeps in
// hipMalloc'ed staging buffers — the SoA ray queue k_prep_rays writes and the closest-hit kernel reads, the kernel's
// hit records, its work-grab counters — lives in ONE static device array, `scratch`, that every GPU the library runs on
// holds from load to exit.  In host code the array's NAME is not a device address: rtx_scene_create asks
// hipGetSymbolAddress for this device's copy once, with the device current, and everything below uses that pointer.
// ---------------------------------------------------------------------------------------------
struct NoAllocScratch {
    float              rays[6 * RTX_NOALLOC_MAX_RAYS]; // ox oy oz dx dy dz, component stride = n of the call
    uint32_t           hits[4 * RTX_NOALLOC_MAX_RAYS]; // ref t u v, component stride = n of the call
    unsigned long long work[64];                       // the persistent kernel's 8 part counters, 64 bytes apart
};
// (no __attribute__((used)) here: clang then keeps the symbol LOCAL in the code object and the runtime cannot find it;
// the host-side use below is what makes it externalise the static)
static __device__ __align__(256) unsigned char scratch[sizeof(NoAllocScratch)];

static int noalloc_scratch_address(rtx_scene* sc) // rtx_scene_create, device current
{
    void* ptr = nullptr;
    HIPCHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(RTX_E_HIP, "hipGetSymbolAddress returned no address for the scratch array");
    sc->noalloc_scratch = (unsigned char*)ptr;
    // the resident grids of the two closest-hit kernels the entry point may launch are queried here as well, so that
    // the entry point itself makes nothing but stream operations
    for (int literal = 0; literal < 2; literal++) {
        const TraceKernel tk = pick_trace_kernel(sc, false, literal != 0, false);
        unsigned          nb;
        if (tk.kern) {
            int rc = resident_blocks(sc, tk, &nb);
            if (rc != RTX_OK) return rc;
        }
    }
    return RTX_OK;
}

extern "C" int rtx_closest_hit_noalloc(rtx_scene* sc, size_t n, const float* d_origin_xyz, const float* d_dir_xyz, uint32_t* d_prim_id, float* d_t,
                                       float* d_position_xyz, float* d_normal_xyz, float* d_uv, uint8_t* d_front_face, void* stream, uint32_t flags)
{
    if (!sc || (n && (!d_origin_xyz || !d_dir_xyz))) return fail(RTX_E_INVALID, "null argument");
    if (n > RTX_NOALLOC_MAX_RAYS)
        return fail(RTX_E_LIMIT, "rtx_closest_hit_noalloc: %zu rays exceed the scratch array's %u (RTX_NOALLOC_MAX_RAYS); split the batch", n,
                    (unsigned)RTX_NOALLOC_MAX_RAYS);
    if (flags & RTX_RENDER_COUNT_VISITS) return fail(RTX_E_INVALID, "rtx_closest_hit_noalloc has no visit counters: use rtx_trace_closest");
    if (!sc->noalloc_scratch) return fail(RTX_E_HIP, "the scratch array has no address on this device");
    if (n == 0) return RTX_OK;
    // asynchronous, and meant for loops next to other GPU code: the caller's current device is put back on every return path
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard;
    int cur = -1;
    HIPCHECK(hipGetDevice(&cur));
    if (cur != sc->device) {
        HIPCHECK(hipSetDevice(sc->device));
        guard.prev = cur;
    }
    hipStream_t     st = (hipStream_t)stream;
    NoAllocScratch* sx = (NoAllocScratch*)sc->noalloc_scratch; // device address: only ever handed to kernels and hipMemsetAsync
    // every ray of the batch overwrites its own slots of rays[] and hits[] (stride n), and launch_trace clears the
    // work counters: nothing of the previous call is read
    hipLaunchKernelGGL(k_prep_rays, dim3(blocks_for(n, 256)), dim3(256), 0, st, d_origin_xyz, d_dir_xyz, (uint64_t)n, sx->rays,
                       (flags & RTX_TRACE_UNIT_DIR) ? 1 : 0);
    HIPCHECK(hipGetLastError());
    int rc = launch_trace(sc, sx->rays, n, n, sx->hits, n, st, false, nullptr, (flags & RTX_TRACE_LITERAL) != 0, sx->work);
    if (rc != RTX_OK) return rc;
    hipLaunchKernelGGL(k_expand_hits, dim3(blocks_for(n, 256)), dim3(256), 0, st, sc->dev, sx->rays, sx->hits, (uint64_t)n, d_prim_id, d_t, d_position_xyz,
                       d_normal_xyz, d_uv, d_front_face);
    HIPCHECK(hipGetLastError());
    return RTX_OK;
}

// ---------------------------------------------------------------------------------------------
// frame rendering
// ---------------------------------------------------------------------------------------------
static int check_params(const rtx_render_params* p, FrameGeom* g, uint32_t* rows_out)
{
    if (!p) return fail(RTX_E_INVALID, "null render params");
    if (p->width < 2 || p->height < 2) return fail(RTX_E_INVALID, "image must be at least 2x2 (u = x/(w-1), renderer.c:101)");
    if (p->width > 65536 || p->height > 65536) return fail(RTX_E_LIMIT, "image larger than 65536 per side");
    if (p->spp == 0) return fail(RTX_E_INVALID, "spp must be positive");
    if (p->max_depth > RTX_MAX_DEPTH) return fail(RTX_E_LIMIT, "max_depth above %d", RTX_MAX_DEPTH);
    uint32_t sb = p->sample_begin, se = p->sample_end;
    if (sb == 0 && se == 0) se = p->spp;
    if (sb >= se || se > p->spp) return fail(RTX_E_INVALID, "bad sample range [%u,%u) for spp %u", sb, se, p->spp);
    if (p->band_rows && (p->band_world == 0 || p->band_rank >= p->band_world)) return fail(RTX_E_INVALID, "bad band partition");
    g->width = p->width, g->height = p->height, g->spp = p->spp;
    g->band_rows  = (p->band_rows && p->band_world > 1) ? p->band_rows : 0;
    g->band_rank  = p->band_rank;
    g->band_world = p->band_world ? p->band_world : 1;
    g->sample_begin = sb, g->n_samples = se - sb;
    g->seed_lo = p->seed_lo, g->seed_hi = p->seed_hi;
    g->pixel_centres = (p->flags & RTX_RENDER_PIXEL_CENTRES) ? 1 : 0;
    g->tile = 0, g->slab_rows = 0, g->tile_h = 0;
    g->jump_tab = nullptr;
    memset(g->jump_b, 0, sizeof g->jump_b);
    uint32_t rows = 0;
    if (g->band_rows == 0) {
        rows = p->height;
    } else {
        uint32_t nb = (p->height + g->band_rows - 1) / g->band_rows;
        for (uint32_t b = g->band_rank; b < nb; b += g->band_world) rows += std::min(g->band_rows, p->height - b * g->band_rows);
    }
    *rows_out = rows;
    geom_finish(*g);
    return RTX_OK;
}

#ifndef PATH_TILE
#define PATH_TILE 32u
#endif

static int get_event(rtx_scene* sc, int i, hipEvent_t* ev)
{
    while (sc->n_ev <= i) {
        if (sc->n_ev >= (int)(sizeof sc->ev / sizeof sc->ev[0])) return fail(RTX_E_LIMIT, "too many timed launches in one call");
        HIPCHECK(hipEventCreate(&sc->ev[sc->n_ev]));
        sc->n_ev++;
    }
    *ev = sc->ev[i];
    return RTX_OK;
}

static int render_device_impl(rtx_scene* sc, const rtx_camera* cam, const rtx_render_params* p, float* d_accum, void* stream, rtx_stats* stats);

extern "C" int rtx_render_device(rtx_scene* sc, const rtx_camera* cam, const rtx_render_params* p, float* d_accum, void* stream, rtx_stats* stats)
{
    const int rc = render_device_impl(sc, cam, p, d_accum, stream, stats);
    if (rc != RTX_OK && sc) {
        // A failure in the middle of a frame leaves kernels and the copies of the survivor counts in flight on the caller's
        // stream and on the lanes' own streams; the next call may regrow (free) the workspace they use or reuse their events.
        // Drain them before reporting — errors of the draining itself are ignored, the message is the first failure's.
        char first[sizeof g_err];
        memcpy(first, g_err, sizeof first);
        (void)hipSetDevice(sc->device);
        (void)hipStreamSynchronize((hipStream_t)stream);
        for (int l = 0; l < RTX_MAX_LANES - 1; l++)
            if (sc->aux_stream[l]) (void)hipStreamSynchronize(sc->aux_stream[l]);
        (void)hipGetLastError();
        memcpy(g_err, first, sizeof first);
    }
    return rc;
}

static int render_device_impl(rtx_scene* sc, const rtx_camera* cam, const rtx_render_params* p, float* d_accum, void* stream, rtx_stats* stats)
{
    if (!sc || !cam || !d_accum) return fail(RTX_E_INVALID, "null argument");
    FrameGeom g;
    uint32_t  rows = 0;
    int       rc   = check_params(p, &g, &rows);
    if (rc != RTX_OK) return rc;
    HIPCHECK(hipSetDevice(sc->device));
    if (knob(K_JUMP_TABLE) != 0) { // 0: the literal 128-step jump in raygen (the tests compare the two)
        g.jump_tab = sc->d_jump_tab;
        host_jump(0, p->seed_hi, g.jump_b);
    }
    hipStream_t st = (hipStream_t)stream;
    rtx_stats   local;
    memset(&local, 0, sizeof local);
    const bool count   = (p->flags & RTX_RENDER_COUNT_VISITS) != 0;
    const bool literal = (p->flags & RTX_TRACE_LITERAL) != 0;

    const uint64_t n_pixels = (uint64_t)rows * g.width;
    const uint64_t total    = n_pixels * g.n_samples;
    if (total == 0 || p->max_depth == 0) { // depth 0 → every RayColor is black: accum stays as is
        if (stats) *stats = local;
        return RTX_OK;
    }
    // Lanes (RTX_OVERLAP=1..3, off by default).  The assigned pixels can be cut into 2-4 parts that go through the bounce
    // loop on as many streams, a pass of one overlapping the end of another's launch.  The parts are independent slabs,
    // so the frame is the same, bit for bit (tests).  It paid 4-8 % on a rank's share of a frame split over 2-8 GPUs
    // while a launch ended with some waves a whole 1024-ray grab behind the others; with 128-ray grabs what is left
    // of a launch's end is its longest ray, which overlap cannot shorten, and one lane is as fast (tools/dbg/bands.py).
    const int   overlap = (int)knob(K_OVERLAP);
    int lanes = overlap >= 1 ? std::min(overlap + 1, RTX_MAX_LANES) : 1; // off unless asked for: see above
    if ((uint32_t)lanes > rows) lanes = (int)rows;
    // slab sizing: whole pixels, at most `budget` paths in flight
    uint64_t budget = p->max_paths_in_flight ? p->max_paths_in_flight : (160ull << 20); // whole 1080p/64spp frame: 133 M paths, 18.6 GB of 288
    if (budget > 0xFFFFFF00ull) budget = 0xFFFFFF00ull; // path ids are u32
    uint64_t px_per_slab = std::max<uint64_t>(1, budget / lanes / g.n_samples);
    px_per_slab          = std::min(px_per_slab, (uint64_t)((rows + lanes - 1) / lanes) * g.width); // whole rows when the frame is halved
    const uint64_t cap   = px_per_slab * g.n_samples;
    if (cap > 0xFFFFFF00ull) return fail(RTX_E_LIMIT, "spp too large for one slab");

    const uint64_t qbytes     = cap * Q_RECORD * 4;
    const uint64_t bin_bytes  = (4ull << BIN_BITS) + 4ull * BIN_CHUNKS; // counts, then the chunk totals of the scan
    const uint64_t lane_bytes = (2 * qbytes + cap * 16 + cap * 12 + cap * 8 + bin_bytes + 255) & ~255ull;
    char*          ws         = nullptr;
    if ((rc = workspace(sc, lanes * lane_bytes, &ws)) != RTX_OK) return rc;
    // bin the queues that feed depth >= sort_depth (knob SORT_DEPTH; 0 = never)
    const int   sort_depth = (int)knob(K_SORT_DEPTH);
    // binning a queue costs a pass over it plus a fixed ~0.2 ms (histogram scan, launches): below ~1.5 M rays the closest-hit
    // pass it speeds up is too short to pay that back (tools/dbg/bands.py: a rank's share under an 8-way split)
    const long long bin_min    = knob(K_SORT_MIN); // (4 M before the scatter got cheaper, 2.5 M until v25: tools/dbg/share_sortmin.py)
    const int       sort_every = (int)std::max(1ll, knob(K_SORT_EVERY)); // experiments: bin every k-th eligible depth (huge = once)
    if (!sc->h_alive) HIPCHECK(hipHostMalloc((void**)&sc->h_alive, RTX_MAX_LANES * sizeof(unsigned long long)));
    if (lanes > 1 && !sc->ev_fork) HIPCHECK(hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming));
    for (int l = 1; l < lanes; l++) {
        if (sc->aux_stream[l - 1]) continue;
        HIPCHECK(hipStreamCreateWithFlags(&sc->aux_stream[l - 1], hipStreamNonBlocking));
        HIPCHECK(hipEventCreateWithFlags(&sc->ev_join[l - 1], hipEventDisableTiming));
    }

    unsigned long long* d_shaded = (unsigned long long*)&sc->d_counters[RTX_MAX_LANES * 80 + 80];
    HIPCHECK(hipMemsetAsync(sc->d_counters, 0, 16 * sizeof(uint64_t), st));
    // The tail of the bounce loop in one launch (k_trace_flat<..., TAIL>): once a lane's queue is too short to fill the
    // chip, a pass per depth costs more in ramps and drains than in work.  RTX_TAIL_MAX = longest queue handed to it
    // (0: never); read per call, the tests switch it.  Same paths, same results: a path's RNG state and result slot are its own.
    unsigned long long* d_tail = (unsigned long long*)&sc->d_counters[RTX_MAX_LANES * 80];
    const long long     tail_max = (count || !tail_capable(sc, literal)) ? 0 : knob(K_TAIL_MAX);
    const uint32_t      tail_shade_min = (uint32_t)std::min(64ll, std::max(1ll, knob(K_TAIL_SHADE_MIN)));
    bool                tail_used = false;
    // RTX_RAYGEN_LEAN=0: k_raygen stores whole path records and the first shade pass reads them (the tests compare the two)
    const bool          lean_raygen = knob(K_RAYGEN_LEAN) != 0;
    HIPCHECK(hipMemsetAsync(d_tail, 0, (TAIL_COUNTERS + EARLY_COUNTERS) * sizeof(uint64_t), st));
    // rays that miss the box of the whole tree and every unbounded object are answered by the shade pass that creates
    // them (see k_shade); not under the literal walk, which is there to walk, nor when asked not to (the tests compare)
    unsigned long long* d_early   = d_tail + TAIL_COUNTERS;
    const bool          early_miss = !literal && knob(K_EARLY_MISS) != 0;

    int        nev = 0;
    hipEvent_t evA, evB;
    if ((rc = get_event(sc, nev++, &evA)) != RTX_OK) return rc;
    if ((rc = get_event(sc, nev++, &evB)) != RTX_OK) return rc;
    HIPCHECK(hipEventRecord(evA, st));
    if (lanes > 1) { // the other streams start after what the caller queued on the first (the zeroed frame buffer)
        HIPCHECK(hipEventRecord(sc->ev_fork, st));
        for (int l = 1; l < lanes; l++) HIPCHECK(hipStreamWaitEvent(sc->aux_stream[l - 1], sc->ev_fork, 0));
    }
    struct Span { int a, b, kind; uint32_t depth; };
    std::vector<Span> spans;
    auto timed = [&](int kind, uint32_t depth, hipStream_t s, auto&& launch) -> int {
        if (nev + 2 > (int)(sizeof sc->ev / sizeof sc->ev[0])) { // a frame cut into very many slabs: the rest goes untimed
            launch();
            HIPCHECK(hipGetLastError());
            return RTX_OK;
        }
        hipEvent_t a, b;
        int        ia = nev, ib = nev + 1, r;
        if ((r = get_event(sc, nev++, &a)) != RTX_OK) return r;
        if ((r = get_event(sc, nev++, &b)) != RTX_OK) return r;
        HIPCHECK(hipEventRecord(a, s));
        launch();
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipEventRecord(b, s));
        spans.push_back({ia, ib, kind, depth});
        return RTX_OK;
    };

    struct Lane {
        hipStream_t         s;
        float*              qb[2];
        Hits                hits;
        float*              result;
        uint2*              bin_keys;
        uint32_t*           bin_cnt;
        unsigned long long *d_cnt, *work, *h_alive;
        uint64_t            px0, npx, n;
        FrameGeom           gs;
        int                 cur;
        Queue               qin;
        uint32_t            depth;
        bool                active, bin, last;
        bool                ahead;   // the next depth's binning and closest-hit pass are already queued (on the count the device holds)
        hipEvent_t          ev_cnt;  // the survivors' count has reached the host
    } L[RTX_MAX_LANES];
    // A depth's closest-hit pass does not wait for the host (knob AHEAD): the kernel is persistent — its grid does not depend
    // on the queue's length — and reads that length from the counter the shade pass leaves on the device, so it is queued
    // right behind that shade pass (and the binning pass, whose scatter reads the same counter).  The host reads the count
    // while the closest-hit pass runs and sizes the NEXT shade pass with it; the device never idles through a round trip.
    const bool ahead_ok = knob(K_AHEAD) != 0 && pick_trace_kernel(sc, count, literal, false).variant >= 2;
    for (int l = 0; l < lanes; l++) {
        char* w       = ws + l * lane_bytes;
        L[l].s        = l == 0 ? st : sc->aux_stream[l - 1];
        L[l].qb[0]    = (float*)w, L[l].qb[1] = (float*)(w + qbytes);
        L[l].hits     = Hits{(uint32_t*)(w + 2 * qbytes), cap};
        L[l].bin_keys = (uint2*)(w + 2 * qbytes + cap * 16); // 8-byte entries on a 16-byte boundary
        L[l].result   = (float*)(w + 2 * qbytes + cap * 16 + cap * 8);
        L[l].bin_cnt  = (uint32_t*)(w + 2 * qbytes + cap * 16 + cap * 8 + cap * 12);
        L[l].d_cnt    = (unsigned long long*)&sc->d_counters[80 * l + 6];
        L[l].work     = (unsigned long long*)&sc->d_counters[80 * l + 16];
        L[l].h_alive  = sc->h_alive + l;
        L[l].active   = false;
        L[l].ahead    = false;
        if ((rc = get_event(sc, nev++, &L[l].ev_cnt)) != RTX_OK) return rc;
    }
    auto soa = [&](const Lane& a, int b) { return Queue{a.qb[b], cap, 1}; };
    auto aos = [&](const Lane& a, int b) { return Queue{a.qb[b], 1, Q_RECORD}; };

    // The camera pass's queue is in image order (tiles).  Cut into eighths — one per XCD — the eighths of the picture that hold
    // the model take several times longer than those that hold ground and sky; the XCDs that finish early go on with the
    // others' parts and the pass ends on the slow eighths' last rays.  Runs of 4096 rays (one pixel block of a tile) dealt out
    // to the parts in turn give every XCD a share of every region: 10.7 -> 10.0 ms for the bench frame's camera pass (sweep:
    // 64 k rays 10.3, 16 k 10.1, 4 k 10.0, 1 k 10.0).  The bounce passes keep their eighths: interleaved, the first bounce
    // (whose queue is in image order too) lost 0.6 ms, and the binned queues' order IS their locality.  Only which wave
    // traces which ray changes; frames are the same bit for bit (test_tile_rows_dealt_out...).
    auto il_len = [&](const Lane&) -> uint64_t { return (uint64_t)std::max(0ll, knob(K_TILE_INTERLEAVE)); };
    auto launch_binning = [&](Lane& a, uint64_t n, const unsigned long long* n_dev, Queue from, Queue to) {
        uint32_t* chunk = a.bin_cnt + (1u << BIN_BITS);
        hipLaunchKernelGGL(k_bin_scan, dim3(BIN_CHUNKS), dim3(1024), 0, a.s, a.bin_cnt, chunk);
        hipLaunchKernelGGL(k_bin_scan_chunks, dim3(1), dim3(BIN_CHUNKS), 0, a.s, chunk);
        hipLaunchKernelGGL(k_bin_scatter, dim3((unsigned)std::min<uint64_t>(blocks_for(n, 256), 1u << 16)), dim3(256), 0, a.s, from, n, n_dev, a.bin_cnt, chunk,
                           a.bin_keys, to);
    };
    // one bounce pass of a lane: closest hit, shade + compaction, and the survivors' count on its way to the host
    auto enqueue_depth = [&](Lane& a) -> int {
        const uint32_t depth = a.depth;
        local.rays_at_depth[depth] += a.n;
        local.rays_total += a.n;
        int lrc = RTX_OK, r = RTX_OK;
        if (!a.ahead) { // (else: queued behind the previous depth's shade pass)
            r = timed(1, depth, a.s, [&] {
                lrc = launch_trace(sc, a.qin.base, a.qin.es == 1 ? cap : 0, a.n, a.hits.base, cap, a.s, count, sc->d_counters, literal, a.work, depth > 0, nullptr, nullptr,
                                   depth == 0 ? il_len(a) : 0);
            });
            if (r != RTX_OK || lrc != RTX_OK) return r != RTX_OK ? r : lrc;
        }
        a.ahead = false;
        local.trace_launches++;
        HIPCHECK(hipMemsetAsync(a.d_cnt, 0, sizeof(unsigned long long), a.s));
        a.last = depth + 1 == p->max_depth;
        a.bin  = !a.last && sort_depth > 0 && (int)depth + 1 >= sort_depth && ((int)depth + 1 - sort_depth) % sort_every == 0 && (long long)a.n >= bin_min;
        if (a.bin) HIPCHECK(hipMemsetAsync(a.bin_cnt, 0, bin_bytes, a.s));
        r = timed(2, depth, a.s, [&] {
            hipLaunchKernelGGL(k_shade, dim3(blocks_for(a.n, SHADE_BLOCK)), dim3(SHADE_BLOCK), 0, a.s, sc->dev, a.qin, a.hits, a.n, soa(a, a.cur ^ 1), a.d_cnt,
                               d_shaded, a.result, cap, a.last ? 1 : 0, a.bin ? a.bin_cnt : nullptr, a.bin_keys, a.gs, a.px0, depth == 0 && lean_raygen ? 1 : 0,
                               early_miss && !a.last ? d_early + depth + 1 : nullptr);
        });
        if (r != RTX_OK) return r;
        *a.h_alive = 0;
        if (!a.last) HIPCHECK(hipMemcpyAsync(a.h_alive, a.d_cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, a.s));
        HIPCHECK(hipEventRecord(a.ev_cnt, a.s));
        // (not when this queue could already have gone to the tail launch: then the next one may, and that is another kernel)
        if (ahead_ok && !a.last && (long long)a.n > tail_max) {
            Queue nq = soa(a, a.cur ^ 1);
            if (a.bin) {
                r = timed(4, depth + 1, a.s, [&] { launch_binning(a, a.n, a.d_cnt, nq, aos(a, a.cur)); });
                if (r != RTX_OK) return r;
                nq = aos(a, a.cur);
            }
            r = timed(1, depth + 1, a.s, [&] {
                lrc = launch_trace(sc, nq.base, nq.es == 1 ? cap : 0, a.n, a.hits.base, cap, a.s, count, sc->d_counters, literal, a.work, true, nullptr, a.d_cnt,
                                   0);
            });
            if (r != RTX_OK || lrc != RTX_OK) return r != RTX_OK ? r : lrc;
            a.ahead = true;
        }
        return RTX_OK;
    };
    uint64_t next_px = 0;
    auto start_slab = [&](Lane& a) -> int { // the lane's next slab: camera rays, then its first pass
        a.active = next_px < n_pixels;
        if (!a.active) return RTX_OK;
        a.px0 = next_px, a.npx = std::min(px_per_slab, n_pixels - next_px), a.n = a.npx * g.n_samples;
        next_px += a.npx;
        local.passes++;
        local.camera_paths += a.n;
        // (no clearing of the result slab: every path writes its slot exactly once — on a miss, when its material ends
        // it, or at the last depth — before the slab is resolved)
        a.gs = g; // tile-major path order when the slab is made of whole rows
        if (a.px0 % g.width == 0 && a.npx % g.width == 0 && a.npx / g.width >= 2 * PATH_TILE) {
            a.gs.tile = a.gs.tile_h = PATH_TILE, a.gs.slab_rows = (uint32_t)(a.npx / g.width);
            // a rank's rows under the band split are bands from all over the image: a tile as high as a band (and as
            // much wider) stays one compact region instead of a stack of strips band_world bands apart
            const bool band_tiles = knob(K_BAND_TILES) != 0;
            if (band_tiles && g.band_rows && g.band_rows < PATH_TILE && a.px0 == 0 && a.gs.slab_rows % g.band_rows == 0)
                a.gs.tile_h = g.band_rows, a.gs.tile = std::max<uint32_t>(PATH_TILE, PATH_TILE * PATH_TILE / g.band_rows);
        }
        geom_finish(a.gs);
        a.cur = 0, a.qin = soa(a, 0), a.depth = 0; // SoA out of raygen / shade, AoS out of the binning pass
        int r = timed(0, 0, a.s, [&] { hipLaunchKernelGGL(k_raygen, dim3(blocks_for(a.n, 256)), dim3(256), 0, a.s, *cam, a.gs, a.px0, a.n, soa(a, 0), lean_raygen ? 1 : 0); });
        return r != RTX_OK ? r : enqueue_depth(a);
    };
    auto advance = [&](Lane& a) -> int { // the lane's pass has finished: bin and go one deeper, or close the slab
        a.n   = a.last ? 0 : *a.h_alive;
        if (a.ahead) { // the queue has been binned (a.bin) and its closest-hit pass is running: on to that depth's shade pass
            if (!a.bin) a.cur ^= 1;
            a.qin = a.bin ? aos(a, a.cur) : soa(a, a.cur);
            a.depth++;
            if (a.n > 0) return enqueue_depth(a);
            a.ahead = false; // (it found an empty queue)
            int r = timed(3, 0, a.s, [&] {
                hipLaunchKernelGGL(k_resolve_slab, dim3(blocks_for(a.npx, 256)), dim3(256), 0, a.s, a.result, cap, a.gs, a.px0, a.npx, d_accum);
            });
            return r != RTX_OK ? r : start_slab(a);
        }
        a.cur ^= 1;
        a.qin = soa(a, a.cur);
        if (a.n > 0 && (long long)a.n <= tail_max && a.depth + 1 < p->max_depth) {
            // all remaining depths of these paths in one launch; the kernel counts the rays it starts beyond this queue's
            a.depth++;
            local.rays_at_depth[a.depth] += a.n;
            local.rays_total += a.n;
            const TailArgs ta = {a.result, cap, a.depth, p->max_depth, d_tail, tail_shade_min};
            int            lrc = RTX_OK;
            int            r   = timed(1, a.depth, a.s, [&] {
                lrc = launch_trace(sc, a.qin.base, cap, a.n, a.hits.base, cap, a.s, false, sc->d_counters, literal, a.work, true, &ta);
            });
            if (r != RTX_OK || lrc != RTX_OK) return r != RTX_OK ? r : lrc;
            local.trace_launches++;
            tail_used = true;
            a.n       = 0; // nothing is left for a further pass
            a.depth   = p->max_depth;
        }
        if (a.bin && a.n > 0) {
            int r = timed(4, a.depth, a.s, [&] { launch_binning(a, a.n, nullptr, a.qin, aos(a, a.cur ^ 1)); });
            if (r != RTX_OK) return r;
            a.cur ^= 1;
            a.qin = aos(a, a.cur);
        }
        a.depth++;
        if (a.depth < p->max_depth && a.n > 0) return enqueue_depth(a);
        int r = timed(3, 0, a.s, [&] {
            hipLaunchKernelGGL(k_resolve_slab, dim3(blocks_for(a.npx, 256)), dim3(256), 0, a.s, a.result, cap, a.gs, a.px0, a.npx, d_accum);
        });
        return r != RTX_OK ? r : start_slab(a);
    };
    for (int l = 0; l < lanes; l++)
        if ((rc = start_slab(L[l])) != RTX_OK) return rc;
    // round robin: while the host waits for one lane's pass, the other lane's next pass is already queued
    auto any_active = [&] {
        for (int l = 0; l < lanes; l++)
            if (L[l].active) return true;
        return false;
    };
    while (any_active()) {
        for (int l = 0; l < lanes; l++) {
            if (!L[l].active) continue;
            HIPCHECK(hipEventSynchronize(L[l].ev_cnt)); // the lane's shade pass is done and its count is here
            if ((rc = advance(L[l])) != RTX_OK) return rc;
        }
    }
    for (int l = 1; l < lanes; l++) {
        HIPCHECK(hipEventRecord(sc->ev_join[l - 1], sc->aux_stream[l - 1]));
        HIPCHECK(hipStreamWaitEvent(st, sc->ev_join[l - 1], 0));
    }
    HIPCHECK(hipEventRecord(evB, st));
    HIPCHECK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, evA, evB));
    local.ms_total = ms;
    for (const Span& s : spans) {
        HIPCHECK(hipEventElapsedTime(&ms, sc->ev[s.a], sc->ev[s.b]));
        (s.kind == 0 ? local.ms_raygen : s.kind == 1 ? local.ms_trace : s.kind == 2 ? local.ms_shade : s.kind == 3 ? local.ms_resolve : local.ms_sort) += ms;
        if (s.kind == 1) local.ms_trace_at_depth[s.depth] += ms;
    }
    uint64_t c[16];
    HIPCHECK(hipMemcpy(c, sc->d_counters, sizeof c, hipMemcpyDeviceToHost));
    local.visits_inner = c[0], local.visits_leaf = c[1], local.visits_prim = c[2], local.visits_culled = c[3];
    {
        uint64_t sh[64];
        HIPCHECK(hipMemcpy(sh, d_shaded, sizeof sh, hipMemcpyDeviceToHost));
        for (int k = 0; k < 8; k++) local.hits_shaded += sh[8 * k];
    }
    for (int i = 0; i < 8; i++) local.wave_stats[i] = c[8 + i];
    if (tail_used) {
        uint64_t tc[RTX_MAX_DEPTH + 1];
        HIPCHECK(hipMemcpy(tc, d_tail, sizeof tc, hipMemcpyDeviceToHost));
        for (int i = 0; i < RTX_MAX_DEPTH; i++) local.rays_at_depth[i] += tc[i], local.rays_total += tc[i];
        local.hits_shaded += tc[RTX_MAX_DEPTH];
    }
    if (early_miss) {
        uint64_t ec[EARLY_COUNTERS];
        HIPCHECK(hipMemcpy(ec, d_early, sizeof ec, hipMemcpyDeviceToHost));
        for (int k = 0; k < 8; k++)
            for (int dpt = 0; dpt < RTX_MAX_DEPTH; dpt++) {
                const uint64_t c = ec[RTX_MAX_DEPTH * k + dpt];
                local.rays_at_depth[dpt] += c, local.rays_total += c, local.rays_early_miss += c;
            }
    }
    local.samples_done = g.n_samples;
    if (stats) *stats = local;
    return RTX_OK;
}

extern "C" int rtx_resolve_device(rtx_scene* sc, const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_bgr, void* stream)
{
    if (!sc || !d_accum || !d_bgr || !spp) return fail(RTX_E_INVALID, "null argument");
    HIPCHECK(hipSetDevice(sc->device));
    uint64_t n = (uint64_t)width * height;
    hipLaunchKernelGGL(k_tonemap, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, d_accum, n, 1.0f / (float)spp, d_bgr);
    HIPCHECK(hipGetLastError());
    return RTX_OK;
}

static void add_stats(rtx_stats& t, const rtx_stats& s)
{
    t.rays_total += s.rays_total, t.camera_paths += s.camera_paths, t.hits_shaded += s.hits_shaded, t.rays_early_miss += s.rays_early_miss;
    t.visits_inner += s.visits_inner, t.visits_leaf += s.visits_leaf, t.visits_prim += s.visits_prim, t.visits_culled += s.visits_culled;
    t.ms_total += s.ms_total, t.ms_trace += s.ms_trace, t.ms_shade += s.ms_shade, t.ms_raygen += s.ms_raygen, t.ms_resolve += s.ms_resolve;
    t.ms_sort += s.ms_sort, t.trace_launches += s.trace_launches, t.passes += s.passes;
    for (int i = 0; i < RTX_MAX_DEPTH; i++) t.rays_at_depth[i] += s.rays_at_depth[i], t.ms_trace_at_depth[i] += s.ms_trace_at_depth[i];
    for (int i = 0; i < 8; i++) t.wave_stats[i] += s.wave_stats[i];
}

static int render_host(rtx_scene* sc, const rtx_camera* cam, const rtx_render_params* p, uint32_t spp_per_pass, uint8_t* bgr_out, float* accum_out,
                       rtx_progress_fn progress, void* user, rtx_stats* stats)
{
    if (!sc || !cam || !p) return fail(RTX_E_INVALID, "null argument");
    FrameGeom g;
    uint32_t  rows;
    int       rc = check_params(p, &g, &rows);
    if (rc != RTX_OK) return rc;
    HIPCHECK(hipSetDevice(sc->device));
    const uint64_t npx = (uint64_t)p->width * p->height;
    DevBuf         bacc, bbgr;
    HIPCHECK(bacc.alloc(npx * 12));
    HIPCHECK(bbgr.alloc(npx * 3));
    HIPCHECK(hipMemset(bacc.p, 0, npx * 12));
    rtx_stats total;
    memset(&total, 0, sizeof total);
    const uint32_t sb = g.sample_begin, se = g.sample_begin + g.n_samples;
    const uint32_t step = spp_per_pass ? spp_per_pass : g.n_samples; // 0: the whole range at once (rtx_render)
    for (uint32_t s = sb; s < se; s += std::min(step, se - s)) {
        rtx_render_params q = *p;
        q.sample_begin = s, q.sample_end = s + std::min(step, se - s);
        rtx_stats st;
        rc = rtx_render_device(sc, cam, &q, bacc.as<float>(), nullptr, &st);
        if (rc != RTX_OK) return rc;
        add_stats(total, st);
        total.samples_done = q.sample_end - sb;
        if (bgr_out) { // the whole job's average once every sample is in (renderer.c:109), else the average so far
            const uint32_t denom = q.sample_end == se ? p->spp : total.samples_done;
            rc = rtx_resolve_device(sc, bacc.as<float>(), p->width, p->height, denom, bbgr.as<uint8_t>(), nullptr);
            if (rc != RTX_OK) return rc;
            HIPCHECK(hipMemcpy(bgr_out, bbgr.p, npx * 3, hipMemcpyDeviceToHost));
        }
        if (progress && progress(user, total.samples_done, se - sb) != 0) break;
    }
    if (stats) *stats = total;
    if (accum_out) {
        // rows not assigned to this call stay untouched in the caller's buffer
        if (g.band_rows == 0) {
            HIPCHECK(hipMemcpy(accum_out, bacc.p, npx * 12, hipMemcpyDeviceToHost));
        } else {
            uint32_t nb = (p->height + g.band_rows - 1) / g.band_rows;
            for (uint32_t b = g.band_rank; b < nb; b += g.band_world) {
                uint64_t r0 = (uint64_t)b * g.band_rows, nr = std::min<uint64_t>(g.band_rows, p->height - r0);
                HIPCHECK(hipMemcpy(accum_out + r0 * p->width * 3, bacc.as<float>() + r0 * p->width * 3, nr * p->width * 12, hipMemcpyDeviceToHost));
            }
        }
    }
    return RTX_OK;
}

extern "C" int rtx_render(rtx_scene* sc, const rtx_camera* cam, const rtx_render_params* p, uint8_t* bgr_out, float* accum_out, rtx_stats* stats)
{
    return render_host(sc, cam, p, 0, bgr_out, accum_out, nullptr, nullptr, stats);
}

extern "C" int rtx_render_progressive(rtx_scene* sc, const rtx_camera* cam, const rtx_render_params* p, uint32_t spp_per_pass, uint8_t* bgr_out,
                                      float* accum_out, rtx_progress_fn progress, void* user, rtx_stats* stats)
{
    if (spp_per_pass == 0) return fail(RTX_E_INVALID, "rtx_render_progressive: spp_per_pass must be positive");
    return render_host(sc, cam, p, spp_per_pass, bgr_out, accum_out, progress, user, stats);
}

// ---------------------------------------------------------------------------------------------
// One frame over several devices behind the C ABI.
//
// The reference's Render_Waiter spawns NUM_HYPERTHREADS workers that share the frame and joins them
// (src/rt/renderer.c:186-237); here the waiter's workers are one host thread per DEVICE: each renders its share — the
// interleaved bands b = r (mod n) of `band_rows` rows, or the samples [r spp/n, (r+1) spp/n) of every pixel — with its own
// replicated scene into a full-frame f32 buffer on its device, and the shares meet on the root (scenes[0]'s device):
//   * tile split: the bands are disjoint, so a share's rows are COPIED into the root's buffer, band by band, device to device
//     (hipMemcpyPeerAsync; 1/n of the frame per peer): the frame is rtx_render's bit for bit;
//   * sample split: a peer's buffer is copied whole into a staging buffer on the root and ADDED there in rank order
//     (k_accum_add): partial sums of the per-pixel sum of renderer.c:106 in another association, as any reduce gives.
// Why peer copies and not RCCL here: it is ONE reduction to ONE root of 25-100 MB per frame, inside one process.  On the
// xGMI mesh the n-1 peers write over n-1 different links into the root at the same time (0.16-0.65 ms), the sum
// order is fixed (rank order: reproducible frames), nothing needs a communicator (ncclCommInitAll takes seconds and pulls
// a gigabyte library into a C renderer) and a C caller links nothing but libamdhip64.  The multi-PROCESS path
// (bench.py under torch.distributed.run, raytracer_amd/distributed.py) is the one that reduces through RCCL.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_accum_add(float* __restrict__ acc, const float* __restrict__ part, uint64_t n4, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        float4 a = ((float4*)acc)[i];
        const float4 b = ((const float4*)part)[i];
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        ((float4*)acc)[i] = a;
    } else if (i == n4) {
        for (uint64_t k = 4 * n4; k < n; k++) acc[k] += part[k];
    }
}

static int multi_buffer(rtx_scene* sc, int which, uint64_t bytes, float** out) // device current
{
    if (sc->multi_bytes[which] < bytes) {
        if (sc->multi_buf[which]) (void)hipFree(sc->multi_buf[which]);
        sc->multi_buf[which]   = nullptr;
        sc->multi_bytes[which] = 0;
        hipError_t e = hipMalloc(&sc->multi_buf[which], bytes);
        if (e != hipSuccess) return fail(RTX_E_NOMEM, "rtx_render_multi: cannot allocate %llu bytes on device %d: %s", (unsigned long long)bytes, sc->device, hipGetErrorString(e));
        sc->multi_bytes[which] = bytes;
    }
    *out = (float*)sc->multi_buf[which];
    return RTX_OK;
}

struct MultiWorker {
    rtx_scene*        sc;
    rtx_render_params p;
    float*            d_share;
    rtx_stats         st;
    int               rc;
    char              err[sizeof g_err];
};

extern "C" int rtx_render_multi(rtx_scene* const* scenes, uint32_t n, const rtx_camera* cam, const rtx_render_params* params, uint32_t split_mode,
                                uint32_t band_rows, uint8_t* bgr_out, float* accum_out, rtx_stats* stats)
{
    if (!scenes || n == 0 || !cam || !params) return fail(RTX_E_INVALID, "rtx_render_multi: null argument");
    if (n > 64) return fail(RTX_E_LIMIT, "rtx_render_multi: at most 64 scene handles");
    if (split_mode != RTX_SPLIT_TILE && split_mode != RTX_SPLIT_SAMPLE) return fail(RTX_E_INVALID, "rtx_render_multi: split_mode must be RTX_SPLIT_TILE or RTX_SPLIT_SAMPLE");
    for (uint32_t r = 0; r < n; r++) {
        if (!scenes[r]) return fail(RTX_E_INVALID, "rtx_render_multi: scene handle %u is null", r);
        for (uint32_t q = 0; q < r; q++)
            if (scenes[q] == scenes[r]) return fail(RTX_E_INVALID, "rtx_render_multi: scene handle %u is handle %u again (a handle serves one call at a time)", r, q);
    }
    if (params->band_rows || params->band_rank || params->band_world || params->sample_begin || params->sample_end)
        return fail(RTX_E_INVALID, "rtx_render_multi: params describes the WHOLE frame (band_* and sample_* must be 0; the call derives the shares)");
    FrameGeom g;
    uint32_t  rows = 0;
    int       rc   = check_params(params, &g, &rows);
    if (rc != RTX_OK) return rc;
    if (split_mode == RTX_SPLIT_TILE && band_rows == 0) band_rows = 8; // 8-row bands balance best at N = 8
    if (split_mode == RTX_SPLIT_SAMPLE && params->spp < n) return fail(RTX_E_INVALID, "rtx_render_multi: sample split of %u samples over %u devices", params->spp, n);
    const uint64_t npx = (uint64_t)params->width * params->height, fbytes = npx * 12;
    rtx_scene*     root = scenes[0];

    struct InFlight { InFlight() { g_multi_in_flight++; } ~InFlight() { g_multi_in_flight--; } } in_flight; // the tuning table is read-only from here on
    knobs_init();

    std::vector<MultiWorker> w(n);
    for (uint32_t r = 0; r < n; r++) {
        w[r].sc = scenes[r], w[r].p = *params, w[r].rc = RTX_OK, w[r].err[0] = 0, w[r].d_share = nullptr;
        memset(&w[r].st, 0, sizeof w[r].st);
        if (split_mode == RTX_SPLIT_TILE) w[r].p.band_rows = band_rows, w[r].p.band_rank = r, w[r].p.band_world = n;
        else w[r].p.sample_begin = (uint32_t)((uint64_t)r * params->spp / n), w[r].p.sample_end = (uint32_t)((uint64_t)(r + 1) * params->spp / n);
    }
    // one worker per device (renderer.c:205-233: spawn, then join)
    auto work = [&](uint32_t r) {
        MultiWorker& m = w[r];
        auto body = [&]() -> int {
            HIPCHECK(hipSetDevice(m.sc->device));
            if (!m.sc->multi_stream) HIPCHECK(hipStreamCreateWithFlags(&m.sc->multi_stream, hipStreamNonBlocking));
            int rc2 = multi_buffer(m.sc, 0, fbytes, &m.d_share);
            if (rc2 != RTX_OK) return rc2;
            // a one-device tile "split" renders the whole frame; any other share starts from zeros (rows / samples of the
            // others stay zero: the sample split adds whole buffers)
            HIPCHECK(hipMemsetAsync(m.d_share, 0, fbytes, m.sc->multi_stream));
            rc2 = rtx_render_device(m.sc, cam, &m.p, m.d_share, m.sc->multi_stream, &m.st); // synchronises its stream
            return rc2;
        };
        m.rc = body();
        if (m.rc != RTX_OK) memcpy(m.err, g_err, sizeof m.err);
    };
    {
        std::vector<std::thread> th;
        for (uint32_t r = 1; r < n; r++) th.emplace_back(work, r);
        work(0);
        for (auto& t : th) t.join();
    }
    for (uint32_t r = 0; r < n; r++)
        if (w[r].rc != RTX_OK) {
            memcpy(g_err, w[r].err, sizeof g_err);
            return w[r].rc;
        }
    if (stats)
        for (uint32_t r = 0; r < n; r++) stats[r] = w[r].st;

    // the shares meet on the root
    HIPCHECK(hipSetDevice(root->device));
    hipStream_t rs = root->multi_stream;
    float*      d_frame = w[0].d_share;
    auto peer_copy = [&](void* dst, const void* src, int src_dev, uint64_t bytes) -> hipError_t {
        return src_dev == root->device ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rs)
                                       : hipMemcpyPeerAsync(dst, root->device, src, src_dev, bytes, rs);
    };
    if (split_mode == RTX_SPLIT_TILE) {
        const uint32_t nb = (params->height + band_rows - 1) / band_rows;
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t r = b % n;
            if (r == 0) continue;
            const uint64_t r0 = (uint64_t)b * band_rows, nr = std::min<uint64_t>(band_rows, params->height - r0);
            HIPCHECK(peer_copy(d_frame + r0 * params->width * 3, w[r].d_share + r0 * params->width * 3, w[r].sc->device, nr * params->width * 12));
        }
    } else if (n > 1) {
        float* d_stage = nullptr;
        if ((rc = multi_buffer(root, 1, fbytes, &d_stage)) != RTX_OK) return rc;
        const uint64_t nf = npx * 3, n4 = nf / 4;
        for (uint32_t r = 1; r < n; r++) { // rank order: the frame does not depend on which device finished first
            HIPCHECK(peer_copy(d_stage, w[r].d_share, w[r].sc->device, fbytes));
            hipLaunchKernelGGL(k_accum_add, dim3(blocks_for(n4 + 1, 256)), dim3(256), 0, rs, d_frame, d_stage, n4, nf);
            HIPCHECK(hipGetLastError());
        }
    }
    if (bgr_out) {
        DevBuf bbgr;
        HIPCHECK(bbgr.alloc(npx * 3));
        if ((rc = rtx_resolve_device(root, d_frame, params->width, params->height, params->spp, bbgr.as<uint8_t>(), rs)) != RTX_OK) return rc;
        HIPCHECK(hipMemcpyAsync(bgr_out, bbgr.p, npx * 3, hipMemcpyDeviceToHost, rs));
        HIPCHECK(hipStreamSynchronize(rs));
    }
    if (accum_out) HIPCHECK(hipMemcpyAsync(accum_out, d_frame, fbytes, hipMemcpyDeviceToHost, rs));
    HIPCHECK(hipStreamSynchronize(rs));
    return RTX_OK;
}

// ---------------------------------------------------------------------------------------------
// unit-level parity seams
// ---------------------------------------------------------------------------------------------
extern "C" int rtx_camera_rays(rtx_scene* sc, const rtx_camera* cam, const rtx_render_params* p, size_t n, const uint32_t* pixel_index,
                               const uint32_t* sample, float* origin_xyz, float* dir_xyz, uint32_t* rng_state4)
{
    if (!sc || !cam || (n && (!pixel_index || !sample || !origin_xyz || !dir_xyz))) return fail(RTX_E_INVALID, "null argument");
    FrameGeom g;
    uint32_t  rows;
    int       rc = check_params(p, &g, &rows);
    if (rc != RTX_OK) return rc;
    for (size_t i = 0; i < n; i++)
        if (pixel_index[i] >= (uint64_t)p->width * p->height || sample[i] >= p->spp) return fail(RTX_E_INVALID, "path %zu out of range", i);
    if (n == 0) return RTX_OK;
    HIPCHECK(hipSetDevice(sc->device));
    DevBuf bp, bs, bo, bd, bst;
    HIPCHECK(bp.alloc(n * 4));
    HIPCHECK(bs.alloc(n * 4));
    HIPCHECK(bo.alloc(n * 12));
    HIPCHECK(bd.alloc(n * 12));
    HIPCHECK(bst.alloc(n * 16));
    HIPCHECK(hipMemcpy(bp.p, pixel_index, n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(bs.p, sample, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_camera_rays, dim3(blocks_for(n, 64)), dim3(64), 0, 0, *cam, g, (uint64_t)n, bp.as<uint32_t>(), bs.as<uint32_t>(), bo.as<float>(),
                       bd.as<float>(), bst.as<uint32_t>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(origin_xyz, bo.p, n * 12, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(dir_xyz, bd.p, n * 12, hipMemcpyDeviceToHost));
    if (rng_state4) HIPCHECK(hipMemcpy(rng_state4, bst.p, n * 16, hipMemcpyDeviceToHost));
    return RTX_OK;
}

extern "C" int rtx_rng_unilateral(rtx_scene* sc, uint64_t seed_lo, uint64_t seed_hi, size_t n, float* out)
{
    if (!sc || (n && !out)) return fail(RTX_E_INVALID, "null argument");
    if (n > (1u << 20)) return fail(RTX_E_LIMIT, "at most 2^20 draws");
    if (n == 0) return RTX_OK;
    HIPCHECK(hipSetDevice(sc->device));
    DevBuf b;
    HIPCHECK(b.alloc(n * 4));
    hipLaunchKernelGGL(k_rng, dim3(1), dim3(64), 0, 0, seed_lo, seed_hi, (uint64_t)n, b.as<float>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(out, b.p, n * 4, hipMemcpyDeviceToHost));
    return RTX_OK;
}

extern "C" int rtx_material_bounce(rtx_scene* sc, uint32_t mat_index, size_t n, const float* ray_dir_xyz, const float* hit_position_xyz,
                                   const float* hit_normal_xyz, const float* hit_uv, const uint8_t* front_face, uint32_t* rng_state4,
                                   float* surface_rgb, float* emitted_rgb, float* out_origin_xyz, float* out_dir_xyz, uint8_t* cont)
{
    if (!sc || !ray_dir_xyz || !hit_position_xyz || !hit_normal_xyz || !hit_uv || !front_face || !rng_state4 || !surface_rgb || !emitted_rgb
        || !out_origin_xyz || !out_dir_xyz || !cont)
        return fail(RTX_E_INVALID, "null argument");
    if (mat_index >= sc->dev.n_materials) return fail(RTX_E_INVALID, "material index out of range");
    if (n == 0) return RTX_OK;
    HIPCHECK(hipSetDevice(sc->device));
    DevBuf bdir, bpos, bn, buv, bf, bst, bs, be, boo, bod, bc;
    HIPCHECK(bdir.alloc(n * 12));
    HIPCHECK(bpos.alloc(n * 12));
    HIPCHECK(bn.alloc(n * 12));
    HIPCHECK(buv.alloc(n * 8));
    HIPCHECK(bf.alloc(n));
    HIPCHECK(bst.alloc(n * 16));
    HIPCHECK(bs.alloc(n * 12));
    HIPCHECK(be.alloc(n * 12));
    HIPCHECK(boo.alloc(n * 12));
    HIPCHECK(bod.alloc(n * 12));
    HIPCHECK(bc.alloc(n));
    HIPCHECK(hipMemcpy(bdir.p, ray_dir_xyz, n * 12, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(bpos.p, hit_position_xyz, n * 12, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(bn.p, hit_normal_xyz, n * 12, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(buv.p, hit_uv, n * 8, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(bf.p, front_face, n, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(bst.p, rng_state4, n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_bounce, dim3(blocks_for(n, 64)), dim3(64), 0, 0, sc->dev, mat_index, (uint64_t)n, bdir.as<float>(), bpos.as<float>(),
                       bn.as<float>(), buv.as<float>(), bf.as<uint8_t>(), bst.as<uint32_t>(), bs.as<float>(), be.as<float>(), boo.as<float>(),
                       bod.as<float>(), bc.as<uint8_t>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(surface_rgb, bs.p, n * 12, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(emitted_rgb, be.p, n * 12, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(out_origin_xyz, boo.p, n * 12, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(out_dir_xyz, bod.p, n * 12, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(cont, bc.p, n, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(rng_state4, bst.p, n * 16, hipMemcpyDeviceToHost));
    return RTX_OK;
}

extern "C" int rtx_skybox_color(rtx_scene* sc, size_t n, const float* dir_xyz, float* rgb)
{
    if (!sc || (n && (!dir_xyz || !rgb))) return fail(RTX_E_INVALID, "null argument");
    if (n == 0) return RTX_OK;
    HIPCHECK(hipSetDevice(sc->device));
    DevBuf bd, bc;
    HIPCHECK(bd.alloc(n * 12));
    HIPCHECK(bc.alloc(n * 12));
    HIPCHECK(hipMemcpy(bd.p, dir_xyz, n * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sky, dim3(blocks_for(n, 64)), dim3(64), 0, 0, sc->dev, (uint64_t)n, bd.as<float>(), bc.as<float>());
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(rgb, bc.p, n * 12, hipMemcpyDeviceToHost));
    return RTX_OK;
}

#include "kd_build.h"
