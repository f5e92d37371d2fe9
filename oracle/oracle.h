/*
 * oracle.h — C ABI of liboracle.so, the CPU restatement of the reference's ray→scene query path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under raytracer_amd/ may include, link, load or call this;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker.
 *
 * The oracle consumes the same flattened rtx_scene_desc the GPU library consumes and restates, in
 * plain scalar C++ and in the reference's own recursive form, every function of SURVEY §8(a).
 * It is pinned against the REAL reference (oracle/_ref/ref_parity, unmodified sources, same
 * compiler and flags: -O2 -ffp-contract=off) through the fixtures in tests/golden/ — see
 * tests/test_oracle_pinned.py: hit ids / t / normals / uv bit-exact, seeded renders bit-exact.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Scene_ClosestHit for n rays.  unit_dir != 0: directions are used as given (already Ray_Make'd).
 * visits (may be NULL) receives {inner nodes, leaves, primitive tests} summed over the rays. */
int oracle_trace(const rtx_scene_desc* desc, size_t n, const float* origin_xyz, const float* dir_xyz, int unit_dir,
                 uint32_t* prim_id, float* t, float* position_xyz, float* normal_xyz, float* uv, uint8_t* front_face,
                 uint64_t* visits3, uint32_t* steps_per_ray /* inner+leaf+prim visits of each ray, or NULL */);

/* The same query with the GPU library's subtree-box culling replayed on the CPU (checker for DESIGN.md §5;
 * not part of the reference): qbox6 / info14 come from rtx_cull_data_build.  visits receives 4 counters
 * {inner, leaf, prim, culled}.  qbox6 == NULL: plain oracle_trace. */
int oracle_trace_culled(const rtx_scene_desc* desc, const uint16_t* qbox6, const float* info14, const uint8_t* node_flags /* rtx_cull_node_flags, or NULL */, size_t n,
                        const float* origin_xyz, const float* dir_xyz, int unit_dir, uint32_t* prim_id, float* t,
                        float* position_xyz, float* normal_xyz, float* uv, uint8_t* front_face, uint64_t* visits,
                        uint32_t* steps_per_ray);

/* The speculative replay: checker of the GPU library's k_trace_long (not part of the reference).  The walk goes
 * through every subtree the sequential walk could reach, whatever an earlier leaf found, and answers with the accepting leaf
 * that comes first in the literal visiting order (qbox6 == NULL: without the culling).  Must equal oracle_trace on every ray. */
int oracle_trace_speculative(const rtx_scene_desc* desc, const uint16_t* qbox6, const float* info14, const uint8_t* node_flags, size_t n,
                             const float* origin_xyz, const float* dir_xyz, int unit_dir, uint32_t* prim_id, float* t, float* position_xyz,
                             float* normal_xyz, float* uv, uint8_t* front_face, uint32_t* steps_per_ray);

/* Seeded RenderPixel over the frame partition described by params (same meaning as rtx_render).
 * cum_out: W*H*3 f32 sums (rows not assigned are left untouched); bgr_out: W*H*3 u8 or NULL.
 * rays_at_depth: RTX_MAX_DEPTH counters or NULL.  n_threads <= 0: one thread. */
int oracle_render(const rtx_scene_desc* desc, const rtx_camera* cam, const rtx_render_params* params, float* cum_out,
                  uint8_t* bgr_out, uint64_t* rays_at_depth, int n_threads);

/* The same for a list of pixels (y * W + x) of that frame: cum_out[3 i ..] = the f32 sum of pixel_index[i]'s samples. */
int oracle_render_pixels(const rtx_scene_desc* desc, const rtx_camera* cam, const rtx_render_params* params, size_t n,
                         const uint32_t* pixel_index, float* cum_out, uint64_t* rays_at_depth, int n_threads);

int oracle_camera_rays(const rtx_camera* cam, const rtx_render_params* params, size_t n, const uint32_t* pixel_index,
                       const uint32_t* sample, float* origin_xyz, float* dir_xyz, uint32_t* rng_state4);

int oracle_rng_unilateral(uint64_t seed_lo, uint64_t seed_hi, size_t n, float* out);

int oracle_material_bounce(const rtx_scene_desc* desc, uint32_t mat_index, size_t n, const float* ray_dir_xyz,
                           const float* hit_position_xyz, const float* hit_normal_xyz, const float* hit_uv,
                           const uint8_t* front_face, uint32_t* rng_state4, float* surface_rgb, float* emitted_rgb,
                           float* out_origin_xyz, float* out_dir_xyz, uint8_t* cont);

int oracle_skybox_color(const rtx_scene_desc* desc, size_t n, const float* dir_xyz, float* rgb);

/* RGB_FromColor(cum * (1/spp)) for n pixels: cum r,g,b f32 → b,g,r u8 */
int oracle_resolve(const float* cum, size_t n_pixels, uint32_t spp, uint8_t* bgr);

#ifdef __cplusplus
}
#endif
#endif
