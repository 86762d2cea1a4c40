/*
 * volly_oracle.h -- CPU ORACLE (test infrastructure, not product code).
 *
 * Scalar restatement of the reference's per-pixel ray traversal
 * (dylan-mcleod/Volume-Renderer main.cpp:81-291) used ONLY as the checker in
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in
 * the product path (volume-renderer_amd/) links, loads or calls it.
 *
 * PARITY UNPINNED at the GLM boundary: the reference has no tests, golden
 * vectors or fixtures, and cannot be built here (GLM, SDL2 headers absent), so
 * this restatement is pinned only by the reference's source text and by
 * hand-derived known-answer cases (tests/test_oracle_kat.py).  The one part of
 * the reference that does compile stand-alone, triangleCube.c, is built into
 * oracle/_ref and checked against the host voxelizer's restatement of it.
 * GLM semantics are pinned to GLM's pure (non-SIMD) IEEE code paths.
 */
#ifndef VOLLY_ORACLE_H
#define VOLLY_ORACLE_H

#include <stdint.h>
#include "../include/volly_gpu.h"   /* POD descriptors only (frame / dense / svo) */

#ifdef __cplusplus
extern "C" {
#endif

/* RaycastReport<Voxel> (main.cpp:81-86) plus the build-defined by-products. */
typedef struct {
    int32_t  found;
    float    pt_end[3];     /* ptEnd.xyz = `cur` at the hit or at the exit   */
    uint8_t  vox[8];        /* dataAt                                        */
    uint32_t hit_index;     /* x + sx*(y + sy*z) (uint32), VOLLY_MISS_INDEX  */
    uint32_t steps;         /* executions of the sample at main.cpp:137      */
    int32_t  capped;        /* 1 if the iteration cap ended the loop         */
} vo_report;

/* raycast_naiive_vec4<solid> (main.cpp:88-161) */
void vo_raycast_dense(const float start[3], const float dir[3], const volly_dense_desc *grid,
                      int solid, vo_report *out);

/* Build-defined hierarchical DDA over the flattened octree (no reference
 * implementation exists: svo.cpp:14-19 is an empty constructor).  Each
 * iteration is the dense step of main.cpp:123-153 with the cell size 2^L of the
 * largest empty octree cell around the sample point. */
void vo_raycast_svo(const float start[3], const float dir[3], const volly_svo_desc *svo,
                    vo_report *out);

/* raytrace_naiive (main.cpp:170-194; dead code in the reference, its call site is commented
 * out at main.cpp:274-278).  start xyz, direction xyzw (w takes part in reflect / refract /
 * dot exactly as glm::vec4 does).  out = colour xyzw before the clamp of main.cpp:274. */
void vo_raytrace(int iteration, const float start[3], const float direction[4], const volly_dense_desc *grid,
                 float out[4], vo_report *primary);

/* raycast_beamOpt (main.cpp:163-166); target is dense or svo (exactly one non-NULL). */
void vo_raycast_beam(const float start_lo[3], const float dir_hi[3], const float dir_lo[3],
                     const float scale[3], const volly_dense_desc *low,
                     const volly_dense_desc *target, const volly_svo_desc *target_svo,
                     vo_report *lo_out, vo_report *hi_out);

/* State::raycastOntoScreen pixel loop (main.cpp:251-287) over frame->x0..,y0..
 * Planes as in volly_outputs (host pointers, any may be NULL), one frame.
 * n_threads <= 1: the reference's serial order.  n_threads > 1: rows rendered
 * in parallel after the per-row start vectors were accumulated serially
 * (bit-identical results; not the reference's timing). */
int vo_render_frame(const volly_frame_desc *frame, const volly_dense_desc *low,
                    const volly_dense_desc *target, const volly_svo_desc *target_svo,
                    const volly_outputs *outs, int n_threads);

/* Frame constants, main.cpp:201-249.  view is column major m[col][row]
 * (GLM), size_* are the grid sizes of getCurVol() / getCurVol_LowLOD(). */
void vo_frame_from_camera(const float view[16], const float cam_pos[3],
                          const int32_t size_hi[3], const int32_t size_lo[3],
                          int32_t width, int32_t height, volly_frame_desc *out);

/* glm::lookAt(eye, center, up), right handed (main.cpp:397) */
void vo_look_at(const float eye[3], const float center[3], const float up[3], float view[16]);

/* camDir from the mouse, main.cpp:393-395 */
void vo_cam_dir_from_mouse(int mouse_x, int mouse_y, int win_x, int win_y, float cam_dir[3]);

/* shading of one hit voxel, main.cpp:264-269: returns the 4 pixel bytes packed
 * little endian (byte0 = 255, byte1 = r', byte2 = g', byte3 = b'). */
uint32_t vo_shade(const uint8_t vox[8]);

/* VolumeStore::fillSphere (volume.h:95-111) into a caller buffer of sx*sy*sz*8 B */
void vo_fill_sphere(int32_t sx, int32_t sy, int32_t sz, const uint8_t fill[8], uint8_t *voxels);

/* iteration cap shared with the GPU kernels (the reference has none, main.cpp:121) */
uint32_t vo_step_cap(int32_t sx, int32_t sy, int32_t sz);

#ifdef __cplusplus
}
#endif
#endif
