/*
 * volly_host.h -- C view of the host-side scene pipeline (libvolly_host.so).
 *
 * The host pipeline is C++ (volume-renderer_amd/host/volly_scene.h mirrors the
 * reference's classes); these wrappers exist so that tests, bench.py and other
 * non-C++ callers can drive it.  Reference counterparts are cited per function.
 * Nothing here touches the GPU.
 */
#ifndef VOLLY_HOST_H
#define VOLLY_HOST_H

#include "volly_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct volly_host_scene volly_host_scene;

#define VOLLY_SCENE_DENSE 1   /* keep the dense target grid  (volumeFromVoxelMap(m, res, 0), main.cpp:32) */
#define VOLLY_SCENE_SVO   2   /* build the octree of the LoD-0 cells (SparseVoxelOctree, svo.h:38)        */

/* Vol_LOD_Set-style scene for one target resolution (main.cpp:23-50):
 * readPLYFile / OFF -> normalizePoly -> rasterizeVoxelMapFromPoly(res) ->
 * target = LoD 0, low = LoD 2 at res/4 (main.cpp:33).  NULL on error. */
volly_host_scene *volly_host_scene_from_mesh(const char *path, int32_t res, int32_t flags, int32_t n_threads);

/* VolumeStore::fillSphere (volume.h:95-111) target of res^3 filled with `fill`
 * (8 bytes), low = its (res/4)^3 any-of-4^3 reduction. */
volly_host_scene *volly_host_scene_sphere(int32_t res, const uint8_t fill[8], int32_t flags);

/* Scene from caller grids (copied).  low may be NULL. */
volly_host_scene *volly_host_scene_from_dense(const volly_dense_desc *target, const volly_dense_desc *low,
                                              int32_t flags);

int  volly_host_scene_target(const volly_host_scene *s, volly_dense_desc *out);   /* VOLLY_E_UNSUPPORTED if not kept */
int  volly_host_scene_low(const volly_host_scene *s, volly_dense_desc *out);
int  volly_host_scene_svo(const volly_host_scene *s, volly_svo_desc *out);
/* number of entries the voxel map holds at a LoD (the std::map keys with w == lod); -1 if no map */
int64_t volly_host_scene_lod_count(const volly_host_scene *s, int32_t lod);
void volly_host_scene_free(volly_host_scene *s);

/* On-disk scene cache.  The reference re-voxelizes every mesh at every start
 * (main.cpp:369-387) and its .vol readers are dead (volume.cpp:135-275); this stores what
 * volly_host_scene_from_mesh produced -- low grid, dense target and/or octree -- as one
 * little-endian blob with a checksum.  load returns NULL (and an error) for a missing,
 * truncated, corrupted or foreign file. */
int volly_host_scene_save(const volly_host_scene *s, const char *path);
volly_host_scene *volly_host_scene_load(const char *path);

/* A mesh as rasterizeVoxelMapFromPoly would see it, for volume_gpu_voxelize (volly_gpu.h):
 * readPLYFile / the OFF reader, then normalizePoly (PLYfileReader.cpp:141-171) applied
 * n_normalize times -- the reference normalises once in Vol_LOD_Set's constructor
 * (main.cpp:27) and once more at the top of every rasterize call (PLYfileReader.cpp:219),
 * so the grid built first sees 2 (what volly_host_scene_from_mesh does) and `high` sees 3.
 * The descriptor points into the handle. */
typedef struct volly_host_mesh volly_host_mesh;
volly_host_mesh *volly_host_mesh_load(const char *path, int32_t n_normalize);
int  volly_host_mesh_view(const volly_host_mesh *m, volly_mesh_desc *out);
void volly_host_mesh_free(volly_host_mesh *m);

/* The reference's default four-grid set (main.h:79: 16/32/128/256); which = 0..3 = arr[] */
typedef struct volly_host_lod_set volly_host_lod_set;
volly_host_lod_set *volly_host_lod_set_load(const char *path, int32_t sLowR, int32_t lowR, int32_t medR, int32_t highR);
int  volly_host_lod_set_grid(const volly_host_lod_set *s, int32_t which, volly_dense_desc *out);
void volly_host_lod_set_free(volly_host_lod_set *s);

/* main.cpp:201-249.  view: 16 floats column major (glm::mat4). */
void volly_host_frame_from_camera(const float view[16], const float cam_pos[3], const int32_t size_hi[3],
                                  const int32_t size_lo[3], int32_t width, int32_t height, volly_frame_desc *out);
/* glm::lookAt(eye, center, up) as used at main.cpp:397 */
void volly_host_look_at(const float eye[3], const float center[3], const float up[3], float view[16]);
/* main.cpp:393-395 */
void volly_host_cam_dir_from_mouse(int32_t mouse_x, int32_t mouse_y, int32_t win_x, int32_t win_y, float cam_dir[3]);
/* headless orbit camera (replaces the mouse / WASD camera of main.cpp:389-418) */
void volly_host_orbit_camera(int32_t k, int32_t n, float radius, float cam_pos[3], float view[16]);

/* t_c_intersection (triangleCube.c:213): 0 = INSIDE (touches), 1 = OUTSIDE.  tri = v1.xyz v2.xyz v3.xyz */
int  volly_host_tricube(const float tri[9]);

const char *volly_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
