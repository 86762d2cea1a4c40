/*
 * volly_gpu.h -- C-ABI of the MI355X (gfx950) primary-ray traversal path.
 *
 * This is the drop-in boundary for the per-frame render loop of
 * dylan-mcleod/Volume-Renderer.  Nothing like it exists in the reference (it is
 * one C++ translation unit driving everything through C++ objects); every entry
 * point below names the reference code it replaces as file:line into the
 * reference tree.
 *
 * Conventions
 *   - plain C types only; inputs are borrowed for the duration of the call,
 *     outputs are caller allocated;
 *   - every function returns 0 on success or a negative VOLLY_E_* code, and
 *     leaves a message retrievable with volly_gpu_last_error() (thread local);
 *   - one scene handle is bound to one GPU; calls on one handle are not
 *     re-entrant (the reference is single threaded), different handles may be
 *     driven from different host threads / processes (one per GPU).
 */
#ifndef VOLLY_GPU_H
#define VOLLY_GPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOLLY_OK            0
#define VOLLY_E_ARG        -1   /* bad argument (null pointer, bad size, bad mode)          */
#define VOLLY_E_HIP        -2   /* a HIP runtime call failed (message has hipGetErrorString) */
#define VOLLY_E_NOMEM      -3   /* device or host allocation failed                          */
#define VOLLY_E_UNSUPPORTED -4  /* e.g. beam mode on a scene uploaded without a low grid     */

#define VOLLY_MISS_INDEX 0xFFFFFFFFu

/* frame flags */
#define VOLLY_FRAME_BEAM   0    /* raycast_beamOpt, main.cpp:163-166,260 (the live path)      */
#define VOLLY_FRAME_NAIVE  1    /* raycast_naiive_vec4 from camPos_Local, main.cpp:261        */
#define VOLLY_FRAME_RAYTRACE 2  /* raytrace_naiive(3, camPos_Local, ...), main.cpp:170-194 and the
                                   commented call site main.cpp:274-278: reflect / refract recursion
                                   with raycast_naiive_vec4<true>.  Dense targets only.  rgba8 holds
                                   the traced colour; the other planes describe the primary ray.     */

typedef struct volly_gpu_scene volly_gpu_scene;   /* opaque, owns device memory */

/* Dense grid = VolumeStore<Voxel> (volume.h:56-93): sx*sy*sz voxels of 8 bytes
 * (voxel.h:13-16: r,g,b,a,nx,ny,nz,nw), x fastest: idx = x + sx*(y + sy*z)
 * (volume.h:79-81).  A voxel is empty iff a == 0 (main.cpp:140). */
typedef struct {
    int32_t     sx, sy, sz;
    const void *voxels;
} volly_dense_desc;

/* One octree node, 8 bytes.  Flattened, pointer-free form of SVO_Voxel
 * (svo.h:22-30: valid_mask, leaf_mask and the 32-bit "countour" field that the
 * reference comments it uses "as the child pointer for now").
 *   children of a node are stored contiguously, only the valid ones, in octant
 *   order; octant bit = x | y<<1 | z<<2 of the child's position inside the node;
 *   child i lives at child_offset + popcount(valid_mask & ((1<<i)-1))
 *     - in nodes[]  when bit i of leaf_mask is clear,
 *     - in leaves[] when bit i of leaf_mask is set (children of size 1 voxel).
 * nodes[0] is the root and covers [0, 2^depth)^3. */
typedef struct {
    uint32_t child_offset;
    uint8_t  valid_mask;
    uint8_t  leaf_mask;
    uint16_t pad;
} volly_svo_node;

/* SparseVoxelOctree (svo.h:33-39, svo.cpp:14-19: maxScale = ceil(log2(res))). */
typedef struct {
    uint32_t              n_nodes;
    uint32_t              n_leaves;
    int32_t               depth;     /* maxScale: root cell is 2^depth voxels wide          */
    int32_t               res;       /* grid is res^3, res <= 2^depth                        */
    const volly_svo_node *nodes;     /* n_nodes  * 8 B                                       */
    const void           *leaves;    /* n_leaves * 8 B, Voxel payload (voxel.h:13-16)        */
} volly_svo_desc;

/* Per-frame constants, exactly the values State::raycastOntoScreen computes once
 * per frame before its pixel loop (main.cpp:201-249).  They are computed on the
 * host (volly_host: volly_frame_from_camera) and cross the boundary as values. */
typedef struct {
    float   row0_hi[4], xdiff_hi[4], ydiff_hi[4];   /* curYLoop, xdiff, ydiff          main.cpp:219-221,232 */
    float   row0_lo[4], xdiff_lo[4], ydiff_lo[4];   /* curYLooplowp, xdifflowp, ydifflowp  main.cpp:241-244 */
    float   origin_lo[4];                           /* camPos_Local_lowP               main.cpp:247-249 */
    float   origin_hi[4];                           /* camPos_Local                    main.cpp:228-230 */
    float   scale_lo_to_hi[4];                      /* scaleLowptoHighp                main.cpp:239     */
    int32_t width, height;                          /* renderTargetSizeX/Y             main.h:112       */
    int32_t x0, y0, tile_w, tile_h;                 /* sub-rectangle rendered by this call (multi-GPU row bands) */
    int32_t flags;                                  /* VOLLY_FRAME_BEAM (0: raycast_beamOpt, main.cpp:260), VOLLY_FRAME_NAIVE (traversal from the camera,
                                                       the commented call at main.cpp:261) or VOLLY_FRAME_RAYTRACE (raytrace_naiive, main.cpp:170-194 with the
                                                       commented call site main.cpp:274-278; dense targets only, pixel = the blended colour) */
    int32_t reserved;
} volly_frame_desc;

/* Output planes.  Each plane holds n_frames * tile_w * tile_h elements, frame
 * major, then row major inside the sub-rectangle: (x-x0) + (y-y0)*tile_w, which
 * for a full frame is the reference's data_stream[x + y*W] (main.cpp:269).
 * Any pointer may be NULL (plane not wanted).
 *   rgba8     bytes in memory [255, r', g', b'] = glm::u8vec4(255, r*L, g*L, b*L)  main.cpp:269-271
 *   hit_index x + sx*(y + sy*z) of the hit voxel (uint32 arithmetic), VOLLY_MISS_INDEX on a miss
 *   steps_lo  number of executions of the sample at main.cpp:137 in the low-grid pass (0 in NAIVE mode)
 *   steps_hi  same for the target-grid pass; for an SVO target: number of traversal iterations
 *   depth     dot(ptEnd - origin_hi, direction), evaluated (dx*vx + dy*vy) + dz*vz unfused; +inf on a miss
 *   pt_end    THREE floats per ray (x, y, z interleaved): RaycastReport::ptEnd of the target-grid pass
 *             (main.cpp:81-86), the DDA position `cur` at the hit (main.cpp:143) or where the ray
 *             stopped (main.cpp:129), in target-grid cell units */
typedef struct {
    uint32_t *rgba8;
    uint32_t *hit_index;
    uint32_t *steps_lo;
    uint32_t *steps_hi;
    float    *depth;
    float    *pt_end;
} volly_outputs;

/* Upload the target grid (State::getCurVol(), main.h:118) and the beam grid
 * (State::getCurVol_LowLOD() = Vol_LOD_Set::arr[1], main.h:119) once, after
 * State::initAllVolumes (main.cpp:369-387).  `low` may be NULL: the scene then
 * only renders VOLLY_FRAME_NAIVE frames. */
int volume_gpu_upload(int device, const volly_dense_desc *target, const volly_dense_desc *low,
                      volly_gpu_scene **out);

/* Same with the target held as an octree: the GPU-resident form of
 * SparseVoxelOctree (svo.h:33-39).  `low` as above. */
int svo_gpu_upload(int device, const volly_svo_desc *svo, const volly_dense_desc *low,
                   volly_gpu_scene **out);

/* Synthetic dense scene generated on the device (no reference counterpart; it
 * plays the role of VolumeStore::fillSphere, volume.h:95-111, "so we can test
 * without access to files"): voxel (x,y,z) is filled iff
 * volly_noise_hash(x,y,z,seed) < threshold, rgba = 255, norm = hash bits.
 * The low grid (res/4)^3 marks a cell filled iff any of its 4^3 voxels is.
 * The exact hash is volly_noise_hash() below. */
int volume_gpu_generate_noise(int device, int32_t res, uint32_t seed, uint32_t threshold,
                              volly_gpu_scene **out);

/* Replaces the pixel loop of State::raycastOntoScreen (main.cpp:251-287) for
 * n_frames frames (all with the same tile_w x tile_h).  Synchronous; planes in
 * `outs` are HOST pointers (e.g. Texture_SDL::data_stream, main.h:33). */
int volume_gpu_render(volly_gpu_scene *scene, const volly_frame_desc *frames, int n_frames,
                      const volly_outputs *outs);

/* Page-locked host memory for the output planes of volume_gpu_render (the role of the
 * SDL streaming texture the reference locks at main.cpp:199 and unlocks at :290).  When
 * every requested plane lies in such memory (or in memory pinned by hipHostMalloc /
 * hipHostRegister) the call renders in batches and copies each batch back while the next
 * one is traversed; pageable memory works too, through the runtime's staged copy. */
void *volly_gpu_host_alloc(size_t bytes);
void  volly_gpu_host_free(void *p);

/* Same, planes in `outs` are DEVICE pointers on the scene's GPU; work is queued
 * on `hip_stream` (a hipStream_t, NULL = default stream) and the call returns
 * without waiting.  `frames` is consumed before the call returns.
 * A scene's per-call scratch is shared by its calls: calls on ONE stream follow each other; a call
 * on a DIFFERENT stream than the previous call on this scene is made to wait (on the device) for
 * that previous call's kernels (the previous stream may have been destroyed since).  Calls on one scene from
 * several host threads at once are not supported (the handle is not re-entrant). */
int volume_gpu_render_device(volly_gpu_scene *scene, const volly_frame_desc *frames, int n_frames,
                             const volly_outputs *outs, void *hip_stream);

/* volume_gpu_render_device for callers that cannot allocate.  It has NO counterpart in the
 * reference (which never allocates per frame: main.cpp:197-291 works in the locked texture); it is
 * for callers that cannot call hipMalloc: code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  Same arguments, same results bit for bit; it calls no
 * hipMalloc / hipFree / allocating library routine and takes its scratch (frame descriptors, sky
 * spans and direction checkpoints) from a 33 MiB + 4 KiB array that lives in the library's device
 * image -- each GPU the library runs on holds its copy until the process ends.
 * Limit: one sub-rectangle's checkpoints, 6 * ceil(tile_w / 8) * tile_h floats, must fit 32 MiB
 * (up to ~11 Mpixel: a 3840x2160 frame fits, a 7680x4320 frame needs row bands of <= 1456 rows),
 * and tile_h <= 262144 (the spans' 1 MiB); larger ones are rejected with VOLLY_E_ARG.  Any number of frames: they are rendered in launches
 * of as many as fit.  Kernel timing (volly_gpu_set_profiling) is not recorded for these calls.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (different streams,
 * threads or scenes) race; calls queued on one stream do not. */
int volume_gpu_render_device_noalloc(volly_gpu_scene *scene, const volly_frame_desc *frames, int n_frames,
                                     const volly_outputs *outs, void *hip_stream);

/* Scene preparation on the device: the voxelization of Vol_LOD_Set's constructor
 * (main.cpp:23-50) for one target resolution -- rasterizeVoxelMapFromPoly
 * (PLYfileReader.cpp:218-321) with t_c_intersection (triangleCube.c:213-319),
 * volumeFromVoxelMap (volume.h:114-123) for the target (LoD 0, res^3) and the low
 * beam grid (LoD 2, (res/4)^3, main.cpp:33), and the octree builder -- producing a
 * scene ready to render with no voxel ever crossing PCIe.  `mesh` is the polyhedron
 * AFTER normalizePoly (PLYfileReader.cpp:141-171): n_verts x (x,y,z,w) floats and
 * n_tris x 3 vertex indices (volly_host_mesh_* in volly_host.h prepares it the
 * way the host pipeline would).  flags: VOLLY_VOX_DENSE and/or VOLLY_VOX_SVO
 * (rendering uses the octree when both are present).  res: multiple of 4 in
 * [8, 4096].  The grids and the octree equal the host pipeline's byte for byte
 * (tests/test_gpu_voxelize.py); the reference's per-triangle flood fill runs as
 * one breadth-first search over all triangles at once.
 * stats (optional, 8 values): triangles used, cells tested, (cell, triangle)
 * pairs that passed, search levels, distinct LoD-0 cells, microseconds spent in the
 * call, peak bytes of device scratch, 0. */
typedef struct volly_mesh_desc {
    int64_t n_verts;
    const float *verts;             /* n_verts * 4 floats */
    int64_t n_tris;
    const int32_t *tris;            /* n_tris * 3 vertex indices */
} volly_mesh_desc;

#define VOLLY_VOX_DENSE 1
#define VOLLY_VOX_SVO   2

int volume_gpu_voxelize(int device, const volly_mesh_desc *mesh, int32_t res, int32_t flags,
                        volly_gpu_scene **out, int64_t *stats);

/* The whole of Vol_LOD_Set's constructor (main.cpp:23-50, default resolutions 16/32/128/256,
 * main.h:79) on the device: out[2] renders arr[2] = LoD 0 of the rasterisation at medR,
 * out[1] arr[1] = its LoD 2 stored at lowR, out[0] arr[0] = its LoD 3 stored at sLowR,
 * out[3] arr[3] = LoD 0 of a second rasterisation at highR (NULL when highR <= 0) -- each
 * with arr[1] as its beam grid, the pair State::raycastOntoScreen takes for curLoD = i
 * (main.h:118-119).  mesh_med is the polyhedron as the first rasterize call sees it (two
 * normalizePoly passes), mesh_high as the second does (three): volly_host_mesh_load(path, 2 / 3).
 * Dense grids, byte-identical to the host's Vol_LOD_Set (tests/test_gpu_voxelize.py). */
int volume_gpu_voxelize_lod_set(int device, const volly_mesh_desc *mesh_med, const volly_mesh_desc *mesh_high,
                                int32_t sLowR, int32_t lowR, int32_t medR, int32_t highR,
                                volly_gpu_scene *out[4]);

/* t_c_intersection (triangleCube.c:213-319) as the device voxelizer runs it, on its own: n
 * triangles of 9 floats each (three vertices relative to the centre of a unit cube), out[i] = 0
 * (INSIDE: the triangle touches the cube) or 1 (OUTSIDE), the reference's return values
 * (triangleCube.c:28-29).  Lets a test hold the device code against the reference's own function. */
int volly_gpu_tricube(int device, const float *tri9, int64_t n, int32_t *out);

/* Kernel timing with HIP events on the stream the kernels are launched on.
 * While enabled every traversal-kernel launch is bracketed by an event pair;
 * volly_gpu_kernel_time waits for them, returns the summed milliseconds and the
 * number of launches since the last query, and forgets them. */
int volly_gpu_set_profiling(volly_gpu_scene *scene, int enabled);
int volly_gpu_kernel_time(volly_gpu_scene *scene, double *ms_total, int *launches);

/* Copy `count` voxels (8 B each) of the target dense grid starting at linear
 * index `first` back to the host (checks of generated / uploaded scenes). */
int volly_gpu_read_voxels(volly_gpu_scene *scene, uint64_t first, uint64_t count, void *out);

/* Same for the low (beam) grid, and for the octree arrays (either pointer may be
 * null; sizes from volly_gpu_scene_info 10 / 11). */
int volly_gpu_read_low_voxels(volly_gpu_scene *scene, uint64_t first, uint64_t count, void *out);
int volly_gpu_read_svo(volly_gpu_scene *scene, volly_svo_node *nodes, void *leaves);

/* Scene facts: 0 sx, 1 sy, 2 sz, 3 is_svo, 4 has_low, 5 low sx, 6 device,
 * 7 device bytes held (MiB), 8 frame kernel (0 = one tile per wavefront, 3 = blocks of four tiles whose rays are packed into full wavefronts between rounds: dense targets without the region level),
 * 9 target grid uses the 16^3 region level, 10 octree nodes, 11 octree leaves,
 * 12 octree depth, 13 a dense target grid is held,
 * 14 the DDA loop the TARGET grid is traversed with: 0 = octree (not a dense traversal), 1 = the assembly
 *    loop over 16^3 regions and 4^3 bricks, 2 = the assembly loop over bricks only (grids whose regions
 *    nearly all hold voxels), 3 = the fully checked compiled loop -- taken by every grid with an axis
 *    larger than 1024 (the assembly loops pack a cell into 3 x 10 bits), measured ~16 % slower; sides that
 *    are no multiple of 16 cost nothing: the occupancy levels cover the grid padded to whole 16^3 regions
 *    and a ray that samples a cell of the padding has left the grid there; wavefronts holding a
 *    ray with a zero / denormal / NaN direction component take loop 3 whatever the grid,
 * 15 the same for the LOW (beam) grid, 0 = no low grid,
 * 16 octree targets: 1 = traversed through the flattened tree (direct-index table down to 8^3 cells + occupancy
 *    blocks of the lowest three levels, trace_svo_blocks), 0 = through the node chain below the table (trees of
 *    fewer than 3 or more than 11 levels, res not a multiple of 8, nodes not in breadth-first order, stored nodes
 *    without children; wavefronts holding an irregular ray take the node chain whatever the tree),
 * 17 dense targets: voxels with a > 0 as counted at upload (the packed kernel is chosen by this filling),
 * 18 octree targets traversed through the flattened tree: 1 = a hit's leaf is found by its rank among the voxels of its
 *    occupancy word (the tree stores the leaves under each 4^3 cell one after the other in tree order -- octant by
 *    octant, breadth-first builders do), 0 = by walking down from the level-3 node (three more dependent loads). */
int64_t volly_gpu_scene_info(const volly_gpu_scene *scene, int what);

/* Sky tiles (no reference counterpart: the reference traces every pixel, main.cpp:252-287).  For one
 * frame of this scene: the pixel rectangle {x_lo, x_hi, y_lo, y_hi} -- inclusive, relative to the
 * frame's sub-rectangle -- OUTSIDE of which every ray provably ends both passes at its first bounds
 * test (0 steps, no hit, depth +inf, pixel 255,0,0,0).  8x8 tiles that lie wholly outside it write
 * those values without setting their rays up, unless pt_end is requested.  The rectangle is the
 * bounding box of the projected corners of both grids, widened by one cell and by four pixels plus three times
 * the bound on the drift of the reference's serial direction sums (it grows with the frame: 3 px more at 4K, 10 at
 * 7680x4320); it is
 * the whole sub-rectangle (no cull) when a corner is beside or behind the camera, for
 * VOLLY_FRAME_RAYTRACE frames, and in beam mode when the frame's low and target constants do not
 * describe the same rays.  x_lo > x_hi or y_lo > y_hi: nothing can be hit.  VOLLY_SKY_CULL=0 in the
 * environment at scene creation switches the shortcut off. */
int volly_gpu_sky_rect(const volly_gpu_scene *scene, const volly_frame_desc *frame, int32_t rect[4]);
/* The same computation from the two grids' sizes alone (host arithmetic only: needs no scene and no
 * GPU, which lets a CPU test hold it against the oracle).  size_lo may be NULL for naive frames. */
int volly_gpu_sky_rect_for(const int32_t size_hi[3], const int32_t size_lo[3], const volly_frame_desc *frame, int32_t rect[4]);

/* The spans the frame kernel culls with: for tile row r (rows r*block_h .. r*block_h+block_h-1 of the sub-rectangle;
 * block_h = 8 for frames with a beam pass, 2 for launches of naive frames only) spans[2r], spans[2r+1] = the first
 * and last column (relative to the sub-rectangle) that may hold a ray entering a grid; first > last: none.  They are
 * the rectangle above cut by up to five edges of the projected boxes' convex hull (hull_out, optional: 5 x {a, b, c},
 * a*X + b*Y + c <= 0 in absolute pixels; unused entries 0, 0, -1).  Host arithmetic -- the same function the
 * pre-pass kernel runs per tile row.  No reference counterpart. */
int volly_gpu_sky_spans_for(const int32_t size_hi[3], const int32_t size_lo[3], const volly_frame_desc *frame, int32_t block_h,
                            int32_t *spans, float *hull_out);
void        volly_gpu_free(volly_gpu_scene *scene);
const char *volly_gpu_last_error(void);
int         volly_gpu_abi_version(void);
/* What this binary was built from: "src_sha16=<first 16 hex digits of the SHA-256 of volly_gpu.hip, volly_voxelize.hip,
 * volly_kernels.h, volly_internal.h, volly_gpu.h, concatenated in that order>; hipcc=<compiler version>; ...".  The
 * hand-written loops keep state in fixed registers across asm statements, which tools/check_kernel_contract.py verifies
 * on the compiler's output at build time only; this string lets a prebuilt library be told from one that matches the
 * tree (tests/test_abi.py does).  No reference counterpart. */
const char *volly_gpu_build_info(void);

/* The integer mix behind volume_gpu_generate_noise, usable on host and device. */
#if defined(__HIPCC__)
#define VOLLY_HD __host__ __device__
#else
#define VOLLY_HD
#endif
VOLLY_HD static inline uint32_t volly_noise_hash(uint32_t x, uint32_t y, uint32_t z, uint32_t seed)
{
    uint32_t h = seed * 0x9E3779B1u + x * 0x85EBCA77u;
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h += y * 0xC2B2AE3Du;
    h ^= h >> 13; h *= 0x297A2D39u;
    h += z * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x165667B1u;
    h ^= h >> 15;
    return h;
}

#ifdef __cplusplus
}
#endif
#endif /* VOLLY_GPU_H */
