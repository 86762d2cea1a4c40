// volly_gpu.hip -- C-ABI of the MI355X traversal path (include/volly_gpu.h).
// Built for gfx950 only, with -ffp-contract=off (see volly_kernels.h).
#include "volly_internal.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace vimpl {
thread_local std::string g_err;
}

// ---- scratch of volume_gpu_render_device_noalloc ------------------------------------------
// The entry point for callers that cannot allocate keeps what volume_gpu_render_device takes from
// hipMalloc -- the frame descriptors of one launch and its direction checkpoints -- in this array.
// Every GPU the library runs on holds its own copy (at its own address) until the process ends;
// host code never uses the array's name as an address, only what hipGetSymbolAddress returned for
// the device in question (noalloc_scratch below).
namespace {
constexpr size_t kNoAllocDescBytes = 4096;                 // vk::kInlineFrames descriptors of 176 B, padded
constexpr size_t kNoAllocSpans = (size_t)128 << 10;        // 1 MiB of sky spans: one int2 per frame and tile row of a launch
constexpr size_t kNoAllocCkptFloats = (size_t)8 << 20;     // 32 MiB of direction checkpoints: 6 * ceil(tile_w / 8) * tile_h floats per frame
constexpr int kMaxDevices = 64;
}
static __device__ __align__(256) unsigned char scratch[kNoAllocDescBytes + kNoAllocSpans * sizeof(int2) + kNoAllocCkptFloats * sizeof(float)];

using namespace vimpl;

namespace {
constexpr size_t kDirectHostBytes = (size_t)4 << 20;   // volume_gpu_render: calls of at most this many output bytes store straight into pinned host memory
constexpr int kFrameChunk = 64;     // most frames per launch (larger launches run closer to the kernel's steady rate: DESIGN.md section 5a)
static_assert(sizeof(volly_frame_desc) * vk::kInlineFrames <= kNoAllocDescBytes, "descriptor part of the static scratch");

std::mutex g_scratch_mutex;
void *g_scratch_ptr[kMaxDevices] = {};

// device address of `scratch` on `device`, which must be the current device; resolved once per device
void *noalloc_scratch(int device)
{
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        g_scratch_ptr[device] = p;
    }
    return g_scratch_ptr[device];
}
}

namespace vimpl {

int dev_alloc(volly_gpu_scene *s, void **p, size_t bytes)
{
    HIP_TRY(hipMalloc(p, bytes ? bytes : 8));
    s->owned.push_back(*p);
    s->bytes += bytes;
    return VOLLY_OK;
}

void adopt(volly_gpu_scene *s, void *p, size_t bytes)
{
    s->owned.push_back(p);
    s->bytes += bytes;
}

} // namespace vimpl

namespace {

int check_dense(const volly_dense_desc *d, const char *what)
{
    if (!d || !d->voxels) return fail(VOLLY_E_ARG, std::string(what) + ": null grid");
    if (d->sx < 1 || d->sy < 1 || d->sz < 1) return fail(VOLLY_E_ARG, std::string(what) + ": non-positive size");
    if ((uint64_t)d->sx * (uint64_t)d->sy * (uint64_t)d->sz > (1ull << 31))
        return fail(VOLLY_E_ARG, std::string(what) + ": more than 2^31 voxels (the reference indexes with int, volume.h:79-81)");
    return VOLLY_OK;
}

} // namespace

namespace vimpl {

// upload (or adopt) the 8-byte voxels of one grid and derive its occupancy bricks
int make_dense(volly_gpu_scene *s, int sx, int sy, int sz, const void *host_vox, uint2 *dev_vox, vk::DenseDev *out)
{
    const size_t n = (size_t)sx * sy * sz;
    uint2 *vox = dev_vox;
    if (!vox) {
        int rc = dev_alloc(s, (void **)&vox, n * 8);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(vox, host_vox, n * 8, hipMemcpyHostToDevice));
    }
    vk::DenseDev d{};
    d.vox = vox;
    d.sx = sx; d.sy = sy; d.sz = sz;
    d.px = (sx + 15) & ~15; d.py = (sy + 15) & ~15; d.pz = (sz + 15) & ~15;      // whole 16^3 regions; the rim's cells are set
    d.nbx = d.px / 4; d.nby = d.py / 4; d.nbz = d.pz / 4;
    d.fsx = (float)sx; d.fsy = (float)sy; d.fsz = (float)sz;
    d.cap = 4u * (uint32_t)(sx + sy + sz) + 8u;
    const size_t nb = (size_t)d.nbx * d.nby * d.nbz;
    uint64_t *occ = nullptr;
    int rc = dev_alloc(s, (void **)&occ, nb * 8);
    if (rc) return rc;
    hipLaunchKernelGGL(vk::build_occupancy, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, 0, vox, sx, sy, sz, d.nbx, d.nby,
                       d.nbz, occ);
    HIP_TRY(hipGetLastError());
    d.occ = occ;
    d.nrx = d.px / 16; d.nry = d.py / 16; d.nrz = d.pz / 16;
    const size_t nr = (size_t)d.nrx * d.nry * d.nrz;
    uint64_t *occ2 = nullptr;
    if ((rc = dev_alloc(s, (void **)&occ2, nr * 8 + 16))) return rc;
    uint32_t *counter = (uint32_t *)(occ2 + nr);          // [0] regions that hold voxels; [2..3] filled voxels (64 bit)
    HIP_TRY(hipMemset(counter, 0, 16));
    hipLaunchKernelGGL(vk::build_occupancy2, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, 0, occ, d.nbx, d.nby, d.nbz, d.nrx,
                       d.nry, d.nrz, occ2, counter, (unsigned long long *)(counter + 2));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    uint32_t counts[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(counts, counter, 16, hipMemcpyDeviceToHost));
    const uint32_t nonempty = counts[0];
    d.n_filled = ((uint64_t)counts[2] | ((uint64_t)counts[3] << 32)) - ((uint64_t)d.px * d.py * d.pz - (uint64_t)n);   // less the rim
    d.occ2 = occ2;
    uint4 *occ2x = nullptr;
    if ((rc = dev_alloc(s, (void **)&occ2x, nr * 16))) return rc;
    int max_level = 7;                                    // empty cubes of up to 128^3 voxels; VOLLY_EMPTY_CUBES=4 switches them off (A/B runs)
    if (const char *e = std::getenv("VOLLY_EMPTY_CUBES")) max_level = std::min(std::max(std::atoi(e), 4), 7);
    int boxes = 1;                                        // boxes may grow further along the axes that stay empty; VOLLY_EMPTY_BOXES=0: cubes only
    if (const char *e = std::getenv("VOLLY_EMPTY_BOXES")) boxes = std::atoi(e) != 0;
    hipLaunchKernelGGL(vk::build_region_cubes, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, 0, occ2, d.nrx, d.nry, d.nrz, max_level, boxes, occ2x);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    d.occ2x = occ2x;
    // The region level pays when a fair share of the 16^3 regions is empty (surface scenes:
    // 20-55 % of the regions hold voxels in the bundled meshes, +9 % Mrays/s); in a noise
    // volume every region holds voxels and it is pure overhead (-7 %).  VOLLY_USE_REGIONS=0/1 overrides.
    d.use_regions = (uint64_t)nonempty * 10 <= (uint64_t)nr * 9;
    if (const char *e = std::getenv("VOLLY_USE_REGIONS")) d.use_regions = std::atoi(e) != 0;
    *out = d;
    return VOLLY_OK;
}

void destroy(volly_gpu_scene *s)
{
    if (!s) return;
    DeviceGuard g(s->device);
    (void)hipDeviceSynchronize();
    for (auto &e : s->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (s->scratch_event) (void)hipEventDestroy(s->scratch_event);
    for (void *p : s->owned) (void)hipFree(p);
    if (s->d_frames) (void)hipFree(s->d_frames);
    if (s->d_ckpt) (void)hipFree(s->d_ckpt);
    if (s->d_spans) (void)hipFree(s->d_spans);
    if (s->d_planes) (void)hipFree(s->d_planes);
    if (s->d_queue) (void)hipFree(s->d_queue);
    if (s->s_render) (void)hipStreamDestroy(s->s_render);
    if (s->s_copy) (void)hipStreamDestroy(s->s_copy);
    delete s;
}

int new_scene(int device, volly_gpu_scene **out)
{
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(VOLLY_E_ARG, "no such device");
    volly_gpu_scene *s = new (std::nothrow) volly_gpu_scene();
    if (!s) return fail(VOLLY_E_NOMEM, "host allocation failed");
    s->device = device;
    if (const char *e = std::getenv("VOLLY_SKY_CULL")) s->sky_cull = std::atoi(e);           // 0: every tile sets its rays up, 1: rectangle only (A/B runs)
    {   // set-up is where the static scratch of the entry point that allocates nothing is looked up
        // (once per device, with that device current); a failure is reported by that entry point only
        DeviceGuard g(device);
        if (g.ok) s->noalloc_scratch = noalloc_scratch(device);
    }
    *out = s;
    return VOLLY_OK;
}

static bool host_grid_is_regular(const vk::DenseDev &g)
{
    return g.px <= 1024 && g.py <= 1024 && g.pz <= 1024;
}

// Dense targets without the region level (every 16^3 region holds voxels: the noise volume) have rays of very uneven
// length; their frames run through vk::render_frames_packed, which packs the rays still under way into full
// wavefronts between rounds of `pack_budget` iterations.  VOLLY_PACKED=0 / 1 switches it off / on for every regular
// dense grid, VOLLY_PACK_BUDGET=n sets the round length (A/B runs).
void choose_frame_kernel(volly_gpu_scene *s)
{
    // ... and only when at least one voxel in 600 is filled: in a sparser volume most rays cross it whole, their lengths
    // are set by the geometry and a tile's rays end together anyway (p = 1/1024: the tile kernel is 2 % faster,
    // p = 1/256: the packed one 26 %, p = 1/64: 11 %; profiles/r03_ab/r03_ab_packlen_sweeps.log)
    const uint64_t cells = (uint64_t)s->target.sx * (uint64_t)s->target.sy * (uint64_t)s->target.sz;
    s->packed = !s->is_svo && host_grid_is_regular(s->target) && !s->target.use_regions && s->target.n_filled * 600 >= cells;
    if (const char *e = std::getenv("VOLLY_PACKED")) s->packed = !s->is_svo && host_grid_is_regular(s->target) && std::atoi(e) != 0;
    // round length: in a volume whose rays run long (p <= 1/128) the pooling between passes packs well enough and a
    // compaction inside the first passes only costs (96: one round per 96-iteration pass); in a denser one the rays of a
    // block end early and the rounds pay (profiles/r03_ab/r03_c3_tune4.log: p = 1/32 .. 1/512)
    s->pack_budget = s->target.n_filled * 96 <= cells ? 96 : 48;
    if (const char *e = std::getenv("VOLLY_PACK_BUDGET")) s->pack_budget = (uint32_t)std::max(std::atoi(e), 1);
    // passes (render_chunk): the rays still under way after VOLLY_PACK_PASS_LEN iterations are parked and pooled into
    // full workgroups for the next pass; VOLLY_PACK_PASSES=1: one pass to the end
    s->pack_passes = 6;
    if (const char *e = std::getenv("VOLLY_PACK_PASSES")) s->pack_passes = std::min(std::max(std::atoi(e), 1), 32);
    // passes from this one on (the first is 0) run the look-ahead loop: the passes that hold few, long rays -- the
    // fourth and later at p >= 1/128, the sixth and later in sparser volumes, whose middle passes are still full
    s->pack_lookahead_from = s->target.n_filled * 192 <= cells ? 5 : 3;
    if (const char *e = std::getenv("VOLLY_PACK_LOOKAHEAD_FROM")) s->pack_lookahead_from = std::max(std::atoi(e), 1);
    s->pack_pass_len = 96;
    if (const char *e = std::getenv("VOLLY_PACK_PASS_LEN")) s->pack_pass_len = (uint32_t)std::max(std::atoi(e), 1);
}

int finish_svo(volly_gpu_scene *s, uint2 *nodes, uint2 *leaves, uint32_t n_nodes, uint32_t n_leaves, int depth, int res,
               int64_t l3_first, int64_t l3_count)
{
    s->is_svo = true;
    s->svo.nodes = nodes; s->svo.leaves = leaves;
    s->svo.n_nodes = n_nodes;
    s->n_leaves = n_leaves;
    s->svo.depth = depth; s->svo.res = res; s->svo.fres = (float)res;
    s->svo.cap = 4u * (uint32_t)(3 * res) + 8u;
    s->target.sx = s->target.sy = s->target.sz = res;
    // direct-index table of the upper levels: cells of 2^tshift voxels.  Down to 8^3 cells (tshift = 3) for trees
    // of up to 2048^3 (256^3 entries, 64 MiB), where the occupancy blocks take over; at most 128^3 entries above that
    int tbits = std::max(depth - 3, 0);
    if (tbits > 8) tbits = 7;
    if (const char *e = std::getenv("VOLLY_SVO_TABLE_BITS")) tbits = std::min(std::max(std::atoi(e), 0), std::min(depth - 1, 8));
    s->svo.tbits = tbits;
    s->svo.tshift = depth - tbits;
    const size_t nt = (size_t)1 << (3 * tbits);
    uint32_t *table = nullptr;
    int rc = dev_alloc(s, (void **)&table, nt * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(vk::build_svo_table, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, 0, nodes, depth, tbits, s->svo.tshift, table);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(VOLLY_E_HIP, "build_svo_table failed");
    s->svo.table = table;
    // occupancy blocks of the lowest three levels (vk::build_svo_blocks): needs the level-3 nodes as one contiguous
    // index range (breadth-first trees have that; the caller says where), table cells of 8^3 and whole 8^3 cells
    // inside the grid.  VOLLY_SVO_BLOCKS=0 keeps the node-chain traversal (A/B runs).
    s->svo.blocks = nullptr; s->svo.word_base = nullptr; s->svo.l3_first = 0; s->svo.use_blocks = 0; s->svo.use_rank = 0;
    bool want = depth >= 3 && s->svo.tshift == 3 && (res & 7) == 0 && l3_first >= 0 && l3_count > 0 && l3_first + l3_count <= (int64_t)n_nodes;
    if (const char *e = std::getenv("VOLLY_SVO_BLOCKS")) want = want && std::atoi(e) != 0;
    if (want) {
        uint64_t *blocks = nullptr;
        if ((rc = dev_alloc(s, (void **)&blocks, (size_t)l3_count * 64 + 8))) return rc;
        uint32_t *word_base = nullptr;
        if ((rc = dev_alloc(s, (void **)&word_base, (size_t)l3_count * 32))) return rc;
        uint32_t *bad = (uint32_t *)(blocks + (size_t)l3_count * 8);
        HIP_TRY(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(vk::build_svo_blocks, dim3((unsigned)((l3_count + 255) / 256)), dim3(256), 0, 0, nodes, (uint32_t)l3_first, (uint32_t)l3_count, blocks,
                           word_base, bad);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(VOLLY_E_HIP, "build_svo_blocks failed");
        uint32_t is_bad[2] = {1, 1};
        HIP_TRY(hipMemcpy(is_bad, bad, 8, hipMemcpyDeviceToHost));
        s->svo.blocks = blocks;
        s->svo.word_base = word_base;
        s->svo.l3_first = (uint32_t)l3_first;
        s->svo.use_blocks = is_bad[0] ? 0 : 1;   // a stored node without children: the node chain decides (see build_svo_blocks)
        s->svo.use_rank = is_bad[1] ? 0 : 1;     // leaves not in tree order under some word: a hit walks down to its leaf
        if (const char *e = std::getenv("VOLLY_SVO_LEAF_RANK")) s->svo.use_rank = s->svo.use_rank && std::atoi(e) != 0;      // (A/B runs)
    }
    return VOLLY_OK;
}

} // namespace vimpl

namespace {

int validate_frames(const volly_gpu_scene *s, const volly_frame_desc *frames, int n_frames)
{
    if (!s || !frames || n_frames < 1) return fail(VOLLY_E_ARG, "render: null scene / frames or n_frames < 1");
    const volly_frame_desc &f0 = frames[0];
    for (int i = 0; i < n_frames; ++i) {
        const volly_frame_desc &f = frames[i];
        if (f.width < 1 || f.height < 1 || f.tile_w < 1 || f.tile_h < 1 || f.x0 < 0 || f.y0 < 0 ||
            (int64_t)f.x0 + f.tile_w > f.width || (int64_t)f.y0 + f.tile_h > f.height)
            return fail(VOLLY_E_ARG, "render: sub-rectangle outside the frame");
        if (f.tile_w != f0.tile_w || f.tile_h != f0.tile_h)
            return fail(VOLLY_E_ARG, "render: all frames of a call must share tile_w x tile_h");
        if (f.flags & ~(VOLLY_FRAME_NAIVE | VOLLY_FRAME_RAYTRACE)) return fail(VOLLY_E_ARG, "render: unknown frame flags");
        if ((f.flags & VOLLY_FRAME_RAYTRACE) && s->is_svo)
            return fail(VOLLY_E_UNSUPPORTED, "render: raytrace frames need a dense target (raytrace_naiive takes a VolumeStore)");
        if (!(f.flags & (VOLLY_FRAME_NAIVE | VOLLY_FRAME_RAYTRACE)) && !s->has_low)
            return fail(VOLLY_E_UNSUPPORTED, "render: beam frame on a scene uploaded without a low grid");
    }
    if ((uint64_t)f0.tile_w * (uint64_t)f0.tile_h * (uint64_t)n_frames > (1ull << 40))
        return fail(VOLLY_E_ARG, "render: too many pixels in one call");
    return VOLLY_OK;
}

// ---- sky tiles -----------------------------------------------------------------------------
// The pixel rectangle of a frame outside of which every ray provably ends BOTH passes at its first
// bounds test (main.cpp:126-135 on the first iteration: 0 steps, no hit, depth +inf, pixel 255,0,0,0),
// so that the frame kernel can write those values without setting the ray up.
//
// The serial direction sum of pixel (x, y) is row0 + y*ydiff + x*xdiff up to rounding (width + height
// additions of values in GRID units; how far that can drift, in pixels, is sky_drift_margin: 0.8 px at 4K,
// 3 px at 8K, and it grows with the frame's area), so a box corner Q
// is seen at the pixel (a/c, b/c) where Q - origin = a*xdiff + b*ydiff + c*row0, and it lies in front
// of the camera iff c > 0.  With all eight corners of a box in front, every point of the box is a
// convex combination of them with positive weights c_i, so its pixel lies in the corners' bounding
// rectangle: a ray through a pixel outside that rectangle misses the box.  Boxes are the grids
// widened by a margin (one cell beyond the reference's own (-1, size) test) and the rectangle by
// kSkyPixelMargin pixels plus three times the drift bound, which dwarf every rounding error of the reference's arithmetic for a point
// within 10^6 cells; farther points are outside any grid anyway.  Every point the reference computes
// lies on (or within rounding of) its ray or is not finite, and both kinds fail the bounds test.
// In beam mode the target pass starts on the low ray (scale * ptEnd) and runs along the TARGET
// direction sum: the two sets of frame constants must describe the same rays (checked here at the
// sub-rectangle's corners; a caller's inconsistent constants simply switch the cull off).
// Returns the rectangle relative to the sub-rectangle, inclusive; the whole sub-rectangle = no cull.
// The same holds for ANY half-plane that contains all projected corners: sky_hull_edges adds up to
// vk::kHullEdges slanted ones, and the pre-pass kernel turns rectangle and edges into one span of columns
// per tile row (vk::sky_span), which is what the frame kernel tests.
constexpr double kSkyPixelMargin = 4.0, kSkyCellMargin = 1.0;

struct SkyPass { double o[3], r0[3], xd[3], yd[3], size[3]; };

static bool sky_project(const SkyPass &p, double &xmin, double &xmax, double &ymin, double &ymax, double (*pts)[2] = nullptr)
{
    // inverse of M = [xd | yd | r0] by cofactors
    const double *A = p.xd, *B = p.yd, *C = p.r0;
    const double c0[3] = {B[1] * C[2] - B[2] * C[1], B[2] * C[0] - B[0] * C[2], B[0] * C[1] - B[1] * C[0]};   // B x C
    const double c1[3] = {C[1] * A[2] - C[2] * A[1], C[2] * A[0] - C[0] * A[2], C[0] * A[1] - C[1] * A[0]};   // C x A
    const double c2[3] = {A[1] * B[2] - A[2] * B[1], A[2] * B[0] - A[0] * B[2], A[0] * B[1] - A[1] * B[0]};   // A x B
    const double det = A[0] * c0[0] + A[1] * c0[1] + A[2] * c0[2];
    const double na = std::sqrt(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]), nb = std::sqrt(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]),
                 nc = std::sqrt(C[0] * C[0] + C[1] * C[1] + C[2] * C[2]);
    if (!std::isfinite(det) || !(std::fabs(det) > 1e-9 * na * nb * nc) || !(na > 0) || !(nb > 0) || !(nc > 0)) return false;
    xmin = ymin = 1e300; xmax = ymax = -1e300;
    for (int k = 0; k < 8; ++k) {
        double v[3];
        for (int a = 0; a < 3; ++a) v[a] = (((k >> a) & 1) ? p.size[a] + kSkyCellMargin : -1.0 - kSkyCellMargin) - p.o[a];
        const double a_ = (c0[0] * v[0] + c0[1] * v[1] + c0[2] * v[2]) / det, b_ = (c1[0] * v[0] + c1[1] * v[1] + c1[2] * v[2]) / det,
                     c_ = (c2[0] * v[0] + c2[1] * v[1] + c2[2] * v[2]) / det;
        const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (!std::isfinite(a_) || !std::isfinite(b_) || !(c_ * nc > 1e-3 * len)) return false;   // a corner beside or behind the camera: no cull
        xmin = std::min(xmin, a_ / c_); xmax = std::max(xmax, a_ / c_);
        ymin = std::min(ymin, b_ / c_); ymax = std::max(ymax, b_ / c_);
        if (pts) { pts[k][0] = a_ / c_; pts[k][1] = b_ / c_; }
    }
    return true;
}

// The slanted edges of the convex hull of the projected corners (pts: n points, absolute pixels) as
// half-planes a*X + b*Y + c <= 0 with (a, b) the unit outward normal and the pixel margin in c: every
// pixel whose ray can hit a box satisfies all of them (the box projects into the hull, as into the
// rectangle).  Leaving an edge out only widens the region, so at most vk::kHullEdges are kept: those that
// cut the most off the bounding rectangle (length x how far the edge is from axis-aligned -- the
// rectangle already holds the axis-aligned ones).  Unused entries: (0, 0, -1).
static void sky_hull_edges(const double (*pts)[2], int n, double margin, float out[vk::kHullEdges][3])
{
    // fixed arrays, no heap: the entry point that allocates nothing runs this for every frame
    constexpr int kMaxPts = 16;
    struct P { double x, y; };
    for (int e = 0; e < vk::kHullEdges; ++e) { out[e][0] = 0.f; out[e][1] = 0.f; out[e][2] = -1.f; }
    if (n < 3 || n > kMaxPts) return;
    P p[kMaxPts];
    for (int i = 0; i < n; ++i) {
        if (!std::isfinite(pts[i][0]) || !std::isfinite(pts[i][1]) || std::fabs(pts[i][0]) > 1e7 || std::fabs(pts[i][1]) > 1e7) return;
        p[i] = {pts[i][0], pts[i][1]};
    }
    std::sort(p, p + n, [](const P &a, const P &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });      // in place
    auto cross = [](const P &o, const P &a, const P &b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); };
    P h[2 * kMaxPts];                                                 // Andrew's monotone chain, counter-clockwise
    int k = 0;
    for (int i = 0; i < n; ++i) { while (k >= 2 && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k; h[k++] = p[i]; }
    for (int i = n - 2, t = k + 1; i >= 0; --i) { while (k >= t && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k; h[k++] = p[i]; }
    const int m = k - 1;                                              // hull vertices h[0..m-1]
    if (m < 3) return;
    struct Edge { double a, b, score; };
    Edge edges[2 * kMaxPts];
    int n_edges = 0;
    for (int i = 0; i < m; ++i) {
        const P &u = h[i], &v = h[(i + 1) % m];
        const double dx = v.x - u.x, dy = v.y - u.y, len = std::sqrt(dx * dx + dy * dy);
        if (!(len > 1e-6)) continue;
        const double a = dy / len, b = -dx / len;                      // outward normal of a counter-clockwise polygon
        edges[n_edges++] = {a, b, len * std::min(std::fabs(a), std::fabs(b))};
    }
    std::sort(edges, edges + n_edges, [](const Edge &x, const Edge &y) { return x.score > y.score; });
    for (int e = 0; e < vk::kHullEdges && e < n_edges; ++e) {
        if (!(edges[e].score > 1.0)) break;                           // axis-aligned: the rectangle has it
        // to float, rounded so that the half-plane only grows: c towards -inf.  Every point must lie inside
        // (the hull says so; rounding is covered by taking the support over all points)
        const float a = (float)edges[e].a, b = (float)edges[e].b;
        double support = -1e300;
        for (int j = 0; j < n; ++j) support = std::max(support, (double)a * pts[j][0] + (double)b * pts[j][1]);
        const double norm = std::sqrt((double)a * a + (double)b * b);
        out[e][0] = a; out[e][1] = b;
        out[e][2] = std::nextafterf((float)(-(support + margin * norm + 0.01)), -INFINITY);
    }
}

// Pixels by which the reference's serial direction sums (main.cpp:282-286) can drift from
// row0 + y*ydiff + x*xdiff, times three.  The sums are in grid units (iMV = scale(size) * inverse(view),
// main.cpp:204-221): each of the up to width + height additions rounds a value of magnitude |v| by at
// most 2^-24 |v| per component, and a constant increment rounds the same way inside a binade, so the
// errors can add up rather than cancel.  One pixel spacing is |xdiff| (|ydiff|), so the drift is at most
// (width + height) * 2^-24 * sqrt(3) * max|v| / min(|xdiff|, |ydiff|) pixels across the direction, and
// about as much again through the perspective division at the frame's edge: 0.8 px at 4K, 3 px at
// 7680x4320, 14 px at 16K.  The cull margin is kSkyPixelMargin + three times that bound.
static double sky_drift_margin(const SkyPass &p, const volly_frame_desc &f)
{
    double vmax = 0.0;
    for (int k = 0; k < 4; ++k) {
        const double x = (k & 1) ? f.width - 1 : 0, y = (k & 2) ? f.height - 1 : 0;
        for (int a = 0; a < 3; ++a) vmax = std::max(vmax, std::fabs(p.r0[a] + y * p.yd[a] + x * p.xd[a]));
    }
    const double nx = std::sqrt(p.xd[0] * p.xd[0] + p.xd[1] * p.xd[1] + p.xd[2] * p.xd[2]);
    const double ny = std::sqrt(p.yd[0] * p.yd[0] + p.yd[1] * p.yd[1] + p.yd[2] * p.yd[2]);
    const double spacing = std::min(nx, ny);
    if (!(spacing > 0.0) || !std::isfinite(vmax)) return INFINITY;
    const double drift = ((double)f.width + (double)f.height) * std::ldexp(1.0, -24) * std::sqrt(3.0) * vmax / spacing;
    return std::ceil(3.0 * drift);
}

static void sky_rect_dims(const int32_t size_hi[3], const int32_t size_lo[3], const volly_frame_desc &f, bool enabled, int32_t out[4],
                          float hull[vk::kHullEdges][3] = nullptr)
{
    out[0] = 0; out[1] = f.tile_w - 1; out[2] = 0; out[3] = f.tile_h - 1;            // no cull
    if (hull) for (int e = 0; e < vk::kHullEdges; ++e) { hull[e][0] = 0.f; hull[e][1] = 0.f; hull[e][2] = -1.f; }
    double pts[16][2];
    int n_pts = 0;
    if (!enabled || (f.flags & VOLLY_FRAME_RAYTRACE)) return;
    const bool naive = (f.flags & VOLLY_FRAME_NAIVE) != 0;
    SkyPass hi{}, lo{};
    for (int a = 0; a < 3; ++a) {
        hi.o[a] = f.origin_hi[a]; hi.r0[a] = f.row0_hi[a]; hi.xd[a] = f.xdiff_hi[a]; hi.yd[a] = f.ydiff_hi[a];
        lo.o[a] = f.origin_lo[a]; lo.r0[a] = f.row0_lo[a]; lo.xd[a] = f.xdiff_lo[a]; lo.yd[a] = f.ydiff_lo[a];
    }
    for (int a = 0; a < 3; ++a) { hi.size[a] = size_hi[a]; lo.size[a] = size_lo[a]; }
    double x0, x1, y0, y1;
    // the pixel margin grows with the frame: the serial sums drift (sky_drift_margin); both passes' constants
    // describe the same rays (checked below), so the larger of the two bounds serves both
    const double margin = kSkyPixelMargin + std::max(sky_drift_margin(hi, f), naive ? 0.0 : sky_drift_margin(lo, f));
    if (!std::isfinite(margin) || margin > 4096.0) return;
    if (naive) {
        if (!sky_project(hi, x0, x1, y0, y1, pts)) return;
        n_pts = 8;
    } else {
        // the target pass runs from scale * (a point of the low ray) along the target direction sum: both
        // sets of constants must give the same rays.  D_hi and scale * D_lo are linear in (x, y), so their
        // difference is largest at a corner of the sub-rectangle.
        double sc[3], worst = 0.0, far = 0.0;
        for (int a = 0; a < 3; ++a) sc[a] = f.scale_lo_to_hi[a];
        for (int k = 0; k < 4; ++k) {
            const double x = f.x0 + ((k & 1) ? f.tile_w - 1 : 0), y = f.y0 + ((k & 2) ? f.tile_h - 1 : 0);
            double dh[3], dl[3], nh = 0.0, e = 0.0;
            for (int a = 0; a < 3; ++a) {
                dh[a] = hi.r0[a] + y * hi.yd[a] + x * hi.xd[a];
                dl[a] = sc[a] * (lo.r0[a] + y * lo.yd[a] + x * lo.xd[a]);
                nh += dh[a] * dh[a]; e += (dh[a] - dl[a]) * (dh[a] - dl[a]);
            }
            if (!(nh > 0.0) || !std::isfinite(nh) || !std::isfinite(e)) return;
            worst = std::max(worst, std::sqrt(e / nh));
        }
        SkyPass hb = hi;                                   // the target rays as they are traced: from the scaled low origin
        for (int a = 0; a < 3; ++a) {
            hb.o[a] = sc[a] * lo.o[a];
            const double d0 = std::fabs(-1.0 - hb.o[a]), d1 = std::fabs(hi.size[a] + 1.0 - hb.o[a]);
            far += std::max(d0, d1) * std::max(d0, d1);
        }
        far = std::sqrt(far);
        if (!std::isfinite(far) || !(3.0 * far * worst <= 0.25 * kSkyCellMargin)) return;     // constants disagree: no cull
        double u0, u1, v0, v1;
        if (!sky_project(lo, x0, x1, y0, y1, pts) || !sky_project(hb, u0, u1, v0, v1, pts + 8)) return;
        x0 = std::min(x0, u0); x1 = std::max(x1, u1); y0 = std::min(y0, v0); y1 = std::max(y1, v1);
        n_pts = 16;
    }
    // absolute pixels -> the sub-rectangle's, widened, clamped (an empty range culls everything)
    const double lo_x = std::floor(x0 - margin) - f.x0, hi_x = std::ceil(x1 + margin) - f.x0;
    const double lo_y = std::floor(y0 - margin) - f.y0, hi_y = std::ceil(y1 + margin) - f.y0;
    if (!std::isfinite(lo_x) || !std::isfinite(hi_x) || !std::isfinite(lo_y) || !std::isfinite(hi_y)) return;
    out[0] = (int32_t)std::min(std::max(lo_x, 0.0), (double)f.tile_w);
    out[1] = (int32_t)std::max(std::min(hi_x, (double)f.tile_w - 1.0), -1.0);
    out[2] = (int32_t)std::min(std::max(lo_y, 0.0), (double)f.tile_h);
    out[3] = (int32_t)std::max(std::min(hi_y, (double)f.tile_h - 1.0), -1.0);
    if (hull) sky_hull_edges(pts, n_pts, margin, hull);
}

static void sky_rect(const volly_gpu_scene *s, const volly_frame_desc &f, bool enabled, int32_t out[4], float hull[vk::kHullEdges][3] = nullptr)
{
    const int32_t hi[3] = {s->target.sx, s->target.sy, s->target.sz}, lo[3] = {s->low.sx, s->low.sy, s->low.sz};
    sky_rect_dims(hi, lo, f, enabled, out, hull);
}

size_t checkpoint_floats(const volly_frame_desc &f, int n)
{
    return (size_t)n * (size_t)((f.tile_w + vk::kCkpt - 1) / vk::kCkpt) * (size_t)f.tile_h * 6;
}

// One launch of the pre-pass and the frame kernel for n <= kFrameChunk frames.  fixed_frames /
// fixed_spans / fixed_ckpt: scratch handed in by the entry point that allocates nothing (then n <=
// vk::kInlineFrames, checkpoints and spans fit, and this function calls no allocating routine);
// null: the scene's own scratch, grown with hipMalloc as needed.
int render_chunk(volly_gpu_scene *s, const volly_frame_desc *frames, int n, size_t frame_offset, const volly_outputs *outs,
                 hipStream_t stream, volly_frame_desc *fixed_frames = nullptr, int2 *fixed_spans = nullptr, float *fixed_ckpt = nullptr)
{
    const int tw = frames[0].tile_w, th = frames[0].tile_h;
    const size_t need = checkpoint_floats(frames[0], n);
    if (need >= (1ull << 31)) return fail(VOLLY_E_ARG, "render: sub-rectangles of more than ~2^30 pixels per launch");
    bool any_raytrace = false, all_naive = true;
    for (int i = 0; i < n; ++i) {
        any_raytrace |= (frames[i].flags & VOLLY_FRAME_RAYTRACE) != 0;
        all_naive &= (frames[i].flags & (VOLLY_FRAME_NAIVE | VOLLY_FRAME_RAYTRACE)) == VOLLY_FRAME_NAIVE;
    }
    // launches of naive frames only are bound by the output path: 32 x 2 tiles (128-byte row segments); else 8 x 8
    const bool packed = s->packed && !s->is_svo && !any_raytrace;        // vk::render_frames_packed: blocks of kPackWaves 8 x 8 tiles
    const int tile_w = packed ? vk::kTileBeam : all_naive ? vk::kTileNaive : vk::kTileBeam;
    const int block_w = packed ? tile_w * vk::kPackWaves : tile_w * VK_WAVE_BLOCK, block_h = 64 / tile_w;
    const size_t span_rows = (size_t)((th + block_h - 1) / block_h), need_spans = span_rows * (size_t)n;
    const bool no_alloc = fixed_frames != nullptr;
    volly_frame_desc *d_frames = fixed_frames;
    int2 *d_spans = fixed_spans;
    float *d_ckpt = fixed_ckpt;
    if (!no_alloc) {
        // The scene's scratch (descriptors, spans, checkpoints) is reused by every call.  Calls on one stream
        // are ordered by the stream; when the stream CHANGES, the new stream first waits for the event
        // the previous call recorded behind its last kernel, or its pre-pass would overwrite what that
        // kernel is still reading.  (The previous stream's handle is never touched again: the caller
        // may have destroyed it, and HIP faults on a dead handle.)
        if (s->scratch_in_use && s->scratch_stream != stream) {
            if (hipStreamWaitEvent(stream, s->scratch_event, 0) != hipSuccess) {
                (void)hipGetLastError();
                HIP_TRY(hipDeviceSynchronize());
            }
        }
        s->scratch_stream = stream;
        s->scratch_in_use = true;
        if (!s->d_frames) HIP_TRY(hipMalloc((void **)&s->d_frames, sizeof(volly_frame_desc) * kFrameChunk));
        if (need > s->ckpt_floats) {
            if (s->d_ckpt) { HIP_TRY(hipFree(s->d_ckpt)); s->d_ckpt = nullptr; s->ckpt_floats = 0; }
            HIP_TRY(hipMalloc((void **)&s->d_ckpt, need * sizeof(float)));
            s->ckpt_floats = need;
        }
        if (need_spans > s->span_count) {
            if (s->d_spans) { HIP_TRY(hipFree(s->d_spans)); s->d_spans = nullptr; s->span_count = 0; }
            HIP_TRY(hipMalloc((void **)&s->d_spans, need_spans * sizeof(int2)));
            s->span_count = need_spans;
        }
        d_frames = s->d_frames;
        d_spans = s->d_spans;
        d_ckpt = s->d_ckpt;
    } else if (n > vk::kInlineFrames || need > kNoAllocCkptFloats || need_spans > kNoAllocSpans)
        return fail(VOLLY_E_ARG, "render: batch does not fit the static scratch");
    int cull = outs->pt_end ? 0 : s->sky_cull;               // ptEnd of a miss is the reference's arithmetic: no shortcut
    // The descriptors travel as kernel arguments, kInlineFrames per pre-pass launch: no host-to-device copy (from
    // pageable memory it would block the host and leave the queue empty between calls).  The pre-pass leaves
    // them in the scratch where the frame kernel reads them, and turns each frame's sky rectangle and hull
    // edges into one span of columns per tile row.
    for (int base = 0; base < n; base += vk::kInlineFrames) {
        const int nb = std::min(vk::kInlineFrames, n - base);
        vk::FrameBatch batch;
        std::memcpy(batch.f, frames + base, sizeof(volly_frame_desc) * (size_t)nb);
        std::memset(batch.f + nb, 0, sizeof(volly_frame_desc) * (size_t)(vk::kInlineFrames - nb));
        std::memset(batch.cull, 0, sizeof batch.cull);
        std::memset(batch.hull, 0, sizeof batch.hull);
        for (int i = 0; i < nb; ++i) {
            sky_rect(s, frames[base + i], cull != 0, batch.cull[i], batch.hull[i]);
            if (cull < 2) for (int e = 0; e < vk::kHullEdges; ++e) { batch.hull[i][e][0] = 0.f; batch.hull[i][e][1] = 0.f; batch.hull[i][e][2] = -1.f; }
        }
        const dim3 pre_grid((unsigned)((th + 63) / 64), (unsigned)nb, all_naive ? 3 : 6);      // naive frames need the target sums only
        hipLaunchKernelGGL(vk::direction_checkpoints_inline, pre_grid, dim3(64), 0, stream, batch, nb, d_frames + base, d_spans, d_ckpt, base, block_h);
    }
    HIP_TRY(hipGetLastError());

    const size_t px = (size_t)tw * th * frame_offset;
    vk::OutDev o{};
    o.rgba8 = outs->rgba8 ? outs->rgba8 + px : nullptr;
    o.hit_index = outs->hit_index ? outs->hit_index + px : nullptr;
    o.steps_lo = outs->steps_lo ? outs->steps_lo + px : nullptr;
    o.steps_hi = outs->steps_hi ? outs->steps_hi + px : nullptr;
    o.depth = outs->depth ? outs->depth + px : nullptr;
    o.pt_end = outs->pt_end ? outs->pt_end + 3 * px : nullptr;

    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s->profiling && !no_alloc) {                      // the pair belongs to the scene from here on (destroy() frees it)
        HIP_TRY(hipEventCreate(&e0));
        if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(VOLLY_E_HIP, "render: hipEventCreate failed"); }
        s->events.emplace_back(e0, e1);
        HIP_TRY(hipEventRecord(e0, stream));
    }
    dim3 grid((unsigned)((tw + block_w - 1) / block_w), (unsigned)((th + block_h - 1) / block_h), (unsigned)n);
    if (packed) {
        // Passes: the first renders the blocks up to iteration bounds[0] and parks the rays still under way; every later
        // one continues the parked rays, pooled into full workgroups, up to its bound; the last runs them to the end.
        // A ray that is parked k times is written and read k times (28 bytes).  The entry point that allocates nothing
        // has no queue: one pass to the end.
        const size_t n_px = (size_t)n * tw * th;
        if (n_px >= (1ull << 32)) return fail(VOLLY_E_ARG, "render: more than 2^32 pixels per launch");
        std::vector<uint32_t> bounds;
        if (!no_alloc && s->pack_passes > 1) {
            uint32_t at = 0, len = s->pack_pass_len;
            for (int i = 0; i + 1 < s->pack_passes && at + len < s->target.cap; ++i) { at += len; bounds.push_back(at); if (i >= 2) len *= 2; }
        }
        vk::RayQueue q[2] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
        if (!bounds.empty()) {
            if (n_px > s->queue_rays) {
                if (s->d_queue) { HIP_TRY(hipFree(s->d_queue)); s->d_queue = nullptr; s->queue_rays = 0; }
                HIP_TRY(hipMalloc(&s->d_queue, 64 + 2 * 7 * n_px * sizeof(float)));
                s->queue_rays = n_px;
            }
            uint32_t *counts = (uint32_t *)s->d_queue;
            float *items = (float *)((char *)s->d_queue + 64);
            q[0] = {items, counts, (uint32_t)s->queue_rays};
            q[1] = {items + 7 * s->queue_rays, counts + 1, (uint32_t)s->queue_rays};
            HIP_TRY(hipMemsetAsync(counts, 0, 8, stream));
        }
        const uint32_t first_bound = bounds.empty() ? 0xFFFFFFFFu : bounds[0];
        if (all_naive)
            hipLaunchKernelGGL(vk::render_frames_packed<true>, grid, dim3(64 * vk::kPackWaves), 0, stream, s->target, s->low, d_frames, d_spans, d_ckpt, o,
                               s->pack_budget, first_bound, q[0]);
        else
            hipLaunchKernelGGL(vk::render_frames_packed<false>, grid, dim3(64 * vk::kPackWaves), 0, stream, s->target, s->low, d_frames, d_spans, d_ckpt, o,
                               s->pack_budget, first_bound, q[0]);
        HIP_TRY(hipGetLastError());
        const unsigned cont_blocks = (unsigned)((n_px + 64 * vk::kPackWaves - 1) / (64 * vk::kPackWaves));
        for (size_t i = 0; i < bounds.size(); ++i) {
            const int cur = (int)(i & 1);
            const uint32_t next_bound = i + 1 < bounds.size() ? bounds[i + 1] : 0xFFFFFFFFu;
            HIP_TRY(hipMemsetAsync(q[cur ^ 1].count, 0, 4, stream));
            if ((int)i + 1 >= s->pack_lookahead_from)       // the late passes: few, long rays -- the loop that requests bricks ahead of each ray
                hipLaunchKernelGGL(vk::continue_rays_packed<true>, dim3(cont_blocks), dim3(64 * vk::kPackWaves), 0, stream, s->target, d_frames, (uint32_t)(tw * th), o,
                                   q[cur], bounds[i], s->pack_budget, next_bound, q[cur ^ 1]);
            else
                hipLaunchKernelGGL(vk::continue_rays_packed<false>, dim3(cont_blocks), dim3(64 * vk::kPackWaves), 0, stream, s->target, d_frames, (uint32_t)(tw * th), o,
                                   q[cur], bounds[i], s->pack_budget, next_bound, q[cur ^ 1]);
            HIP_TRY(hipGetLastError());
        }
    }
    else if (s->is_svo && all_naive)
        hipLaunchKernelGGL((vk::render_frames<true, false, vk::kTileNaive>), grid, dim3(vk::kFrameThreads), 0, stream, s->target, s->svo, s->low, d_frames,
                           d_spans, d_ckpt, o);
    else if (s->is_svo)
        hipLaunchKernelGGL(vk::render_frames<true>, grid, dim3(vk::kFrameThreads), 0, stream, s->target, s->svo, s->low, d_frames,
                           d_spans, d_ckpt, o);
    else if (any_raytrace)
        hipLaunchKernelGGL((vk::render_frames<false, true>), grid, dim3(vk::kFrameThreads), 0, stream, s->target, s->svo, s->low,
                           d_frames, d_spans, d_ckpt, o);
    else if (all_naive)
        hipLaunchKernelGGL((vk::render_frames<false, false, vk::kTileNaive>), grid, dim3(vk::kFrameThreads), 0, stream, s->target, s->svo, s->low, d_frames,
                           d_spans, d_ckpt, o);
    else
        hipLaunchKernelGGL(vk::render_frames<false>, grid, dim3(vk::kFrameThreads), 0, stream, s->target, s->svo, s->low, d_frames,
                           d_spans, d_ckpt, o);
    HIP_TRY(hipGetLastError());
    if (e1) HIP_TRY(hipEventRecord(e1, stream));
    if (!no_alloc) {                                      // what a later call on another stream waits for
        if (!s->scratch_event) HIP_TRY(hipEventCreateWithFlags(&s->scratch_event, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(s->scratch_event, stream));
    }
    return VOLLY_OK;
}

} // namespace

extern "C" {

int volly_gpu_abi_version(void) { return 3; }

#ifndef VOLLY_BUILD_SRC_SHA
#define VOLLY_BUILD_SRC_SHA "unknown"
#endif
#ifndef VOLLY_BUILD_HIPCC
#define VOLLY_BUILD_HIPCC "unknown"
#endif
const char *volly_gpu_build_info(void)
{
    return "src_sha16=" VOLLY_BUILD_SRC_SHA "; hipcc=" VOLLY_BUILD_HIPCC
           "; kernel contract (m0, v44-v63, at most 64 VGPRs, no spills) checked on this compiler's assembly by tools/check_kernel_contract.py at build";
}

const char *volly_gpu_last_error(void) { return g_err.c_str(); }

int volume_gpu_upload(int device, const volly_dense_desc *target, const volly_dense_desc *low, volly_gpu_scene **out)
{
    if (!out) return fail(VOLLY_E_ARG, "volume_gpu_upload: out is null");
    *out = nullptr;
    int rc = check_dense(target, "volume_gpu_upload target");
    if (rc) return rc;
    if (low && (rc = check_dense(low, "volume_gpu_upload low"))) return rc;
    volly_gpu_scene *s = nullptr;
    if ((rc = new_scene(device, &s))) return rc;
    DeviceGuard g(device);
    if (!g.ok) { destroy(s); return fail(VOLLY_E_HIP, "hipSetDevice failed"); }
    rc = make_dense(s, target->sx, target->sy, target->sz, target->voxels, nullptr, &s->target);
    if (!rc && low) {
        rc = make_dense(s, low->sx, low->sy, low->sz, low->voxels, nullptr, &s->low);
        s->has_low = true;
    }
    if (rc) { std::string keep = g_err; destroy(s); g_err = keep; return rc; }
    choose_frame_kernel(s);
    *out = s;
    return VOLLY_OK;
}

int svo_gpu_upload(int device, const volly_svo_desc *svo, const volly_dense_desc *low, volly_gpu_scene **out)
{
    if (!out) return fail(VOLLY_E_ARG, "svo_gpu_upload: out is null");
    *out = nullptr;
    if (!svo || !svo->nodes || svo->n_nodes < 1 || (svo->n_leaves && !svo->leaves))
        return fail(VOLLY_E_ARG, "svo_gpu_upload: null octree");
    if (svo->depth < 1 || svo->depth > 13 || svo->res < 1 || svo->res > (1 << svo->depth))
        return fail(VOLLY_E_ARG, "svo_gpu_upload: need 1 <= depth <= 13 and 1 <= res <= 2^depth");
    if (svo->n_nodes >= 0x80000000u) return fail(VOLLY_E_ARG, "svo_gpu_upload: more than 2^31 nodes");
    // The kernel trusts the tree: check that every child reference stays inside the
    // arrays, points forward (breadth-first order) and that leaves sit exactly at
    // level 1, so a walk from the root ends after at most `depth` nodes.
    int64_t l3_first = -1, l3_count = 0;                 // the level-3 nodes (8^3 cells), if they form one index range
    {
        const volly_svo_node *nd = svo->nodes;
        std::vector<int8_t> level(svo->n_nodes, (int8_t)-1);
        level[0] = (int8_t)svo->depth;
        for (uint32_t i = 0; i < svo->n_nodes; ++i) {
            const uint32_t cnt = (uint32_t)__builtin_popcount(nd[i].valid_mask);
            if (level[i] < 1) return fail(VOLLY_E_ARG, "svo_gpu_upload: unreachable node or tree deeper than depth");
            const bool leaf_level = level[i] == 1;
            if (nd[i].leaf_mask != (leaf_level ? nd[i].valid_mask : 0))
                return fail(VOLLY_E_ARG, "svo_gpu_upload: leaf_mask must equal valid_mask at level 1 and be 0 above");
            const uint64_t end = (uint64_t)nd[i].child_offset + cnt;
            if (leaf_level) {
                if (end > svo->n_leaves) return fail(VOLLY_E_ARG, "svo_gpu_upload: leaf offset out of range");
            } else if (cnt) {
                if (end > svo->n_nodes || nd[i].child_offset <= i)
                    return fail(VOLLY_E_ARG, "svo_gpu_upload: child offset out of range");
                for (uint32_t c = 0; c < cnt; ++c) level[nd[i].child_offset + c] = (int8_t)(level[i] - 1);
            }
        }
        int64_t l3_last = -1;
        for (uint32_t i = 0; i < svo->n_nodes; ++i)
            if (level[i] == 3) { if (l3_first < 0) l3_first = i; l3_last = i; ++l3_count; }
        if (l3_count && l3_last - l3_first + 1 != l3_count) l3_first = -1;      // not breadth first: no blocks
    }
    int rc;
    if (low && (rc = check_dense(low, "svo_gpu_upload low"))) return rc;
    volly_gpu_scene *s = nullptr;
    if ((rc = new_scene(device, &s))) return rc;
    DeviceGuard g(device);
    if (!g.ok) { destroy(s); return fail(VOLLY_E_HIP, "hipSetDevice failed"); }
    s->is_svo = true;
    uint2 *nodes = nullptr, *leaves = nullptr;
    rc = dev_alloc(s, (void **)&nodes, (size_t)svo->n_nodes * 8);
    if (!rc) rc = dev_alloc(s, (void **)&leaves, (size_t)svo->n_leaves * 8);
    if (!rc && hipMemcpy(nodes, svo->nodes, (size_t)svo->n_nodes * 8, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(VOLLY_E_HIP, "svo_gpu_upload: node copy failed");
    if (!rc && svo->n_leaves && hipMemcpy(leaves, svo->leaves, (size_t)svo->n_leaves * 8, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(VOLLY_E_HIP, "svo_gpu_upload: leaf copy failed");
    if (!rc) rc = finish_svo(s, nodes, leaves, svo->n_nodes, (uint32_t)svo->n_leaves, svo->depth, svo->res, l3_first, l3_count);
    if (!rc && low) {
        rc = make_dense(s, low->sx, low->sy, low->sz, low->voxels, nullptr, &s->low);
        s->has_low = true;
    }
    if (rc) { std::string keep = g_err; destroy(s); g_err = keep; return rc; }
    *out = s;
    return VOLLY_OK;
}

int volume_gpu_generate_noise(int device, int32_t res, uint32_t seed, uint32_t threshold, volly_gpu_scene **out)
{
    if (!out) return fail(VOLLY_E_ARG, "volume_gpu_generate_noise: out is null");
    *out = nullptr;
    if (res < 4 || res > 1024 || (res % 4) != 0) return fail(VOLLY_E_ARG, "volume_gpu_generate_noise: res must be a multiple of 4 in [4,1024]");
    volly_gpu_scene *s = nullptr;
    int rc = new_scene(device, &s);
    if (rc) return rc;
    DeviceGuard g(device);
    if (!g.ok) { destroy(s); return fail(VOLLY_E_HIP, "hipSetDevice failed"); }
    const size_t n = (size_t)res * res * res;
    uint2 *vox = nullptr, *vox_lo = nullptr;
    rc = dev_alloc(s, (void **)&vox, n * 8);
    if (!rc) {
        hipLaunchKernelGGL(vk::generate_noise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, vox, res, seed, threshold);
        if (hipGetLastError() != hipSuccess) rc = fail(VOLLY_E_HIP, "generate_noise launch failed");
    }
    if (!rc) rc = make_dense(s, res, res, res, nullptr, vox, &s->target);
    const int lr = res / 4;
    const size_t nl = (size_t)lr * lr * lr;
    if (!rc) rc = dev_alloc(s, (void **)&vox_lo, nl * 8);
    if (!rc) {
        hipLaunchKernelGGL(vk::reduce_low, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, 0, s->target.occ, lr, lr, lr, s->target.nbx, s->target.nby, vox_lo);
        if (hipGetLastError() != hipSuccess) rc = fail(VOLLY_E_HIP, "reduce_low launch failed");
    }
    if (!rc) rc = make_dense(s, lr, lr, lr, nullptr, vox_lo, &s->low);
    if (rc) { std::string keep = g_err; destroy(s); g_err = keep; return rc; }
    s->has_low = true;
    choose_frame_kernel(s);
    *out = s;
    return VOLLY_OK;
}

int volume_gpu_render_device(volly_gpu_scene *scene, const volly_frame_desc *frames, int n_frames, const volly_outputs *outs,
                             void *hip_stream)
{
    int rc = validate_frames(scene, frames, n_frames);
    if (rc) return rc;
    if (!outs) return fail(VOLLY_E_ARG, "render: outs is null");
    DeviceGuard g(scene->device);
    if (!g.ok) return fail(VOLLY_E_HIP, "hipSetDevice failed");
    hipStream_t stream = (hipStream_t)hip_stream;
    // one launch per chunk: up to kFrameChunk frames, fewer when their direction checkpoints would pass 2 GiB of scratch
    const size_t per_frame = std::max<size_t>(checkpoint_floats(frames[0], 1), 1);
    const int chunk = (int)std::min<size_t>((size_t)kFrameChunk, std::max<size_t>(((size_t)1 << 29) / per_frame, 1));
    for (int first = 0; first < n_frames; first += chunk) {
        const int n = std::min(chunk, n_frames - first);
        if ((rc = render_chunk(scene, frames + first, n, (size_t)first, outs, stream))) return rc;
    }
    return VOLLY_OK;
}

int volume_gpu_render_device_noalloc(volly_gpu_scene *scene, const volly_frame_desc *frames, int n_frames, const volly_outputs *outs,
                                     void *hip_stream)
{
    int rc = validate_frames(scene, frames, n_frames);
    if (rc) return rc;
    if (!outs) return fail(VOLLY_E_ARG, "render: outs is null");
    DeviceGuard g(scene->device);
    if (!g.ok) return fail(VOLLY_E_HIP, "hipSetDevice failed");
    if (!scene->noalloc_scratch) scene->noalloc_scratch = noalloc_scratch(scene->device);     // first call on this device
    if (!scene->noalloc_scratch) return fail(VOLLY_E_HIP, "volume_gpu_render_device_noalloc: hipGetSymbolAddress(scratch) failed");
    const size_t per_frame = checkpoint_floats(frames[0], 1);
    if (per_frame > kNoAllocCkptFloats)
        return fail(VOLLY_E_ARG, "volume_gpu_render_device_noalloc: one sub-rectangle needs " + std::to_string(per_frame * sizeof(float)) +
                                     " bytes of direction checkpoints (6 * ceil(tile_w / 8) * tile_h floats), the static scratch holds " +
                                     std::to_string(kNoAllocCkptFloats * sizeof(float)) + ": render row bands, or use volume_gpu_render_device");
    volly_frame_desc *d_frames = (volly_frame_desc *)scene->noalloc_scratch;
    int2 *d_spans = (int2 *)((unsigned char *)scene->noalloc_scratch + kNoAllocDescBytes);
    float *d_ckpt = (float *)((unsigned char *)scene->noalloc_scratch + kNoAllocDescBytes + kNoAllocSpans * sizeof(int2));
    const size_t span_rows = ((size_t)frames[0].tile_h + 1) / 2;          // at most: 32 x 2 tiles
    if (span_rows > kNoAllocSpans)
        return fail(VOLLY_E_ARG, "volume_gpu_render_device_noalloc: sub-rectangles of more than " + std::to_string(2 * kNoAllocSpans) + " rows");
    const int chunk = (int)std::min<size_t>(std::min<size_t>((size_t)vk::kInlineFrames, kNoAllocCkptFloats / per_frame), kNoAllocSpans / span_rows);
    hipStream_t stream = (hipStream_t)hip_stream;
    // chunks follow each other on `stream`, so a chunk's pre-pass overwrites the scratch only after
    // the previous chunk's frame kernel has finished with it; nothing in it has to start at zero
    // (descriptors and checkpoints are written in full before they are read)
    for (int first = 0; first < n_frames; first += chunk) {
        const int n = std::min(chunk, n_frames - first);
        if ((rc = render_chunk(scene, frames + first, n, (size_t)first, outs, stream, d_frames, d_spans, d_ckpt))) return rc;
    }
    return VOLLY_OK;
}

int volume_gpu_render(volly_gpu_scene *scene, const volly_frame_desc *frames, int n_frames, const volly_outputs *outs)
{
    int rc = validate_frames(scene, frames, n_frames);
    if (rc) return rc;
    if (!outs) return fail(VOLLY_E_ARG, "render: outs is null");
    DeviceGuard g(scene->device);
    if (!g.ok) return fail(VOLLY_E_HIP, "hipSetDevice failed");
    const size_t px = (size_t)frames[0].tile_w * frames[0].tile_h * (size_t)n_frames;
    constexpr int kPlanes = 6;
    void *host[kPlanes] = {outs->rgba8, outs->hit_index, outs->steps_lo, outs->steps_hi, outs->depth, outs->pt_end};
    const size_t words[kPlanes] = {1, 1, 1, 1, 1, 3};      // 4-byte words per pixel
    size_t wanted = 0;
    for (int i = 0; i < kPlanes; ++i) wanted += host[i] ? words[i] : 0;
    // Output buffers in pinned host memory (volly_gpu_host_alloc, hipHostMalloc, hipHostRegister):
    // render a few frames at a time on one stream and copy each batch back on another, so the
    // PCIe transfer of batch k overlaps the traversal of batch k+1.  Pageable buffers take the
    // plain path below (the runtime stages them through its own pinned buffers).
    bool pinned = wanted > 0;
    for (void *h : host) {
        if (!h) continue;
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, h) != hipSuccess || a.type != hipMemoryTypeHost) { pinned = false; (void)hipGetLastError(); }
    }
    // Small calls into pinned memory -- the reference's own call shape, one 400x300 frame per call (main.h:112,
    // main.cpp:314) -- skip the staging planes altogether: pinned host memory is mapped into the device's address
    // space, so the frame kernel stores its pixels straight into the caller's buffers over PCIe (0.5 MB takes ~10 us)
    // and the call is two launches and one synchronisation: no copy, no second stream, no events.
    if (pinned && wanted * px * 4 <= kDirectHostBytes) {
        void *devp[kPlanes] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        bool mapped = true;
        for (int i = 0; i < kPlanes && mapped; ++i)
            if (host[i] && (hipHostGetDevicePointer(&devp[i], host[i], 0) != hipSuccess || !devp[i])) { mapped = false; (void)hipGetLastError(); }
        if (mapped) {
            if (!scene->s_render) HIP_TRY(hipStreamCreateWithFlags(&scene->s_render, hipStreamNonBlocking));
            volly_outputs direct{(uint32_t *)devp[0], (uint32_t *)devp[1], (uint32_t *)devp[2], (uint32_t *)devp[3], (float *)devp[4], (float *)devp[5]};
            if ((rc = volume_gpu_render_device(scene, frames, n_frames, &direct, scene->s_render))) return rc;
            HIP_TRY(hipStreamSynchronize(scene->s_render));
            return VOLLY_OK;
        }
    }
    // staging planes on the device for everything else
    if (wanted * px * 4 > scene->plane_bytes) {
        if (scene->d_planes) { HIP_TRY(hipFree(scene->d_planes)); scene->d_planes = nullptr; scene->plane_bytes = 0; }
        HIP_TRY(hipMalloc(&scene->d_planes, wanted * px * 4));
        scene->plane_bytes = wanted * px * 4;
    }
    uint32_t *dev[kPlanes] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t k = 0;
    for (int i = 0; i < kPlanes; ++i)
        if (host[i]) { dev[i] = (uint32_t *)scene->d_planes + k * px; k += words[i]; }
    volly_outputs d{dev[0], dev[1], dev[2], dev[3], (float *)dev[4], (float *)dev[5]};

    if (pinned) {
        if (!scene->s_render) HIP_TRY(hipStreamCreateWithFlags(&scene->s_render, hipStreamNonBlocking));
        if (!scene->s_copy) HIP_TRY(hipStreamCreateWithFlags(&scene->s_copy, hipStreamNonBlocking));
        const size_t fpx = (size_t)frames[0].tile_w * frames[0].tile_h;
        const int batch = (int)std::min<size_t>(std::max<size_t>((size_t)(4u << 20) / std::max<size_t>(fpx, 1), 1), (size_t)kFrameChunk);   // ~4 Mpixel per batch
        std::vector<hipEvent_t> evs;
        int status = VOLLY_OK;
        for (int first = 0; first < n_frames && status == VOLLY_OK; first += batch) {
            const int n = std::min(batch, n_frames - first);
            status = render_chunk(scene, frames + first, n, (size_t)first, &d, scene->s_render);
            if (status) break;
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, scene->s_render) != hipSuccess ||
                hipStreamWaitEvent(scene->s_copy, ev, 0) != hipSuccess) {
                status = fail(VOLLY_E_HIP, "render: event set-up failed");
                if (ev) (void)hipEventDestroy(ev);
                break;
            }
            evs.push_back(ev);
            for (int i = 0; i < kPlanes && status == VOLLY_OK; ++i)
                if (host[i] && hipMemcpyAsync((uint32_t *)host[i] + (size_t)first * fpx * words[i], dev[i] + (size_t)first * fpx * words[i],
                                              (size_t)n * fpx * 4 * words[i], hipMemcpyDeviceToHost, scene->s_copy) != hipSuccess)
                    status = fail(VOLLY_E_HIP, "render: copy to pinned host memory failed");
        }
        // render_chunk reuses the frame-descriptor and checkpoint scratch: a batch must not start
        // before the previous one has finished with them -- same stream, so it does not
        hipError_t e1 = hipStreamSynchronize(scene->s_render), e2 = hipStreamSynchronize(scene->s_copy);
        for (hipEvent_t ev : evs) (void)hipEventDestroy(ev);
        if (status) return status;
        if (e1 != hipSuccess || e2 != hipSuccess) return fail(VOLLY_E_HIP, "render: stream synchronisation failed");
        return VOLLY_OK;
    }

    if ((rc = volume_gpu_render_device(scene, frames, n_frames, &d, nullptr))) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    for (int i = 0; i < kPlanes; ++i)
        if (host[i]) HIP_TRY(hipMemcpy(host[i], dev[i], px * 4 * words[i], hipMemcpyDeviceToHost));
    return VOLLY_OK;
}

void *volly_gpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        fail(VOLLY_E_NOMEM, "volly_gpu_host_alloc: hipHostMalloc failed");
        return nullptr;
    }
    return p;
}

void volly_gpu_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int volly_gpu_set_profiling(volly_gpu_scene *scene, int enabled)
{
    if (!scene) return fail(VOLLY_E_ARG, "null scene");
    scene->profiling = enabled != 0;
    return VOLLY_OK;
}

int volly_gpu_kernel_time(volly_gpu_scene *scene, double *ms_total, int *launches)
{
    if (!scene || !ms_total || !launches) return fail(VOLLY_E_ARG, "null argument");
    DeviceGuard g(scene->device);
    double total = 0.0;
    int n = 0;
    for (auto &e : scene->events) {                        // a pair whose launch failed half way is dropped, not counted
        float ms = 0.f;
        if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
            total += ms;
            ++n;
        } else
            (void)hipGetLastError();
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    scene->events.clear();
    *ms_total = total;
    *launches = n;
    return VOLLY_OK;
}

int volly_gpu_read_low_voxels(volly_gpu_scene *scene, uint64_t first, uint64_t count, void *out)
{
    if (!scene || !out) return fail(VOLLY_E_ARG, "null argument");
    if (!scene->has_low) return fail(VOLLY_E_UNSUPPORTED, "scene has no low grid");
    const uint64_t n = (uint64_t)scene->low.sx * scene->low.sy * scene->low.sz;
    if (first > n || count > n - first) return fail(VOLLY_E_ARG, "range outside the grid");
    DeviceGuard g(scene->device);
    HIP_TRY(hipMemcpy(out, scene->low.vox + first, count * 8, hipMemcpyDeviceToHost));
    return VOLLY_OK;
}

int volly_gpu_read_svo(volly_gpu_scene *scene, volly_svo_node *nodes, void *leaves)
{
    if (!scene) return fail(VOLLY_E_ARG, "null argument");
    if (!scene->svo.nodes) return fail(VOLLY_E_UNSUPPORTED, "scene has no octree");
    DeviceGuard g(scene->device);
    if (nodes) HIP_TRY(hipMemcpy(nodes, scene->svo.nodes, (size_t)scene->svo.n_nodes * 8, hipMemcpyDeviceToHost));
    if (leaves && scene->n_leaves) HIP_TRY(hipMemcpy(leaves, scene->svo.leaves, (size_t)scene->n_leaves * 8, hipMemcpyDeviceToHost));
    return VOLLY_OK;
}

int volly_gpu_read_voxels(volly_gpu_scene *scene, uint64_t first, uint64_t count, void *out)
{
    if (!scene || !out) return fail(VOLLY_E_ARG, "null argument");
    if (!scene->target.vox) return fail(VOLLY_E_UNSUPPORTED, "scene has no dense target");
    const uint64_t n = (uint64_t)scene->target.sx * scene->target.sy * scene->target.sz;
    if (first > n || count > n - first) return fail(VOLLY_E_ARG, "range outside the grid");
    DeviceGuard g(scene->device);
    HIP_TRY(hipMemcpy(out, scene->target.vox + first, count * 8, hipMemcpyDeviceToHost));
    return VOLLY_OK;
}

// which DDA loop trace_dense takes for a grid (volly_kernels.h), for wavefronts whose 64 rays are all regular
static int dda_loop_of(const vk::DenseDev &g)
{
    if (!host_grid_is_regular(g)) return 3;               // trace_dense_checked
    return g.use_regions ? 1 : 2;                         // trace_dense_regions_whole_asm : trace_dense_bricks_whole_asm
}

int volly_gpu_sky_rect_for(const int32_t size_hi[3], const int32_t size_lo[3], const volly_frame_desc *frame, int32_t rect[4])
{
    if (!size_hi || !frame || !rect) return fail(VOLLY_E_ARG, "null argument");
    const int32_t none[3] = {1, 1, 1};
    if (!size_lo && !(frame->flags & (VOLLY_FRAME_NAIVE | VOLLY_FRAME_RAYTRACE))) return fail(VOLLY_E_ARG, "a beam frame needs the low grid's size");
    sky_rect_dims(size_hi, size_lo ? size_lo : none, *frame, true, rect);
    return VOLLY_OK;
}

int volly_gpu_sky_spans_for(const int32_t size_hi[3], const int32_t size_lo[3], const volly_frame_desc *frame, int32_t block_h,
                            int32_t *spans, float *hull_out)
{
    if (!size_hi || !frame || !spans || block_h < 1 || frame->tile_h < 1) return fail(VOLLY_E_ARG, "null argument or block_h < 1");
    const int32_t none[3] = {1, 1, 1};
    if (!size_lo && !(frame->flags & (VOLLY_FRAME_NAIVE | VOLLY_FRAME_RAYTRACE))) return fail(VOLLY_E_ARG, "a beam frame needs the low grid's size");
    int32_t rect[4];
    float hull[vk::kHullEdges][3];
    sky_rect_dims(size_hi, size_lo ? size_lo : none, *frame, true, rect, hull);
    const int rows = (frame->tile_h + block_h - 1) / block_h;
    for (int r = 0; r < rows; ++r) {
        const int2 sp = vk::sky_span(*frame, rect, &hull[0][0], r, block_h);
        spans[2 * r] = sp.x; spans[2 * r + 1] = sp.y;
    }
    if (hull_out) std::memcpy(hull_out, hull, sizeof hull);
    return VOLLY_OK;
}

int volly_gpu_sky_rect(const volly_gpu_scene *scene, const volly_frame_desc *frame, int32_t rect[4])
{
    if (!scene || !frame || !rect) return fail(VOLLY_E_ARG, "null argument");
    sky_rect(scene, *frame, scene->sky_cull, rect);
    return VOLLY_OK;
}

int64_t volly_gpu_scene_info(const volly_gpu_scene *s, int what)
{
    if (!s) return -1;
    switch (what) {
    case 0: return s->target.sx;
    case 1: return s->target.sy;
    case 2: return s->target.sz;
    case 3: return s->is_svo;
    case 4: return s->has_low;
    case 5: return s->has_low ? s->low.sx : 0;
    case 6: return s->device;
    case 7: return (int64_t)(s->bytes >> 20);
    case 8: return s->packed ? 3 : 0;
    case 9: return s->is_svo ? 0 : s->target.use_regions;
    case 10: return s->svo.nodes ? (int64_t)s->svo.n_nodes : 0;
    case 11: return s->svo.nodes ? (int64_t)s->n_leaves : 0;
    case 12: return s->svo.nodes ? s->svo.depth : 0;
    case 13: return s->target.vox != nullptr;
    case 14: return s->is_svo ? 0 : dda_loop_of(s->target);
    case 15: return s->has_low ? dda_loop_of(s->low) : 0;
    case 16: return s->is_svo ? s->svo.use_blocks : 0;
    case 17: return s->is_svo ? 0 : (int64_t)s->target.n_filled;
    case 18: return s->is_svo ? (s->svo.use_blocks && s->svo.use_rank) : 0;
    default: return -1;
    }
}

#if VK_COUNT
// measurement build only (make variants, `count`): event counters of the whole-traversal loop
int volly_gpu_debug_counters(unsigned long long *out16, int reset)
{
    if (out16) HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(vk::vk_counters), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long zero[16] = {};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(vk::vk_counters), zero, sizeof zero));
    }
    return VOLLY_OK;
}
#endif

void volly_gpu_free(volly_gpu_scene *scene) { destroy(scene); }

} // extern "C"
