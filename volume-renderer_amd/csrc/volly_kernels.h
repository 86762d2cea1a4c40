// volly_kernels.h -- gfx950 device code of the primary-ray traversal path.
//
// What the kernels compute is fixed by the reference (file:line cited at each
// function); how they compute it is MI355X-first:
//   * one ray per lane, one 64-lane wavefront per 8x8 pixel tile (32x2 for launches of naive
//     frames only, which are bound by the output path), one wavefront per workgroup: a wave
//     retires as soon as its 64 rays are done; tiles whose rays provably miss both grids
//     write their miss values without setting a ray up (per tile row a span of columns, sky_span
//     below, from the projected boxes' rectangle and hull edges: sky_rect / sky_hull_edges, volly_gpu.hip);
//   * the dense grids are traversed through a derived 1-bit occupancy structure
//     (4x4x4-voxel bricks, one 64-bit word each) kept beside the reference's 8-byte
//     voxels; the 8-byte Voxel (voxel.h:13-16) is fetched once, at the hit.  A
//     lane keeps the current brick word in registers, so most DDA steps touch
//     no memory at all; the traversal loops are assembly that nothing leaves (hits and grid
//     exits are recorded under exec masks), a lane in empty space watches the largest empty
//     aligned cube around it, and wavefronts deep inside such cubes run steps blind;
//   * every float operation is the reference's, in the reference's order, unfused
//     (the file is compiled with -ffp-contract=off): hit index, step counts, end
//     point and depth are bit-identical to the CPU restatement;
//   * per-pixel ray directions are the reference's serial "+= xdiff" sums
//     (main.cpp:282-286): a small pre-pass stores the running sum every 8 pixels,
//     each lane finishes with at most 7 adds.
#ifndef VOLLY_KERNELS_H
#define VOLLY_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "volly_gpu.h"

// Build-time switches.  The shipped library is built with the defaults; the others
// exist so that each design choice can be measured against its alternative in one
// process (profiles/ab_bench.py, Makefile target `variants`):
//   VK_FAST_DDA=0     always use the fully checked DDA loop
//   VK_OCC_LEVELS=1   one occupancy level (4^3 bricks) instead of two (bricks + 16^3 regions)
//   VK_SVO_TABLE=0    locate octree cells by walking down from the root instead of starting at
//                     the direct-index table of the upper levels
//   VK_ASM_SVO=0      the compiled loop over the flattened octree (trace_svo_blocks)
//   VK_SVO_BLOCKS=0   no occupancy blocks for the lowest three octree levels: every locate walks the node chain
//   VK_XCD_GROUPS=0   tiles in dispatch order: neighbouring 8 x 8 tiles on different XCDs
//   VK_SKY_BLOCKS=0   sky tiles store their own pixels (eight 32-byte segments per store) instead of whole 32 x 8 blocks
//   VK_CKPT_STRIDE=4  a direction checkpoint every 4 pixels instead of 8 (at most 3 instead of 7
//                     serial adds per lane and component, twice the checkpoint traffic: measured
//                     1.4 % slower on config 2, 4 % on config 1)
//   VK_ASM_QUIET=0    the compiled region loop only (no assembly loop for the quiet steps)
//   VK_RUN_AHEAD=K    steps the region loop takes without looking when every lane is deep inside its empty cube (0: never)
//   VK_ASM_EXTRA=0    the compiled loop for the serial adds past a checkpoint
//   VK_ASM_BRICKS=0   the compiled loop for grids without the region level
//   VK_SHORT_DIV=0    the compiler's full IEEE expansions of 1/d, sqrt and /255 everywhere
//   VK_WAVE_BLOCK=N   N wavefronts (N tiles side by side) per workgroup, each storing its tile
//                     straight from registers: no LDS, no barrier, 1/N of the workgroup dispatches
// Designs that were measured and rejected in round 1 -- sampling the 8-byte voxels at every step,
// a 32x2 wavefront tile, output rows staged through LDS, an XCD-contiguous block order, several
// tiles per wavefront, the top of the octree in LDS -- are no longer in this file: DESIGN.md has their
// numbers, profiles/r01_ab/ the logs, the history (commit cfca876) the code.  The ray-refill and parked-ray
// frame kernels (volly_refill.h, volly_requeue.h: measured, not adopted) left the tree in round 3: commit 732586d holds them.  Round 2's brick layout with a 2x2x2 block of bricks per
// 64-byte line (-31 % L2 reads, +2.4 % time on the noise volume; profiles/r02_ab/ab_brick_blocks_c3.log)
// is in commit e1ece63; round 1's assembly loops, which handed hits and grid exits to compiled code, are
// in commit 7641e49 (profiles/r02_ab/r02_ab_whole_c*.log).
#ifndef VK_FAST_DDA
#define VK_FAST_DDA 1
#endif
#ifndef VK_OCC_LEVELS
#define VK_OCC_LEVELS 2             // 1: bricks only; 2: bricks + one word per 4x4x4 bricks
#endif
#ifndef VK_SVO_TABLE
#define VK_SVO_TABLE 1              // 0: walk down from the root at every iteration
#endif
#ifndef VK_SVO_BLOCKS
#define VK_SVO_BLOCKS VK_SVO_TABLE  // 0: locate every cell through the node chain below the table (trace_svo only)
#endif
#ifndef VK_WAVE_BLOCK
#define VK_WAVE_BLOCK 1
#endif
#ifndef VK_SHORT_DIV
#define VK_SHORT_DIV 1
#endif
#ifndef VK_ASM_QUIET
#define VK_ASM_QUIET 1
#endif
#ifndef VK_ASM_EXTRA
#define VK_ASM_EXTRA 1
#endif
#ifndef VK_ASM_BRICKS
#define VK_ASM_BRICKS 1
#endif
#ifndef VK_CKPT_STRIDE
#define VK_CKPT_STRIDE 8
#endif
#ifndef VK_XCD_GROUPS
#define VK_XCD_GROUPS 4             // tiles side by side per XCD; 0: tile = blockIdx.x (a row chunk's four tiles stored through four XCDs' L2s)
#endif
#ifndef VK_SKY_BLOCKS
#define VK_SKY_BLOCKS 1             // 0: every sky tile stores its own 8 x 8 pixels
#endif

static_assert(VK_WAVE_BLOCK >= 1, "wavefronts per workgroup");

namespace vk {

constexpr int kCkpt = VK_CKPT_STRIDE;   // a direction checkpoint every kCkpt pixels of a row (at most kCkpt - 1 adds per lane)
// The wavefront's tile in the frame kernel is TW x 64/TW pixels (template parameter of render_frames):
// 8 x 8 for frames with a beam pass -- the most compact ray bundle, fewest attention steps -- and 32 x 2 for
// launches of naive frames only, which are bound by the output path: 128-byte instead of 32-byte row
// segments per store (0.64 -> 0.52 ms on the headline frames in naive mode, profiles/r02_ab/ab_tile_shapes.log;
// in beam mode 16 x 4 is neutral and 32 x 2 is 5-12 % slower).
constexpr int kTileBeam = 8, kTileNaive = 32;       // (64 x 1 for naive launches measured slower: profiles/r03_ab/r03_ab_naive_only.log)
constexpr int kThreads = 256;
constexpr int kFrameThreads = 64 * VK_WAVE_BLOCK;   // workgroup size of render_frames

struct DenseDev {
    const uint64_t *occ;            // one word per 4x4x4 brick, bit = x&3 | (y&3)<<2 | (z&3)<<4
    const uint64_t *occ2;           // one word per 4x4x4 bricks (16^3 voxels): bit set iff that brick has a voxel
    const uint4 *occ2x;             // per region {occ2 word, key mask of the largest empty aligned cube (16^3 .. 128^3 voxels) around it, 0}:
                                    // what the whole-traversal loop loads on a region change (build_region_cubes)
    const uint2 *vox;               // the reference's 8-byte voxels, x + sx*(y + sy*z)
    int sx, sy, sz;
    int px, py, pz;                 // the sizes rounded up to whole 16^3 regions: what occ / occ2 / occ2x cover.  The cells of
                                    // the rim between (sx, sy, sz) and (px, py, pz) are SET in occ: a sample there looks like a
                                    // hit to the fast loops, which is how they notice that the ray has left the grid (rim_exit)
    int nbx, nby, nbz;              // bricks per axis (px / 4 ...)
    int nrx, nry, nrz;              // 16^3 regions per axis (px / 16 ...)
    int use_regions;                // consult occ2 (set at upload when most regions are empty; useless for noise)
    float fsx, fsy, fsz;
    uint32_t cap;                   // iteration cap 4*(sx+sy+sz)+8 (the reference has none, main.cpp:121)
    uint64_t n_filled;              // voxels with a > 0 (counted at upload; the host chooses the frame kernel by the filling)
};

struct SvoDev {
    const uint2 *nodes;             // .x = child_offset, .y = valid_mask | leaf_mask<<8
    const uint2 *leaves;
    const uint32_t *table;          // upper levels flattened: (2^tbits)^3 entries, see build_svo_table
    const uint64_t *blocks;         // lowest three levels flattened: 8 occupancy words (one per 4^3 octant, the dense bricks' bit
                                    // order) per level-3 node (an 8^3 cell), indexed by node - l3_first; see build_svo_blocks
    const uint32_t *word_base;      // per block word: index in leaves[] of the first leaf under it (build_svo_blocks)
    uint32_t l3_first;              // index of the first level-3 node (breadth-first order: the nodes of a level are contiguous)
    int use_rank;                   // the leaves under each block word are stored one after the other in tree order: the leaf
                                    // of a hit is word_base + its rank among the word's bits (svo_leaf_by_rank), no walk
    int use_blocks;                 // the table ends at 8^3 cells (tshift == 3), blocks are built, res is a multiple of 8: trace_svo_blocks
    int tbits, tshift;              // table cell = octree cell of 2^tshift voxels; tbits + tshift = depth
    uint32_t n_nodes;
    int depth, res;
    float fres;
    uint32_t cap;
};

struct OutDev {
    uint32_t *rgba8, *hit_index, *steps_lo, *steps_hi;
    float *depth;
    float *pt_end;                  // 3 floats per ray
};

#ifndef VOLLY_TYPES_ONLY            // (the other translation units of the library need only the types above)

struct Ray {
    float dx, dy, dz;               // direction
    float ix, iy, iz;               // 1/direction
    float hx, hy, hz;               // heaviside: 0 for a negative component, else 1
    float ex, ey, ez;               // sign * eps
    bool regular;                   // every component a normal float with 2^-100 <= |d| <= 2
};

// ---- correctly rounded 1/d, sqrt(x), n/255 in a few instructions ------------------
// The compiler expands an IEEE division into 10 and a square root into 17 vector
// instructions, most of them there for denormal / huge / special operands.  For operands in
// [2^-100, 2^100] the short sequences below give the SAME bits: tools/exhaustive_math.hip
// compares them with the full expansions on every one of the 2^32 binary32 patterns on the
// device (tests/test_gpu_parity.py::test_short_division_sequences_are_exact), 0 mismatches in
// that range.  Operands outside it take the full expansion.
constexpr float kFastLo = 7.888609052210118e-31f;   // 2^-100
constexpr float kFastHi = 1.2676506002282294e30f;   // 2^100

__device__ __forceinline__ bool in_fast_range(float x)
{
    float a = fabsf(x);
    return (a >= kFastLo) & (a <= kFastHi);              // NaN fails
}

__device__ __forceinline__ float rcp_in_range(float d)
{
    float r = __builtin_amdgcn_rcpf(d);                   // 1 ulp
    float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

__device__ __forceinline__ float sqrt_in_range(float x)  // x > 0
{
    float s = __builtin_amdgcn_sqrtf(x);                  // 1 ulp: the result is s or a neighbour
    float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    float em = __builtin_fmaf(-sm, s, x), ep = __builtin_fmaf(-sp, s, x);
    float r = (em <= 0.0f) ? sm : s;
    return (ep > 0.0f) ? sp : r;
}

// x / 255.0f for x = 0, 1, ..., 255 (all 256 cases checked by the same tool)
__device__ __forceinline__ float over_255(float x)
{
#if VK_SHORT_DIV
    const float r = 1.0f / 255.0f;
    float q = x * r;
    float e = __builtin_fmaf(-255.0f, q, x);
    return __builtin_fmaf(e, r, q);
#else
    return x / 255.0f;
#endif
}

// main.cpp:96-102
__device__ __forceinline__ Ray make_ray(float dx, float dy, float dz)
{
    const float eps = 0.00030f;
    Ray r;
    r.dx = dx; r.dy = dy; r.dz = dz;
    {
        const float lo = kFastLo, hi = 2.0f;
        float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
        r.regular = (ax >= lo) & (ax <= hi) & (ay >= lo) & (ay <= hi) & (az >= lo) & (az <= hi);     // NaN fails
    }
    if (VK_SHORT_DIV && r.regular) {
        r.ix = rcp_in_range(dx); r.iy = rcp_in_range(dy); r.iz = rcp_in_range(dz);
    } else {
        r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
    }
    // heaviside = round((sign+1)/2): 0 for sign -1, 1 for sign +1 and, rounding half away from zero, for sign 0
    r.hx = (dx < 0.0f) ? 0.0f : 1.0f;
    r.hy = (dy < 0.0f) ? 0.0f : 1.0f;
    r.hz = (dz < 0.0f) ? 0.0f : 1.0f;
    if (__all(r.regular)) {                               // no zero, no NaN: sign(d) * eps is eps with d's sign bit (one v_bfi_b32 each)
        r.ex = __builtin_copysignf(eps, dx); r.ey = __builtin_copysignf(eps, dy); r.ez = __builtin_copysignf(eps, dz);
        return r;
    }
    float sx = (float)((0.0f < dx) - (dx < 0.0f));
    float sy = (float)((0.0f < dy) - (dy < 0.0f));
    float sz = (float)((0.0f < dz) - (dz < 0.0f));
    r.ex = sx * eps; r.ey = sy * eps; r.ez = sz * eps;
    return r;
}

// main.cpp:105-108
__device__ __forceinline__ void enter_volume(const Ray &r, float ox, float oy, float oz, float vx, float vy, float vz,
                                             float &cx, float &cy, float &cz)
{
    float ndx = (-(vx * (r.hx - 1.0f) + ox)) * r.ix;
    float ndy = (-(vy * (r.hy - 1.0f) + oy)) * r.iy;
    float ndz = (-(vz * (r.hz - 1.0f) + oz)) * r.iz;
    float t0 = fmaxf(fmaxf(fmaxf(ndx, ndy), ndz), 0.0f);
    cx = t0 * r.dx + ox;
    cy = t0 * r.dy + oy;
    cz = t0 * r.dz + oz;
}

// main.cpp:149-153, unit cells
__device__ __forceinline__ void advance_unit(const Ray &r, float px, float py, float pz, float &cx, float &cy, float &cz)
{
    float ndx = ((floorf(px) + r.hx) - cx) * r.ix;
    float ndy = ((floorf(py) + r.hy) - cy) * r.iy;
    float ndz = ((floorf(pz) + r.hz) - cz) * r.iz;
    float m = fminf(fminf(ndx, ndy), ndz);
    cx = m * r.dx + cx;
    cy = m * r.dy + cy;
    cz = m * r.dz + cz;
}

// the same step across an empty octree cell of 2^k voxels (s = 2^k, is = 2^-k: both scalings are exact)
__device__ __forceinline__ void advance_cell(const Ray &r, float s, float is, float px, float py, float pz,
                                             float &cx, float &cy, float &cz)
{
    float ndx = ((floorf(px * is) * s + r.hx * s) - cx) * r.ix;
    float ndy = ((floorf(py * is) * s + r.hy * s) - cy) * r.iy;
    float ndz = ((floorf(pz * is) * s + r.hz * s) - cz) * r.iz;
    float m = fminf(fminf(ndx, ndy), ndz);
    cx = m * r.dx + cx;
    cy = m * r.dy + cy;
    cz = m * r.dz + cz;
}

struct Hit {
    float cx, cy, cz;               // ptEnd (cur at the hit or at the exit)
    uint32_t steps;
    uint32_t index;                 // dense linear index of the hit voxel, VOLLY_MISS_INDEX on a miss
    uint32_t payload;               // dense: = index; octree: index into leaves[]
};

// raycast_naiive_vec4<false> (main.cpp:88-161) on the occupancy bricks of a dense grid.
// The reference tests roundPos = (int)curPlusEpsilon against [0,size): for the
// x86 conversion (truncation; NaN / out of range -> INT_MIN) that is exactly
// -1 < curPlusEpsilon < size, NaN failing, which is what is tested here.
// This is the fully checked loop: it is right for every input, including direction
// components that are zero, denormal or NaN (1/0 = inf, 0*inf = NaN inside the DDA).
template <bool SOLID = false>
__device__ __forceinline__ Hit trace_dense_checked(const DenseDev &g, const Ray &r, float ox, float oy, float oz)
{
    Hit h;
    enter_volume(r, ox, oy, oz, g.fsx, g.fsy, g.fsz, h.cx, h.cy, h.cz);
    h.steps = 0;
    h.index = VOLLY_MISS_INDEX;
    h.payload = VOLLY_MISS_INDEX;
    int last_brick = -1;
    uint64_t bits = 0;
    for (;;) {
        float px = h.cx + r.ex, py = h.cy + r.ey, pz = h.cz + r.ez;
        bool inside = (px > -1.0f) & (px < g.fsx) & (py > -1.0f) & (py < g.fsy) & (pz > -1.0f) & (pz < g.fsz);
        if (!inside) break;
        if (h.steps >= g.cap) break;
        int x = (int)px, y = (int)py, z = (int)pz;       // truncation toward zero, like glm::ivec4(vec4)
        ++h.steps;
        int brick = (x >> 2) + g.nbx * ((y >> 2) + g.nby * (z >> 2));
        if (brick != last_brick) {
            bits = g.occ[brick];
            last_brick = brick;
        }
        uint32_t bit = (uint32_t)(x & 3) | ((uint32_t)(y & 3) << 2) | ((uint32_t)(z & 3) << 4);
        if ((((bits >> bit) & 1ull) != 0) != SOLID) {    // main.cpp:139-147: solid looks for the first EMPTY voxel
            h.index = (uint32_t)x + (uint32_t)g.sx * ((uint32_t)y + (uint32_t)g.sy * (uint32_t)z);
            h.payload = h.index;
            break;
        }
        advance_unit(r, px, py, pz, h.cx, h.cy, h.cz);
    }
    return h;
}

// A ray is "regular" when every direction component is a normal float of magnitude
// >= 2^-100 (so 1/d is finite).  For such a ray, once one bounds check has passed,
// every quantity of the DDA stays finite and each step moves every coordinate by at
// most one cell; the loop below relies on both.
__device__ __forceinline__ bool ray_is_regular(const Ray &r) { return r.regular; }

__device__ __forceinline__ bool grid_is_regular(const DenseDev &g)
{
    // coordinates that fit 10 bits (the occupancy levels always cover whole 16^3 regions: DenseDev::px)
    return g.px <= 1024 && g.py <= 1024 && g.pz <= 1024;
}

// The fast loops run on the padded grid (px, py, pz), whose rim cells are set.  A "hit" they report in a rim cell is
// the sample at which the reference's bounds test fails (main.cpp:126-135): the ray left the grid there, one step
// count less than a hit's (which counts its own sample, main.cpp:137), no voxel.  Returns true for such a cell.
__device__ __forceinline__ bool rim_exit(const DenseDev &g, uint32_t x, uint32_t y, uint32_t z)
{
    return !((x < (uint32_t)g.sx) & (y < (uint32_t)g.sy) & (z < (uint32_t)g.sz));
}

// bit index x&3 | (y&3)<<2 | (z&3)<<4 of a cell packed as x | y<<10 | z<<20 (fields already
// masked to two bits): one 24-bit multiply moves the fields at bits 0, 10, 20 to bits 18..23
__device__ __forceinline__ uint32_t gather_bits(uint32_t masked)
{
    return (__umul24(masked, 0x00040404u) >> 18) & 63u;
}

// x + nx * (y + ny * z) for the bricks / regions of a regular grid (at most 256 per axis, so the
// products stay below 2^24 and the 24-bit multiply-add does: 64-bit multiplies are quarter rate)
__device__ __forceinline__ uint32_t cell_index24(int x, int y, int z, int nx, int ny)
{
    return __umul24((uint32_t)nx, __umul24((uint32_t)ny, (uint32_t)z) + (uint32_t)y) + (uint32_t)x;
}

// x | y << 10 | z << 20 in two v_lshl_or_b32 (left alone the compiler shifts twice and ORs three
// values: one instruction more in a loop that is bound by instruction issue)
__device__ __forceinline__ uint32_t pack_cell(int x, int y, int z)
{
    uint32_t t, c;
    asm("v_lshl_or_b32 %0, %1, 10, %2" : "=v"(t) : "v"(y), "v"(x));
    asm("v_lshl_or_b32 %0, %1, 20, %2" : "=v"(c) : "v"(z), "v"(t));
    return c;
}

__device__ __forceinline__ bool test_word(uint32_t w_lo, uint32_t w_hi, uint32_t bit)
{
    uint32_t w = (bit & 32u) ? w_hi : w_lo;
    return ((w >> (bit & 31u)) & 1u) != 0;
}

// The same loop for regular rays in regular grids, with less work and far fewer loads per
// step; every decision it takes is the one trace_dense_checked takes on the same values.
//   * the cell is packed as c = x | y<<10 | z<<20.  Clearing the low 4 (2) bits of each field
//     names the 16^3 region (4^3 brick) the cell lies in.  While the region does not change
//     the cell is inside the grid (regions are whole), so the bounds test runs only on a region
//     change.  A cell outside the grid always changes the key: a coordinate of -1 sets bits
//     30-31, a coordinate equal to the size leaves the last region's field (a step moves each
//     coordinate by one cell at most);
//   * a lane keeps the region word (which of its 64 bricks hold voxels) and the word of the
//     last non-empty brick in registers: steps through empty bricks read no memory, and a
//     wavefront of neighbouring rays reloads only when it enters a new region or a brick with
//     content -- the dependent-load latency that otherwise paces the loop.
template <bool REGIONS>
__device__ __forceinline__ Hit trace_dense_regular(const DenseDev &g, const Ray &r, float ox, float oy, float oz)
{
    Hit h;
    enter_volume(r, ox, oy, oz, g.fsx, g.fsy, g.fsz, h.cx, h.cy, h.cz);
    h.steps = 0;
    h.index = VOLLY_MISS_INDEX;
    h.payload = VOLLY_MISS_INDEX;
    if (!(fabsf(h.cx) < 3.0e38f && fabsf(h.cy) < 3.0e38f && fabsf(h.cz) < 3.0e38f)) return h;   // inf/NaN entry: the first bounds test fails
    uint32_t last_brick = 0xFFFFFFFFu;                   // no brick yet: valid keys have bits 0-1 clear
    uint32_t w_lo = 0, w_hi = 0;
    uint32_t last_region = 0xFFFFFFFFu;
    uint32_t r_lo = 0, r_hi = 0;
    uint32_t r_any = 0;                                  // the current region holds at least one voxel
    uint32_t live = 1;
    // A lane samples once per loop iteration from the first one until it leaves the loop, so
    // its step count is the iteration number at which it leaves: no per-lane counter, and the
    // iteration cap is a wave-uniform test.  Lane state changes only inside the rare blocks.
    for (uint32_t iter = 0;; ++iter) {
        float px = h.cx + r.ex, py = h.cy + r.ey, pz = h.cz + r.ez;
        int x = (int)px, y = (int)py, z = (int)pz;       // truncation toward zero, saturating
        uint32_t c = pack_cell(x, y, z);
        if (iter >= g.cap) { h.steps = iter; break; }    // uniform
        uint32_t region = c & 0xFF0FC3F0u;
        if (REGIONS && region != last_region) {
            bool inside = ((uint32_t)x < (uint32_t)g.px) & ((uint32_t)y < (uint32_t)g.py) & ((uint32_t)z < (uint32_t)g.pz);
            if (inside) {
                uint64_t bits = g.occ2[cell_index24(x >> 4, y >> 4, z >> 4, g.nrx, g.nry)];
                r_lo = (uint32_t)bits;
                r_hi = (uint32_t)(bits >> 32);
                r_any = (r_lo | r_hi) != 0u;
                last_region = region;
            } else {                                     // left the grid: main.cpp:126-135
                r_any = 0;
                live = 0;
                h.steps = iter;
            }
        }
        if (!REGIONS || r_any) {
            if (!REGIONS || test_word(r_lo, r_hi, gather_bits((c >> 2) & 0x00300C03u))) {      // the brick holds voxels
                uint32_t brick = c & 0xFFCFF3FCu;
                bool inside = true;
                if (brick != last_brick) {
                    if (!REGIONS) inside = ((uint32_t)x < (uint32_t)g.px) & ((uint32_t)y < (uint32_t)g.py) & ((uint32_t)z < (uint32_t)g.pz);
                    if (inside) {
                        uint64_t bits = g.occ[cell_index24(x >> 2, y >> 2, z >> 2, g.nbx, g.nby)];
                        w_lo = (uint32_t)bits;
                        w_hi = (uint32_t)(bits >> 32);
                        last_brick = brick;
                    } else {
                        live = 0;
                        h.steps = iter;
                    }
                }
                if (inside && test_word(w_lo, w_hi, gather_bits(c & 0x00300C03u))) {   // main.cpp:139-142
                    live = 0;
                    if (rim_exit(g, (uint32_t)x, (uint32_t)y, (uint32_t)z)) {
                        h.steps = iter;
                    } else {
                        h.steps = iter + 1;
                        h.index = (uint32_t)x + (uint32_t)g.sx * ((uint32_t)y + (uint32_t)g.sy * (uint32_t)z);
                        h.payload = h.index;
                    }
                }
            }
        }
        if (!live) break;
        advance_unit(r, px, py, pz, h.cx, h.cy, h.cz);
    }
    return h;
}

// trace_dense_regular<true> in assembly, with nothing left to compiled code (VK_ASM_QUIET).
//
// A lane is *quiet* while (cell & wmask) == watch: watch / wmask name the largest empty aligned cube
// (16^3 .. 128^3 voxels) around its region while that region is empty, its 4^3 brick while the region
// holds voxels but the brick does not, and watch is 0xFFFFFFFF (which no key equals) inside a brick
// with voxels.  A step in which every live lane is quiet needs no per-lane test at all: 29 vector and
// 5 scalar instructions (35 and ~20 compiled).  Round 1's form of this loop (commit 7641e49 and before)
// left for a per-lane C++ block whenever a lane entered a brick that holds voxels or left the grid --
// about 95 vector instructions for the whole wavefront each time, ~1.5 M times per launch of the
// headline workload.  Here those events are handled in place, under exec masks of the loop's own, and
// a finished lane simply drops out of the loop's live mask:
//   * a lane whose new region lies outside the grid (main.cpp:126-135) records its step count;
//   * a lane in a brick that holds voxels loads the brick word when the brick is new to it and
//     tests its cell's bit every step (watch = none); a set bit is the hit (main.cpp:139-142):
//     step count and packed cell are recorded.
// The loop ends when no lane is live or at the iteration cap.  All per-lane traversal state lives
// in fixed registers (v44-v45 brick word, v56 cell, v57 region key, v58-v59 region word, v60
// empty-cube mask, v62 / v63 watch key and mask, v46 brick key) beside the temporaries v47-v55: 64 vector registers, still
// eight wavefronts per SIMD (tools/check_kernel_contract.py).  Region / brick byte offsets come
// from the grid's sizes alone (whole 16^3 regions: sy and sx are multiples of 16, so
// ((z>>4)*sy + (y&~15)) * sx is 256 * the region row index), which keeps the statement within
// the 30 operands an asm statement may take.
#ifndef VK_RUN_AHEAD
#define VK_RUN_AHEAD 4               // steps the whole-traversal loop takes without looking when every lane is deep inside its empty cube (0: never)
#endif
#ifndef VK_RUN_LONG
#define VK_RUN_LONG 16               // (VK_RUN_AHEAD << n) steps tried first after every attention step, by the same corner test; 0: off
#endif
#if VK_RUN_LONG == 0
#define VK_RUN_ARMED "1"
#else
#define VK_RUN_ARMED "2"
#if VK_RUN_LONG == 2 * VK_RUN_AHEAD
#define VK_RUN_LONG_SHIFT "1"
#elif VK_RUN_LONG == 4 * VK_RUN_AHEAD
#define VK_RUN_LONG_SHIFT "2"
#elif VK_RUN_LONG == 8 * VK_RUN_AHEAD
#define VK_RUN_LONG_SHIFT "3"
#else
#error "VK_RUN_LONG must be 2, 4 or 8 times VK_RUN_AHEAD"
#endif
#endif
#define VK_STR2(x) #x
#define VK_STR(x) VK_STR2(x)
// advance_unit (main.cpp:149-153) on the sample point in v47-v49, for the loops below
#define VK_ADVANCE_UNIT \
        "v_floor_f32 v53, v47\n\t" \
        "v_floor_f32 v54, v48\n\t" \
        "v_floor_f32 v55, v49\n\t" \
        "v_add_f32 v53, v53, %[hx]\n\t" \
        "v_add_f32 v54, v54, %[hy]\n\t" \
        "v_add_f32 v55, v55, %[hz]\n\t" \
        "v_sub_f32 v53, v53, %[cx]\n\t" \
        "v_sub_f32 v54, v54, %[cy]\n\t" \
        "v_sub_f32 v55, v55, %[cz]\n\t" \
        "v_mul_f32 v53, v53, %[ix]\n\t" \
        "v_mul_f32 v54, v54, %[iy]\n\t" \
        "v_mul_f32 v55, v55, %[iz]\n\t" \
        "v_min3_f32 v53, v53, v54, v55\n\t" \
        "v_mul_f32 v54, v53, %[dx]\n\t" \
        "v_mul_f32 v55, v53, %[dy]\n\t" \
        "v_mul_f32 v53, v53, %[dz]\n\t" \
        "v_add_f32 %[cx], v54, %[cx]\n\t" \
        "v_add_f32 %[cy], v55, %[cy]\n\t" \
        "v_add_f32 %[cz], v53, %[cz]\n\t"
// one more step without cell, key or test: the sample point, then the advance
#define VK_BLIND_STEP \
        "s_add_u32 m0, m0, 1\n\t" \
        "v_add_f32 v47, %[cx], %[ex]\n\t" \
        "v_add_f32 v48, %[cy], %[ey]\n\t" \
        "v_add_f32 v49, %[cz], %[ez]\n\t" \
        VK_ADVANCE_UNIT
#if VK_RUN_AHEAD == 2
#define VK_RUN_AHEAD_MORE VK_BLIND_STEP
#elif VK_RUN_AHEAD == 3
#define VK_RUN_AHEAD_MORE VK_BLIND_STEP VK_BLIND_STEP
#elif VK_RUN_AHEAD == 4
#define VK_RUN_AHEAD_MORE VK_BLIND_STEP VK_BLIND_STEP VK_BLIND_STEP
#elif VK_RUN_AHEAD == 6
#define VK_RUN_AHEAD_MORE VK_BLIND_STEP VK_BLIND_STEP VK_BLIND_STEP VK_BLIND_STEP VK_BLIND_STEP
#elif VK_RUN_AHEAD == 8
#define VK_RUN_AHEAD_MORE VK_BLIND_STEP VK_BLIND_STEP VK_BLIND_STEP VK_BLIND_STEP VK_BLIND_STEP VK_BLIND_STEP VK_BLIND_STEP
#else
#define VK_RUN_AHEAD_MORE
#endif
#ifndef VK_COUNT
#define VK_COUNT 0                   // 1: measurement build, the whole-traversal loop counts its events
#endif
#if VK_COUNT
#define VK_CNT(x) x
__device__ unsigned long long vk_counters[16];   // waves, wave-steps, attention steps, region changes, grid exits, steps with a lane in a region with voxels, ... in a brick with voxels, brick loads, hits, run-ahead blocks
#else
#define VK_CNT(x)
#endif
__device__ __forceinline__ Hit trace_dense_regions_whole_asm(const DenseDev &g, const Ray &r, float ox, float oy, float oz)
{
    Hit h;
    enter_volume(r, ox, oy, oz, g.fsx, g.fsy, g.fsz, h.cx, h.cy, h.cz);
    h.steps = 0;
    h.index = VOLLY_MISS_INDEX;
    h.payload = VOLLY_MISS_INDEX;
    if (!(fabsf(h.cx) < 3.0e38f && fabsf(h.cy) < 3.0e38f && fabsf(h.cz) < 3.0e38f)) return h;   // inf/NaN entry: the first bounds test fails
    const uint32_t cap = g.cap;
    const uint32_t sx = (uint32_t)g.px, sy = (uint32_t)g.py, sz = (uint32_t)g.pz;    // the loop's grid is the padded one
    const uint4 *occ2 = g.occ2x;
    const uint64_t *occ = g.occ;
    uint32_t steps = 0xFFFFFFFFu;                         // still live
    uint32_t hitc = 0xFFFFFFFFu;                          // packed cell of the hit
    uint64_t mask_e, mask_a, mask_b, mask_c;
    uint32_t tmp, tr;
    // the packed-cell offset of VK_RUN_AHEAD cells along the ray's octant (fields wrap: a field that leaves
    // [0, 1024) changes its own high bits, so the masked compare fails, which is the safe answer)
    const uint32_t k_ahead = (uint32_t)VK_RUN_AHEAD;
    const uint32_t ahead = (r.dx < 0.0f ? 0u - k_ahead : k_ahead) + ((r.dy < 0.0f ? 0u - k_ahead : k_ahead) << 10) +
                           ((r.dz < 0.0f ? 0u - k_ahead : k_ahead) << 20);
    asm volatile(
        "s_mov_b64 %[se], exec\n\t"
        VK_CNT("s_mov_b32 s88, 0\n\ts_mov_b32 s89, 0\n\ts_mov_b32 s90, 0\n\ts_mov_b32 s91, 0\n\ts_mov_b32 s92, 0\n\ts_mov_b32 s93, 0\n\ts_mov_b32 s94, 0\n\ts_mov_b32 s95, 0\n\t")
        "s_mov_b32 m0, 0\n\t"
        "s_mov_b32 %[tr], " VK_RUN_ARMED "\n\t"
        "v_mov_b32 v57, -1\n\t"                       // no region yet
        "v_mov_b32 v58, 0\n\t"
        "v_mov_b32 v59, 0\n\t"
        "v_mov_b32 v62, -1\n\t"                       // watch: a key no cell has
        "v_mov_b32 v63, 0xff0fc3f0\n\t"
        "v_mov_b32 v46, -1\n\t"                       // no brick yet
        "v_mov_b32 v44, 0\n\t"
        "v_mov_b32 v45, 0\n"
        "1:\n\t"                                      // ---- one step: sample point, cell, key (exec = live lanes)
        "v_add_f32 v47, %[cx], %[ex]\n\t"
        "v_add_f32 v48, %[cy], %[ey]\n\t"
        "v_add_f32 v49, %[cz], %[ez]\n\t"
        "v_cvt_i32_f32 v50, v47\n\t"
        "v_cvt_i32_f32 v51, v48\n\t"
        "v_cvt_i32_f32 v52, v49\n\t"
        "v_lshl_or_b32 v56, v51, 10, v50\n\t"
        "v_lshl_or_b32 v56, v52, 20, v56\n\t"
        "v_and_b32 v53, v63, v56\n\t"
        "v_cmp_ne_u32 vcc, v53, v62\n\t"
        "s_cbranch_vccnz 3f\n\t"
        "s_cmp_ge_u32 m0, %[cap]\n\t"
        "s_cbranch_scc1 9f\n\t"
#if VK_RUN_AHEAD
        // ---- every live lane is quiet.  If each one's cell VK_RUN_AHEAD steps further along its ray (a step moves
        // every coordinate by at most one cell, always in the ray's direction) still lies in its empty cube, the
        // next VK_RUN_AHEAD steps can hit nothing and cannot leave the grid: they run without cell, key or test.
        // `tr` remembers which tests are still worth making: 2 after an attention step (try VK_RUN_LONG steps, and
        // again after every run that passed), 1 once the long test has failed (some lane is within VK_RUN_LONG
        // cells of its cube's face: runs of VK_RUN_AHEAD), 0 once that has failed too (plain steps until the next
        // attention step).  Headline +1.7 %, configs 4 / 5 +2.5-3 % over short runs alone (r03_ab_runlong*.log).
        "s_cmp_eq_u32 %[tr], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
#if VK_RUN_LONG
        // the long run first (once per attention step: a failure leaves the short runs armed): the same test with
        // the offset of VK_RUN_LONG cells (the packed offset is linear: a shift multiplies every field)
        "s_cmp_lg_u32 %[tr], 2\n\t"
        "s_cbranch_scc1 12f\n\t"
        "v_lshlrev_b32 v53, " VK_RUN_LONG_SHIFT ", %[ahead]\n\t"
        "v_add_u32 v53, v56, v53\n\t"
        "v_and_b32 v53, v63, v53\n\t"
        "v_cmp_ne_u32 vcc, v53, v62\n\t"
        "s_cbranch_vccnz 13f\n\t"
        "s_add_u32 %[tmp], m0, " VK_STR(VK_RUN_LONG) "\n\t"
        "s_cmp_gt_u32 %[tmp], %[cap]\n\t"
        "s_cbranch_scc1 13f\n\t"
        "10:\n\t"                                    // VK_RUN_AHEAD steps per turn, until iteration tmp
        VK_ADVANCE_UNIT
        VK_RUN_AHEAD_MORE
        "s_add_u32 m0, m0, 1\n\t"
        "s_cmp_ge_u32 m0, %[tmp]\n\t"
        "s_cbranch_scc1 1b\n\t"
        "v_add_f32 v47, %[cx], %[ex]\n\t"
        "v_add_f32 v48, %[cy], %[ey]\n\t"
        "v_add_f32 v49, %[cz], %[ez]\n\t"
        "s_branch 10b\n"
        "13:\n\t"
        "s_mov_b32 %[tr], 1\n"
        "12:\n\t"
#endif
        "v_add_u32 v53, v56, %[ahead]\n\t"
        "v_and_b32 v53, v63, v53\n\t"
        "v_cmp_ne_u32 vcc, v53, v62\n\t"
        "s_cbranch_vccnz 8f\n\t"
        "s_add_u32 %[tmp], m0, " VK_STR(VK_RUN_AHEAD) "\n\t"
        "s_cmp_gt_u32 %[tmp], %[cap]\n\t"
        "s_cbranch_scc1 2f\n\t"
        VK_CNT("s_add_u32 s95, s95, 1\n\t")
        VK_ADVANCE_UNIT
        VK_RUN_AHEAD_MORE
        "s_add_u32 m0, m0, 1\n\t"
        "s_branch 1b\n"
        "8:\n\t"
        "s_mov_b32 %[tr], 0\n"
#endif
        "2:\n\t"                                      // ---- advance_unit (main.cpp:149-153), every live lane
        VK_ADVANCE_UNIT
        "s_add_u32 m0, m0, 1\n\t"
        "s_branch 1b\n"
        "3:\n\t"                                      // ---- vcc = the lanes that are not quiet
        "s_cmp_ge_u32 m0, %[cap]\n\t"
        "s_cbranch_scc1 9f\n\t"
        "s_mov_b64 %[sa], exec\n\t"                   // sa: the live lanes
        VK_CNT("s_add_u32 s88, s88, 1\n\t")
        "s_mov_b64 exec, vcc\n\t"
        "v_and_b32 v53, 0xff0fc3f0, v56\n\t"          // region key
        "v_cmp_ne_u32 vcc, v53, v57\n\t"
        "s_and_saveexec_b64 %[sb], vcc\n\t"           // sb: the lanes to look at; exec: those in a new region
        "s_cbranch_execz 5f\n\t"
        VK_CNT("s_add_u32 s89, s89, 1\n\t")
        "v_cmp_gt_u32 vcc, %[sx], v50\n\t"
        "v_cmp_gt_u32 %[sc], %[sy], v51\n\t"
        "s_and_b64 vcc, vcc, %[sc]\n\t"
        "v_cmp_gt_u32 %[sc], %[sz], v52\n\t"
        "s_and_b64 vcc, vcc, %[sc]\n\t"               // vcc: new region inside the grid
        "s_andn2_b64 %[sc], exec, vcc\n\t"            // sc: new region outside the grid
        "s_cbranch_scc0 4f\n\t"
        "s_mov_b64 exec, %[sc]\n\t"                   // ---- lanes leaving the grid: main.cpp:126-135
        VK_CNT("s_add_u32 s90, s90, 1\n\t")
        "s_mov_b32 %[tmp], m0\n\t"
        "v_mov_b32 %[steps], %[tmp]\n\t"
        "s_andn2_b64 %[sa], %[sa], %[sc]\n\t"         // no longer live
        "s_andn2_b64 %[sb], %[sb], %[sc]\n"           // nor to be looked at
        "4:\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "s_cbranch_execz 5f\n\t"
        "v_lshrrev_b32 v54, 4, v52\n\t"               // byte offset of the region entry: 16 * (x>>4 + nrx * (y>>4 + nry * (z>>4)))
        "v_and_b32 v55, 0xfffffff0, v51\n\t"
        "v_mad_u32_u24 v54, v54, %[sy], v55\n\t"      // 16 * (y>>4 + nry * (z>>4))
        "v_mul_u32_u24 v54, v54, %[sx]\n\t"           // 256 * nrx * (...)
        "v_lshrrev_b32 v54, 4, v54\n\t"
        "v_lshrrev_b32 v55, 4, v50\n\t"
        "v_lshl_add_u32 v54, v55, 4, v54\n\t"
        "global_load_dwordx4 v[58:61], v54, %[occ2]\n\t"   // region word, and the key mask of the empty cube around an empty region
        "v_mov_b32 v57, v53\n\t"
        "s_waitcnt vmcnt(0)\n"
        "5:\n\t"
        "s_mov_b64 exec, %[sb]\n\t"                   // the lanes to look at, all inside the grid
        "s_cbranch_execz 7f\n\t"
        "v_and_b32 v54, v60, v56\n\t"                 // key of the largest empty aligned cube (16^3 .. 128^3) around the region
        "v_cmp_eq_u64 vcc, 0, v[58:59]\n\t"           // empty region: quiet until the lane leaves that cube
        "s_nop 1\n\t"                                 // gfx950: a vector compare's vcc needs two wait states before a vector instruction reads it
        "v_cndmask_b32 v62, v62, v54, vcc\n\t"
        "v_cndmask_b32 v63, v63, v60, vcc\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"             // lanes in a region with voxels: look at the brick's bit
        "s_cbranch_execz 7f\n\t"
        VK_CNT("s_add_u32 s91, s91, 1\n\t")
        "v_lshrrev_b32 v54, 2, v56\n\t"
        "v_and_b32 v54, 0x300c03, v54\n\t"
        "v_mul_u32_u24 v54, 0x40404, v54\n\t"
        "v_bfe_u32 v54, v54, 18, 6\n\t"
        "v_lshrrev_b64 v[54:55], v54, v[58:59]\n\t"
        "v_and_b32 v54, 1, v54\n\t"
        "v_cmp_eq_u32 vcc, 1, v54\n\t"                // vcc: the brick holds voxels
        "s_andn2_b64 %[sc], exec, vcc\n\t"
        "s_mov_b64 %[sb], vcc\n\t"
        "s_mov_b64 exec, %[sc]\n\t"                   // an empty brick: quiet until the lane leaves it
        "v_and_b32 v62, 0xffcff3fc, v56\n\t"
        "v_mov_b32 v63, 0xffcff3fc\n\t"
        "s_mov_b64 exec, %[sb]\n\t"                   // ---- lanes in a brick that holds voxels
        "s_cbranch_execz 7f\n\t"
        VK_CNT("s_add_u32 s92, s92, 1\n\t")
        "v_mov_b32 v62, -1\n\t"                       // every step is looked at
        "v_and_b32 v53, 0xffcff3fc, v56\n\t"          // brick key
        "v_cmp_ne_u32 vcc, v53, v46\n\t"
        "s_and_saveexec_b64 %[sc], vcc\n\t"           // sc: lanes in such a brick; exec: those that need its word
        "s_cbranch_execz 6f\n\t"
        VK_CNT("s_add_u32 s93, s93, 1\n\t")
        "v_lshrrev_b32 v54, 2, v52\n\t"               // byte offset of the brick word: 8 * (x>>2 + nbx * (y>>2 + nby * (z>>2)))
        "v_and_b32 v55, 0xfffffffc, v51\n\t"
        "v_mad_u32_u24 v54, v54, %[sy], v55\n\t"      // 4 * (y>>2 + nby * (z>>2))
        "v_mul_u32_u24 v54, v54, %[sx]\n\t"           // 16 * nbx * (...)
        "v_lshrrev_b32 v54, 1, v54\n\t"
        "v_lshrrev_b32 v55, 2, v50\n\t"
        "v_lshl_add_u32 v54, v55, 3, v54\n\t"
        "global_load_dwordx2 v[44:45], v54, %[occ]\n\t"
        "v_mov_b32 v46, v53\n\t"
        "s_waitcnt vmcnt(0)\n"
        "6:\n\t"
        "s_mov_b64 exec, %[sc]\n\t"
        "v_and_b32 v54, 0x300c03, v56\n\t"            // the cell's bit in the brick word
        "v_mul_u32_u24 v54, 0x40404, v54\n\t"
        "v_bfe_u32 v54, v54, 18, 6\n\t"
        "v_lshrrev_b64 v[54:55], v54, v[44:45]\n\t"
        "v_and_b32 v54, 1, v54\n\t"
        "v_cmp_eq_u32 vcc, 1, v54\n\t"
        "s_cbranch_vccz 7f\n\t"
        "s_mov_b64 exec, vcc\n\t"                     // ---- hits: main.cpp:139-142
        VK_CNT("s_add_u32 s94, s94, 1\n\t")
        "s_add_u32 %[tmp], m0, 1\n\t"
        "v_mov_b32 %[steps], %[tmp]\n\t"
        "v_mov_b32 %[hitc], v56\n\t"
        "s_andn2_b64 %[sa], %[sa], vcc\n"
        "7:\n\t"
        "s_mov_b32 %[tr], " VK_RUN_ARMED "\n\t"
        "s_mov_b64 exec, %[sa]\n\t"
        "s_cbranch_execnz 2b\n"                       // the step itself, for the lanes still live
        "9:\n\t"
        "s_mov_b64 exec, %[se]"
        : [cx] "+v"(h.cx), [cy] "+v"(h.cy), [cz] "+v"(h.cz), [steps] "+v"(steps), [hitc] "+v"(hitc),
          [se] "=&s"(mask_e), [sa] "=&s"(mask_a), [sb] "=&s"(mask_b), [sc] "=&s"(mask_c), [tmp] "=&s"(tmp), [tr] "=&s"(tr)
        : [ex] "v"(r.ex), [ey] "v"(r.ey), [ez] "v"(r.ez), [hx] "v"(r.hx), [hy] "v"(r.hy), [hz] "v"(r.hz),
          [ix] "v"(r.ix), [iy] "v"(r.iy), [iz] "v"(r.iz), [dx] "v"(r.dx), [dy] "v"(r.dy), [dz] "v"(r.dz),
          [ahead] "v"(ahead), [cap] "s"(cap), [sx] "s"(sx), [sy] "s"(sy), [sz] "s"(sz), [occ2] "s"(occ2), [occ] "s"(occ)
        : "vcc", "scc", "memory", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
          "v60", "v61", "v62", "v63"
#if VK_COUNT
          , "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95"
#endif
          );
    uint32_t iter;
    asm volatile("s_mov_b32 %0, m0" : "=s"(iter));
#if VK_COUNT
    {   // measurement build only: what the loop did, summed over the launch (volly_gpu_debug_counters)
        uint32_t k[8];
        asm volatile("s_mov_b32 %0, s88\n\ts_mov_b32 %1, s89\n\ts_mov_b32 %2, s90\n\ts_mov_b32 %3, s91\n\ts_mov_b32 %4, s92\n\ts_mov_b32 %5, s93\n\ts_mov_b32 %6, s94\n\ts_mov_b32 %7, s95"
                     : "=s"(k[0]), "=s"(k[1]), "=s"(k[2]), "=s"(k[3]), "=s"(k[4]), "=s"(k[5]), "=s"(k[6]), "=s"(k[7]));
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {      // first active lane
            atomicAdd(&vk_counters[0], 1ull);
            atomicAdd(&vk_counters[1], (unsigned long long)iter);
            for (int i = 0; i < 8; ++i) atomicAdd(&vk_counters[2 + i], (unsigned long long)k[i]);
        }
    }
#endif
    h.steps = (steps == 0xFFFFFFFFu) ? iter : steps;     // lanes still live left at the iteration cap
    if (hitc != 0xFFFFFFFFu) {
        const uint32_t x = hitc & 1023u, y = (hitc >> 10) & 1023u, z = hitc >> 20;
        if (rim_exit(g, x, y, z)) {
            h.steps -= 1;
        } else {
            h.index = x + (uint32_t)g.sx * (y + (uint32_t)g.sy * z);
            h.payload = h.index;
        }
    }
    return h;
}

// trace_dense_regular<false> (bricks only: grids whose 16^3 regions nearly all hold voxels, e.g. the
// noise volume) in assembly, with nothing left to compiled code (VK_ASM_BRICKS).  In such a grid some
// lane of the wavefront enters a new 4^3 brick on almost every step, so there are no quiet steps to run
// ahead with; what the assembly saves is the exec-mask bookkeeping of the compiled per-lane tests: a
// step is the sample point and cell (11 vector instructions), for the lanes in a new brick the bounds
// test, brick offset and load of its word, for the lanes whose brick holds voxels the test of the cell's
// bit, and the advance (19).  A lane that leaves the grid or samples a voxel records its result under an
// exec mask and drops out of the loop's live mask.  Fixed registers: v47-v55 temporaries, v56 packed cell, v46 brick key, v44-v45 brick word.
// The loop itself, for the active lanes: iterations iter0 .. iter_end - 1 of rays whose position is cur (cx, cy, cz).
// Returns the iteration number it stopped at (iter_end, or earlier when no lane was left); a lane that left the grid or
// sampled a voxel has its step count in `steps` (0xffffffff: still live) and the packed cell of a hit in `hitc`
// (0xffffffff: none).  A live lane's cur is where iteration iter_end starts, so the loop can be re-entered there
// (render_frames_packed does, with other rays in the lanes).
// The text of the loop (a macro, so that the look-ahead form below shares it).  Labels: 1 a step begins: sample point,
// cell, packed cell, iteration-cap test, brick key; lanes in a new brick: bounds test, leaving lanes record their step
// count (main.cpp:126-135); 3 the others request their brick's word; 4 lanes whose brick holds voxels test the
// cell's bit, hits record step count and packed cell (main.cpp:139-142); 6 advance_unit (main.cpp:149-153);
// 9 the end: no lane left, or the cap.  PREFETCH stands where a lane has just requested the word of a new brick
// (after it the word must have arrived); EXITWAIT in front of the loop's end.
#define VK_BRICKS_LOOP_TEXT(PREFETCH, EXITWAIT) \
        "s_mov_b64 %[se], exec\n\t" \
        "s_mov_b32 m0, %[iter0]\n\t" \
        "v_mov_b32 v46, -1\n\t" \
        "v_mov_b32 v44, 0\n\t" \
        "v_mov_b32 v45, 0\n" \
        "1:\n\t" \
        "v_add_f32 v47, %[cx], %[ex]\n\t" \
        "v_add_f32 v48, %[cy], %[ey]\n\t" \
        "v_add_f32 v49, %[cz], %[ez]\n\t" \
        "v_cvt_i32_f32 v50, v47\n\t" \
        "v_cvt_i32_f32 v51, v48\n\t" \
        "v_cvt_i32_f32 v52, v49\n\t" \
        "v_lshl_or_b32 v56, v51, 10, v50\n\t" \
        "v_lshl_or_b32 v56, v52, 20, v56\n\t" \
        "s_cmp_ge_u32 m0, %[cap]\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "v_and_b32 v53, 0xffcff3fc, v56\n\t" \
        "v_cmp_ne_u32 vcc, v53, v46\n\t" \
        "s_cbranch_vccz 4f\n\t" \
        "s_mov_b64 %[sa], exec\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "v_cmp_gt_u32 vcc, %[sx], v50\n\t" \
        "v_cmp_gt_u32 %[sc], %[sy], v51\n\t" \
        "s_and_b64 vcc, vcc, %[sc]\n\t" \
        "v_cmp_gt_u32 %[sc], %[sz], v52\n\t" \
        "s_and_b64 vcc, vcc, %[sc]\n\t" \
        "s_andn2_b64 %[sc], exec, vcc\n\t" \
        "s_cbranch_scc0 3f\n\t" \
        "s_mov_b64 exec, %[sc]\n\t" \
        "s_mov_b32 %[tmp], m0\n\t" \
        "v_mov_b32 %[steps], %[tmp]\n\t" \
        "s_andn2_b64 %[sa], %[sa], %[sc]\n" \
        "3:\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "s_cbranch_execz 5f\n\t" \
        "v_lshrrev_b32 v54, 2, v52\n\t" \
        "v_and_b32 v55, 0xfffffffc, v51\n\t" \
        "v_mad_u32_u24 v54, v54, %[sy], v55\n\t" \
        "v_mul_u32_u24 v54, v54, %[sx]\n\t" \
        "v_lshrrev_b32 v54, 1, v54\n\t" \
        "v_lshrrev_b32 v55, 2, v50\n\t" \
        "v_lshl_add_u32 v54, v55, 3, v54\n\t" \
        "global_load_dwordx2 v[44:45], v54, %[occ]\n\t" \
        "v_mov_b32 v46, v53\n\t" \
        PREFETCH \
        "5:\n\t" \
        "s_mov_b64 exec, %[sa]\n\t" \
        "s_cbranch_execz 9f\n" \
        "4:\n\t" \
        "v_cmp_ne_u64 vcc, 0, v[44:45]\n\t" \
        "s_cbranch_vccz 6f\n\t" \
        "s_mov_b64 %[sa], exec\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "v_and_b32 v54, 0x300c03, v56\n\t" \
        "v_mul_u32_u24 v54, 0x40404, v54\n\t" \
        "v_bfe_u32 v54, v54, 18, 6\n\t" \
        "v_lshrrev_b64 v[54:55], v54, v[44:45]\n\t" \
        "v_and_b32 v54, 1, v54\n\t" \
        "v_cmp_eq_u32 vcc, 1, v54\n\t" \
        "s_cbranch_vccz 7f\n\t" \
        "s_mov_b64 exec, vcc\n\t" \
        "s_add_u32 %[tmp], m0, 1\n\t" \
        "v_mov_b32 %[steps], %[tmp]\n\t" \
        "v_mov_b32 %[hitc], v56\n\t" \
        "s_andn2_b64 %[sa], %[sa], vcc\n" \
        "7:\n\t" \
        "s_mov_b64 exec, %[sa]\n\t" \
        "s_cbranch_execz 9f\n" \
        "6:\n\t" \
        VK_ADVANCE_UNIT \
        "s_add_u32 m0, m0, 1\n\t" \
        "s_branch 1b\n" \
        "9:\n\t" \
        EXITWAIT \
        "s_mov_b64 exec, %[se]"

// In the plain loop a lane waits for its brick word at once.
#define VK_BRICKS_WAIT "s_waitcnt vmcnt(0)\n"
// The look-ahead form, for the passes that continue few, long rays (continue_rays_packed): there the chip is nearly
// idle and a ray's time is the latency of its dependent brick loads, one per ~2.5 steps, most of them from beyond
// the L2.  A lane that enters a new brick also requests -- into a register nobody reads -- the brick that holds the
// point 16 cells further along its ray, clamped into the grid: the ray passes through that point, so when it gets
// there ~6 bricks later the word is in the L2.  The request changes no result; the lane waits for its own word only
// (loads return in order: vmcnt(1) leaves the look-ahead request in flight).  17 more vector instructions per brick
// change: not for the first pass, which is bound by vector issue.
#define VK_BRICKS_LOOKAHEAD \
        "v_mul_f32 v53, 0x41800000, %[dx]\n\t" \
        "v_mul_f32 v54, 0x41800000, %[dy]\n\t" \
        "v_mul_f32 v55, 0x41800000, %[dz]\n\t" \
        "v_add_f32 v53, v53, %[cx]\n\t" \
        "v_add_f32 v54, v54, %[cy]\n\t" \
        "v_add_f32 v55, v55, %[cz]\n\t" \
        "v_cvt_i32_f32 v53, v53\n\t" \
        "v_cvt_i32_f32 v54, v54\n\t" \
        "v_cvt_i32_f32 v55, v55\n\t" \
        "s_sub_u32 %[tmp], %[sx], 1\n\t" \
        "v_med3_i32 v53, v53, 0, %[tmp]\n\t" \
        "s_sub_u32 %[tmp], %[sy], 1\n\t" \
        "v_med3_i32 v54, v54, 0, %[tmp]\n\t" \
        "s_sub_u32 %[tmp], %[sz], 1\n\t" \
        "v_med3_i32 v55, v55, 0, %[tmp]\n\t" \
        "v_lshrrev_b32 v55, 2, v55\n\t" \
        "v_and_b32 v54, 0xfffffffc, v54\n\t" \
        "v_mad_u32_u24 v55, v55, %[sy], v54\n\t" \
        "v_mul_u32_u24 v55, v55, %[sx]\n\t" \
        "v_lshrrev_b32 v55, 1, v55\n\t" \
        "v_lshrrev_b32 v53, 2, v53\n\t" \
        "v_lshl_add_u32 v55, v53, 3, v55\n\t" \
        "global_load_dword v57, v55, %[occ]\n\t" \
        "s_waitcnt vmcnt(1)\n"

template <bool LOOKAHEAD = false>
__device__ __forceinline__ uint32_t bricks_loop_asm(const DenseDev &g, const Ray &r, float &cx, float &cy, float &cz, uint32_t iter0, uint32_t iter_end,
                                                    uint32_t &steps, uint32_t &hitc)
{
    const uint32_t cap = iter_end;
    const uint32_t sx = (uint32_t)g.px, sy = (uint32_t)g.py, sz = (uint32_t)g.pz;    // the loop's grid is the padded one
    const uint64_t *occ = g.occ;
    uint64_t mask_e, mask_a, mask_c;
    uint32_t tmp;
    if constexpr (LOOKAHEAD)
    asm volatile(
        VK_BRICKS_LOOP_TEXT(VK_BRICKS_LOOKAHEAD, "s_waitcnt vmcnt(0)\n\t")
        : [cx] "+v"(cx), [cy] "+v"(cy), [cz] "+v"(cz), [steps] "+v"(steps), [hitc] "+v"(hitc),
          [se] "=&s"(mask_e), [sa] "=&s"(mask_a), [sc] "=&s"(mask_c), [tmp] "=&s"(tmp)
        : [ex] "v"(r.ex), [ey] "v"(r.ey), [ez] "v"(r.ez), [hx] "v"(r.hx), [hy] "v"(r.hy), [hz] "v"(r.hz),
          [ix] "v"(r.ix), [iy] "v"(r.iy), [iz] "v"(r.iz), [dx] "v"(r.dx), [dy] "v"(r.dy), [dz] "v"(r.dz),
          [cap] "s"(cap), [sx] "s"(sx), [sy] "s"(sy), [sz] "s"(sz), [occ] "s"(occ), [iter0] "s"(iter0)
        : "vcc", "scc", "memory", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
    else
    asm volatile(
        VK_BRICKS_LOOP_TEXT(VK_BRICKS_WAIT, "")
        : [cx] "+v"(cx), [cy] "+v"(cy), [cz] "+v"(cz), [steps] "+v"(steps), [hitc] "+v"(hitc),
          [se] "=&s"(mask_e), [sa] "=&s"(mask_a), [sc] "=&s"(mask_c), [tmp] "=&s"(tmp)
        : [ex] "v"(r.ex), [ey] "v"(r.ey), [ez] "v"(r.ez), [hx] "v"(r.hx), [hy] "v"(r.hy), [hz] "v"(r.hz),
          [ix] "v"(r.ix), [iy] "v"(r.iy), [iz] "v"(r.iz), [dx] "v"(r.dx), [dy] "v"(r.dy), [dz] "v"(r.dz),
          [cap] "s"(cap), [sx] "s"(sx), [sy] "s"(sy), [sz] "s"(sz), [occ] "s"(occ), [iter0] "s"(iter0)
        : "vcc", "scc", "memory", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56");
    uint32_t iter;
    asm volatile("s_mov_b32 %0, m0" : "=s"(iter));
    return iter;
}

__device__ __forceinline__ Hit trace_dense_bricks_whole_asm(const DenseDev &g, const Ray &r, float ox, float oy, float oz)
{
    Hit h;
    enter_volume(r, ox, oy, oz, g.fsx, g.fsy, g.fsz, h.cx, h.cy, h.cz);
    h.steps = 0;
    h.index = VOLLY_MISS_INDEX;
    h.payload = VOLLY_MISS_INDEX;
    if (!(fabsf(h.cx) < 3.0e38f && fabsf(h.cy) < 3.0e38f && fabsf(h.cz) < 3.0e38f)) return h;   // inf/NaN entry: the first bounds test fails
    uint32_t steps = 0xFFFFFFFFu;                         // still live
    uint32_t hitc = 0xFFFFFFFFu;                          // packed cell of the hit
    const uint32_t iter = bricks_loop_asm(g, r, h.cx, h.cy, h.cz, 0u, g.cap, steps, hitc);
    h.steps = (steps == 0xFFFFFFFFu) ? iter : steps;     // lanes still live left at the iteration cap
    if (hitc != 0xFFFFFFFFu) {
        const uint32_t x = hitc & 1023u, y = (hitc >> 10) & 1023u, z = hitc >> 20;
        if (rim_exit(g, x, y, z)) {
            h.steps -= 1;
        } else {
            h.index = x + (uint32_t)g.sx * (y + (uint32_t)g.sy * z);
            h.payload = h.index;
        }
    }
    return h;
}

__device__ __forceinline__ Hit trace_dense(const DenseDev &g, const Ray &r, float ox, float oy, float oz)
{
#if VK_FAST_DDA
    // one path per wavefront, so the two loops never run interleaved in a wave
    if (grid_is_regular(g) && __all(ray_is_regular(r))) {
        // the 16^3 region level is chosen per grid at upload (DenseDev::use_regions)
#if VK_ASM_QUIET
        if (VK_OCC_LEVELS == 2 && g.use_regions) return trace_dense_regions_whole_asm(g, r, ox, oy, oz);
#else
        if (VK_OCC_LEVELS == 2 && g.use_regions) return trace_dense_regular<true>(g, r, ox, oy, oz);
#endif
#if VK_ASM_BRICKS
        return trace_dense_bricks_whole_asm(g, r, ox, oy, oz);
#else
        return trace_dense_regular<false>(g, r, ox, oy, oz);
#endif
    }
#endif
    return trace_dense_checked<false>(g, r, ox, oy, oz);
}

// Octree traversal (no reference implementation: svo.cpp:14-19 is empty).  Each
// iteration locates the sample cell in the tree and either hits a leaf voxel or steps
// across the largest empty octree cell around the sample point.
//
// Locating a cell by walking down from the root costs one dependent load per level, and
// that latency -- not arithmetic or bandwidth -- paces the loop.  So the upper levels are
// flattened at upload into a direct-index table (build_svo_table): one load gives either
// "empty cell of 2^k voxels" or the node that covers the 2^tshift cell, from which at most
// tshift more levels remain.  A lane keeps the last table entry and its node in registers.
// With VK_SVO_TABLE=0 the walk starts at the root.
__device__ __forceinline__ Hit trace_svo(const SvoDev &t, const Ray &r, float ox, float oy, float oz)
{
    Hit h;
    enter_volume(r, ox, oy, oz, t.fres, t.fres, t.fres, h.cx, h.cy, h.cz);
    h.steps = 0;
    h.index = VOLLY_MISS_INDEX;
    h.payload = VOLLY_MISS_INDEX;
#if VK_SVO_TABLE
    uint32_t last_cell = 0xFFFFFFFFu, entry = 0;
    uint2 entry_node = make_uint2(0u, 0u);
#endif
    for (;;) {
        float px = h.cx + r.ex, py = h.cy + r.ey, pz = h.cz + r.ez;
        bool inside = (px > -1.0f) & (px < t.fres) & (py > -1.0f) & (py < t.fres) & (pz > -1.0f) & (pz < t.fres);
        if (!inside) break;
        if (h.steps >= t.cap) break;
        int x = (int)px, y = (int)py, z = (int)pz;
        ++h.steps;

        bool hit = false;
        int sh;
#if VK_SVO_TABLE
        uint32_t tcell = (uint32_t)(x >> t.tshift) | ((uint32_t)(y >> t.tshift) << t.tbits) | ((uint32_t)(z >> t.tshift) << (2 * t.tbits));
        if (tcell != last_cell) {
            entry = t.table[tcell];
            if (!(entry >> 31)) entry_node = t.nodes[entry];
            last_cell = tcell;
        }
        uint2 n = entry_node;
        sh = t.tshift - 1;
        if (entry >> 31) {
            sh = (int)((entry >> 23) & 0xFFu) - 127;      // empty cell of 2^sh voxels, above the table's level: the entry is -(2^sh) as a float
        } else
#else
        uint2 n = t.nodes[0];
        sh = t.depth - 1;
#endif
        for (;;) {
            uint32_t oct = ((uint32_t)(x >> sh) & 1u) | (((uint32_t)(y >> sh) & 1u) << 1) | (((uint32_t)(z >> sh) & 1u) << 2);
            uint32_t valid = n.y & 0xFFu;
            uint32_t bit = 1u << oct;
            if (!(valid & bit)) break;                    // empty cell of 2^sh voxels
            uint32_t child = n.x + (uint32_t)__popc(valid & (bit - 1u));
            if ((n.y >> 8) & bit) {                       // leaf voxel
                h.payload = child;
                hit = true;
                break;
            }
            n = t.nodes[child];
            --sh;
        }
        if (hit) {
            h.index = (uint32_t)x + (uint32_t)t.res * ((uint32_t)y + (uint32_t)t.res * (uint32_t)z);
            break;
        }
        float s = __uint_as_float((uint32_t)(127 + sh) << 23);
        float is = __uint_as_float((uint32_t)(127 - sh) << 23);
        advance_cell(r, s, is, px, py, pz, h.cx, h.cy, h.cz);
    }
    return h;
}

// The same step with the three multiplications by s folded into one: floor(p/s)*s + h*s == (floor(p/s) + h)*s bit
// for bit -- floor(p/s) is an integer of magnitude <= 2^13, h is 0 or 1, s a power of two, so both sides are the
// exact product (no rounding happens on either side; -0 + 0 and -s + s are +0 on both).
__device__ __forceinline__ void advance_cell_folded(const Ray &r, float s, float is, float px, float py, float pz,
                                                    float &cx, float &cy, float &cz)
{
    float ndx = (((floorf(px * is) + r.hx) * s) - cx) * r.ix;
    float ndy = (((floorf(py * is) + r.hy) * s) - cy) * r.iy;
    float ndz = (((floorf(pz * is) + r.hz) * s) - cz) * r.iz;
    float m = fminf(fminf(ndx, ndy), ndz);
    cx = m * r.dx + cx;
    cy = m * r.dy + cy;
    cz = m * r.dz + cz;
}

__device__ __forceinline__ bool svo_is_regular(const SvoDev &t) { return t.use_blocks != 0; }

// The leaf of the hit voxel (x, y, z) under level-3 node l3, for the flattened tree.  Walking down from the node costs
// three dependent loads per hit ray -- with the leaf itself, four latencies at the end of every hit wavefront's life,
// which is what paces the naive frames and the reference's start-up camera (few iterations per ray, every ray a hit).
// When the tree stores the leaves under a block word one after the other in tree order (use_rank; breadth-first
// builders do), the leaf is word_base + the number of the word's voxels that come before this one in that order:
// two independent loads and three popcounts.  Tree order under a word: 2^3 sub-octant (z1 y1 x1), then voxel
// (z0 y0 x0); the word's bits are x0 x1 y0 y1 z0 z1 (LSB first), so "before" is a mask per sub-octant plus the
// sub-octant's own lower voxels.
__device__ __forceinline__ uint32_t svo_leaf_by_walk(const SvoDev &t, uint32_t l3, uint32_t x, uint32_t y, uint32_t z)
{
    uint2 n = t.nodes[l3];
    uint32_t child = 0;
    #pragma unroll
    for (int sh = 2; sh >= 0; --sh) {
        const uint32_t oct = ((x >> sh) & 1u) | (((y >> sh) & 1u) << 1) | (((z >> sh) & 1u) << 2);
        const uint32_t valid = n.y & 0xFFu, bit = 1u << oct;
        child = n.x + (uint32_t)__popc(valid & (bit - 1u));
        if (sh) n = t.nodes[child];
    }
    return child;
}

__device__ __forceinline__ uint32_t svo_leaf_by_rank(const SvoDev &t, uint32_t l3, uint32_t x, uint32_t y, uint32_t z)
{
    const uint32_t o4 = ((x >> 2) & 1u) | (((y >> 2) & 1u) << 1) | (((z >> 2) & 1u) << 2);
    const size_t wi = (size_t)(l3 - t.l3_first) * 8 + o4;
    const uint32_t base = t.word_base[wi];
    const uint64_t w = t.blocks[wi];
    const uint32_t b = (x & 3u) | ((y & 3u) << 2) | ((z & 3u) << 4);
    // voxels of the sub-octants before this one: within a 32-bit half (one z1), those with (y1, x1) smaller
    const uint32_t half = ((b & 8u) ? 0x00FF00FFu : 0u) | (((b & 8u) ? 0xFF00FF00u : 0x00FF00FFu) & ((b & 2u) ? 0x33333333u : 0u));
    const uint32_t lt_lo = (b & 32u) ? 0xFFFFFFFFu : half, lt_hi = (b & 32u) ? half : 0u;
    // this sub-octant's voxels (offsets 0,1,4,5,16,17,20,21 from its first bit, in voxel order) below the hit one
    const uint32_t sub = (uint32_t)(w >> (b & 0x2Au)) & 0x00330033u;
    const uint32_t before = sub & ((1u << (b & 0x15u)) - 1u);
    return base + (uint32_t)__popc((uint32_t)w & lt_lo) + (uint32_t)__popc((uint32_t)(w >> 32) & lt_hi) + (uint32_t)__popc(before);
}

__device__ __forceinline__ uint32_t svo_leaf_of_hit(const SvoDev &t, uint32_t l3, uint32_t x, uint32_t y, uint32_t z)
{
    return t.use_rank ? svo_leaf_by_rank(t, l3, x, y, z) : svo_leaf_by_walk(t, l3, x, y, z);
}

// Octree traversal over the flattened tree (VK_SVO_BLOCKS): the same iterations as trace_svo -- every decision and
// every float equal -- with at most two dependent loads per iteration instead of up to five, and most iterations none:
//   * upper levels: the direct-index table gives, per 8^3 cell, "empty cell of 2^k voxels" (as the float -(2^k)) or
//     the level-3 node that covers it; a lane keeps that entry while its sample cell stays in the 8^3 cell;
//   * lowest three levels: the node's block holds one 64-bit occupancy word per 4^3 octant.  word == 0: the octant is
//     an empty cell of 4 voxels; the 2^3 sub-octant's eight bits clear: an empty cell of 2; else the voxel's own bit
//     decides between a hit and an empty cell of 1.  A lane keeps the word while it stays in the 4^3 brick.
// The leaf index of a hit is found afterwards: svo_leaf_of_hit.
// Preconditions (svo_is_regular + regular rays, checked by the caller): res a multiple of 8 (so a cell outside the
// grid never shares its 8^3 key with the cell before it), table cells of 8^3, every direction component regular.
// Cells are kept as the pair xy = x | y << 16 (saturating conversion: a negative or huge coordinate becomes 0xffff,
// which no key inside a grid of <= 2048 has) and z.
__device__ __forceinline__ Hit trace_svo_blocks(const SvoDev &t, const Ray &r, float ox, float oy, float oz)
{
    Hit h;
    enter_volume(r, ox, oy, oz, t.fres, t.fres, t.fres, h.cx, h.cy, h.cz);
    h.steps = 0;
    h.index = VOLLY_MISS_INDEX;
    h.payload = VOLLY_MISS_INDEX;
    if (!(fabsf(h.cx) < 3.0e38f && fabsf(h.cy) < 3.0e38f && fabsf(h.cz) < 3.0e38f)) return h;   // inf/NaN entry: the first bounds test fails
    const uint32_t res = (uint32_t)t.res;
    uint32_t key_xy = 0xFFFFFFFFu, key_z = 0xFFFFFFFFu;  // the 8^3 cell the cached entry belongs to
    uint32_t l3 = 0;                                     // sign bit: empty cell of -l3 voxels (a float); else level-3 node index
    uint32_t bkey_xy = 0xFFFFFFFFu, bkey_z = 0xFFFFFFFFu; // the 4^3 brick the cached word belongs to
    uint32_t w_lo = 0, w_hi = 0;
    bool hit = false;
    int x = 0, y = 0, z = 0;
    for (;;) {
        float px = h.cx + r.ex, py = h.cy + r.ey, pz = h.cz + r.ez;
        x = (int)px; y = (int)py; z = (int)pz;           // truncation toward zero, saturating
        if (h.steps >= t.cap) break;
        const uint32_t k_xy = ((uint32_t)x & 0xFFF8u) | (((uint32_t)y & 0xFFF8u) << 16), k_z = (uint32_t)z & 0xFFFFFFF8u;
        const bool oob16 = ((uint32_t)x > 0xFFFFu) | ((uint32_t)y > 0xFFFFu);       // what the saturating pack turns into 0xffff
        if (oob16 | (k_xy != key_xy) | (k_z != key_z)) {
            if (!(((uint32_t)x < res) & ((uint32_t)y < res) & ((uint32_t)z < res))) break;   // main.cpp:126-135
            key_xy = k_xy; key_z = k_z;
            l3 = t.table[((uint32_t)x >> 3) | (((uint32_t)y >> 3) << t.tbits) | (((uint32_t)z >> 3) << (2 * t.tbits))];
        }
        ++h.steps;
        float s;
        if (l3 >> 31) {
            s = __uint_as_float(l3 & 0x7FFFFFFFu);
        } else {
            const uint32_t b_xy = ((uint32_t)x & 0xFFFCu) | (((uint32_t)y & 0xFFFCu) << 16), b_z = (uint32_t)z & 0xFFFFFFFCu;
            if ((b_xy != bkey_xy) | (b_z != bkey_z)) {
                bkey_xy = b_xy; bkey_z = b_z;
                const uint32_t o4 = (((uint32_t)x >> 2) & 1u) | ((((uint32_t)y >> 2) & 1u) << 1) | ((((uint32_t)z >> 2) & 1u) << 2);
                const uint64_t w = t.blocks[(size_t)(l3 - t.l3_first) * 8 + o4];
                w_lo = (uint32_t)w; w_hi = (uint32_t)(w >> 32);
            }
            if ((w_lo | w_hi) == 0u) {
                s = 4.0f;
            } else {
                const uint32_t b = ((uint32_t)x & 3u) | (((uint32_t)y & 3u) << 2) | (((uint32_t)z & 3u) << 4);
                const uint64_t w = ((uint64_t)w_hi << 32) | w_lo;
                const uint32_t oct = (uint32_t)(w >> (b & 0x2Au));          // the 2^3 sub-octant's bits at offsets 0,1,4,5,16,17,20,21
                if ((oct & 0x00330033u) == 0u) s = 2.0f;
                else if ((oct >> (b & 0x15u)) & 1u) { hit = true; break; }  // leaf voxel
                else s = 1.0f;
            }
        }
        const float is = __uint_as_float(0x7F000000u - __float_as_uint(s));  // 2^-k for s = 2^k
        advance_cell_folded(r, s, is, px, py, pz, h.cx, h.cy, h.cz);
    }
    if (hit) {
        h.index = (uint32_t)x + res * ((uint32_t)y + res * (uint32_t)z);
        h.payload = svo_leaf_of_hit(t, l3, (uint32_t)x, (uint32_t)y, (uint32_t)z);
    }
    return h;
}

// trace_svo_blocks in assembly, with nothing left to compiled code (VK_ASM_SVO).  The compiled loop spends ~90 vector
// and ~80 scalar instructions per wave-iteration, most of the scalar ones exec-mask bookkeeping of its nested
// per-lane tests (profiles/r02: 726 M scalar beside 989 M vector instructions per launch of the naive headline: the CU's
// shared scalar pipe is as busy as its four vector pipes).  Here an iteration is 38 vector + ~12 scalar instructions
// while every live lane sits in an empty cell above the blocks' level, 58 + ~25 with lanes inside 8^3 cells that hold
// voxels -- the wave-coherent skips the north star asks for are the scalar branches below: a wavefront whose lanes
// all kept their 8^3 cell skips the table block, one whose lanes are all in empty upper-level cells skips the block /
// brick tests altogether (s_cbranch_vccz on the ballot of "my entry is a node").
//   exec = live lanes; m0 = iteration number = samples taken so far (the lane's step count when it leaves).
//   v47-v49 sample point, v50-v52 cell, v56 = x | y << 16 (saturating pack: a coordinate outside [0, 65535] gives 0xffff,
//   which no key of a grid <= 2048 has), v57 table entry (sign set: -(cell size) of an empty cell; else the level-3 node),
//   v58:v59 key of the 8^3 cell it belongs to (xy, z), v60:v61 key of the 4^3 brick whose word is in v44:v45,
//   v62 / v63 the step's cell size s and 1/s, v46 and v53-v55 temporaries.
// Results: steps (0xffffffff: still live at the cap) and hitl3 (the level-3 node of a hit, 0xffffffff: no hit); the
// hit cell is recomputed afterwards from cur, which a finished lane no longer changes.
#ifndef VK_ASM_SVO
#define VK_ASM_SVO 1
#endif
__device__ __forceinline__ Hit trace_svo_blocks_asm(const SvoDev &t, const Ray &r, float ox, float oy, float oz)
{
    Hit h;
    enter_volume(r, ox, oy, oz, t.fres, t.fres, t.fres, h.cx, h.cy, h.cz);
    h.steps = 0;
    h.index = VOLLY_MISS_INDEX;
    h.payload = VOLLY_MISS_INDEX;
    if (!(fabsf(h.cx) < 3.0e38f && fabsf(h.cy) < 3.0e38f && fabsf(h.cz) < 3.0e38f)) return h;   // inf/NaN entry: the first bounds test fails
    const uint32_t cap = t.cap, res = (uint32_t)t.res, tbits = (uint32_t)t.tbits, l3_first = t.l3_first;
    const uint32_t *table = t.table;
    const uint64_t *blocks = t.blocks;
    uint32_t steps = 0xFFFFFFFFu;                         // still live
    uint32_t hitl3 = 0xFFFFFFFFu;                         // level-3 node of the hit
    uint64_t mask_e, mask_a, mask_b, mask_c;
    uint32_t tmp;
    asm volatile(
        "s_mov_b64 %[se], exec\n\t"
        "s_mov_b32 m0, 0\n\t"
        "v_mov_b32 v58, -1\n\t"                       // no 8^3 cell yet (a key no cell has)
        "v_mov_b32 v59, -1\n\t"
        "v_mov_b32 v60, -1\n\t"                       // no brick yet
        "v_mov_b32 v61, -1\n\t"
        "v_mov_b32 v57, 0\n\t"
        "v_mov_b32 v44, 0\n\t"
        "v_mov_b32 v45, 0\n"
        "1:\n\t"                                      // ---- sample point, cell, 8^3 key (exec = live lanes)
        "v_add_f32 v47, %[cx], %[ex]\n\t"
        "v_add_f32 v48, %[cy], %[ey]\n\t"
        "v_add_f32 v49, %[cz], %[ez]\n\t"
        "v_cvt_i32_f32 v50, v47\n\t"
        "v_cvt_i32_f32 v51, v48\n\t"
        "v_cvt_i32_f32 v52, v49\n\t"
        "v_cvt_pk_u16_u32 v56, v50, v51\n\t"
        "s_cmp_ge_u32 m0, %[cap]\n\t"
        "s_cbranch_scc1 9f\n\t"
        "v_and_b32 v54, 0xfff8fff8, v56\n\t"
        "v_and_b32 v55, -8, v52\n\t"
        "v_cmp_ne_u64 vcc, v[54:55], v[58:59]\n\t"
        "s_cbranch_vccz 4f\n\t"
        "s_mov_b64 %[sa], exec\n\t"                   // sa: the live lanes
        "s_mov_b64 exec, vcc\n\t"                     // ---- lanes in a new 8^3 cell: bounds test (main.cpp:126-135), table entry
        "v_cmp_gt_u32 vcc, %[res], v50\n\t"
        "v_cmp_gt_u32 %[sc], %[res], v51\n\t"
        "s_and_b64 vcc, vcc, %[sc]\n\t"
        "v_cmp_gt_u32 %[sc], %[res], v52\n\t"
        "s_and_b64 vcc, vcc, %[sc]\n\t"               // vcc: inside the grid
        "s_andn2_b64 %[sc], exec, vcc\n\t"            // sc: outside
        "s_cbranch_scc0 2f\n\t"
        "s_mov_b64 exec, %[sc]\n\t"
        "s_mov_b32 %[tmp], m0\n\t"
        "v_mov_b32 %[steps], %[tmp]\n\t"
        "s_andn2_b64 %[sa], %[sa], %[sc]\n"           // no longer live
        "2:\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "s_cbranch_execz 3f\n\t"
        "s_lshl_b32 %[tmp], %[tbits], 1\n\t"
        "v_lshrrev_b32 v53, 3, v50\n\t"               // table index x>>3 | (y>>3) << tbits | (z>>3) << 2 tbits
        "v_lshrrev_b32 v46, 3, v51\n\t"
        "v_lshl_or_b32 v53, v46, %[tbits], v53\n\t"
        "v_lshrrev_b32 v46, 3, v52\n\t"
        "v_lshl_or_b32 v53, v46, %[tmp], v53\n\t"
        "v_lshlrev_b32 v53, 2, v53\n\t"
        "global_load_dword v57, v53, %[table]\n\t"
        "v_mov_b32 v58, v54\n\t"
        "v_mov_b32 v59, v55\n\t"
        "s_waitcnt vmcnt(0)\n"
        "3:\n\t"
        "s_mov_b64 exec, %[sa]\n\t"
        "s_cbranch_execz 9f\n"
        "4:\n\t"                                      // ---- the cell size of an empty upper-level cell; which lanes sit in a cell with voxels
        "v_and_b32 v62, 0x7fffffff, v57\n\t"
        "v_cmp_le_i32 vcc, 0, v57\n\t"
        "s_cbranch_vccz 8f\n\t"                       // every live lane is in an empty cell of >= 8 voxels: straight to the step
        "s_mov_b64 %[sa], exec\n\t"
        "s_mov_b64 exec, vcc\n\t"                     // ---- lanes in an 8^3 cell that holds voxels: the 4^3 brick's word
        "v_and_b32 v54, 0xfffcfffc, v56\n\t"
        "v_and_b32 v55, -4, v52\n\t"
        "v_cmp_ne_u64 vcc, v[54:55], v[60:61]\n\t"
        "s_and_saveexec_b64 %[sb], vcc\n\t"           // sb: lanes at a node; exec: those that need the word
        "s_cbranch_execz 5f\n\t"
        "v_sub_u32 v53, v57, %[l3f]\n\t"              // byte offset 64 * (node - l3_first) + 8 * octant
        "v_lshlrev_b32 v53, 6, v53\n\t"
        "v_bfe_u32 v46, v50, 2, 1\n\t"
        "v_lshl_or_b32 v53, v46, 3, v53\n\t"
        "v_bfe_u32 v46, v51, 2, 1\n\t"
        "v_lshl_or_b32 v53, v46, 4, v53\n\t"
        "v_bfe_u32 v46, v52, 2, 1\n\t"
        "v_lshl_or_b32 v53, v46, 5, v53\n\t"
        "global_load_dwordx2 v[44:45], v53, %[blocks]\n\t"
        "v_mov_b32 v60, v54\n\t"
        "v_mov_b32 v61, v55\n\t"
        "s_waitcnt vmcnt(0)\n"
        "5:\n\t"
        "s_mov_b64 exec, %[sb]\n\t"
        "v_mov_b32 v62, 4.0\n\t"                      // an empty 4^3 octant unless the word has bits
        "v_cmp_ne_u64 vcc, 0, v[44:45]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 7f\n\t"
        "v_and_b32 v53, 3, v50\n\t"                   // bit index x&3 | (y&3)<<2 | (z&3)<<4
        "v_and_b32 v54, 3, v51\n\t"
        "v_lshl_or_b32 v53, v54, 2, v53\n\t"
        "v_and_b32 v54, 3, v52\n\t"
        "v_lshl_or_b32 v53, v54, 4, v53\n\t"
        "v_and_b32 v54, 0x2a, v53\n\t"                // the 2^3 sub-octant's first bit
        "v_lshrrev_b64 v[54:55], v54, v[44:45]\n\t"
        "v_and_b32 v55, 0x330033, v54\n\t"            // its eight voxels: offsets 0,1,4,5,16,17,20,21
        "v_mov_b32 v62, 2.0\n\t"
        "v_cmp_ne_u32 vcc, 0, v55\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 7f\n\t"
        "v_and_b32 v53, 0x15, v53\n\t"                // the voxel's own bit
        "v_bfe_u32 v54, v54, v53, 1\n\t"
        "v_mov_b32 v62, 1.0\n\t"
        "v_cmp_eq_u32 vcc, 1, v54\n\t"
        "s_cbranch_vccz 7f\n\t"
        "s_mov_b64 exec, vcc\n\t"                     // ---- hits
        "s_add_u32 %[tmp], m0, 1\n\t"
        "v_mov_b32 %[steps], %[tmp]\n\t"
        "v_mov_b32 %[hitl3], v57\n\t"
        "s_andn2_b64 %[sa], %[sa], vcc\n"
        "7:\n\t"
        "s_mov_b64 exec, %[sa]\n\t"
        "s_cbranch_execz 9f\n"
        "8:\n\t"                                      // ---- the step across the empty cell of s voxels (advance_cell_folded)
        "v_sub_u32 v63, 0x7f000000, v62\n\t"          // 1/s
        "v_mul_f32 v53, v47, v63\n\t"
        "v_mul_f32 v54, v48, v63\n\t"
        "v_mul_f32 v55, v49, v63\n\t"
        "v_floor_f32 v53, v53\n\t"
        "v_floor_f32 v54, v54\n\t"
        "v_floor_f32 v55, v55\n\t"
        "v_add_f32 v53, v53, %[hx]\n\t"
        "v_add_f32 v54, v54, %[hy]\n\t"
        "v_add_f32 v55, v55, %[hz]\n\t"
        "v_mul_f32 v53, v53, v62\n\t"
        "v_mul_f32 v54, v54, v62\n\t"
        "v_mul_f32 v55, v55, v62\n\t"
        "v_sub_f32 v53, v53, %[cx]\n\t"
        "v_sub_f32 v54, v54, %[cy]\n\t"
        "v_sub_f32 v55, v55, %[cz]\n\t"
        "v_mul_f32 v53, v53, %[ix]\n\t"
        "v_mul_f32 v54, v54, %[iy]\n\t"
        "v_mul_f32 v55, v55, %[iz]\n\t"
        "v_min3_f32 v53, v53, v54, v55\n\t"
        "v_mul_f32 v54, v53, %[dx]\n\t"
        "v_mul_f32 v55, v53, %[dy]\n\t"
        "v_mul_f32 v53, v53, %[dz]\n\t"
        "v_add_f32 %[cx], v54, %[cx]\n\t"
        "v_add_f32 %[cy], v55, %[cy]\n\t"
        "v_add_f32 %[cz], v53, %[cz]\n\t"
        "s_add_u32 m0, m0, 1\n\t"
        "s_branch 1b\n"
        "9:\n\t"
        "s_mov_b64 exec, %[se]"
        : [cx] "+v"(h.cx), [cy] "+v"(h.cy), [cz] "+v"(h.cz), [steps] "+v"(steps), [hitl3] "+v"(hitl3),
          [se] "=&s"(mask_e), [sa] "=&s"(mask_a), [sb] "=&s"(mask_b), [sc] "=&s"(mask_c), [tmp] "=&s"(tmp)
        : [ex] "v"(r.ex), [ey] "v"(r.ey), [ez] "v"(r.ez), [hx] "v"(r.hx), [hy] "v"(r.hy), [hz] "v"(r.hz),
          [ix] "v"(r.ix), [iy] "v"(r.iy), [iz] "v"(r.iz), [dx] "v"(r.dx), [dy] "v"(r.dy), [dz] "v"(r.dz),
          [cap] "s"(cap), [res] "s"(res), [tbits] "s"(tbits), [l3f] "s"(l3_first), [table] "s"(table), [blocks] "s"(blocks)
        : "vcc", "scc", "memory", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
          "v60", "v61", "v62", "v63");
    uint32_t iter;
    asm volatile("s_mov_b32 %0, m0" : "=s"(iter));
    h.steps = (steps == 0xFFFFFFFFu) ? iter : steps;     // lanes still live left at the iteration cap
    if (hitl3 != 0xFFFFFFFFu) {
        // the hit cell again from cur, which the lane has not changed since (the same two operations the loop did)
        const uint32_t x = (uint32_t)(int)(h.cx + r.ex), y = (uint32_t)(int)(h.cy + r.ey), z = (uint32_t)(int)(h.cz + r.ez);
        h.index = x + res * (y + res * z);
        h.payload = svo_leaf_of_hit(t, hitl3, x, y, z);
    }
    return h;
}

// glm::normalize(vec4(x,y,z,0)): v * (1/sqrt((xx+yy)+(zz+ww))); returns the w component 0 * inversesqrt
__device__ __forceinline__ float normalize_w0(float &x, float &y, float &z)
{
    float d = (x * x + y * y) + z * z;                   // (z*z + w*w) with w = 0: z*z is never -0, so adding +0 changes no bit
    float inv;
    if (VK_SHORT_DIV && in_fast_range(d)) inv = rcp_in_range(sqrt_in_range(d));   // sqrt of [2^-100, 2^100] lies in [2^-50, 2^50]
    else inv = 1.0f / sqrtf(d);
    x = x * inv; y = y * inv; z = z * inv;
    return 0.0f * inv;
}

// main.cpp:264-269
__device__ __forceinline__ uint32_t shade(uint2 vox)
{
    float inv = 1.0f / sqrtf((1.0f + 1.0f) + (1.0f + 0.0f));
    float sun = 1.0f * inv;
    float sun_w = 0.0f * inv;
    float nx = over_255((float)(vox.y & 0xFFu));
    float ny = over_255((float)((vox.y >> 8) & 0xFFu));
    float nz = over_255((float)((vox.y >> 16) & 0xFFu));
    float nw = over_255((float)(vox.y >> 24));
    float d = (sun * nx + sun * ny) + (sun * nz + sun_w * nw);
    float l = d / 2.0f + 0.5f;
    l = fminf(fmaxf(l, 0.0f), 1.0f);
    uint32_t cr = (uint32_t)((float)(vox.x & 0xFFu) * l) & 0xFFu;
    uint32_t cg = (uint32_t)((float)((vox.x >> 8) & 0xFFu) * l) & 0xFFu;
    uint32_t cb = (uint32_t)((float)((vox.x >> 16) & 0xFFu) * l) & 0xFFu;
    return 255u | (cr << 8) | (cg << 16) | (cb << 24);
}

// ---- raytrace_naiive, main.cpp:170-194 (dead code in the reference; its call site is the
// commented block main.cpp:274-278).  glm::vec4 arithmetic including the w lanes: the decoded
// normal has w = 0*(1/127) - 1 = -1, so w takes part in every dot / reflect / refract below.
struct V4 { float x, y, z, w; };

__device__ __forceinline__ float dot4(const V4 &a, const V4 &b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

template <int ITER>
__device__ V4 raytrace(const DenseDev &g, float sx, float sy, float sz, V4 dir, Hit *primary)
{
    const float one_over_127 = 1.f / 127.f, one_over_255 = 1.f / 255.f;
    const Ray r = make_ray(dir.x, dir.y, dir.z);
    const Hit h = trace_dense_checked<false>(g, r, sx, sy, sz);
    if (primary) *primary = h;
    const bool found = h.index != VOLLY_MISS_INDEX;
    uint2 vox = make_uint2(0xFF000000u, 0u);              // Voxel(bgColor) = (0,0,0,255), main.cpp:159
    if (found) vox = g.vox[h.payload];
    const float cr = (float)(vox.x & 0xFFu), cg = (float)((vox.x >> 8) & 0xFFu), cb = (float)((vox.x >> 16) & 0xFFu), ca = (float)(vox.x >> 24);
    V4 out;
    if constexpr (ITER <= 1) {
        out.x = cr * one_over_255; out.y = cg * one_over_255; out.z = cb * one_over_255; out.w = ca * one_over_255;
        return out;
    } else {
        if (!found) {                                     // main.cpp:177
            out.x = cr * one_over_255; out.y = cg * one_over_255; out.z = cb * one_over_255; out.w = ca * one_over_255;
            return out;
        }
        V4 n;                                             // main.cpp:179-180, unfused
        n.x = (float)(vox.y & 0xFFu) * one_over_127 + -1.0f;
        n.y = (float)((vox.y >> 8) & 0xFFu) * one_over_127 + -1.0f;
        n.z = (float)((vox.y >> 16) & 0xFFu) * one_over_127 + -1.0f;
        n.w = (float)(vox.y >> 24) * one_over_127 + -1.0f;
        const float d = dot4(n, dir);
        V4 refl;                                          // glm::reflect: I - N * dot(N, I) * 2
        refl.x = dir.x - (n.x * d) * 2.f; refl.y = dir.y - (n.y * d) * 2.f;
        refl.z = dir.z - (n.z * d) * 2.f; refl.w = dir.w - (n.w * d) * 2.f;
        V4 refr = {0.f, 0.f, 0.f, 0.f};                   // glm::refract(I, N, 2)
        const float eta = 2.f;
        const float k = 1.f - (eta * eta) * (1.f - d * d);
        if (k >= 0.f) {
            const float f = eta * d + sqrtf(k);
            refr.x = eta * dir.x - f * n.x; refr.y = eta * dir.y - f * n.y;
            refr.z = eta * dir.z - f * n.z; refr.w = eta * dir.w - f * n.w;
        }
        const Ray rr = make_ray(refr.x, refr.y, refr.z);
        const Hit hs = trace_dense_checked<true>(g, rr, h.cx, h.cy, h.cz);       // main.cpp:185, solid
        const V4 c_refl = raytrace<ITER - 1>(g, h.cx, h.cy, h.cz, refl, nullptr);
        const V4 c_refr = raytrace<ITER - 1>(g, hs.cx, hs.cy, hs.cz, dir, nullptr);
        const float amt_reflect = 0.35f, amt_refract = 0.00f, amt_diffuse = 0.65f;
        out.x = (amt_reflect * c_refl.x + amt_refract * c_refr.x) + (amt_diffuse * cr) * one_over_255;   // main.cpp:193
        out.y = (amt_reflect * c_refl.y + amt_refract * c_refr.y) + (amt_diffuse * cg) * one_over_255;
        out.z = (amt_reflect * c_refl.z + amt_refract * c_refr.z) + (amt_diffuse * cb) * one_over_255;
        out.w = (amt_reflect * c_refl.w + amt_refract * c_refr.w) + (amt_diffuse * ca) * one_over_255;
        return out;
    }
}

// ---- pre-pass: serial ray-direction sums (main.cpp:254-255,282-286) -----------
// One thread per (frame, row of the sub-rectangle, component): the six components (target
// xyz, low xyz) are independent chains of dependent adds, so each gets its own thread.
// A thread walks its chain -- y adds down to its row, then along the row -- and stores the
// running sum every kCkpt pixels.  ckpt layout: [frame][column group][component][row], so
// the 64 rows of a workgroup store 256 contiguous bytes.
// g_lo .. g_hi: the 8-pixel groups of this row whose checkpoints a tile that is not sky will read (sky_span);
// the others are neither computed past nor stored.
__device__ __forceinline__ void checkpoint_chain(const volly_frame_desc &fd, int f, int comp, int yrel, float *__restrict__ ckpt, int g_lo, int g_hi)
{
    const int tile_w = fd.tile_w, tile_h = fd.tile_h, x0 = fd.x0;
    if (yrel >= tile_h || g_lo > g_hi) return;
    const int y = fd.y0 + yrel;
    const int lo = comp >= 3, k = comp - 3 * lo;
    float v = lo ? fd.row0_lo[k] : fd.row0_hi[k];
    const float dy = lo ? fd.ydiff_lo[k] : fd.ydiff_hi[k];
    const float dx = lo ? fd.xdiff_lo[k] : fd.xdiff_hi[k];
    int i = 0;
    for (; i + 8 <= y; i += 8) {                          // main.cpp:285-286
        #pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) v = v + dy;
    }
    for (; i < y; ++i) v = v + dy;
    for (int x = 0; x < x0; ++x) v = v + dx;              // main.cpp:282-283, columns left of the sub-rectangle
    const int groups = (tile_w + kCkpt - 1) / kCkpt;
    float *o = ckpt + ((size_t)f * (size_t)groups * 6 + (size_t)comp) * (size_t)tile_h + (size_t)yrel;
    const size_t stride = (size_t)tile_h * 6;
    const int last = g_hi < groups - 1 ? g_hi : groups - 1;
    #pragma unroll 4
    for (int g = 0; g <= last; ++g) {
        if (g >= g_lo) o[(size_t)g * stride] = v;
        #pragma unroll
        for (int i = 0; i < kCkpt; ++i) v = v + dx;
    }
}

// The pre-pass kernel.  The frame descriptors of a batch travel as kernel arguments: no
// host-to-device copy ahead of the pre-pass (4 us + a launch gap per call, and a blocked host
// when the caller's array is pageable).  The kernel also leaves the descriptors in frames_out,
// where the frame kernel reads them.
constexpr int kInlineFrames = 16;
constexpr int kHullEdges = 5;
// cull: per frame the pixel rectangle {x_lo, x_hi, y_lo, y_hi} (inclusive, relative to the sub-rectangle)
// outside of which every ray provably misses both grids (host: sky_rect in volly_gpu.hip).
// hull: per frame up to kHullEdges half-planes a*X + b*Y + c <= 0 (absolute pixel coordinates, (a, b) a unit
// vector, margin included) that hold the same pixels: the slanted edges of the projected boxes' convex hull
// (host: sky_hull_edges in volly_gpu.hip); an unused entry is (0, 0, -1).
struct FrameBatch { volly_frame_desc f[kInlineFrames]; int32_t cull[kInlineFrames][4]; float hull[kInlineFrames][kHullEdges][3]; };
static_assert(sizeof(FrameBatch) <= 4040, "kernel arguments are limited to 4 KB");

// Sky spans: for the tile row `row` (rows row*block_h .. +block_h-1 of the sub-rectangle) the range of columns
// [lo, hi] (relative to the sub-rectangle, inclusive) outside of which every ray of those rows misses both
// grids; (INT_MAX, -1): the whole tile row is sky (the frame kernel's two compares then cull every
// tile of it).  The frame kernel's sky test is two compares on one
// 8-byte scalar load.  Double precision and two pixels of slack on top of the margin the half-planes carry.
// (Host and device: volly_gpu_sky_spans_for runs the same function, for the CPU tests.)
__host__ __device__ __forceinline__ int2 sky_span(const volly_frame_desc &fd, const int32_t *rect, const float *hull, int row, int block_h)
{
    const int y0 = row * block_h, y1 = (y0 + block_h - 1 < fd.tile_h - 1) ? y0 + block_h - 1 : fd.tile_h - 1;
    int lo = rect[0], hi = rect[1];
    if (y1 < rect[2] || y0 > rect[3]) return make_int2(0x7fffffff, -1);
    const double Y0 = (double)(fd.y0 + y0), Y1 = (double)(fd.y0 + y1), X0 = (double)fd.x0, X1 = (double)(fd.x0 + fd.tile_w - 1);
    for (int e = 0; e < kHullEdges; ++e) {
        const double a = hull[3 * e], b = hull[3 * e + 1], c = hull[3 * e + 2];
        const double rhs = -c - fmin(b * Y0, b * Y1);     // a*X <= rhs for a pixel of these rows that may hit
        if (a > 1e-3) {
            const double x = floor(rhs / a + 2.0) - X0;
            if (x < (double)hi) hi = (int)fmax(x, -1.0);
        } else if (a < -1e-3) {
            const double x = ceil(rhs / a - 2.0) - X0;
            if (x > (double)lo) lo = (int)fmin(x, (double)fd.tile_w);
        } else if (fmin(a * X0, a * X1) > rhs + 0.5)
            return make_int2(0x7fffffff, -1);
    }
    return lo > hi ? make_int2(0x7fffffff, -1) : make_int2(lo, hi);
}

// base: index of the batch's first frame in the launch (a launch of more than kInlineFrames frames takes one
// pre-pass per kInlineFrames of them; frames_out points at the batch's first entry, spans_out at the launch's first).
__global__ void __launch_bounds__(64)
direction_checkpoints_inline(FrameBatch batch, int n_frames, volly_frame_desc *__restrict__ frames_out, int2 *__restrict__ spans_out,
                             float *__restrict__ ckpt, int base, int block_h)
{
    const int f = blockIdx.y;
    if (f >= n_frames) return;
    // the batch is the first argument: read it where it lies in the kernel-argument segment
    // (indexing the by-value copy with a run-time index would move it to scratch memory)
    const volly_frame_desc *args = (const volly_frame_desc *)__builtin_amdgcn_kernarg_segment_ptr();
    (void)batch;
    if (blockIdx.x == 0 && blockIdx.z == 0) {
        constexpr unsigned words = sizeof(volly_frame_desc) / 4;
        const uint32_t *src = (const uint32_t *)(args + f);
        uint32_t *dst = (uint32_t *)(frames_out + f);
        for (unsigned w = threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t *rect = (const int32_t *)(args + kInlineFrames) + 4 * f;
    const float *hull = (const float *)((const int32_t *)(args + kInlineFrames) + 4 * kInlineFrames) + 3 * kHullEdges * f;
    if (blockIdx.z == 0) {                                // one thread per tile row
        const int span_rows = (args[f].tile_h + block_h - 1) / block_h;
        if (t < span_rows) spans_out[(size_t)(base + f) * (size_t)span_rows + (size_t)t] = sky_span(args[f], rect, hull, t, block_h);
    }
    if (t >= args[f].tile_h) return;
    // the groups of row t that tiles inside its tile row's span cover (a tile is 64 / block_h pixels wide)
    const int2 sp = sky_span(args[f], rect, hull, t / block_h, block_h);
    const int tile_groups = 64 / block_h / kCkpt, tw_px = 64 / block_h;
    const int g_lo = sp.x > sp.y ? 1 : (sp.x < 0 ? 0 : sp.x / tw_px) * tile_groups;
    const int g_hi = sp.x > sp.y ? 0 : (sp.y < 0 ? 0 : sp.y / tw_px) * tile_groups + tile_groups - 1;
    checkpoint_chain(args[f], base + f, blockIdx.z, t, ckpt, g_lo, g_hi);
}

// what one ray leaves behind (RaycastReport + the pixel, main.cpp:81-86,269-271)
struct PixelOut {
    uint32_t rgba = 255u, hit = VOLLY_MISS_INDEX, slo = 0, shi = 0;
    float depth = __uint_as_float(0x7F800000u);
    float ex = 0.f, ey = 0.f, ez = 0.f;                  // RaycastReport::ptEnd
};

// the six direction sums of a pixel's checkpoint (target xyz, low xyz)
struct Ckpt { float v[6]; };

// NAIVE_ONLY (launches of naive frames only): the low-grid sums are neither stored by the pre-pass nor read
template <bool NAIVE_ONLY>
__device__ __forceinline__ Ckpt load_checkpoint(const float *__restrict__ ckpt, const volly_frame_desc &fd, int f, int xrel, int yrel)
{
    // checkpoint scratch of one launch stays below 2^31 floats (render_chunk): 32-bit indices
    const uint32_t groups = (uint32_t)(fd.tile_w + kCkpt - 1) / (uint32_t)kCkpt;
    const uint32_t cs = (uint32_t)fd.tile_h;
    const float *c = ckpt + (((uint32_t)f * groups + (uint32_t)xrel / (uint32_t)kCkpt) * 6u * cs + (uint32_t)yrel);
    Ckpt k;
    #pragma unroll
    for (int i = 0; i < (NAIVE_ONLY ? 3 : 6); ++i) k.v[i] = c[(uint32_t)i * cs];
    if (NAIVE_ONLY) k.v[3] = k.v[4] = k.v[5] = 0.0f;
    return k;
}

// main.cpp:256-272 for one pixel, from its checkpoint
// The serial adds from a pixel's checkpoint to the pixel (main.cpp:282-283) for the six direction sums in
// hx..lz (NAIVE_ONLY: the three target sums only).
template <bool NAIVE_ONLY>
__device__ __forceinline__ void direction_sums_from_checkpoint(const volly_frame_desc &fd, int xrel, float &hx, float &hy, float &hz,
                                                               float &lx, float &ly, float &lz)
{
#if VK_ASM_EXTRA && VK_CKPT_STRIDE == 8
    // main.cpp:282-283: lane l of an 8-pixel row segment is l & 7 serial adds past its checkpoint.
    // Which lanes still add in round i is known at compile time (every eighth lane drops out per
    // round), so the rounds run under constant exec masks: 7 x (2 scalar + 6 vector)
    // instructions, where the compiled loop over a per-lane trip count spends 7 x (10 + 8).
    {
        const float ah = fd.xdiff_hi[0], bh = fd.xdiff_hi[1], ch = fd.xdiff_hi[2];
        const float al = fd.xdiff_lo[0], bl = fd.xdiff_lo[1], cl = fd.xdiff_lo[2];
        (void)xrel;
#define VK_EXTRA_ROUND_HI(MASK)                                                                  \
            "s_and_b32 exec_lo, %[save0], " MASK "\n\t"                                          \
            "s_and_b32 exec_hi, %[save1], " MASK "\n\t"                                          \
            "v_add_f32 %[hx], %[hx], %[ah]\n\t"                                                  \
            "v_add_f32 %[hy], %[hy], %[bh]\n\t"                                                  \
            "v_add_f32 %[hz], %[hz], %[ch]\n\t"
#define VK_EXTRA_ROUND(MASK)                                                                     \
            VK_EXTRA_ROUND_HI(MASK)                                                              \
            "v_add_f32 %[lx], %[lx], %[al]\n\t"                                                  \
            "v_add_f32 %[ly], %[ly], %[bl]\n\t"                                                  \
            "v_add_f32 %[lz], %[lz], %[cl]\n\t"
        uint32_t save0, save1;
        if constexpr (NAIVE_ONLY) {                       // the low-grid sums are not needed: three adds per round
            asm volatile(
                "s_mov_b32 %[save0], exec_lo\n\t"
                "s_mov_b32 %[save1], exec_hi\n\t"
                VK_EXTRA_ROUND_HI("0xfefefefe")
                VK_EXTRA_ROUND_HI("0xfcfcfcfc")
                VK_EXTRA_ROUND_HI("0xf8f8f8f8")
                VK_EXTRA_ROUND_HI("0xf0f0f0f0")
                VK_EXTRA_ROUND_HI("0xe0e0e0e0")
                VK_EXTRA_ROUND_HI("0xc0c0c0c0")
                VK_EXTRA_ROUND_HI("0x80808080")
                "s_mov_b32 exec_lo, %[save0]\n\t"
                "s_mov_b32 exec_hi, %[save1]"
                : [hx] "+v"(hx), [hy] "+v"(hy), [hz] "+v"(hz), [save0] "=&s"(save0), [save1] "=&s"(save1)
                : [ah] "s"(ah), [bh] "s"(bh), [ch] "s"(ch)
                : "scc");
            (void)al; (void)bl; (void)cl;
        } else
        asm volatile(
            "s_mov_b32 %[save0], exec_lo\n\t"
            "s_mov_b32 %[save1], exec_hi\n\t"
            VK_EXTRA_ROUND("0xfefefefe")
            VK_EXTRA_ROUND("0xfcfcfcfc")
            VK_EXTRA_ROUND("0xf8f8f8f8")
            VK_EXTRA_ROUND("0xf0f0f0f0")
            VK_EXTRA_ROUND("0xe0e0e0e0")
            VK_EXTRA_ROUND("0xc0c0c0c0")
            VK_EXTRA_ROUND("0x80808080")
            "s_mov_b32 exec_lo, %[save0]\n\t"
            "s_mov_b32 exec_hi, %[save1]"
            : [hx] "+v"(hx), [hy] "+v"(hy), [hz] "+v"(hz), [lx] "+v"(lx), [ly] "+v"(ly), [lz] "+v"(lz),
              [save0] "=&s"(save0), [save1] "=&s"(save1)
            : [ah] "s"(ah), [bh] "s"(bh), [ch] "s"(ch), [al] "s"(al), [bl] "s"(bl), [cl] "s"(cl)
            : "scc");
#undef VK_EXTRA_ROUND
#undef VK_EXTRA_ROUND_HI
    }
#else
    const int extra = xrel % kCkpt;
    for (int i = 0; i < extra; ++i) {                     // main.cpp:282-283
        hx = hx + fd.xdiff_hi[0]; hy = hy + fd.xdiff_hi[1]; hz = hz + fd.xdiff_hi[2];
        lx = lx + fd.xdiff_lo[0]; ly = ly + fd.xdiff_lo[1]; lz = lz + fd.xdiff_lo[2];
    }
#endif
}

template <bool SVO, bool RAYTRACE, bool NAIVE_ONLY>
__device__ __forceinline__ PixelOut render_pixel(const DenseDev &target, const SvoDev &svo, const DenseDev &low, const volly_frame_desc &fd,
                                                 const Ckpt &k, int xrel)
{
    PixelOut o;
    float hx = k.v[0], hy = k.v[1], hz = k.v[2], lx = k.v[3], ly = k.v[4], lz = k.v[5];
    direction_sums_from_checkpoint<NAIVE_ONLY>(fd, xrel, hx, hy, hz, lx, ly, lz);
    const float hw = normalize_w0(hx, hy, hz);            // main.cpp:260
    const Ray ray_hi = make_ray(hx, hy, hz);
    if constexpr (!SVO && RAYTRACE) {
        if (fd.flags & VOLLY_FRAME_RAYTRACE) {            // main.cpp:274-278 (commented call site)
            Hit prim;
            const V4 col = raytrace<3>(target, fd.origin_hi[0], fd.origin_hi[1], fd.origin_hi[2], V4{hx, hy, hz, hw}, &prim);
            const uint32_t tr = (uint32_t)(fminf(fmaxf(col.x, 0.0f), 1.0f) * 255.f) & 0xFFu;
            const uint32_t tg = (uint32_t)(fminf(fmaxf(col.y, 0.0f), 1.0f) * 255.f) & 0xFFu;
            const uint32_t tb = (uint32_t)(fminf(fmaxf(col.z, 0.0f), 1.0f) * 255.f) & 0xFFu;
            o.rgba = 255u | (tr << 8) | (tg << 16) | (tb << 24);
            o.shi = prim.steps;
            o.hit = prim.index;
            o.ex = prim.cx; o.ey = prim.cy; o.ez = prim.cz;
            if (prim.index != VOLLY_MISS_INDEX) {
                float ddx = prim.cx - fd.origin_hi[0], ddy = prim.cy - fd.origin_hi[1], ddz = prim.cz - fd.origin_hi[2];
                o.depth = (ddx * ray_hi.dx + ddy * ray_hi.dy) + ddz * ray_hi.dz;
            }
            return o;
        }
    }
    (void)hw;

    float sx, sy, sz;                                     // start of the target-grid pass
    if (NAIVE_ONLY || (fd.flags & VOLLY_FRAME_NAIVE)) {   // main.cpp:261
        sx = fd.origin_hi[0]; sy = fd.origin_hi[1]; sz = fd.origin_hi[2];
    } else {                                              // raycast_beamOpt, main.cpp:163-166
        normalize_w0(lx, ly, lz);
        const Ray ray_lo = make_ray(lx, ly, lz);
        Hit lo = trace_dense(low, ray_lo, fd.origin_lo[0], fd.origin_lo[1], fd.origin_lo[2]);
        o.slo = lo.steps;
        sx = fd.scale_lo_to_hi[0] * lo.cx; sy = fd.scale_lo_to_hi[1] * lo.cy; sz = fd.scale_lo_to_hi[2] * lo.cz;
    }

    Hit hi;
    if constexpr (SVO) {
#if VK_SVO_BLOCKS
        // one path per wavefront, as in trace_dense
#if VK_ASM_SVO
        if (svo_is_regular(svo) && __all(ray_is_regular(ray_hi))) hi = trace_svo_blocks_asm(svo, ray_hi, sx, sy, sz);
#else
        if (svo_is_regular(svo) && __all(ray_is_regular(ray_hi))) hi = trace_svo_blocks(svo, ray_hi, sx, sy, sz);
#endif
        else
#endif
            hi = trace_svo(svo, ray_hi, sx, sy, sz);
    } else
        hi = trace_dense(target, ray_hi, sx, sy, sz);
    o.shi = hi.steps;
    o.hit = hi.index;
    o.ex = hi.cx; o.ey = hi.cy; o.ez = hi.cz;
    if (hi.index != VOLLY_MISS_INDEX) {
        uint2 vox = SVO ? svo.leaves[hi.payload] : target.vox[hi.payload];
        o.rgba = shade(vox);
        float ddx = hi.cx - fd.origin_hi[0], ddy = hi.cy - fd.origin_hi[1], ddz = hi.cz - fd.origin_hi[2];
        o.depth = (ddx * ray_hi.dx + ddy * ray_hi.dy) + ddz * ray_hi.dz;
    }
    return o;
}

// ---- the frame kernel: main.cpp:256-272 for every pixel of the sub-rectangle --
// RAYTRACE: the instantiation that can also serve VOLLY_FRAME_RAYTRACE frames (dense targets);
// it needs 96 VGPRs, so frames of the live path never run through it.
template <bool SVO, bool RAYTRACE = false, int TW = kTileBeam>
__global__ void __launch_bounds__(kFrameThreads)
render_frames(DenseDev target, SvoDev svo, DenseDev low, const volly_frame_desc *__restrict__ frames,
              const int2 *__restrict__ spans, const float *__restrict__ ckpt, OutDev out)
{
    const int f = blockIdx.z;
    // uniform, read with scalar loads where it is used: a by-value copy keeps ~50 scalar registers
    // busy for the whole kernel, and measured 0.5-1 % slower (profiles/r01_ab/ab_tiles_per_wave_c2.log, "new" vs "base")
    const volly_frame_desc &fd = frames[f];
    // a wavefront covers a TW x 64/TW tile; workgroups stay in dispatch order (round-robin over the XCDs
    // balances sky and object tiles; an XCD-contiguous order measured 12-60 % slower)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int kBlockW = TW * VK_WAVE_BLOCK, kBlockH = 64 / TW;       // workgroup covers VK_WAVE_BLOCK tiles side by side
    // Which tile of its row this workgroup renders.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share
    // one), so with tile = blockIdx.x the four 32-byte segments of a 128-byte row chunk are stored through four different,
    // mutually incoherent L2s and none of them can merge the chunk: 83 M write requests of 32 bytes per launch instead of
    // 21 M of 128 (profiles/r03/pmc_summary.csv, TCC_EA0_WRREQ).  With 8 x 8 tiles the columns of a row are therefore taken
    // in groups of 32: workgroups r, r + 8, r + 16, r + 24 of a group -- one XCD -- render four tiles side by side.  Every
    // XCD still gets every eighth four-tile chunk of a row, and a different one in the next row (without the rotation a
    // narrow upright model lands on two of the eight XCDs: config 5 was 2.5 % slower): the load stays balanced, unlike
    // with an XCD-contiguous order (12-60 % slower in round 1).  Needs a row of a multiple of 8 workgroups, so that a
    // row's first workgroup always falls on the same XCD; the last, shorter group of a row gives each XCD fewer tiles.
    int tile_x = (int)blockIdx.x;
#if VK_XCD_GROUPS
    if constexpr (TW == kTileBeam && VK_WAVE_BLOCK == 1) {
        const unsigned gx = gridDim.x, x = blockIdx.x;
        if ((gx & 7u) == 0u) {
            constexpr unsigned kG = 8u * VK_XCD_GROUPS;                       // workgroups per group: VK_XCD_GROUPS tiles side by side per XCD
            const unsigned full = gx - gx % kG;
            const unsigned base = x < full ? x - x % kG : full;
            const unsigned k = x < full ? (unsigned)VK_XCD_GROUPS : (gx - full) >> 3;   // tiles per XCD in this group
            const unsigned r = x - base;
            tile_x = (int)(base + k * ((r + blockIdx.y) & 7u) + (r >> 3));   // which XCD takes which chunk rotates from row to row
        }
    }
#endif
    const int xrel = tile_x * kBlockW + wave * TW + (lane & (TW - 1));
    const int yrel = (int)blockIdx.y * kBlockH + (lane / TW);
    if ((xrel < fd.tile_w) & (yrel < fd.tile_h)) {       // eight 32-byte row segments per store instruction
        // Sky tiles: outside its tile row's span (sky_span) every ray is known to end both passes at its
        // first bounds test (0 steps, no hit -- what PixelOut starts as), so the 288-instruction
        // set-up of direction, rays and entry points is skipped.  Wave-uniform: the tile's corner.
        const int2 span = spans[(uint32_t)f * gridDim.y + blockIdx.y];
        const int tx = tile_x * kBlockW + (VK_WAVE_BLOCK > 1 ? __builtin_amdgcn_readfirstlane(wave) * TW : 0);
        const bool sky = (tx + TW - 1 < span.x) | (tx > span.y);
#if VK_SKY_BLOCKS
        if constexpr (TW == kTileBeam && VK_WAVE_BLOCK == 1) {
            // Four sky tiles side by side (a 32 x 8 block inside the sub-rectangle, all of it outside the span): the first
            // tile's wavefront writes the whole block, 16 bytes per lane, so that every store instruction fills eight
            // 128-byte rows instead of eight 32-byte segments; the other three wavefronts have nothing to do.
            const int bx = tx & ~(4 * TW - 1), ty = (int)blockIdx.y * kBlockH;
            if ((bx + 4 * TW - 1 < fd.tile_w) & (ty + kBlockH - 1 < fd.tile_h) && ((bx + 4 * TW - 1 < span.x) | (bx > span.y))) {
                if (tx != bx) return;
                const size_t q = (size_t)f * (size_t)fd.tile_w * (size_t)fd.tile_h + (size_t)(ty + (lane >> 3)) * (size_t)fd.tile_w + (size_t)(bx + 4 * (lane & 7));
                const PixelOut miss;
                if (out.rgba8) *(uint4 *)(out.rgba8 + q) = make_uint4(miss.rgba, miss.rgba, miss.rgba, miss.rgba);
                if (out.hit_index) *(uint4 *)(out.hit_index + q) = make_uint4(miss.hit, miss.hit, miss.hit, miss.hit);
                if (out.steps_lo) *(uint4 *)(out.steps_lo + q) = make_uint4(0u, 0u, 0u, 0u);
                if (out.steps_hi) *(uint4 *)(out.steps_hi + q) = make_uint4(0u, 0u, 0u, 0u);
                if (out.depth) *(float4 *)(out.depth + q) = make_float4(miss.depth, miss.depth, miss.depth, miss.depth);
                return;                                   // (pt_end is never requested where tiles are culled: render_chunk)
            }
        }
        if constexpr (TW == kTileNaive && VK_WAVE_BLOCK == 1) {
            // The same for the 32 x 2 tiles of naive launches, stacked: four sky tiles one above the other (each tile row has
            // its own span) are one 32 x 8 block, written by the topmost tile's wavefront.
            const int by = (int)blockIdx.y & ~3, ty = by * kBlockH;
            if ((tx + TW - 1 < fd.tile_w) & (ty + 4 * kBlockH - 1 < fd.tile_h)) {
                const int2 *sp = spans + (uint32_t)f * gridDim.y + by;
                const int2 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
                const bool all_sky = ((tx + TW - 1 < s0.x) | (tx > s0.y)) & ((tx + TW - 1 < s1.x) | (tx > s1.y)) &
                                     ((tx + TW - 1 < s2.x) | (tx > s2.y)) & ((tx + TW - 1 < s3.x) | (tx > s3.y));
                if (all_sky) {
                    if ((int)blockIdx.y != by) return;
                    const size_t q = (size_t)f * (size_t)fd.tile_w * (size_t)fd.tile_h + (size_t)(ty + (lane >> 3)) * (size_t)fd.tile_w + (size_t)(tx + 4 * (lane & 7));
                    const PixelOut miss;
                    if (out.rgba8) *(uint4 *)(out.rgba8 + q) = make_uint4(miss.rgba, miss.rgba, miss.rgba, miss.rgba);
                    if (out.hit_index) *(uint4 *)(out.hit_index + q) = make_uint4(miss.hit, miss.hit, miss.hit, miss.hit);
                    if (out.steps_lo) *(uint4 *)(out.steps_lo + q) = make_uint4(0u, 0u, 0u, 0u);
                    if (out.steps_hi) *(uint4 *)(out.steps_hi + q) = make_uint4(0u, 0u, 0u, 0u);
                    if (out.depth) *(float4 *)(out.depth + q) = make_float4(miss.depth, miss.depth, miss.depth, miss.depth);
                    return;
                }
            }
        }
#endif
        PixelOut o;
        if (!sky) {
            constexpr bool kNaiveOnly = TW == kTileNaive && !RAYTRACE;     // this instantiation serves launches of naive frames only
            const Ckpt k = load_checkpoint<kNaiveOnly>(ckpt, fd, f, xrel, yrel);
            o = render_pixel<SVO, RAYTRACE, kNaiveOnly>(target, svo, low, fd, k, xrel);
        }
        const size_t p = (size_t)f * (size_t)fd.tile_w * (size_t)fd.tile_h + (size_t)yrel * (size_t)fd.tile_w + (size_t)xrel;
        if (out.rgba8) out.rgba8[p] = o.rgba;
        if (out.hit_index) out.hit_index[p] = o.hit;
        if (out.steps_lo) out.steps_lo[p] = o.slo;
        if (out.steps_hi) out.steps_hi[p] = o.shi;
        if (out.depth) out.depth[p] = o.depth;
        if (out.pt_end) { out.pt_end[3 * p] = o.ex; out.pt_end[3 * p + 1] = o.ey; out.pt_end[3 * p + 2] = o.ez; }
    }
}

// ---- the frame kernel for dense targets whose rays have very uneven lengths (VK_PACKED) ----------------
// In a grid without the region level -- the noise volume: every 16^3 region holds voxels -- ray lengths are
// geometric: a wavefront of render_frames lives as long as the longest of its 64 rays and spends its vector
// instructions at 17 of 64 lanes (profiles/r02: config 3).  Here a workgroup of kPackWaves wavefronts renders a
// block of kPackWaves 8x8 tiles side by side.  Set-up, low-grid pass and the entry into the target grid run tile
// by tile as in render_frames; the target pass then runs in rounds of `budget` iterations (the assembly loop
// over the bricks, re-entered), and between rounds the rays still under way are packed through LDS into as few
// full wavefronts as hold them -- in pixel order, so that neighbouring rays stay neighbours in neighbouring
// lanes and keep sharing brick words -- while the wavefronts left without rays end.  A ray's arithmetic is
// untouched: position and direction travel as they are, the ray constants are recomputed by the same make_ray
// from the same direction, and every ray of the block is at the same iteration number (all start the target pass
// at iteration 0), so the step count is still the iteration at which the lane leaves.
// State per ray: position, direction, pixel = 28 bytes; 7 KiB of LDS per workgroup.
constexpr int kPackWaves = 4;                         // (2, 8 and 16 wavefronts per workgroup measured slower: profiles/r03_ab/r03_ab_packwaves_c3.log)

// Rays parked between two passes of the packed kernels: position, direction and pixel (index into the launch's
// planes) as seven arrays of `capacity` words, and how many are in it.
struct RayQueue { float *q; uint32_t *count; uint32_t capacity; };

// what a finished ray of the target pass leaves behind (render_pixel's tail, main.cpp:262-272): every plane but
// steps_lo.  p: the pixel's index in the launch's planes; its frame (for camPos_Local) is p / frame_px.
__device__ __forceinline__ void store_finished_ray(const DenseDev &target, const volly_frame_desc *__restrict__ frames, uint32_t frame_px,
                                                   const OutDev &out, uint32_t p, float cx, float cy, float cz, float dx, float dy, float dz,
                                                   uint32_t steps, uint32_t hitc)
{
    uint32_t rgba = 255u, index = VOLLY_MISS_INDEX;
    float depth = __uint_as_float(0x7F800000u);
    if (hitc != 0xFFFFFFFFu) {
        const uint32_t x = hitc & 1023u, y = (hitc >> 10) & 1023u, z = hitc >> 20;
        if (rim_exit(target, x, y, z)) {
            steps -= 1;
        } else {
            index = x + (uint32_t)target.sx * (y + (uint32_t)target.sy * z);
            rgba = shade(target.vox[index]);
            const volly_frame_desc &fd = frames[p / frame_px];
            const float ddx = cx - fd.origin_hi[0], ddy = cy - fd.origin_hi[1], ddz = cz - fd.origin_hi[2];
            depth = (ddx * dx + ddy * dy) + ddz * dz;
        }
    }
    if (out.rgba8) out.rgba8[p] = rgba;
    if (out.hit_index) out.hit_index[p] = index;
    if (out.steps_hi) out.steps_hi[p] = steps;
    if (out.depth) out.depth[p] = depth;
    if (out.pt_end) { out.pt_end[3 * (size_t)p] = cx; out.pt_end[3 * (size_t)p + 1] = cy; out.pt_end[3 * (size_t)p + 2] = cz; }
}

struct PackShared {
    float st[7][64 * kPackWaves];
    uint32_t wave_live[2][kPackWaves];
    uint32_t queue_base;
};

// The rounds of one pass for the workgroup's rays (live lanes hold one each: position c, direction d, pixel pix), all at
// iteration `iter0`: pack, run `budget` iterations, store what finished, until no ray is left or iteration `park_at` is
// reached -- then the rays still under way are appended to `park` (all at iteration park_at) for the next pass.
// park_at >= the grid's iteration cap: the pass runs every ray to its end.
template <bool LOOKAHEAD>
__device__ __forceinline__ void packed_rounds(const DenseDev &target, const volly_frame_desc *__restrict__ frames, uint32_t frame_px, const OutDev &out,
                                              PackShared &sh, int wave, int lane, bool live, float cx, float cy, float cz, float dx, float dy, float dz,
                                              uint32_t pix, uint32_t iter0, uint32_t budget, uint32_t park_at, const RayQueue &park)
{
    int n_waves = kPackWaves, parity = 0;
    for (;; parity ^= 1) {
        const uint64_t mine = __ballot(live);
        if (lane == 0) sh.wave_live[parity][wave] = (uint32_t)__popcll(mine);
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (int w = 0; w < n_waves; ++w) {               // (wavefronts >= n_waves have ended)
            const uint32_t c = sh.wave_live[parity][w];
            if (w < wave) base += c;
            total += c;
        }
        if (total == 0) return;
        const uint32_t rank = base + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
        if (iter0 >= park_at) {                           // ---- park: this pass ends here for every ray of the block
            if (wave == 0 && lane == 0) sh.queue_base = atomicAdd(park.count, total);
            __syncthreads();
            if (live) {
                const size_t slot = (size_t)sh.queue_base + rank, n = park.capacity;
                park.q[slot] = cx; park.q[n + slot] = cy; park.q[2 * n + slot] = cz;
                park.q[3 * n + slot] = dx; park.q[4 * n + slot] = dy; park.q[5 * n + slot] = dz;
                park.q[6 * n + slot] = __uint_as_float(pix);
            }
            return;
        }
        if (live) {
            sh.st[0][rank] = cx; sh.st[1][rank] = cy; sh.st[2][rank] = cz;
            sh.st[3][rank] = dx; sh.st[4][rank] = dy; sh.st[5][rank] = dz;
            sh.st[6][rank] = __uint_as_float(pix);
        }
        __syncthreads();
        n_waves = (int)((total + 63u) >> 6);
        if (wave >= n_waves) return;                      // no rays left for this wavefront
        const uint32_t slot = (uint32_t)wave * 64u + (uint32_t)lane;
        live = slot < total;
        uint32_t iter_end = (iter0 + budget > iter0 && iter0 + budget < target.cap) ? iter0 + budget : target.cap;
        if (iter_end > park_at) iter_end = park_at;
        if (live) {
            cx = sh.st[0][slot]; cy = sh.st[1][slot]; cz = sh.st[2][slot];
            dx = sh.st[3][slot]; dy = sh.st[4][slot]; dz = sh.st[5][slot];
            pix = __float_as_uint(sh.st[6][slot]);
            const Ray r = make_ray(dx, dy, dz);
            uint32_t steps = 0xFFFFFFFFu, hitc = 0xFFFFFFFFu;
            const uint32_t iter = bricks_loop_asm<LOOKAHEAD>(target, r, cx, cy, cz, iter0, iter_end, steps, hitc);
            if (steps == 0xFFFFFFFFu && iter_end >= target.cap) steps = iter;     // still live at the iteration cap: a miss with that many steps
            if (steps != 0xFFFFFFFFu) {
                store_finished_ray(target, frames, frame_px, out, pix, cx, cy, cz, dx, dy, dz, steps, hitc);
                live = false;
            }
        }
        iter0 = iter_end;
    }
}

template <bool NAIVE_ONLY>
__global__ void __launch_bounds__(64 * kPackWaves)
render_frames_packed(DenseDev target, DenseDev low, const volly_frame_desc *__restrict__ frames, const int2 *__restrict__ spans,
                     const float *__restrict__ ckpt, OutDev out, uint32_t budget, uint32_t park_at, RayQueue park)
{
    __shared__ PackShared sh;
    const int f = blockIdx.z;
    const volly_frame_desc &fd = frames[f];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int xrel = (int)blockIdx.x * (kTileBeam * kPackWaves) + wave * kTileBeam + (lane & (kTileBeam - 1));
    const int yrel = (int)blockIdx.y * kTileBeam + (lane / kTileBeam);
    const uint32_t frame_px = (uint32_t)fd.tile_w * (uint32_t)fd.tile_h;          // (a launch holds fewer than 2^32 pixels: render_chunk)
    bool live = false;
    float cx = 0.f, cy = 0.f, cz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    uint32_t pix = 0;
    if ((xrel < fd.tile_w) & (yrel < fd.tile_h)) {
        const int2 span = spans[(uint32_t)f * gridDim.y + blockIdx.y];
        const int tx = (int)blockIdx.x * (kTileBeam * kPackWaves) + wave * kTileBeam;
        const bool sky = (tx + kTileBeam - 1 < span.x) | (tx > span.y);
        pix = (uint32_t)f * frame_px + (uint32_t)yrel * (uint32_t)fd.tile_w + (uint32_t)xrel;
        PixelOut o;
        if (!sky) {
            const Ckpt k = load_checkpoint<NAIVE_ONLY>(ckpt, fd, f, xrel, yrel);
            float hx = k.v[0], hy = k.v[1], hz = k.v[2], lx = k.v[3], ly = k.v[4], lz = k.v[5];
            direction_sums_from_checkpoint<NAIVE_ONLY>(fd, xrel, hx, hy, hz, lx, ly, lz);
            normalize_w0(hx, hy, hz);                     // main.cpp:260
            const Ray ray_hi = make_ray(hx, hy, hz);
            float sx, sy, sz;                             // start of the target-grid pass
            if (NAIVE_ONLY || (fd.flags & VOLLY_FRAME_NAIVE)) {   // main.cpp:261
                sx = fd.origin_hi[0]; sy = fd.origin_hi[1]; sz = fd.origin_hi[2];
            } else {                                      // raycast_beamOpt, main.cpp:163-166
                normalize_w0(lx, ly, lz);
                const Ray ray_lo = make_ray(lx, ly, lz);
                const Hit lo = trace_dense(low, ray_lo, fd.origin_lo[0], fd.origin_lo[1], fd.origin_lo[2]);
                o.slo = lo.steps;
                sx = fd.scale_lo_to_hi[0] * lo.cx; sy = fd.scale_lo_to_hi[1] * lo.cy; sz = fd.scale_lo_to_hi[2] * lo.cz;
            }
            if (__all(ray_is_regular(ray_hi))) {          // (the host launches this kernel for regular grids only)
                enter_volume(ray_hi, sx, sy, sz, target.fsx, target.fsy, target.fsz, cx, cy, cz);
                dx = hx; dy = hy; dz = hz;
                o.ex = cx; o.ey = cy; o.ez = cz;          // an inf / NaN entry fails its first bounds test: a miss that ends here
                live = fabsf(cx) < 3.0e38f && fabsf(cy) < 3.0e38f && fabsf(cz) < 3.0e38f;
            } else {                                      // a wavefront with an irregular ray: the checked loop, as in trace_dense
                const Hit hi = trace_dense_checked<false>(target, ray_hi, sx, sy, sz);
                o.shi = hi.steps;
                o.hit = hi.index;
                o.ex = hi.cx; o.ey = hi.cy; o.ez = hi.cz;
                if (hi.index != VOLLY_MISS_INDEX) {
                    o.rgba = shade(target.vox[hi.payload]);
                    const float ddx = hi.cx - fd.origin_hi[0], ddy = hi.cy - fd.origin_hi[1], ddz = hi.cz - fd.origin_hi[2];
                    o.depth = (ddx * ray_hi.dx + ddy * ray_hi.dy) + ddz * ray_hi.dz;
                }
            }
        }
        const size_t p = pix;
        if (out.steps_lo) out.steps_lo[p] = o.slo;
        if (!live) {
            if (out.rgba8) out.rgba8[p] = o.rgba;
            if (out.hit_index) out.hit_index[p] = o.hit;
            if (out.steps_hi) out.steps_hi[p] = o.shi;
            if (out.depth) out.depth[p] = o.depth;
            if (out.pt_end) { out.pt_end[3 * p] = o.ex; out.pt_end[3 * p + 1] = o.ey; out.pt_end[3 * p + 2] = o.ez; }
        }
    }
    packed_rounds<false>(target, frames, frame_px, out, sh, wave, lane, live, cx, cy, cz, dx, dy, dz, pix, 0u, budget, park_at, park);
}

// A later pass: the rays parked by the pass before (all at iteration `iter0`), 64 * kPackWaves of them per workgroup in
// the order they were parked (a block's rays stay together), through the same rounds.  LOOKAHEAD: with the loop that
// requests the bricks ahead of each ray (the late passes: few, long rays, bound by load latency).
template <bool LOOKAHEAD>
__global__ void __launch_bounds__(64 * kPackWaves)
continue_rays_packed(DenseDev target, const volly_frame_desc *__restrict__ frames, uint32_t frame_px, OutDev out, RayQueue from,
                     uint32_t iter0, uint32_t budget, uint32_t park_at, RayQueue park)
{
    __shared__ PackShared sh;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t count = *from.count, n = from.capacity;
    const uint32_t i = blockIdx.x * (64u * kPackWaves) + threadIdx.x;          // (the grid covers the queue's capacity: most workgroups end here)
    if (blockIdx.x * (64u * kPackWaves) >= count) return;
    const bool live = i < count;
    float cx = 0.f, cy = 0.f, cz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    uint32_t pix = 0;
    if (live) {
        cx = from.q[i]; cy = from.q[n + i]; cz = from.q[2 * (size_t)n + i];
        dx = from.q[3 * (size_t)n + i]; dy = from.q[4 * (size_t)n + i]; dz = from.q[5 * (size_t)n + i];
        pix = __float_as_uint(from.q[6 * (size_t)n + i]);
    }
    packed_rounds<LOOKAHEAD>(target, frames, frame_px, out, sh, wave, lane, live, cx, cy, cz, dx, dy, dz, pix, iter0, budget, park_at, park);
}

// ---- scene preparation --------------------------------------------------------

// occupancy bricks from the reference's voxels: bit set iff rgba.a > 0 (main.cpp:140)
// (bricks cover the padded grid: the rim cells beyond sx, sy, sz are set, DenseDev::px)
__global__ void __launch_bounds__(256) build_occupancy(const uint2 *vox, int sx, int sy, int sz, int nbx, int nby, int nbz,
                                                       uint64_t *occ)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nb = (size_t)nbx * nby * nbz;
    if (b >= nb) return;
    const int bx = (int)(b % (size_t)nbx), by = (int)((b / (size_t)nbx) % (size_t)nby), bz = (int)(b / ((size_t)nbx * nby));
    uint64_t bits = 0;
    for (int k = 0; k < 64; ++k) {
        const int x = bx * 4 + (k & 3), y = by * 4 + ((k >> 2) & 3), z = bz * 4 + (k >> 4);
        if (x < sx && y < sy && z < sz) {
            const uint2 v = vox[(size_t)x + (size_t)sx * ((size_t)y + (size_t)sy * (size_t)z)];
            if ((v.x >> 24) != 0) bits |= 1ull << k;
        } else {
            bits |= 1ull << k;
        }
    }
    occ[b] = bits;
}

// second occupancy level: one word per 4x4x4 bricks, bit set iff the brick holds a voxel
__global__ void __launch_bounds__(256) build_occupancy2(const uint64_t *occ, int nbx, int nby, int nbz, int nrx, int nry, int nrz,
                                                        uint64_t *occ2, uint32_t *nonempty_regions, unsigned long long *filled_voxels)
{
    const size_t rg = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nr = (size_t)nrx * nry * nrz;
    if (rg >= nr) return;
    const int rx = (int)(rg % (size_t)nrx), ry = (int)((rg / (size_t)nrx) % (size_t)nry), rz = (int)(rg / ((size_t)nrx * nry));
    uint64_t bits = 0;
    unsigned filled = 0;
    for (int k = 0; k < 64; ++k) {
        const int bx = rx * 4 + (k & 3), by = ry * 4 + ((k >> 2) & 3), bz = rz * 4 + (k >> 4);
        if (bx < nbx && by < nby && bz < nbz) {
            const uint64_t w = occ[(size_t)bx + (size_t)nbx * ((size_t)by + (size_t)nby * (size_t)bz)];
            if (w) { bits |= 1ull << k; filled += (unsigned)__popcll(w); }
        }
    }
    occ2[rg] = bits;
    if (bits) { atomicAdd(nonempty_regions, 1u); atomicAdd(filled_voxels, (unsigned long long)filled); }
}

// Region entries of the whole-traversal loop: the occ2 word plus, for an empty region, the key mask of
// a large aligned box of 2^Lx x 2^Ly x 2^Lz voxels (L = 4..max_level per axis) around it that is empty
// and lies inside the grid: a lane that enters such a region stays quiet until its cell leaves the whole
// box, not just the 16^3 region.  With cubes (Lx = Ly = Lz) the headline workload has a region change in
// 15 % of its wave-steps instead of 25 % (tools/wave_step_census.py); the box grows one axis at a time,
// x, y, z in turn, as long as the doubled box is still empty (boxes = 0: all three together, cubes only),
// so it holds the largest empty cube and extends beyond it along the axes that stay free (the slabs of
// empty space above, below and beside a model).  Mask = the bits of the packed cell x | y<<10 | z<<20
// above each axis' L: the loop's test `(cell & mask) == key` does not care that the levels differ.
__global__ void __launch_bounds__(256) build_region_cubes(const uint64_t *occ2, int nrx, int nry, int nrz, int max_level, int boxes, uint4 *occ2x)
{
    const size_t rg = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nr = (size_t)nrx * nry * nrz;
    if (rg >= nr) return;
    const int r[3] = {(int)(rg % (size_t)nrx), (int)((rg / (size_t)nrx) % (size_t)nry), (int)(rg / ((size_t)nrx * nry))};
    const int n[3] = {nrx, nry, nrz};
    const uint64_t word = occ2[rg];
    int L[3] = {4, 4, 4};
    if (word == 0) {
        auto empty_box = [&](const int *lv) {
            int lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                const int b = 1 << (lv[a] - 4);               // box side in regions
                lo[a] = r[a] & ~(b - 1);
                hi[a] = lo[a] + b;
                if (hi[a] > n[a]) return false;
            }
            for (int z = lo[2]; z < hi[2]; ++z)
                for (int y = lo[1]; y < hi[1]; ++y)
                    for (int x = lo[0]; x < hi[0]; ++x)
                        if (occ2[(size_t)x + (size_t)nrx * ((size_t)y + (size_t)nry * (size_t)z)]) return false;
            return true;
        };
        if (!boxes) {
            for (int lv = 5; lv <= max_level; ++lv) {
                const int t[3] = {lv, lv, lv};
                if (!empty_box(t)) break;
                L[0] = L[1] = L[2] = lv;
            }
        } else {
            // (the best of the six axis orders by volume measured the same: r02_ab_boxorder_c*.log)
            bool grew = true;
            while (grew) {
                grew = false;
                for (int a = 0; a < 3; ++a) {
                    if (L[a] >= max_level) continue;
                    int t[3] = {L[0], L[1], L[2]};
                    ++t[a];
                    if (empty_box(t)) { L[a] = t[a]; grew = true; }
                }
            }
        }
    }
    const uint32_t fx = (1023u >> L[0]) << L[0], fy = (1023u >> L[1]) << L[1], fz = (1023u >> L[2]) << L[2];
    const uint32_t mask = 0xC0000000u | (fz << 20) | (fy << 10) | fx;
    occ2x[rg] = make_uint4((uint32_t)word, (uint32_t)(word >> 32), mask, 0u);
}

// Upper octree levels as a direct-index table: entry (tx,ty,tz) describes the octree cell
// of 2^tshift voxels at (tx,ty,tz) << tshift: the index of the node covering it, or, when the
// walk from the root ends earlier in an empty cell of 2^k voxels, the float -(2^k) (sign bit = empty;
// its magnitude is the step's cell size as the advance uses it).
__global__ void __launch_bounds__(256) build_svo_table(const uint2 *nodes, int depth, int tbits, int tshift, uint32_t *table)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)1 << (3 * tbits);
    if (i >= n) return;
    const uint32_t m = (1u << tbits) - 1u;
    const uint32_t x = ((uint32_t)i & m) << tshift, y = (((uint32_t)(i >> tbits)) & m) << tshift, z = ((uint32_t)(i >> (2 * tbits)) & m) << tshift;
    uint32_t node = 0, entry = 0;
    bool done = false;
    for (int sh = depth - 1; sh >= tshift; --sh) {
        const uint2 nd = nodes[node];
        const uint32_t oct = ((x >> sh) & 1u) | (((y >> sh) & 1u) << 1) | (((z >> sh) & 1u) << 2);
        const uint32_t valid = nd.y & 0xFFu, bit = 1u << oct;
        if (!(valid & bit)) { entry = 0x80000000u | ((uint32_t)(127 + sh) << 23); done = true; break; }
        node = nd.x + (uint32_t)__popc(valid & (bit - 1u));
    }
    table[i] = done ? entry : node;
}

// Lowest three octree levels as occupancy blocks: for the level-3 node l3_first + j (an 8^3 cell) eight words, one
// per 4^3 octant in octant order, bit x&3 | (y&3)<<2 | (z&3)<<4 set iff that voxel is a leaf -- the dense bricks'
// layout.  A word is zero exactly when the octant's valid bit is clear, PROVIDED no stored node is empty; a valid
// child without children would end the oracle's walk one level further down (a smaller empty cell), so such a
// tree sets *bad and keeps the node-chain traversal.  word_base: per word the leaf index of its first voxel in tree
// order; bad[1] is set when the leaves under some word are NOT stored one after the other in that order (then a hit's
// leaf is found by walking down from the level-3 node, svo_leaf_by_walk).
__global__ void __launch_bounds__(256) build_svo_blocks(const uint2 *nodes, uint32_t l3_first, uint32_t l3_count, uint64_t *blocks,
                                                        uint32_t *word_base, uint32_t *bad)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= l3_count) return;
    const uint2 n3 = nodes[l3_first + j];
    const uint32_t valid3 = n3.y & 0xFFu;
    if ((n3.y >> 8) & 0xFFu) *bad = 1u;                                  // leaves above level 1
    for (uint32_t o4 = 0; o4 < 8; ++o4) {
        uint64_t word = 0;
        uint32_t first_leaf = 0, next_leaf = 0;
        bool any = false;
        if (valid3 & (1u << o4)) {
            const uint2 n2 = nodes[n3.x + (uint32_t)__popc(valid3 & ((1u << o4) - 1u))];
            const uint32_t valid2 = n2.y & 0xFFu;
            if (!valid2 || ((n2.y >> 8) & 0xFFu)) *bad = 1u;
            for (uint32_t o2 = 0; o2 < 8; ++o2) {
                if (!(valid2 & (1u << o2))) continue;
                const uint2 n1 = nodes[n2.x + (uint32_t)__popc(valid2 & ((1u << o2) - 1u))];
                const uint32_t leaf = (n1.y >> 8) & 0xFFu;
                if (!leaf || leaf != (n1.y & 0xFFu)) *bad = 1u;
                if (!any) { first_leaf = n1.x; any = true; }
                else if (n1.x != next_leaf) bad[1] = 1u;
                next_leaf = n1.x + (uint32_t)__popc(leaf);
                for (uint32_t o1 = 0; o1 < 8; ++o1) {
                    if (!(leaf & (1u << o1))) continue;
                    const uint32_t x = ((o2 & 1u) << 1) | (o1 & 1u), y = (((o2 >> 1) & 1u) << 1) | ((o1 >> 1) & 1u),
                                   z = (((o2 >> 2) & 1u) << 1) | ((o1 >> 2) & 1u);
                    word |= 1ull << (x | (y << 2) | (z << 4));
                }
            }
        }
        blocks[(size_t)j * 8 + o4] = word;
        word_base[(size_t)j * 8 + o4] = first_leaf;
    }
}

// synthetic noise voxels (volume_gpu_generate_noise)
__global__ void __launch_bounds__(256) generate_noise(uint2 *vox, int res, uint32_t seed, uint32_t threshold)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)res * res * res;
    if (i >= n) return;
    const uint32_t x = (uint32_t)(i % (size_t)res), y = (uint32_t)((i / (size_t)res) % (size_t)res), z = (uint32_t)(i / ((size_t)res * res));
    const uint32_t h = volly_noise_hash(x, y, z, seed);
    uint2 v = make_uint2(0u, 0u);
    if (h < threshold) { v.x = 0xFFFFFFFFu; v.y = (h >> 8) & 0x00FFFFFFu; }
    vox[i] = v;
}

// low grid of a synthetic scene: cell filled (white, zero normal) iff any of its 4^3 voxels is
// (lx, ly, lz: the low grid's sizes = the target's / 4, so none of its cells is a brick with rim cells; nbx, nby: the
// target's bricks per axis, padded)
__global__ void __launch_bounds__(256) reduce_low(const uint64_t *occ_hi, int lx, int ly, int lz, int nbx, int nby, uint2 *vox_lo)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nb = (size_t)lx * ly * lz;
    if (b >= nb) return;
    const size_t x = b % (size_t)lx, y = (b / (size_t)lx) % (size_t)ly, z = b / ((size_t)lx * ly);
    vox_lo[b] = occ_hi[x + (size_t)nbx * (y + (size_t)nby * z)] ? make_uint2(0xFFFFFFFFu, 0u) : make_uint2(0u, 0u);
}

#endif // VOLLY_TYPES_ONLY

} // namespace vk
#endif
