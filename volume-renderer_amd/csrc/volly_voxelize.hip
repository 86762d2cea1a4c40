// volly_voxelize.hip -- scene preparation on the device (volume_gpu_voxelize, include/volly_gpu.h).
//
// What it computes is the host pipeline's Vol_LOD_Set construction for one target resolution
// (volume-renderer_amd/host/voxelize.cpp, svo.cpp; reference: main.cpp:23-50,
// PLYfileReader.cpp:218-321, triangleCube.c:213-319, volume.h:114-123), byte for byte:
//
//   reference / host                                  here
//   per triangle: flood fill from the three vertex    ONE breadth-first search over all triangles at
//   cells through the 26-neighbourhood, a cell is     once: a work item is a (triangle, cell) pair, a
//   expanded iff t_c_intersection passes              level is one kernel launch, "already seen" is a
//   (std::set + std::stack)                           bit in the triangle's own box-shaped bitmap
//   std::map insert, later triangles overwrite        sort the (cell, triangle) words, keep the last
//   LoD chain: toPlace /= 2 while x > 1               map + sort + keep-last per level
//   volumeFromVoxelMap                                scatter kernels
//   octree: Morton sort, unique per level             radix sort by Morton key, head flags + scan
//
// The result of a flood fill does not depend on the order cells are visited in (it is the set
// of cells reachable from the seeds through passing cells), so the search order is free.
// Sorting and scanning use rocPRIM (plain library primitives of a preparation step); the
// triangle-cube test and everything that decides a voxel's bytes is written out below with the
// host code's operations in the host code's order, unfused (-ffp-contract=off).
#include <cstring>
#include <string.h>
#define VOLLY_TYPES_ONLY
#include "volly_internal.h"
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cstdlib>

using namespace vimpl;

namespace vk {

constexpr int kVoxCoordBits = 13;
constexpr int kVoxTriBits = 25;
constexpr uint64_t kVoxCoordMask = (1ull << kVoxCoordBits) - 1;
constexpr uint64_t kVoxTriMask = (1ull << kVoxTriBits) - 1;

struct TriRec {
    float a[3], b[3], c[3];         // vertices * (res - 1), PLYfileReader.cpp:229-231
    int ox, oy, oz;                 // first cell of the triangle's bitmap box
    int nx, ny, nz;                 // box size in cells; nx == 0: triangle skipped
    uint32_t norm;                  // Voxel::norm as the high dword of the 8-byte voxel
};

struct P3 { float x, y, z; };

// ---- t_c_intersection (triangleCube.c:213-319), as restated in host/tricube.cpp ----
__device__ __forceinline__ uint32_t face_code(P3 p)
{
    uint32_t c = 0;
    if (p.x > .5f) c |= 0x01;
    if (p.x < -.5f) c |= 0x02;
    if (p.y > .5f) c |= 0x04;
    if (p.y < -.5f) c |= 0x08;
    if (p.z > .5f) c |= 0x10;
    if (p.z < -.5f) c |= 0x20;
    return c;
}

__device__ __forceinline__ uint32_t edge_code(P3 p)
{
    uint32_t c = 0;
    if (p.x + p.y > 1.0f) c |= 0x001;
    if (p.x - p.y > 1.0f) c |= 0x002;
    if (-p.x + p.y > 1.0f) c |= 0x004;
    if (-p.x - p.y > 1.0f) c |= 0x008;
    if (p.x + p.z > 1.0f) c |= 0x010;
    if (p.x - p.z > 1.0f) c |= 0x020;
    if (-p.x + p.z > 1.0f) c |= 0x040;
    if (-p.x - p.z > 1.0f) c |= 0x080;
    if (p.y + p.z > 1.0f) c |= 0x100;
    if (p.y - p.z > 1.0f) c |= 0x200;
    if (-p.y + p.z > 1.0f) c |= 0x400;
    if (-p.y - p.z > 1.0f) c |= 0x800;
    return c;
}

__device__ __forceinline__ uint32_t corner_code(P3 p)
{
    uint32_t c = 0;
    if ((p.x + p.y + p.z) > 1.5f) c |= 0x01;
    if ((p.x + p.y - p.z) > 1.5f) c |= 0x02;
    if ((p.x - p.y + p.z) > 1.5f) c |= 0x04;
    if ((p.x - p.y - p.z) > 1.5f) c |= 0x08;
    if ((-p.x + p.y + p.z) > 1.5f) c |= 0x10;
    if ((-p.x + p.y - p.z) > 1.5f) c |= 0x20;
    if ((-p.x - p.y + p.z) > 1.5f) c |= 0x40;
    if ((-p.x - p.y - p.z) > 1.5f) c |= 0x80;
    return c;
}

__device__ __forceinline__ bool on_cube_at(P3 p1, P3 p2, float t, uint32_t mask)
{
    P3 q;
    q.x = p1.x + t * (p2.x - p1.x);
    q.y = p1.y + t * (p2.y - p1.y);
    q.z = p1.z + t * (p2.z - p1.z);
    return (face_code(q) & mask) == 0;
}

__device__ __forceinline__ bool segment_hits_cube(P3 p1, P3 p2, uint32_t diff)
{
    if ((diff & 0x01) && on_cube_at(p1, p2, (0.5f - p1.x) / (p2.x - p1.x), 0x3e)) return true;
    if ((diff & 0x02) && on_cube_at(p1, p2, (-0.5f - p1.x) / (p2.x - p1.x), 0x3d)) return true;
    if ((diff & 0x04) && on_cube_at(p1, p2, (0.5f - p1.y) / (p2.y - p1.y), 0x3b)) return true;
    if ((diff & 0x08) && on_cube_at(p1, p2, (-0.5f - p1.y) / (p2.y - p1.y), 0x37)) return true;
    if ((diff & 0x10) && on_cube_at(p1, p2, (0.5f - p1.z) / (p2.z - p1.z), 0x2f)) return true;
    if ((diff & 0x20) && on_cube_at(p1, p2, (-0.5f - p1.z) / (p2.z - p1.z), 0x1f)) return true;
    return false;
}

__device__ __forceinline__ float tc_max3(float a, float b, float c) { return ((a > b) && (a > c)) ? a : ((b > c) ? b : c); }
__device__ __forceinline__ float tc_min3(float a, float b, float c) { return ((a < b) && (a < c)) ? a : ((b < c) ? b : c); }
__device__ __forceinline__ P3 tc_sub(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ P3 tc_crs(P3 a, P3 b)
{
    P3 c;
    c.x = a.y * b.z - a.z * b.y;
    c.y = -a.x * b.z + a.z * b.x;
    c.z = a.x * b.y - a.y * b.x;
    return c;
}

// the reference's SIGN3 macro as it parses (a chain of conditionals); its tolerance is a
// double, so the floats are widened before they are compared (triangleCube.c:8-12)
__device__ __forceinline__ uint32_t sign_code(P3 a)
{
    const double tol = 10e-5;
    if ((double)a.x < tol) return 4;
    if ((double)a.x > -tol) return 32;
    if ((double)a.y < tol) return 2;
    if ((double)a.y > -tol) return 16;
    if ((double)a.z < tol) return 1;
    if ((double)a.z > -tol) return 8;
    return 0;
}

__device__ __forceinline__ bool point_in_triangle(P3 p, P3 v1, P3 v2, P3 v3)
{
    if (p.x > tc_max3(v1.x, v2.x, v3.x)) return false;
    if (p.y > tc_max3(v1.y, v2.y, v3.y)) return false;
    if (p.z > tc_max3(v1.z, v2.z, v3.z)) return false;
    if (p.x < tc_min3(v1.x, v2.x, v3.x)) return false;
    if (p.y < tc_min3(v1.y, v2.y, v3.y)) return false;
    if (p.z < tc_min3(v1.z, v2.z, v3.z)) return false;
    uint32_t s12 = sign_code(tc_crs(tc_sub(v1, v2), tc_sub(v1, p)));
    uint32_t s23 = sign_code(tc_crs(tc_sub(v2, v3), tc_sub(v2, p)));
    uint32_t s31 = sign_code(tc_crs(tc_sub(v3, v1), tc_sub(v3, p)));
    return (s12 & s23 & s31) != 0;
}

__device__ bool triangle_touches_unit_cube(P3 v1, P3 v2, P3 v3)
{
    uint32_t t1 = face_code(v1);
    if (t1 == 0) return true;
    uint32_t t2 = face_code(v2);
    if (t2 == 0) return true;
    uint32_t t3 = face_code(v3);
    if (t3 == 0) return true;
    if (t1 & t2 & t3) return false;

    t1 |= edge_code(v1) << 8;
    t2 |= edge_code(v2) << 8;
    t3 |= edge_code(v3) << 8;
    if (t1 & t2 & t3) return false;

    t1 |= corner_code(v1) << 24;
    t2 |= corner_code(v2) << 24;
    t3 |= corner_code(v3) << 24;
    if (t1 & t2 & t3) return false;

    if ((t1 & t2) == 0 && segment_hits_cube(v1, v2, t1 | t2)) return true;
    if ((t1 & t3) == 0 && segment_hits_cube(v1, v3, t1 | t3)) return true;
    if ((t2 & t3) == 0 && segment_hits_cube(v2, v3, t2 | t3)) return true;

    const double tol = 10e-5;
    P3 n = tc_crs(tc_sub(v1, v2), tc_sub(v1, v3));
    float d = n.x * v1.x + n.y * v1.y + n.z * v1.z;
    float den;
    den = n.x + n.y + n.z;
    if ((double)fabsf(den) > tol) {
        float h = d / den;
        if (fabsf(h) <= 0.5f && point_in_triangle({h, h, h}, v1, v2, v3)) return true;
    }
    den = n.x + n.y - n.z;
    if ((double)fabsf(den) > tol) {
        float h = d / den;
        if (fabsf(h) <= 0.5f && point_in_triangle({h, h, -h}, v1, v2, v3)) return true;
    }
    den = n.x - n.y + n.z;
    if ((double)fabsf(den) > tol) {
        float h = d / den;
        if (fabsf(h) <= 0.5f && point_in_triangle({h, -h, h}, v1, v2, v3)) return true;
    }
    den = n.x - n.y - n.z;
    if ((double)fabsf(den) > tol) {
        float h = d / den;
        if (fabsf(h) <= 0.5f && point_in_triangle({h, -h, -h}, v1, v2, v3)) return true;
    }
    return false;
}

// the cube test on its own: n triangles of 9 floats (already relative to the cube centre)
__global__ void __launch_bounds__(256) vox_tricube_batch(const float *tri9, uint64_t n, int32_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *t = tri9 + 9 * i;
    out[i] = triangle_touches_unit_cube({t[0], t[1], t[2]}, {t[3], t[4], t[5]}, {t[6], t[7], t[8]}) ? 0 : 1;
}

// ---- per-triangle set-up: PLYfileReader.cpp:225-248 ---------------------------------------
// The bitmap box of a triangle holds every cell the search can look at: a cell passes only if
// the triangle's bounding box reaches its cube (the first rejection of t_c_intersection), so
// passing cells lie in [floor(min) - 1, floor(max)] per axis; their neighbours one further;
// one more cell of slack for the rounding of (vertex - cell centre).  vox_level reports any
// cell outside the box it would have needed (none: tests assert the counter stays 0).
__global__ void __launch_bounds__(256)
vox_tri_setup(const float *verts, int64_t n_verts, const int32_t *tris, int n_tris, float resF, int res, TriRec *rec,
              uint64_t *bits)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    TriRec r;
    r.nx = r.ny = r.nz = 0; r.ox = r.oy = r.oz = 0; r.norm = 0;
    for (int k = 0; k < 3; ++k) r.a[k] = r.b[k] = r.c[k] = 0.f;
    const int64_t ia = tris[3 * i], ib = tris[3 * i + 1], ic = tris[3 * i + 2];
    bool ok = ia >= 0 && ib >= 0 && ic >= 0 && ia < n_verts && ib < n_verts && ic < n_verts;
    if (ok) {
        for (int k = 0; k < 3; ++k) {
            r.a[k] = verts[4 * ia + k] * resF;
            r.b[k] = verts[4 * ib + k] * resF;
            r.c[k] = verts[4 * ic + k] * resF;
            ok = ok && isfinite(r.a[k]) && isfinite(r.b[k]) && isfinite(r.c[k]);
        }
    }
    if (ok) {
        // normalize(cross(b - a, b - c)), clamp(.,0,1) * 255, truncated (PLYfileReader.cpp:233-237)
        const float ux = r.b[0] - r.a[0], uy = r.b[1] - r.a[1], uz = r.b[2] - r.a[2];
        const float wx = r.b[0] - r.c[0], wy = r.b[1] - r.c[1], wz = r.b[2] - r.c[2];
        const float cx = uy * wz - wy * uz, cy = uz * wx - wz * ux, cz = ux * wy - wx * uy;
        float d = cx * cx + cy * cy;
        d = d + cz * cz;
        const float inv = 1.0f / sqrtf(d);
        const float n[3] = {cx * inv, cy * inv, cz * inv};
        uint32_t packed = 0;
        for (int k = 0; k < 3; ++k) {
            float m = (n[k] < 0.f) ? 0.f : n[k];
            m = (1.f < m) ? 1.f : m;
            const float v = m * 255.f;
            const uint32_t byte = (v == v) ? (uint32_t)(uint8_t)v : 0u;     // NaN (degenerate face) -> 0
            packed |= byte << (8 * k);
        }
        r.norm = packed;
        int lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            const int ca = (int)r.a[k], cb = (int)r.b[k], cc = (int)r.c[k];
            lo[k] = max(min(min(ca, cb), cc) - 3, -2);
            hi[k] = min(max(max(ca, cb), cc) + 2, res + 1);
        }
        r.ox = lo[0]; r.oy = lo[1]; r.oz = lo[2];
        r.nx = hi[0] - lo[0] + 1; r.ny = hi[1] - lo[1] + 1; r.nz = hi[2] - lo[2] + 1;
        if (r.nx < 1 || r.ny < 1 || r.nz < 1) r.nx = r.ny = r.nz = 0;      // cannot happen for finite vertices in [0, res-1]
    }
    rec[i] = r;
    const uint64_t nb = (uint64_t)r.nx * (uint64_t)r.ny * (uint64_t)r.nz;
    bits[i] = (nb + 63) & ~63ull;
}

__device__ __forceinline__ uint64_t vox_item(uint32_t tri, int x, int y, int z)
{
    return ((uint64_t)tri << 39) | ((uint64_t)(z + 2) << 26) | ((uint64_t)(y + 2) << 13) | (uint64_t)(x + 2);
}

// one thread per triangle: the three vertex cells start the search (PLYfileReader.cpp:242-248)
__global__ void __launch_bounds__(256)
vox_seed(const TriRec *rec, const uint64_t *bitoff, int n_tris, uint32_t *seen, uint64_t *queue, uint32_t *counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    const TriRec r = rec[i];
    if (r.nx == 0) return;
    atomicAdd(&counters[3], 1u);
    const float *v[3] = {r.a, r.b, r.c};
    for (int k = 0; k < 3; ++k) {
        const int x = (int)v[k][0], y = (int)v[k][1], z = (int)v[k][2];
        const int lx = x - r.ox, ly = y - r.oy, lz = z - r.oz;
        if (lx < 0 || ly < 0 || lz < 0 || lx >= r.nx || ly >= r.ny || lz >= r.nz) { atomicAdd(&counters[2], 1u); continue; }
        const uint64_t bit = bitoff[i] + (uint64_t)lx + (uint64_t)r.nx * ((uint64_t)ly + (uint64_t)r.ny * (uint64_t)lz);
        const uint32_t m = 1u << (bit & 31);
        if (!(atomicOr(&seen[bit >> 5], m) & m)) queue[atomicAdd(&counters[0], 1u)] = vox_item((uint32_t)i, x, y, z);
    }
}

// exclusive prefix sum over the lanes of a wavefront; total = sum over all lanes
__device__ __forceinline__ uint32_t wave_exclusive_scan(uint32_t v, uint32_t &total)
{
    const int lane = (int)(threadIdx.x & 63);
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += t;
    }
    total = (uint32_t)__shfl((int)incl, 63);
    return incl - v;
}

// One search level: test every (triangle, cell) of the frontier; a passing cell is recorded and
// its 26 neighbours not yet seen by this triangle form the next frontier
// (PLYfileReader.cpp:250-275).  counters: [0] next frontier size, [1] queue overflow,
// [2] cells outside a triangle's box; tests: [0] cube tests run, [1] passing cells with x > 1.
__global__ void __launch_bounds__(256)
vox_level(const TriRec *rec, const uint64_t *bitoff, const uint64_t *in, uint32_t n_in, uint64_t *out, uint32_t cap, uint32_t *seen,
          uint32_t *pass, int res, uint32_t *counters, unsigned long long *tests)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = i < n_in;
    const int lane = (int)(threadIdx.x & 63);
    bool passed = false;
    uint32_t tri = 0;
    int x = 0, y = 0, z = 0, ox = 0, oy = 0, oz = 0, nx = 1, ny = 1, nz = 1;
    uint64_t base = 0;
    if (have) {
        const uint64_t it = in[i];
        tri = (uint32_t)(it >> 39);
        x = (int)(it & kVoxCoordMask) - 2;
        y = (int)((it >> 13) & kVoxCoordMask) - 2;
        z = (int)((it >> 26) & kVoxCoordMask) - 2;
        const TriRec r = rec[tri];
        ox = r.ox; oy = r.oy; oz = r.oz; nx = r.nx; ny = r.ny; nz = r.nz;
        base = bitoff[tri];
        const float cx = (float)x + 0.5f, cy = (float)y + 0.5f, cz = (float)z + 0.5f;       // PLYfileReader.cpp:257-262
        passed = triangle_touches_unit_cube({r.a[0] - cx, r.a[1] - cy, r.a[2] - cz}, {r.b[0] - cx, r.b[1] - cy, r.b[2] - cz},
                                            {r.c[0] - cx, r.c[1] - cy, r.c[2] - cz});
        if (passed) {
            const uint64_t bit = base + (uint64_t)(x - ox) + (uint64_t)nx * ((uint64_t)(y - oy) + (uint64_t)ny * (uint64_t)(z - oz));
            atomicOr(&pass[bit >> 5], 1u << (bit & 31));
        }
    }
    {
        const uint64_t m = __ballot(have);
        if (lane == 0 && m) atomicAdd(&tests[0], (unsigned long long)__popcll(m));
        const uint64_t k = __ballot(passed && x > 1);     // pairs vox_emit will write (PLYfileReader.cpp:278)
        if (lane == 0 && k) atomicAdd(&tests[1], (unsigned long long)__popcll(k));
    }
    // The 26 neighbours, one row of three (dx = -1, 0, 1) at a time: the three "seen" bits of a
    // row are adjacent, so a row costs one read and, only if a bit looks clear, one atomic OR
    // (two when the row straddles a word).  The read may be stale -- bits are only ever set, so
    // a stale word can only send a cell to the atomic that did not need it.  Device-scope
    // atomics are the expensive operation here (they complete beyond the per-XCD L2).
    uint32_t fresh = 0;                                   // bit 3 * row + (dx + 1): first seen now
    if (passed) {
        #pragma unroll
        for (int row = 0; row < 9; ++row) {
            const int dy = row % 3 - 1, dz = row / 3 - 1;
            const int qy = y + dy, qz = z + dz;
            if (qy < -2 || qz < -2 || qy > res + 1 || qz > res + 1) continue;
            const int ly = qy - oy, lz = qz - oz;
            uint32_t want = 0;                            // which of the three cells to look at
            #pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int qx = x + k - 1;
                if (row == 4 && k == 1) continue;         // the cell itself
                if (qx < -2 || qx > res + 1) continue;
                want |= 1u << k;
            }
            if (!want) continue;
            const int lx0 = x - 1 - ox;
            const bool row_in_box = ly >= 0 && lz >= 0 && ly < ny && lz < nz;
            uint32_t in_box = 0;
            #pragma unroll
            for (int k = 0; k < 3; ++k)
                if (row_in_box && lx0 + k >= 0 && lx0 + k < nx) in_box |= 1u << k;
            if (want & ~in_box) atomicAdd(&counters[2], 1u);
            want &= in_box;
            if (!want) continue;
            // bit of cell lx0 (may be "before" the box when lx0 == -1: then bit 0 of want is clear)
            const int64_t bit0 = (int64_t)base + (int64_t)lx0 + (int64_t)nx * ((int64_t)ly + (int64_t)ny * (int64_t)lz);
            const int64_t w0 = bit0 >> 5;                 // arithmetic shift: floor
            const uint32_t sh = (uint32_t)(bit0 & 31);
            const uint64_t m64 = (uint64_t)want << sh;
            const uint32_t m_lo = (uint32_t)m64, m_hi = (uint32_t)(m64 >> 32);
            uint32_t got = 0;                             // bits of want that were clear before our OR
            if (m_lo) {
                uint32_t cur = seen[w0];
                if (~cur & m_lo) cur = atomicOr(&seen[w0], m_lo);
                got |= (uint32_t)(((uint64_t)(~cur & m_lo)) >> sh);
            }
            if (m_hi) {
                uint32_t cur = seen[w0 + 1];
                if (~cur & m_hi) cur = atomicOr(&seen[w0 + 1], m_hi);
                got |= (uint32_t)((((uint64_t)(~cur & m_hi)) << 32) >> sh);
            }
            fresh |= (got & 7u) << (3 * row);
        }
    }
    // next frontier: one counter update per wavefront
    const uint32_t mine = (uint32_t)__popc(fresh);
    uint32_t total;
    const uint32_t before = wave_exclusive_scan(mine, total);
    if (total) {
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(&counters[0], total);
        first = (uint32_t)__shfl((int)first, 0);
        uint32_t pos = first + before;
        for (uint32_t f = fresh; f; f &= f - 1) {
            const int k = __ffs((int)f) - 1;
            const int row = k / 3, dx = k % 3 - 1, dy = row % 3 - 1, dz = row / 3 - 1;
            if (pos < cap) out[pos] = vox_item(tri, x + dx, y + dy, z + dz);
            else counters[1] = 1u;
            ++pos;
        }
    }
}

__device__ __forceinline__ uint64_t vox_word(int x, int y, int z, uint32_t tri)
{
    const uint64_t cell = ((uint64_t)(z + 1) << (2 * kVoxCoordBits)) | ((uint64_t)(y + 1) << kVoxCoordBits) | (uint64_t)(x + 1);
    return (cell << kVoxTriBits) | (uint64_t)tri;
}

__device__ __forceinline__ void vox_unword(uint64_t w, int &x, int &y, int &z, uint32_t &tri)
{
    tri = (uint32_t)(w & kVoxTriMask);
    const uint64_t cell = w >> kVoxTriBits;
    x = (int)(cell & kVoxCoordMask) - 1;
    y = (int)((cell >> kVoxCoordBits) & kVoxCoordMask) - 1;
    z = (int)((cell >> (2 * kVoxCoordBits)) & kVoxCoordMask) - 1;
}

// passing cells -> (cell, triangle) words, kept while x > 1 (PLYfileReader.cpp:278).  With
// out == nullptr only counts.  One bitmap word per thread; the triangle a word belongs to is
// found by bisection of the box offsets.
__global__ void __launch_bounds__(256)
vox_emit(const TriRec *rec, const uint64_t *bitoff, int n_tris, const uint32_t *pass, uint64_t n_words, uint64_t *out,
         uint64_t out_cap, unsigned long long *count)
{
    for (uint64_t first = (uint64_t)blockIdx.x * blockDim.x; first < n_words; first += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = first + threadIdx.x;
        uint32_t bits = w < n_words ? pass[w] : 0u;
        if (!__ballot(bits != 0u)) continue;              // most of a box is empty space
        uint32_t tri = 0;
        int ox = 0, oy = 0, oz = 0, nx = 1, ny = 1;
        uint64_t local0 = 0;
        uint32_t mine = 0;
        if (bits) {
            const uint64_t bit0 = w * 32;
            int lo = 0, hi = n_tris;                      // largest t with bitoff[t] <= bit0
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (bitoff[mid] <= bit0) lo = mid; else hi = mid;
            }
            tri = (uint32_t)lo;
            const TriRec &r = rec[tri];
            ox = r.ox; oy = r.oy; oz = r.oz; nx = r.nx; ny = r.ny;
            local0 = bit0 - bitoff[tri];
            uint32_t keep = 0;
            for (uint32_t b = bits; b; b &= b - 1) {
                const int k = __ffs((int)b) - 1;
                const uint64_t l = local0 + (uint64_t)k;
                const int x = ox + (int)(l % (uint64_t)nx);
                if (x > 1) keep |= 1u << k;
            }
            bits = keep;
            mine = (uint32_t)__popc(bits);
        }
        uint32_t total;
        const uint32_t before = wave_exclusive_scan(mine, total);
        if (!total) continue;
        unsigned long long start = 0;
        if ((threadIdx.x & 63) == 0) start = atomicAdd(count, (unsigned long long)total);
        start = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(start >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)start, 0);
        if (out && bits) {
            uint64_t o = start + before;
            for (uint32_t b = bits; b; b &= b - 1) {
                const int k = __ffs((int)b) - 1;
                const uint64_t l = local0 + (uint64_t)k;
                const int x = ox + (int)(l % (uint64_t)nx);
                const int y = oy + (int)((l / (uint64_t)nx) % (uint64_t)ny);
                const int z = oz + (int)(l / ((uint64_t)nx * (uint64_t)ny));
                if (o < out_cap) out[o] = vox_word(x, y, z, tri);
                ++o;
            }
        }
    }
}

// sorted words: flag the last word of every cell (= its highest triangle: "the later insert
// overwrites", PLYfileReader.cpp:283)
__global__ void __launch_bounds__(256) vox_flag_last(const uint64_t *w, uint64_t n, uint8_t *flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i + 1 == n) || ((w[i + 1] >> kVoxTriBits) != (w[i] >> kVoxTriBits));
}

// next LoD: toPlace /= 2 with C++ division (-1/2 == 0), kept while x > 1 (PLYfileReader.cpp:284-286)
__global__ void __launch_bounds__(256) vox_parent(const uint64_t *w, uint64_t n, uint64_t *out, uint8_t *flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x, y, z;
    uint32_t tri;
    vox_unword(w[i], x, y, z, tri);
    x /= 2; y /= 2; z /= 2;
    flag[i] = x > 1;
    out[i] = vox_word(x, y, z, tri);
}

// volumeFromVoxelMap (volume.h:114-123): cells of LoD level `r` into a store of `store`^3 voxels
__global__ void __launch_bounds__(256) vox_scatter(const uint64_t *w, uint64_t n, int r, int store, const TriRec *rec, uint2 *vox)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x, y, z;
    uint32_t tri;
    vox_unword(w[i], x, y, z, tri);
    if ((x < 0) | (y < 0) | (z < 0) | (x >= r) | (y >= r) | (z >= r)) return;          // PLYfileReader.cpp:279
    if (x >= store || y >= store || z >= store) return;
    vox[(size_t)x + (size_t)store * ((size_t)y + (size_t)store * (size_t)z)] = make_uint2(0xFFFFFFFFu, rec[tri].norm);
}

__device__ __forceinline__ uint64_t vox_spread3(uint64_t v)
{
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// LoD-0 cells inside the grid -> (Morton key, voxel high dword); others get a key that sorts last
__global__ void __launch_bounds__(256)
vox_morton(const uint64_t *w, uint64_t n, int r, const TriRec *rec, uint64_t *key, uint32_t *val, unsigned long long *n_inside)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool inside = false;
    if (i < n) {
        int x, y, z;
        uint32_t tri;
        vox_unword(w[i], x, y, z, tri);
        inside = !((x < 0) | (y < 0) | (z < 0) | (x >= r) | (y >= r) | (z >= r));
        key[i] = inside ? (vox_spread3((uint64_t)x) | (vox_spread3((uint64_t)y) << 1) | (vox_spread3((uint64_t)z) << 2)) : (1ull << 39);
        val[i] = rec[tri].norm;
    }
    const uint64_t m = __ballot(inside);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_inside, (unsigned long long)__popcll(m));
}

__global__ void __launch_bounds__(256) vox_leaves(const uint32_t *val, uint64_t n, uint2 *leaves)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) leaves[i] = make_uint2(0xFFFFFFFFu, val[i]);
}

// octree level: children sorted by key; a node starts where key >> 3 changes
__global__ void __launch_bounds__(256) svo_flag_heads(const uint64_t *ck, uint64_t n, uint32_t *flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0) || ((ck[i] >> 3) != (ck[i - 1] >> 3));
}

__global__ void __launch_bounds__(256)
svo_make_parents(const uint64_t *ck, const uint32_t *flag, const uint32_t *incl, uint64_t n, uint64_t *pk, uint32_t *first, uint32_t *mask)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = incl[i] - 1u;
    if (flag[i]) { pk[p] = ck[i] >> 3; first[p] = (uint32_t)i; }
    atomicOr(&mask[p], 1u << (uint32_t)(ck[i] & 7u));
}

// nodes of one level into the breadth-first array (host/svo.cpp build())
__global__ void __launch_bounds__(256)
svo_write_level(const uint32_t *first, const uint32_t *mask, uint32_t n, uint32_t base, uint32_t child_base, int leaf_level, uint2 *nodes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = mask[i] & 0xFFu;
    nodes[base + i] = make_uint2(child_base + first[i], m | (leaf_level ? (m << 8) : 0u));
}

} // namespace vk

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    void reset() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    hipError_t alloc(size_t n)
    {
        reset();
        hipError_t e = hipMalloc(&p, n ? n : 8);
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    void *release() { void *q = p; p = nullptr; bytes = 0; return q; }
    template <class T> T *as() const { return (T *)p; }
};

struct Scratch {
    size_t peak = 0, cur = 0;
    void add(size_t n) { cur += n; peak = std::max(peak, cur); }
    void sub(size_t n) { cur -= std::min(cur, n); }
};

inline unsigned blocks_for(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, 0x7FFFFFFFull); }

// ascending sort of 64-bit words on bits [0, end_bit); result in `io` (buffers may be swapped)
int sort_words(DevBuf &io, DevBuf &alt, uint64_t n, int end_bit, Scratch &sc)
{
    if (n < 2) return VOLLY_OK;
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, io.as<uint64_t>(), alt.as<uint64_t>(), n, 0, end_bit, 0));
    DevBuf tmp;
    HIP_TRY(tmp.alloc(tmp_bytes));
    sc.add(tmp_bytes);
    HIP_TRY(rocprim::radix_sort_keys(tmp.p, tmp_bytes, io.as<uint64_t>(), alt.as<uint64_t>(), n, 0, end_bit, 0));
    HIP_TRY(hipDeviceSynchronize());
    sc.sub(tmp_bytes);
    std::swap(io.p, alt.p);
    std::swap(io.bytes, alt.bytes);
    return VOLLY_OK;
}

// out = in[i] where flag[i]; returns the number kept
int select_words(const uint64_t *in, const uint8_t *flag, uint64_t n, uint64_t *out, uint64_t *kept, Scratch &sc)
{
    *kept = 0;
    if (!n) return VOLLY_OK;
    DevBuf cnt, tmp;
    HIP_TRY(cnt.alloc(8));
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::select(nullptr, tmp_bytes, in, flag, out, cnt.as<size_t>(), n, (hipStream_t)0));
    HIP_TRY(tmp.alloc(tmp_bytes));
    sc.add(tmp_bytes);
    HIP_TRY(rocprim::select(tmp.p, tmp_bytes, in, flag, out, cnt.as<size_t>(), n, (hipStream_t)0));
    size_t h = 0;
    HIP_TRY(hipMemcpy(&h, cnt.p, sizeof h, hipMemcpyDeviceToHost));
    sc.sub(tmp_bytes);
    *kept = h;
    return VOLLY_OK;
}

// sorted words -> one word per cell (the highest triangle); result in `io`, count in *n
int keep_last_per_cell(DevBuf &io, DevBuf &alt, uint64_t *n, Scratch &sc)
{
    if (!*n) return VOLLY_OK;
    DevBuf flag;
    HIP_TRY(flag.alloc(*n));
    sc.add(*n);
    hipLaunchKernelGGL(vk::vox_flag_last, dim3(blocks_for(*n)), dim3(256), 0, 0, io.as<uint64_t>(), *n, flag.as<uint8_t>());
    HIP_TRY(hipGetLastError());
    uint64_t kept = 0;
    int rc = select_words(io.as<uint64_t>(), flag.as<uint8_t>(), *n, alt.as<uint64_t>(), &kept, sc);
    sc.sub(*n);
    if (rc) return rc;
    std::swap(io.p, alt.p);
    std::swap(io.bytes, alt.bytes);
    *n = kept;
    return VOLLY_OK;
}

int ceil_log2(int v)
{
    int d = 0;
    while ((1 << d) < v) ++d;
    return d;
}

// breadth-first octree over LoD-0 cells inside the grid (host/svo.cpp), arrays handed to the scene
int build_octree(volly_gpu_scene *s, const uint64_t *words, uint64_t n_words, int res, const vk::TriRec *rec, Scratch &sc)
{
    const int depth = std::max(1, ceil_log2(res));
    DevBuf key, key2, val, val2, cnt;
    HIP_TRY(key.alloc(n_words * 8)); HIP_TRY(key2.alloc(n_words * 8));
    HIP_TRY(val.alloc(n_words * 4)); HIP_TRY(val2.alloc(n_words * 4));
    HIP_TRY(cnt.alloc(8));
    sc.add(n_words * 24);
    HIP_TRY(hipMemset(cnt.p, 0, 8));
    uint64_t n_leaves = 0;
    if (n_words) {
        hipLaunchKernelGGL(vk::vox_morton, dim3(blocks_for(n_words)), dim3(256), 0, 0, words, n_words, res, rec, key.as<uint64_t>(),
                           val.as<uint32_t>(), cnt.as<unsigned long long>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(&n_leaves, cnt.p, 8, hipMemcpyDeviceToHost));
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key.as<uint64_t>(), key2.as<uint64_t>(), val.as<uint32_t>(),
                                          val2.as<uint32_t>(), n_words, 0, 40, 0));
        DevBuf tmp;
        HIP_TRY(tmp.alloc(tmp_bytes));
        sc.add(tmp_bytes);
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, key.as<uint64_t>(), key2.as<uint64_t>(), val.as<uint32_t>(),
                                          val2.as<uint32_t>(), n_words, 0, 40, 0));
        HIP_TRY(hipDeviceSynchronize());
        sc.sub(tmp_bytes);
    }
    if (n_leaves >= 0x80000000ull) return fail(VOLLY_E_UNSUPPORTED, "volume_gpu_voxelize: more than 2^31 octree leaves");
    uint2 *leaves = nullptr;
    int rc = dev_alloc(s, (void **)&leaves, n_leaves * 8);
    if (rc) return rc;
    if (n_leaves) {
        hipLaunchKernelGGL(vk::vox_leaves, dim3(blocks_for(n_leaves)), dim3(256), 0, 0, val2.as<uint32_t>(), n_leaves, leaves);
        HIP_TRY(hipGetLastError());
    }
    val.reset(); val2.reset();

    // levels bottom-up: level L holds the nodes whose cell is 2^L voxels wide
    struct Level { DevBuf first, mask; uint32_t n = 0; };
    std::vector<Level> level((size_t)depth + 1);
    DevBuf *ck = &key2, *pk = &key;
    uint64_t n_child = n_leaves;
    for (int L = 1; L <= depth && n_child; ++L) {
        DevBuf flag, incl;
        HIP_TRY(flag.alloc(n_child * 4)); HIP_TRY(incl.alloc(n_child * 4));
        hipLaunchKernelGGL(vk::svo_flag_heads, dim3(blocks_for(n_child)), dim3(256), 0, 0, ck->as<uint64_t>(), n_child, flag.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::inclusive_scan(nullptr, tmp_bytes, flag.as<uint32_t>(), incl.as<uint32_t>(), n_child, rocprim::plus<uint32_t>(), 0));
        DevBuf tmp;
        HIP_TRY(tmp.alloc(tmp_bytes));
        HIP_TRY(rocprim::inclusive_scan(tmp.p, tmp_bytes, flag.as<uint32_t>(), incl.as<uint32_t>(), n_child, rocprim::plus<uint32_t>(), 0));
        uint32_t n_parent = 0;
        HIP_TRY(hipMemcpy(&n_parent, incl.as<uint32_t>() + (n_child - 1), 4, hipMemcpyDeviceToHost));
        Level &lv = level[(size_t)L];
        lv.n = n_parent;
        HIP_TRY(lv.first.alloc((size_t)n_parent * 4)); HIP_TRY(lv.mask.alloc((size_t)n_parent * 4));
        HIP_TRY(hipMemset(lv.mask.p, 0, (size_t)n_parent * 4));
        hipLaunchKernelGGL(vk::svo_make_parents, dim3(blocks_for(n_child)), dim3(256), 0, 0, ck->as<uint64_t>(), flag.as<uint32_t>(),
                           incl.as<uint32_t>(), n_child, pk->as<uint64_t>(), lv.first.as<uint32_t>(), lv.mask.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        std::swap(ck, pk);
        n_child = n_parent;
    }
    // breadth first: root level first
    std::vector<uint32_t> base((size_t)depth + 2, 0);
    uint64_t at = 0;
    for (int L = depth; L >= 1; --L) { base[(size_t)L] = (uint32_t)at; at += level[(size_t)L].n; }
    const bool empty = level[(size_t)depth].n == 0;
    if (empty) at = 1;                                   // empty scene: bare root
    if (at >= 0x80000000ull) return fail(VOLLY_E_UNSUPPORTED, "volume_gpu_voxelize: more than 2^31 octree nodes");
    uint2 *nodes = nullptr;
    if ((rc = dev_alloc(s, (void **)&nodes, at * 8))) return rc;
    if (empty) {                                          // host/svo.cpp: the bare root points just past itself
        const uint2 root = make_uint2(depth == 1 ? 0u : 1u, 0u);
        HIP_TRY(hipMemcpy(nodes, &root, 8, hipMemcpyHostToDevice));
    } else {
        for (int L = depth; L >= 1; --L) {
            const Level &lv = level[(size_t)L];
            hipLaunchKernelGGL(vk::svo_write_level, dim3(blocks_for(lv.n)), dim3(256), 0, 0, lv.first.as<uint32_t>(), lv.mask.as<uint32_t>(),
                               lv.n, base[(size_t)L], L == 1 ? 0u : base[(size_t)L - 1], L == 1 ? 1 : 0, nodes);
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    sc.sub(n_words * 24);
    // the levels were written one after the other: level 3 is the index range base[3] .. base[3] + level[3].n
    const bool has_l3 = depth >= 3 && !empty;
    return finish_svo(s, nodes, leaves, (uint32_t)at, (uint32_t)n_leaves, depth, res, has_l3 ? (int64_t)base[3] : -1, has_l3 ? (int64_t)level[3].n : 0);
}

// The search and its reduction to LoD 0: sorted (cell, last triangle) words
struct Raster {
    DevBuf rec;                     // TriRec per triangle (the scatter kernels read the normals)
    DevBuf wa, wb;                  // wa: the current level's words, sorted, one per cell; wb: spare of the same size
    uint64_t n = 0;                 // words in wa
    int res = 0, level = 0;         // wa holds LoD `level` of a rasterisation at `res`
    uint32_t tris_used = 0;
    unsigned long long tests = 0, hits = 0;
    int levels = 0;
    uint64_t cells0 = 0;
};

int rasterize(const volly_mesh_desc *mesh, int res, Raster &out, Scratch &sc)
{
    const int n_tris = (int)mesh->n_tris;
    const float resF = (float)(res - 1);                  // PLYfileReader.cpp:222
    out.res = res;
    out.level = 0;

    DevBuf d_verts, d_tris, d_bits, d_off;
    DevBuf &d_rec = out.rec;
    HIP_TRY(d_verts.alloc((size_t)mesh->n_verts * 16));
    HIP_TRY(d_tris.alloc((size_t)n_tris * 12));
    HIP_TRY(d_rec.alloc((size_t)n_tris * sizeof(vk::TriRec)));
    HIP_TRY(d_bits.alloc(((size_t)n_tris + 1) * 8));
    HIP_TRY(d_off.alloc(((size_t)n_tris + 1) * 8));
    sc.add(d_verts.bytes + d_tris.bytes + d_rec.bytes + d_bits.bytes + d_off.bytes);
    HIP_TRY(hipMemcpy(d_verts.p, mesh->verts, (size_t)mesh->n_verts * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tris.p, mesh->tris, (size_t)n_tris * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_bits.p, 0, ((size_t)n_tris + 1) * 8));
    hipLaunchKernelGGL(vk::vox_tri_setup, dim3(blocks_for((uint64_t)n_tris)), dim3(256), 0, 0, d_verts.as<float>(), mesh->n_verts,
                       d_tris.as<int32_t>(), n_tris, resF, res, d_rec.as<vk::TriRec>(), d_bits.as<uint64_t>());
    HIP_TRY(hipGetLastError());
    {   // box offsets: exclusive scan over n_tris + 1 entries (the last one is the total)
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, d_bits.as<uint64_t>(), d_off.as<uint64_t>(), (uint64_t)0, (size_t)n_tris + 1,
                                        rocprim::plus<uint64_t>(), 0));
        DevBuf tmp;
        HIP_TRY(tmp.alloc(tmp_bytes));
        HIP_TRY(rocprim::exclusive_scan(tmp.p, tmp_bytes, d_bits.as<uint64_t>(), d_off.as<uint64_t>(), (uint64_t)0, (size_t)n_tris + 1,
                                        rocprim::plus<uint64_t>(), 0));
        HIP_TRY(hipDeviceSynchronize());
    }
    uint64_t total_bits = 0;
    HIP_TRY(hipMemcpy(&total_bits, d_off.as<uint64_t>() + n_tris, 8, hipMemcpyDeviceToHost));
    if (total_bits > (1ull << 38))
        return fail(VOLLY_E_NOMEM, "volume_gpu_voxelize: the triangles' bounding boxes hold more than 2^38 cells (64 GiB of search bitmaps)");
    const uint64_t n_bitwords = total_bits / 32;

    DevBuf seen, pass;
    HIP_TRY(seen.alloc(n_bitwords * 4)); HIP_TRY(pass.alloc(n_bitwords * 4));
    sc.add(n_bitwords * 8);

    // Frontier queues.  A level's frontier is the set of cells first seen in the level before it,
    // so it can never hold more than total_bits entries; small searches get exactly that, large
    // ones start with an eighth (their frontiers are surfaces, not volumes) and the search is run
    // again with a larger queue should one ever fill up.
    const uint64_t safe = total_bits + 3ull * (uint64_t)n_tris + 1024;
    uint64_t cap64 = safe <= (1ull << 26) ? safe : std::min<uint64_t>(std::max<uint64_t>(total_bits / 8, 1ull << 26), 1ull << 28);
    if (const char *e = std::getenv("VOLLY_VOX_QUEUE")) cap64 = std::max<uint64_t>((uint64_t)std::atoll(e), 3ull * (uint64_t)n_tris + 1024);
    cap64 = std::max<uint64_t>(cap64, 3ull * (uint64_t)n_tris + 1024);       // vox_seed writes up to three entries per triangle
    DevBuf q0, q1, counters, tests;
    HIP_TRY(counters.alloc(16)); HIP_TRY(tests.alloc(16));
    uint32_t hc[4] = {0, 0, 0, 0};
    for (;;) {
        const uint32_t cap = (uint32_t)std::min<uint64_t>(cap64, 0x7FFFFFFFull);
        HIP_TRY(q0.alloc((size_t)cap * 8)); HIP_TRY(q1.alloc((size_t)cap * 8));
        sc.add((size_t)cap * 16);
        HIP_TRY(hipMemset(seen.p, 0, n_bitwords * 4));
        HIP_TRY(hipMemset(pass.p, 0, n_bitwords * 4));
        HIP_TRY(hipMemset(counters.p, 0, 16));
        HIP_TRY(hipMemset(tests.p, 0, 16));
        hipLaunchKernelGGL(vk::vox_seed, dim3(blocks_for((uint64_t)n_tris)), dim3(256), 0, 0, d_rec.as<vk::TriRec>(), d_off.as<uint64_t>(), n_tris,
                           seen.as<uint32_t>(), q0.as<uint64_t>(), counters.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(hc, counters.p, 16, hipMemcpyDeviceToHost));
        out.tris_used = hc[3];
        out.levels = 0;
        uint32_t count = hc[0];
        bool overflow = false;
        DevBuf *qin = &q0, *qout = &q1;
        while (count && !overflow) {
            HIP_TRY(hipMemsetAsync(counters.p, 0, 4, 0));
            hipLaunchKernelGGL(vk::vox_level, dim3(blocks_for(count)), dim3(256), 0, 0, d_rec.as<vk::TriRec>(), d_off.as<uint64_t>(),
                               qin->as<uint64_t>(), count, qout->as<uint64_t>(), cap, seen.as<uint32_t>(), pass.as<uint32_t>(), res,
                               counters.as<uint32_t>(), tests.as<unsigned long long>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(hc, counters.p, 16, hipMemcpyDeviceToHost));
            overflow = hc[1] || hc[0] > cap;
            std::swap(qin, qout);
            count = hc[0];
            ++out.levels;
        }
        sc.sub((size_t)cap * 16);
        if (!overflow) break;
        if (cap64 >= safe || cap64 >= 0x7FFFFFFFull)
            return fail(VOLLY_E_NOMEM, "volume_gpu_voxelize: search frontier larger than 2^31 entries");
        cap64 = std::min<uint64_t>(cap64 * 4, safe);
    }
    if (hc[2]) return fail(VOLLY_E_HIP, "volume_gpu_voxelize: internal error, a search left its triangle's box");
    unsigned long long h_tests[2] = {0, 0};
    HIP_TRY(hipMemcpy(h_tests, tests.p, 16, hipMemcpyDeviceToHost));
    out.tests = h_tests[0];
    q0.reset(); q1.reset(); seen.reset();
    sc.sub(n_bitwords * 4);

    // passing (cell, triangle) pairs with x > 1 -> words
    DevBuf n_hits;
    HIP_TRY(n_hits.alloc(8));
    HIP_TRY(hipMemset(n_hits.p, 0, 8));
    const unsigned emit_blocks = (unsigned)std::min<uint64_t>((n_bitwords + 255) / 256, 1u << 16);
    const unsigned long long hits = h_tests[1];          // counted by the search itself
    out.hits = hits;
    DevBuf &wa = out.wa, &wb = out.wb;
    HIP_TRY(wa.alloc((size_t)hits * 8)); HIP_TRY(wb.alloc((size_t)hits * 8));
    sc.add((size_t)hits * 16);
    if (hits) {
        HIP_TRY(hipMemset(n_hits.p, 0, 8));
        hipLaunchKernelGGL(vk::vox_emit, dim3(emit_blocks), dim3(256), 0, 0, d_rec.as<vk::TriRec>(), d_off.as<uint64_t>(), n_tris,
                           pass.as<uint32_t>(), n_bitwords, wa.as<uint64_t>(), (uint64_t)hits, n_hits.as<unsigned long long>());
        HIP_TRY(hipGetLastError());
        unsigned long long written = 0;
        HIP_TRY(hipMemcpy(&written, n_hits.p, 8, hipMemcpyDeviceToHost));
        if (written != hits) return fail(VOLLY_E_HIP, "volume_gpu_voxelize: internal error, pair count mismatch");
    }
    pass.reset();
    sc.sub(n_bitwords * 4);

    // LoD 0: the last triangle of every cell
    out.n = hits;
    int rc = sort_words(wa, wb, out.n, 64, sc);
    if (!rc) rc = keep_last_per_cell(wa, wb, &out.n, sc);
    if (rc) return rc;
    out.cells0 = out.n;
    return VOLLY_OK;
}

// LoD k -> LoD k + 1 (PLYfileReader.cpp:284-286)
int next_lod(Raster &r, Scratch &sc)
{
    int rc;
    if (r.n) {
        DevBuf flag;
        HIP_TRY(flag.alloc(r.n));
        hipLaunchKernelGGL(vk::vox_parent, dim3(blocks_for(r.n)), dim3(256), 0, 0, r.wa.as<uint64_t>(), r.n, r.wb.as<uint64_t>(), flag.as<uint8_t>());
        HIP_TRY(hipGetLastError());
        uint64_t kept = 0;
        if ((rc = select_words(r.wb.as<uint64_t>(), flag.as<uint8_t>(), r.n, r.wa.as<uint64_t>(), &kept, sc))) return rc;
        r.n = kept;
    }
    if ((rc = sort_words(r.wa, r.wb, r.n, 64, sc))) return rc;
    if ((rc = keep_last_per_cell(r.wa, r.wb, &r.n, sc))) return rc;
    ++r.level;
    return VOLLY_OK;
}

// volumeFromVoxelMap(map, store, level) (volume.h:114-123) into a grid owned by scene s
int dense_from_level(volly_gpu_scene *s, const Raster &r, int store, vk::DenseDev *out)
{
    if (store < 1 || (uint64_t)store * store * store > (1ull << 31))
        return fail(VOLLY_E_ARG, "volume_gpu_voxelize: a dense grid of more than 2^31 voxels (ask for VOLLY_VOX_SVO only)");
    const int level_res = r.res / (1 << r.level);
    const size_t n = (size_t)store * store * store;
    uint2 *vox = nullptr;
    int rc = dev_alloc(s, (void **)&vox, n * 8);
    if (rc) return rc;
    HIP_TRY(hipMemset(vox, 0, n * 8));
    if (r.n) {
        hipLaunchKernelGGL(vk::vox_scatter, dim3(blocks_for(r.n)), dim3(256), 0, 0, r.wa.as<uint64_t>(), r.n, level_res, store,
                           r.rec.as<vk::TriRec>(), vox);
        HIP_TRY(hipGetLastError());
    }
    return make_dense(s, store, store, store, nullptr, vox, out);
}

int voxelize(volly_gpu_scene *s, const volly_mesh_desc *mesh, int res, int flags, int64_t *stats)
{
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    Scratch sc;
    Raster r;
    int rc = rasterize(mesh, res, r, sc);
    if (rc) return rc;
    if (flags & VOLLY_VOX_DENSE)
        if ((rc = dense_from_level(s, r, res, &s->target))) return rc;
    if (flags & VOLLY_VOX_SVO)
        if ((rc = build_octree(s, r.wa.as<uint64_t>(), r.n, res, r.rec.as<vk::TriRec>(), sc))) return rc;
    // the low beam grid is LoD 2 at res / 4 (main.cpp:33)
    if ((rc = next_lod(r, sc)) || (rc = next_lod(r, sc))) return rc;
    if ((rc = dense_from_level(s, r, res / 4, &s->low))) return rc;
    s->has_low = true;
    HIP_TRY(hipDeviceSynchronize());
    if (stats) {
        stats[0] = r.tris_used;
        stats[1] = (int64_t)r.tests;
        stats[2] = (int64_t)r.hits;
        stats[3] = r.levels;
        stats[4] = (int64_t)r.cells0;
        stats[5] = (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t0).count();
        stats[6] = (int64_t)sc.peak;
        stats[7] = 0;
    }
    return VOLLY_OK;
}

// Vol_LOD_Set's constructor (main.cpp:23-50): arr[2] = LoD 0 of the medR rasterisation, arr[1] =
// its LoD 2 stored at lowR, arr[0] = its LoD 3 stored at sLowR, arr[3] = LoD 0 of a second
// rasterisation at highR.  Scene i renders arr[i] with arr[1] as its beam grid, the pair
// State::raycastOntoScreen uses for curLoD = i (main.h:118-119).
int voxelize_lod_set(volly_gpu_scene *s[4], const volly_mesh_desc *mesh_med, const volly_mesh_desc *mesh_high, int sLowR, int lowR,
                     int medR, int highR)
{
    Scratch sc;
    int rc;
    {
        Raster r;
        if ((rc = rasterize(mesh_med, medR, r, sc))) return rc;
        if ((rc = dense_from_level(s[2], r, medR, &s[2]->target))) return rc;
        if ((rc = next_lod(r, sc)) || (rc = next_lod(r, sc))) return rc;
        if ((rc = dense_from_level(s[1], r, lowR, &s[1]->target))) return rc;
        for (int i = 0; i < 4; ++i) {
            if (!s[i]) continue;
            if ((rc = dense_from_level(s[i], r, lowR, &s[i]->low))) return rc;
            s[i]->has_low = true;
        }
        if ((rc = next_lod(r, sc))) return rc;
        if ((rc = dense_from_level(s[0], r, sLowR, &s[0]->target))) return rc;
    }
    if (s[3]) {
        Raster r;
        if ((rc = rasterize(mesh_high, highR, r, sc))) return rc;
        if ((rc = dense_from_level(s[3], r, highR, &s[3]->target))) return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    return VOLLY_OK;
}

} // namespace

extern "C" int volume_gpu_voxelize(int device, const volly_mesh_desc *mesh, int32_t res, int32_t flags, volly_gpu_scene **out,
                                   int64_t *stats)
{
    if (!out) return fail(VOLLY_E_ARG, "volume_gpu_voxelize: out is null");
    *out = nullptr;
    if (!mesh || !mesh->verts || !mesh->tris || mesh->n_verts < 1 || mesh->n_tris < 1)
        return fail(VOLLY_E_ARG, "volume_gpu_voxelize: empty mesh");
    if (mesh->n_tris >= (1ll << vk::kVoxTriBits) || mesh->n_verts >= (1ll << 31))
        return fail(VOLLY_E_ARG, "volume_gpu_voxelize: at most 2^25 - 1 triangles");
    if (res < 8 || res > 4096 || (res % 4) != 0) return fail(VOLLY_E_ARG, "volume_gpu_voxelize: res must be a multiple of 4 in [8, 4096]");
    if (!(flags & (VOLLY_VOX_DENSE | VOLLY_VOX_SVO)) || (flags & ~(VOLLY_VOX_DENSE | VOLLY_VOX_SVO)))
        return fail(VOLLY_E_ARG, "volume_gpu_voxelize: flags must be VOLLY_VOX_DENSE and/or VOLLY_VOX_SVO");
    volly_gpu_scene *s = nullptr;
    int rc = new_scene(device, &s);
    if (rc) return rc;
    DeviceGuard g(device);
    if (!g.ok) { destroy(s); return fail(VOLLY_E_HIP, "hipSetDevice failed"); }
    rc = voxelize(s, mesh, res, flags, stats);
    if (rc) { std::string keep = g_err; destroy(s); g_err = keep; return rc; }
    choose_frame_kernel(s);
    *out = s;
    return VOLLY_OK;
}

extern "C" int volume_gpu_voxelize_lod_set(int device, const volly_mesh_desc *mesh_med, const volly_mesh_desc *mesh_high, int32_t sLowR,
                                           int32_t lowR, int32_t medR, int32_t highR, volly_gpu_scene *out[4])
{
    if (!out) return fail(VOLLY_E_ARG, "volume_gpu_voxelize_lod_set: out is null");
    for (int i = 0; i < 4; ++i) out[i] = nullptr;
    auto bad_mesh = [](const volly_mesh_desc *m) {
        return !m || !m->verts || !m->tris || m->n_verts < 1 || m->n_tris < 1 || m->n_tris >= (1ll << vk::kVoxTriBits) || m->n_verts >= (1ll << 31);
    };
    if (bad_mesh(mesh_med) || (highR > 0 && bad_mesh(mesh_high))) return fail(VOLLY_E_ARG, "volume_gpu_voxelize_lod_set: empty or oversized mesh");
    if (sLowR < 1 || lowR < 1 || medR < 8 || medR > 1024 || highR > 1024 || sLowR > 1024 || lowR > 1024)
        return fail(VOLLY_E_ARG, "volume_gpu_voxelize_lod_set: resolutions must lie in [1, 1024], medR >= 8 (dense grids)");
    volly_gpu_scene *s[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = VOLLY_OK;
    for (int i = 0; i < 4 && !rc; ++i)
        if (i < 3 || highR > 0) rc = new_scene(device, &s[i]);
    if (!rc) {
        DeviceGuard g(device);
        if (!g.ok) rc = fail(VOLLY_E_HIP, "hipSetDevice failed");
        else rc = voxelize_lod_set(s, mesh_med, mesh_high, sLowR, lowR, medR, highR);
    }
    if (rc) {
        std::string keep = g_err;
        for (auto *p : s) if (p) destroy(p);
        g_err = keep;
        return rc;
    }
    for (int i = 0; i < 4; ++i)
        if (s[i]) { choose_frame_kernel(s[i]); out[i] = s[i]; }
    return VOLLY_OK;
}

extern "C" int volly_gpu_tricube(int device, const float *tri9, int64_t n, int32_t *out)
{
    if (!tri9 || !out || n < 0) return fail(VOLLY_E_ARG, "volly_gpu_tricube: null argument");
    if (n == 0) return VOLLY_OK;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(VOLLY_E_ARG, "no such device");
    DeviceGuard g(device);
    if (!g.ok) return fail(VOLLY_E_HIP, "hipSetDevice failed");
    DevBuf d_in, d_out;
    HIP_TRY(d_in.alloc((size_t)n * 36));
    HIP_TRY(d_out.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(d_in.p, tri9, (size_t)n * 36, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(vk::vox_tricube_batch, dim3(blocks_for((uint64_t)n)), dim3(256), 0, 0, d_in.as<float>(), (uint64_t)n, d_out.as<int32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return VOLLY_OK;
}
