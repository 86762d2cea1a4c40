// volly_refill.h -- frame kernel for INCOHERENT dense scenes (ray refill).
//
// render_frames gives every wavefront one 8x8 pixel tile and runs until the tile's
// longest ray ends.  On the bundled meshes the rays of a tile end within 6 % of each
// other, so that is the right shape.  In a noise volume (BASELINE config 3) ray lengths
// are geometric: the longest of 64 rays is ~4.7x the mean and the wavefront runs at
// ~22 % lane utilisation (measured: 14 250 VALU instructions per wave for a mean of 69
// steps per ray).  Here a wavefront owns a strip of 512 rays instead: whenever at least
// kRefillAt lanes are idle they take the next rays of the strip, so lanes stay busy until
// the strip is drained.  Each lane is a small state machine (idle / low-grid pass /
// target-grid pass); both passes run the same DDA step with the grid picked per lane.
//
// Same arithmetic, same order, same results as render_frames (tests compare the two
// and the oracle); only the assignment of rays to lanes and the store pattern differ.
// Used for dense targets whose grids are "regular" (see grid_is_regular); rays that are
// not regular, or whose entry point is not finite, are finished on the spot with the fully
// checked loops.
#ifndef VOLLY_REFILL_H
#define VOLLY_REFILL_H

#include "volly_kernels.h"

namespace vk {

constexpr int kRaysPerWave = 512;   // 8 tiles of 8x8 pixels
constexpr int kRefillAt = 16;       // refill when this many lanes are idle (or all are)

struct LaneOut {
    uint32_t rgba, hit, slo, shi;
    float depth;
    float ex, ey, ez;
};

__device__ __forceinline__ void store_pixel(const OutDev &out, size_t p, const LaneOut &o)
{
    if (out.rgba8) out.rgba8[p] = o.rgba;
    if (out.hit_index) out.hit_index[p] = o.hit;
    if (out.steps_lo) out.steps_lo[p] = o.slo;
    if (out.steps_hi) out.steps_hi[p] = o.shi;
    if (out.depth) out.depth[p] = o.depth;
    if (out.pt_end) { out.pt_end[3 * p] = o.ex; out.pt_end[3 * p + 1] = o.ey; out.pt_end[3 * p + 2] = o.ez; }
}

// main.cpp:262-272 for one finished ray
__device__ __forceinline__ void finish_pixel(const DenseDev &target, const volly_frame_desc &fd, const OutDev &out, size_t p,
                                             uint32_t slo, uint32_t shi, uint32_t hit_index, float cx, float cy, float cz,
                                             float dx, float dy, float dz)
{
    LaneOut o;
    o.rgba = 255u; o.hit = hit_index; o.slo = slo; o.shi = shi;
    o.depth = __uint_as_float(0x7F800000u);
    o.ex = cx; o.ey = cy; o.ez = cz;
    if (hit_index != VOLLY_MISS_INDEX) {
        o.rgba = shade(target.vox[hit_index]);
        float ddx = cx - fd.origin_hi[0], ddy = cy - fd.origin_hi[1], ddz = cz - fd.origin_hi[2];
        o.depth = (ddx * dx + ddy * dy) + ddz * dz;
    }
    store_pixel(out, p, o);
}

__global__ void __launch_bounds__(kThreads)
render_frames_refill(DenseDev target, DenseDev low, const volly_frame_desc *__restrict__ frames,
                     const float *__restrict__ ckpt, OutDev out)
{
    const int f = blockIdx.y;
    const volly_frame_desc fd = frames[f];
    const int tiles_x = (fd.tile_w + kTile - 1) / kTile, tiles_y = (fd.tile_h + kTile - 1) / kTile;
    const uint32_t n_tiles = (uint32_t)tiles_x * (uint32_t)tiles_y;
    const uint32_t wave = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const uint32_t first_tile = wave * (kRaysPerWave / 64);
    if (first_tile >= n_tiles) return;
    const uint32_t my_tiles = min((uint32_t)(kRaysPerWave / 64), n_tiles - first_tile);
    const uint32_t total = my_tiles * 64u;
    const size_t frame_px = (size_t)f * (size_t)fd.tile_w * (size_t)fd.tile_h;
    const int groups = (fd.tile_w + kCkpt - 1) / kCkpt;         // checkpoint groups per row
    const bool naive = (fd.flags & VOLLY_FRAME_NAIVE) != 0;

    // lane state
    uint32_t phase = 0;                                  // 0 idle, 1 low-grid pass, 2 target-grid pass,
                                                         // 3 low pass over (waits for service), 4 target pass over
    uint32_t pix = 0, start_iter = 0, slo = 0, shi = 0, hit_index = VOLLY_MISS_INDEX;
    float cx = 0, cy = 0, cz = 0;
    Ray r = make_ray(1.0f, 1.0f, 1.0f);
    float hdx = 0, hdy = 0, hdz = 0;                     // target-grid direction (kept through the low pass)
    uint32_t last_key = 0xFFFFFFFFu, w_lo = 0, w_hi = 0;

    uint32_t next = 0;                                   // wave uniform: next unassigned ray of the strip
    for (uint32_t iter = 0;; ++iter) {
        // ---- service: phase changes, pixel stores and refills are expensive and touch few lanes,
        // so lanes wait for them until kRefillAt lanes have something to do (or nothing else runs)
        const uint64_t stepping = __ballot((phase == 1) | (phase == 2));
        const uint32_t n_pending = (uint32_t)__popcll(__ballot(phase >= 3));
        const uint32_t n_idle = (uint32_t)__popcll(__ballot(phase == 0));
        const uint32_t serviceable = n_pending + (next < total ? n_idle : 0u);
        if (serviceable >= (uint32_t)kRefillAt || (stepping == 0 && serviceable > 0)) {
            if (phase == 3) {                             // raycast_beamOpt: restart at the low-grid end point
                const float sx = fd.scale_lo_to_hi[0] * cx, sy = fd.scale_lo_to_hi[1] * cy, sz = fd.scale_lo_to_hi[2] * cz;
                r = make_ray(hdx, hdy, hdz);
                enter_volume(r, sx, sy, sz, target.fsx, target.fsy, target.fsz, cx, cy, cz);
                if ((fabsf(cx) < 3.0e38f) & (fabsf(cy) < 3.0e38f) & (fabsf(cz) < 3.0e38f)) {
                    phase = 2;
                    start_iter = iter;
                    last_key = 0xFFFFFFFFu;
                } else {                                  // first bounds test fails: a miss with no step
                    shi = 0;
                    hit_index = VOLLY_MISS_INDEX;
                    phase = 4;
                }
            }
            if (phase == 4) {
                finish_pixel(target, fd, out, frame_px + pix, slo, shi, hit_index, cx, cy, cz, hdx, hdy, hdz);
                phase = 0;
            }
            const uint64_t idle_mask = __ballot(phase == 0);
            if (next < total && idle_mask) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const uint32_t mine = next + rank;
                const bool take = (phase == 0) & (mine < total);
                next = min(total, next + (uint32_t)__popcll(idle_mask));
                if (take) {
                    const uint32_t tile = first_tile + (mine >> 6), in_tile = mine & 63u;
                    const int xrel = (int)(tile % (uint32_t)tiles_x) * kTile + (int)(in_tile & 7u);
                    const int yrel = (int)(tile / (uint32_t)tiles_x) * kTile + (int)(in_tile >> 3);
                    if (xrel < fd.tile_w && yrel < fd.tile_h) {
                        pix = (uint32_t)yrel * (uint32_t)fd.tile_w + (uint32_t)xrel;
                        const size_t cs = (size_t)fd.tile_h;
                        const float *c = ckpt + ((size_t)f * (size_t)groups + (size_t)(xrel / kCkpt)) * 6 * cs + (size_t)yrel;
                        float hx = c[0], hy = c[cs], hz = c[2 * cs], lx = c[3 * cs], ly = c[4 * cs], lz = c[5 * cs];
                        const int extra = xrel % kCkpt;
                        for (int i = 0; i < extra; ++i) { // main.cpp:282-283
                            hx = hx + fd.xdiff_hi[0]; hy = hy + fd.xdiff_hi[1]; hz = hz + fd.xdiff_hi[2];
                            lx = lx + fd.xdiff_lo[0]; ly = ly + fd.xdiff_lo[1]; lz = lz + fd.xdiff_lo[2];
                        }
                        normalize_w0(hx, hy, hz);         // main.cpp:260
                        hdx = hx; hdy = hy; hdz = hz;
                        const Ray ray_hi = make_ray(hx, hy, hz);
                        bool regular = ray_is_regular(ray_hi);
                        Ray ray_first = ray_hi;
                        float ox = fd.origin_hi[0], oy = fd.origin_hi[1], oz = fd.origin_hi[2];
                        if (!naive) {
                            normalize_w0(lx, ly, lz);
                            ray_first = make_ray(lx, ly, lz);
                            regular = regular & ray_is_regular(ray_first);
                            ox = fd.origin_lo[0]; oy = fd.origin_lo[1]; oz = fd.origin_lo[2];
                        }
                        const DenseDev &g0 = naive ? target : low;
                        float ex, ey, ez;
                        enter_volume(ray_first, ox, oy, oz, g0.fsx, g0.fsy, g0.fsz, ex, ey, ez);
                        regular = regular & (fabsf(ex) < 3.0e38f) & (fabsf(ey) < 3.0e38f) & (fabsf(ez) < 3.0e38f);
                        if (regular) {
                            r = ray_first;
                            cx = ex; cy = ey; cz = ez;
                            phase = naive ? 2u : 1u;
                            slo = 0;
                            start_iter = iter;
                            last_key = 0xFFFFFFFFu;
                        } else {
                            // not a regular ray: finish it here with the fully checked loops
                            float sx = ox, sy = oy, sz = oz;
                            uint32_t steps_lo = 0;
                            if (!naive) {
                                Hit lo = trace_dense_checked(low, ray_first, ox, oy, oz);
                                steps_lo = lo.steps;
                                sx = fd.scale_lo_to_hi[0] * lo.cx; sy = fd.scale_lo_to_hi[1] * lo.cy; sz = fd.scale_lo_to_hi[2] * lo.cz;
                            }
                            Hit hi = trace_dense_checked(target, ray_hi, sx, sy, sz);
                            finish_pixel(target, fd, out, frame_px + pix, steps_lo, hi.steps, hi.index, hi.cx, hi.cy, hi.cz, hdx, hdy, hdz);
                        }
                    }
                }
            }
            if (__ballot(phase != 0) == 0 && next >= total) break;
        }

        // ---- one DDA step of every stepping lane (trace_dense_regular's step, grid picked per lane)
        if ((phase == 1) | (phase == 2)) {
            const bool in_low = phase == 1;
            float px = cx + r.ex, py = cy + r.ey, pz = cz + r.ez;
            int x = (int)px, y = (int)py, z = (int)pz;
            uint32_t c = pack_cell(x, y, z);
            const uint32_t steps = iter - start_iter;
            bool done = steps >= (in_low ? low.cap : target.cap);
            bool hit = false;
            if (!done) {
                uint32_t key = c & 0xFFCFF3FCu;
                if (key != last_key) {
                    const uint32_t gx = in_low ? (uint32_t)low.sx : (uint32_t)target.sx;
                    const uint32_t gy = in_low ? (uint32_t)low.sy : (uint32_t)target.sy;
                    const uint32_t gz = in_low ? (uint32_t)low.sz : (uint32_t)target.sz;
                    const bool inside = ((uint32_t)x < gx) & ((uint32_t)y < gy) & ((uint32_t)z < gz);
                    if (inside) {
                        const uint64_t *occ = in_low ? low.occ : target.occ;
                        const uint32_t nbx = in_low ? (uint32_t)low.nbx : (uint32_t)target.nbx;
                        const uint32_t nby = in_low ? (uint32_t)low.nby : (uint32_t)target.nby;
                        uint64_t bits = occ[(uint32_t)(x >> 2) + nbx * ((uint32_t)(y >> 2) + nby * (uint32_t)(z >> 2))];
                        w_lo = (uint32_t)bits;
                        w_hi = (uint32_t)(bits >> 32);
                        last_key = key;
                    } else {
                        done = true;                      // left the grid, main.cpp:126-135
                    }
                }
                if (!done) {
                    hit = test_word(w_lo, w_hi, gather_bits(c & 0x00300C03u));
                    done = hit;
                }
            }
            if (!done) {
                advance_unit(r, px, py, pz, cx, cy, cz);
            } else if (in_low) {
                slo = steps + (hit ? 1u : 0u);
                phase = 3;
            } else {
                shi = steps + (hit ? 1u : 0u);
                hit_index = hit ? (uint32_t)x + (uint32_t)target.sx * ((uint32_t)y + (uint32_t)target.sy * (uint32_t)z) : VOLLY_MISS_INDEX;
                phase = 4;
            }
        }
    }
}

} // namespace vk
#endif
