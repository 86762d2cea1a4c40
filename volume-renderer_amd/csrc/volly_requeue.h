// volly_requeue.h -- frame kernels for scenes whose rays have very uneven lengths.
//
// In the tile kernel (render_frames) a wavefront runs until the longest of its 64 rays is done.
// For meshes neighbouring rays are alike and that costs 6 %; in a noise volume ray lengths are
// geometric and lanes idle about half of the time.  Here a
// wavefront traces its rays only up to an iteration budget; rays still under way are parked in
// a queue (position, direction, pixel: 32 bytes) and a follow-up launch continues them, 64
// parked rays to a wavefront, again up to a budget that doubles from round to round.  The loop
// iteration number is wave-uniform in every round (all rays of a round were parked at the same
// iteration), so the step count stays the iteration number and nothing is counted per lane.
//
// A parked ray resumes with exactly the position it was parked with and the same direction, so
// every float operation of the traversal is the one the uninterrupted loop performs: the five
// output planes are bit-identical (tests run both kernels on the same frames).  Only the
// per-lane cache of occupancy words is rebuilt, which cannot change a result.
//
// Dense targets made of whole 16^3 regions only.  An experiment kept for measurement, like
// volly_refill.h: opt-in with VOLLY_REQUEUE=1 (budget: VOLLY_REQUEUE_BUDGET), 5 % faster than
// the tile kernel on the 1024^3 noise volume at best -- see choose_frame_kernel().
#ifndef VOLLY_REQUEUE_H
#define VOLLY_REQUEUE_H

#include "volly_kernels.h"

namespace vk {

struct RayState {
    float cx, cy, cz;               // position the DDA was parked at (Hit::cx..cz)
    float dx, dy, dz;               // normalised target-grid direction
    uint32_t pix;                   // index into the output planes of this launch
    uint32_t steps_lo;
};
static_assert(sizeof(RayState) == 32, "two 16-byte stores per parked ray");

struct QueueDev {
    RayState *items;
    uint32_t *count;
    uint32_t capacity;
};

// trace_dense_regular (volly_kernels.h) from a given position, over iterations [iter0, iter_end):
// a lane still under way at iter_end leaves with parked = true and its position in the Hit
template <bool REGIONS>
__device__ __forceinline__ Hit trace_dense_span(const DenseDev &g, const Ray &r, float cx, float cy, float cz, uint32_t iter0,
                                                uint32_t iter_end, bool &parked)
{
    Hit h;
    h.cx = cx; h.cy = cy; h.cz = cz;
    h.steps = iter0;
    h.index = VOLLY_MISS_INDEX;
    h.payload = VOLLY_MISS_INDEX;
    parked = false;
    uint32_t last_brick = 0xFFFFFFFFu, w_lo = 0, w_hi = 0;
    uint32_t last_region = 0xFFFFFFFFu, r_lo = 0, r_hi = 0, r_any = 0;
    uint32_t live = 1;
    for (uint32_t iter = iter0;; ++iter) {
        float px = h.cx + r.ex, py = h.cy + r.ey, pz = h.cz + r.ez;
        int x = (int)px, y = (int)py, z = (int)pz;
        uint32_t c = pack_cell(x, y, z);
        if (iter >= g.cap) { h.steps = iter; break; }    // uniform
        if (iter >= iter_end) { h.steps = iter; parked = true; break; }   // uniform: everyone still here is parked
        uint32_t region = c & 0xFF0FC3F0u;
        if (REGIONS && region != last_region) {
            bool inside = ((uint32_t)x < (uint32_t)g.sx) & ((uint32_t)y < (uint32_t)g.sy) & ((uint32_t)z < (uint32_t)g.sz);
            if (inside) {
                uint64_t bits = g.occ2[cell_index24(x >> 4, y >> 4, z >> 4, g.nrx, g.nry)];
                r_lo = (uint32_t)bits;
                r_hi = (uint32_t)(bits >> 32);
                r_any = (r_lo | r_hi) != 0u;
                last_region = region;
            } else {
                r_any = 0;
                live = 0;
                h.steps = iter;
            }
        }
        if (!REGIONS || r_any) {
            if (!REGIONS || test_word(r_lo, r_hi, gather_bits((c >> 2) & 0x00300C03u))) {
                uint32_t brick = c & 0xFFCFF3FCu;
                bool inside = true;
                if (brick != last_brick) {
                    if (!REGIONS) inside = ((uint32_t)x < (uint32_t)g.sx) & ((uint32_t)y < (uint32_t)g.sy) & ((uint32_t)z < (uint32_t)g.sz);
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
                if (inside && test_word(w_lo, w_hi, gather_bits(c & 0x00300C03u))) {
                    live = 0;
                    h.steps = iter + 1;
                    h.index = (uint32_t)x + (uint32_t)g.sx * ((uint32_t)y + (uint32_t)g.sy * (uint32_t)z);
                    h.payload = h.index;
                }
            }
        }
        if (!live) break;
        advance_unit(r, px, py, pz, h.cx, h.cy, h.cz);
    }
    return h;
}

// park the lanes named by `parked` at the end of the queue: one counter update per wavefront
__device__ __forceinline__ void park_rays(const QueueDev &q, bool parked, const RayState &st)
{
    const uint64_t m = __ballot(parked);
    if (!m) return;
    const int lane = (int)(threadIdx.x & 63);
    const int leader = __ffsll((long long)m) - 1;
    uint32_t first = 0;
    if (lane == leader) first = atomicAdd(q.count, (uint32_t)__popcll(m));
    first = (uint32_t)__shfl((int)first, leader);
    if (parked) {
        const uint32_t pos = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < q.capacity) {                           // the host sizes the queue for every ray of the launch
            uint4 *o = (uint4 *)(q.items + pos);
            o[0] = make_uint4(__float_as_uint(st.cx), __float_as_uint(st.cy), __float_as_uint(st.cz), __float_as_uint(st.dx));
            o[1] = make_uint4(__float_as_uint(st.dy), __float_as_uint(st.dz), st.pix, st.steps_lo);
        }
    }
}

__device__ __forceinline__ void write_pixel(const OutDev &out, size_t p, const DenseDev &target, const Hit &hi, uint32_t steps_lo,
                                            const float *origin_hi, const Ray &ray_hi)
{
    uint32_t o_rgba = 255u;
    float o_depth = __uint_as_float(0x7F800000u);
    if (hi.index != VOLLY_MISS_INDEX) {
        o_rgba = shade(target.vox[hi.payload]);
        float ddx = hi.cx - origin_hi[0], ddy = hi.cy - origin_hi[1], ddz = hi.cz - origin_hi[2];
        o_depth = (ddx * ray_hi.dx + ddy * ray_hi.dy) + ddz * ray_hi.dz;
    }
    if (out.rgba8) out.rgba8[p] = o_rgba;
    if (out.hit_index) out.hit_index[p] = hi.index;
    if (out.steps_lo) out.steps_lo[p] = steps_lo;
    if (out.steps_hi) out.steps_hi[p] = hi.steps;
    if (out.depth) out.depth[p] = o_depth;
    if (out.pt_end) { out.pt_end[3 * p] = hi.cx; out.pt_end[3 * p + 1] = hi.cy; out.pt_end[3 * p + 2] = hi.cz; }
}

// First round: render_frames for a dense target (beam and naive frames), the target pass cut at
// `budget` iterations.  One wavefront (8x8 pixels) per workgroup.
__global__ void __launch_bounds__(64)
render_frames_budget(DenseDev target, DenseDev low, const volly_frame_desc *__restrict__ frames, const float *__restrict__ ckpt, OutDev out,
                     QueueDev queue, uint32_t budget)
{
    const int f = blockIdx.z;
    const volly_frame_desc fd = frames[f];
    const int lane = (int)threadIdx.x;
    const int xrel = (int)blockIdx.x * kTile + (lane & 7);
    const int yrel = (int)blockIdx.y * kTile + (lane >> 3);
    const bool active = (xrel < fd.tile_w) & (yrel < fd.tile_h);
    bool parked = false;
    RayState st{};
    if (active) {
        const uint32_t groups = (uint32_t)(fd.tile_w + kCkpt - 1) / (uint32_t)kCkpt;
        const uint32_t cs = (uint32_t)fd.tile_h;
        const float *c = ckpt + (((uint32_t)f * groups + (uint32_t)xrel / (uint32_t)kCkpt) * 6u * cs + (uint32_t)yrel);
        float hx = c[0], hy = c[cs], hz = c[2 * cs], lx = c[3 * cs], ly = c[4 * cs], lz = c[5 * cs];
        const int extra = xrel % kCkpt;
        for (int i = 0; i < extra; ++i) {                 // main.cpp:282-283
            hx = hx + fd.xdiff_hi[0]; hy = hy + fd.xdiff_hi[1]; hz = hz + fd.xdiff_hi[2];
            lx = lx + fd.xdiff_lo[0]; ly = ly + fd.xdiff_lo[1]; lz = lz + fd.xdiff_lo[2];
        }
        normalize_w0(hx, hy, hz);                         // main.cpp:260
        const Ray ray_hi = make_ray(hx, hy, hz);
        uint32_t o_slo = 0;
        float sx, sy, sz;
        if (fd.flags & VOLLY_FRAME_NAIVE) {               // main.cpp:261
            sx = fd.origin_hi[0]; sy = fd.origin_hi[1]; sz = fd.origin_hi[2];
        } else {                                          // raycast_beamOpt, main.cpp:163-166
            normalize_w0(lx, ly, lz);
            const Ray ray_lo = make_ray(lx, ly, lz);
            Hit lo = trace_dense(low, ray_lo, fd.origin_lo[0], fd.origin_lo[1], fd.origin_lo[2]);
            o_slo = lo.steps;
            sx = fd.scale_lo_to_hi[0] * lo.cx; sy = fd.scale_lo_to_hi[1] * lo.cy; sz = fd.scale_lo_to_hi[2] * lo.cz;
        }
        Hit hi;
        if (__all(ray_is_regular(ray_hi))) {              // the grid is regular (checked on the host)
            float cx, cy, cz;
            enter_volume(ray_hi, sx, sy, sz, target.fsx, target.fsy, target.fsz, cx, cy, cz);
            if (!(fabsf(cx) < 3.0e38f && fabsf(cy) < 3.0e38f && fabsf(cz) < 3.0e38f)) {      // as trace_dense_regular
                hi.cx = cx; hi.cy = cy; hi.cz = cz;
                hi.steps = 0; hi.index = VOLLY_MISS_INDEX; hi.payload = VOLLY_MISS_INDEX;
            } else if (target.use_regions) {
                hi = trace_dense_span<true>(target, ray_hi, cx, cy, cz, 0u, budget, parked);
            } else {
                hi = trace_dense_span<false>(target, ray_hi, cx, cy, cz, 0u, budget, parked);
            }
        } else {
            hi = trace_dense_checked<false>(target, ray_hi, sx, sy, sz);
        }
        const size_t p = (size_t)f * (size_t)fd.tile_w * (size_t)fd.tile_h + (size_t)yrel * (size_t)fd.tile_w + (size_t)xrel;
        if (parked) {
            st.cx = hi.cx; st.cy = hi.cy; st.cz = hi.cz;
            st.dx = hx; st.dy = hy; st.dz = hz;
            st.pix = (uint32_t)p;
            st.steps_lo = o_slo;
        } else {
            write_pixel(out, p, target, hi, o_slo, fd.origin_hi, ray_hi);
        }
    }
    park_rays(queue, parked, st);
}

// Later rounds: 64 parked rays per wavefront, iterations [iter0, iter_end) of each
__global__ void __launch_bounds__(64)
continue_rays(DenseDev target, const volly_frame_desc *__restrict__ frames, uint32_t px_per_frame, QueueDev in, QueueDev out_queue, OutDev out,
              uint32_t iter0, uint32_t iter_end)
{
    const uint32_t n = min(*in.count, in.capacity);
    for (uint32_t base = blockIdx.x * 64u; base < n; base += gridDim.x * 64u) {       // uniform trip count
        const uint32_t i = base + threadIdx.x;
        bool parked = false;
        RayState st{};
        if (i < n) {
            const uint4 *s = (const uint4 *)(in.items + i);
            const uint4 a = s[0], b = s[1];
            st.cx = __uint_as_float(a.x); st.cy = __uint_as_float(a.y); st.cz = __uint_as_float(a.z);
            st.dx = __uint_as_float(a.w); st.dy = __uint_as_float(b.x); st.dz = __uint_as_float(b.y);
            st.pix = b.z; st.steps_lo = b.w;
            const Ray ray_hi = make_ray(st.dx, st.dy, st.dz);          // same direction, same reciprocals
            Hit hi;
            if (target.use_regions) hi = trace_dense_span<true>(target, ray_hi, st.cx, st.cy, st.cz, iter0, iter_end, parked);
            else hi = trace_dense_span<false>(target, ray_hi, st.cx, st.cy, st.cz, iter0, iter_end, parked);
            if (parked) {
                st.cx = hi.cx; st.cy = hi.cy; st.cz = hi.cz;
            } else {
                const volly_frame_desc &fd = frames[st.pix / px_per_frame];
                write_pixel(out, (size_t)st.pix, target, hi, st.steps_lo, fd.origin_hi, ray_hi);
            }
        }
        park_rays(out_queue, parked, st);
    }
}

} // namespace vk
#endif
