/*
 * volly_oracle.c -- CPU ORACLE (test infrastructure only; see volly_oracle.h).
 *
 * Scalar, in-order restatement of the reference hot path.  Build with
 *   gcc -O3 -mavx -ffp-contract=off -fno-fast-math
 * (reference Makefile:3 has -mavx but neither -mfma nor fast-math, so every
 * glm::fma there is an unfused multiply then add).  All arithmetic is IEEE
 * binary32, written one operation per statement where the order matters.
 *
 * PARITY UNPINNED (no reference fixtures exist; GLM version unpinned): GLM's
 * pure code paths are assumed: normalize = v * (1/sqrt(dot)), round = half away
 * from zero, fma = a*b + c with two roundings.
 */
#include "volly_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ helpers */

/* glm::sign: (0 < x) - (x < 0) */
static float sgn_f(float x) { return (float)((0.0f < x) - (x < 0.0f)); }

/* float -> int as the reference's x86 build does it (cvttss2si): truncation
 * toward zero; NaN and out-of-range give INT_MIN ("integer indefinite"). */
static int32_t trunc_i32(float f)
{
    if (!(f > -2147483648.0f && f < 2147483648.0f)) return INT_MIN;
    return (int32_t)f;
}

uint32_t vo_step_cap(int32_t sx, int32_t sy, int32_t sz)
{
    return 4u * (uint32_t)(sx + sy + sz) + 8u;
}

typedef struct {
    float dir[3], inv[3], heav[3], sgn_eps[3];
} ray_consts;

/* main.cpp:96-102 */
static void ray_setup(const float dir[3], ray_consts *rc)
{
    const float eps = 0.00030f;
    for (int c = 0; c < 3; ++c) {
        float s = sgn_f(dir[c]);
        rc->dir[c] = dir[c];
        rc->inv[c] = 1.0f / dir[c];
        float h = (s + 1.0f) / 2.0f;
        rc->heav[c] = roundf(h);          /* half away from zero: 0.5 -> 1 */
        rc->sgn_eps[c] = s * eps;
    }
}

/* main.cpp:105-108: clip the start to the volume's entry plane */
static void ray_enter(const float start[3], const ray_consts *rc, const float vol_sz[3], float cur[3])
{
    float nd[3];
    for (int c = 0; c < 3; ++c) {
        float hm1 = rc->heav[c] - 1.0f;
        float prod = vol_sz[c] * hm1;
        float sum = prod + start[c];          /* glm::fma, unfused */
        float dist = -sum;
        nd[c] = dist * rc->inv[c];
    }
    float t0 = fmaxf(fmaxf(fmaxf(nd[0], nd[1]), nd[2]), 0.0f);
    for (int c = 0; c < 3; ++c) {
        float prod = t0 * rc->dir[c];
        cur[c] = prod + start[c];
    }
}

/* main.cpp:149-153 with cell size s (s = 1 for the dense grid) */
static void ray_advance(const float cpe[3], const ray_consts *rc, float s, float inv_s, float cur[3])
{
    float nd[3];
    for (int c = 0; c < 3; ++c) {
        float fl;
        if (s == 1.0f) {
            fl = floorf(cpe[c]);
        } else {
            float scaled = cpe[c] * inv_s;    /* exact: power of two */
            fl = floorf(scaled) * s;          /* exact */
        }
        float hs = rc->heav[c] * s;           /* exact */
        float a = fl + hs;
        float dn = a - cur[c];
        nd[c] = dn * rc->inv[c];
    }
    float m = fminf(fminf(nd[0], nd[1]), nd[2]);
    for (int c = 0; c < 3; ++c) {
        float prod = m * rc->dir[c];
        cur[c] = prod + cur[c];               /* glm::fma, unfused */
    }
}

static void report_miss(vo_report *out, const float cur[3], uint32_t steps, int capped)
{
    out->found = 0;
    memcpy(out->pt_end, cur, sizeof(float) * 3);
    /* Voxel(bgColor.r,g,b) = (0,0,0,255), norm 0  (main.cpp:159, voxel.h:23) */
    memset(out->vox, 0, 8);
    out->vox[3] = 255;
    out->hit_index = VOLLY_MISS_INDEX;
    out->steps = steps;
    out->capped = capped;
}

/* ------------------------------------------------- dense DDA main.cpp:88-161 */

void vo_raycast_dense(const float start[3], const float dir[3], const volly_dense_desc *g,
                      int solid, vo_report *out)
{
    ray_consts rc;
    ray_setup(dir, &rc);
    const float vol_sz[3] = { (float)g->sx, (float)g->sy, (float)g->sz };
    float cur[3];
    ray_enter(start, &rc, vol_sz, cur);

    const uint8_t *data = (const uint8_t *)g->voxels;
    const uint32_t cap = vo_step_cap(g->sx, g->sy, g->sz);
    uint32_t steps = 0;
    for (;;) {
        float cpe[3];
        int32_t cell[3];
        for (int c = 0; c < 3; ++c) {
            cpe[c] = cur[c] + rc.sgn_eps[c];
            cell[c] = trunc_i32(cpe[c]);          /* glm::ivec4 roundPos(curPlusEpsilon) */
        }
        if ((cell[0] >= g->sx) | (cell[1] >= g->sy) | (cell[2] >= g->sz) |
            (cell[0] < 0) | (cell[1] < 0) | (cell[2] < 0))
            break;
        if (steps >= cap) { report_miss(out, cur, steps, 1); return; }

        /* VolumeStore::sample, volume.h:79-81 */
        size_t idx = (size_t)cell[0] + (size_t)g->sx * ((size_t)cell[1] + (size_t)g->sy * (size_t)cell[2]);
        const uint8_t *vox = data + idx * 8;
        ++steps;
        int hit = solid ? (vox[3] == 0) : (vox[3] > 0);
        if (hit) {
            out->found = 1;
            memcpy(out->pt_end, cur, sizeof(float) * 3);      /* returns cur, not cur+eps */
            memcpy(out->vox, vox, 8);
            out->hit_index = (uint32_t)cell[0] +
                             (uint32_t)g->sx * ((uint32_t)cell[1] + (uint32_t)g->sy * (uint32_t)cell[2]);
            out->steps = steps;
            out->capped = 0;
            return;
        }
        ray_advance(cpe, &rc, 1.0f, 1.0f, cur);
    }
    report_miss(out, cur, steps, 0);
}

/* ------------------------------------------- octree DDA (build-defined) ---- */

static int popcount8(unsigned v) { return __builtin_popcount(v & 0xFFu); }

void vo_raycast_svo(const float start[3], const float dir[3], const volly_svo_desc *svo,
                    vo_report *out)
{
    ray_consts rc;
    ray_setup(dir, &rc);
    const int32_t res = svo->res;
    const float vol_sz[3] = { (float)res, (float)res, (float)res };
    float cur[3];
    ray_enter(start, &rc, vol_sz, cur);

    const uint32_t cap = vo_step_cap(res, res, res);
    uint32_t steps = 0;
    for (;;) {
        float cpe[3];
        int32_t cell[3];
        for (int c = 0; c < 3; ++c) {
            cpe[c] = cur[c] + rc.sgn_eps[c];
            cell[c] = trunc_i32(cpe[c]);
        }
        if ((cell[0] >= res) | (cell[1] >= res) | (cell[2] >= res) |
            (cell[0] < 0) | (cell[1] < 0) | (cell[2] < 0))
            break;
        if (steps >= cap) { report_miss(out, cur, steps, 1); return; }
        ++steps;

        /* locate the sample cell: walk down from the root until an empty
         * octant (size 2^(level-1)) or a leaf voxel is found */
        uint32_t node = 0;
        int level = svo->depth;              /* node covers 2^level voxels per side */
        int empty_log2 = -1;
        for (;;) {
            const volly_svo_node *n = &svo->nodes[node];
            int sh = level - 1;
            unsigned oct = ((unsigned)(cell[0] >> sh) & 1u) |
                           (((unsigned)(cell[1] >> sh) & 1u) << 1) |
                           (((unsigned)(cell[2] >> sh) & 1u) << 2);
            unsigned bit = 1u << oct;
            if (!(n->valid_mask & bit)) { empty_log2 = sh; break; }
            uint32_t child = n->child_offset + (uint32_t)popcount8(n->valid_mask & (bit - 1u));
            if (n->leaf_mask & bit) {
                const uint8_t *vox = (const uint8_t *)svo->leaves + (size_t)child * 8;
                out->found = 1;
                memcpy(out->pt_end, cur, sizeof(float) * 3);
                memcpy(out->vox, vox, 8);
                out->hit_index = (uint32_t)cell[0] +
                                 (uint32_t)res * ((uint32_t)cell[1] + (uint32_t)res * (uint32_t)cell[2]);
                out->steps = steps;
                out->capped = 0;
                return;
            }
            node = child;
            level = sh;
        }
        float s = (float)(1u << empty_log2);
        float inv_s = 1.0f / s;
        ray_advance(cpe, &rc, s, inv_s, cur);
    }
    report_miss(out, cur, steps, 0);
}

/* ------------------------------------- raytrace_naiive, main.cpp:170-194 --- */

static float dot4(const float a[4], const float b[4])
{
    /* glm::dot(vec4): (x + y) + (z + w) of the products */
    float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2], p3 = a[3] * b[3];
    return (p0 + p1) + (p2 + p3);
}

void vo_raytrace(int iteration, const float start[3], const float direction[4], const volly_dense_desc *grid,
                 float out[4], vo_report *primary)
{
    const float one_over_127 = 1.f / 127.f, one_over_255 = 1.f / 255.f;
    vo_report rep;
    vo_raycast_dense(start, direction, grid, 0, &rep);
    if (primary) *primary = rep;
    if (iteration <= 1 || !rep.found) {                       /* main.cpp:177 */
        for (int c = 0; c < 4; ++c) out[c] = (float)rep.vox[c] * one_over_255;
        return;
    }
    float norm[4];
    for (int c = 0; c < 4; ++c) {                             /* main.cpp:179-180, unfused */
        float p = (float)rep.vox[4 + c] * one_over_127;
        norm[c] = p + -1.0f;
    }
    /* glm::reflect: I - N * dot(N, I) * 2 */
    float d = dot4(norm, direction);
    float reflect_dir[4], refract_dir[4];
    for (int c = 0; c < 4; ++c) {
        float nd = norm[c] * d;
        float nd2 = nd * 2.f;
        reflect_dir[c] = direction[c] - nd2;
    }
    /* glm::refract(I, N, eta = 2): k = 1 - eta*eta*(1 - d*d); k >= 0 ? eta*I - (eta*d + sqrt(k))*N : 0 */
    {
        const float eta = 2.f;
        float dd = d * d;
        float k = 1.f - (eta * eta) * (1.f - dd);
        if (k >= 0.f) {
            float f = eta * d + sqrtf(k);
            for (int c = 0; c < 4; ++c) refract_dir[c] = eta * direction[c] - f * norm[c];
        } else {
            for (int c = 0; c < 4; ++c) refract_dir[c] = 0.f;
        }
    }
    vo_report refr;
    vo_raycast_dense(rep.pt_end, refract_dir, grid, 1, &refr);        /* main.cpp:185, solid */
    float reflect_col[4], refract_col[4];
    vo_raytrace(iteration - 1, rep.pt_end, reflect_dir, grid, reflect_col, 0);
    vo_raytrace(iteration - 1, refr.pt_end, direction, grid, refract_col, 0);
    const float amt_reflect = 0.35f, amt_refract = 0.00f, amt_diffuse = 0.65f;
    for (int c = 0; c < 4; ++c) {                             /* main.cpp:193 */
        float a = amt_reflect * reflect_col[c];
        float b = amt_refract * refract_col[c];
        float e = (amt_diffuse * (float)rep.vox[c]) * one_over_255;
        out[c] = (a + b) + e;
    }
}

/* ----------------------------------------------- beam, main.cpp:163-166 ---- */

void vo_raycast_beam(const float start_lo[3], const float dir_hi[3], const float dir_lo[3],
                     const float scale[3], const volly_dense_desc *low,
                     const volly_dense_desc *target, const volly_svo_desc *target_svo,
                     vo_report *lo_out, vo_report *hi_out)
{
    vo_raycast_dense(start_lo, dir_lo, low, 0, lo_out);
    float start_hi[3];
    for (int c = 0; c < 3; ++c) start_hi[c] = scale[c] * lo_out->pt_end[c];
    if (target)
        vo_raycast_dense(start_hi, dir_hi, target, 0, hi_out);
    else
        vo_raycast_svo(start_hi, dir_hi, target_svo, hi_out);
}

/* ------------------------------------------------------- small vector math */

/* glm::normalize(vec4) with w = 0: v * inversesqrt(dot), dot = (xx+yy)+(zz+ww) */
static void normalize3_w0(const float v[3], float o[3])
{
    float xx = v[0] * v[0], yy = v[1] * v[1], zz = v[2] * v[2];
    float ww = 0.0f;
    float d = (xx + yy) + (zz + ww);
    float inv = 1.0f / sqrtf(d);
    o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
}

/* glm::normalize(vec3): dot = xx + yy + zz left to right */
static void normalize3(const float v[3], float o[3])
{
    float d = v[0] * v[0] + v[1] * v[1];
    d = d + v[2] * v[2];
    float inv = 1.0f / sqrtf(d);
    o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
}

static void cross3(const float x[3], const float y[3], float o[3])
{
    o[0] = x[1] * y[2] - y[1] * x[2];
    o[1] = x[2] * y[0] - y[2] * x[0];
    o[2] = x[0] * y[1] - y[0] * x[1];
}

static float dot3(const float a[3], const float b[3])
{
    float d = a[0] * b[0] + a[1] * b[1];
    return d + a[2] * b[2];
}

uint32_t vo_shade(const uint8_t vox[8])
{
    /* main.cpp:264-269 */
    float inv = 1.0f / sqrtf((1.0f + 1.0f) + (1.0f + 0.0f));
    float sun = 1.0f * inv;
    float sun_w = 0.0f * inv;
    float nx = (float)vox[4] / 255.0f, ny = (float)vox[5] / 255.0f;
    float nz = (float)vox[6] / 255.0f, nw = (float)vox[7] / 255.0f;
    float d = (sun * nx + sun * ny) + (sun * nz + sun_w * nw);
    float l = d / 2.0f + 0.5f;
    l = fminf(fmaxf(l, 0.0f), 1.0f);
    uint32_t r = (uint8_t)((float)vox[0] * l);
    uint32_t g = (uint8_t)((float)vox[1] * l);
    uint32_t b = (uint8_t)((float)vox[2] * l);
    return 255u | (r << 8) | (g << 16) | (b << 24);
}

/* --------------------------------------- frame loop, main.cpp:251-287 ------ */

static void render_row(const volly_frame_desc *f, int y, const float row_hi[3], const float row_lo[3],
                       const volly_dense_desc *low, const volly_dense_desc *target,
                       const volly_svo_desc *target_svo, const volly_outputs *o)
{
    float cx_hi[3] = { row_hi[0], row_hi[1], row_hi[2] };
    float cx_lo[3] = { row_lo[0], row_lo[1], row_lo[2] };
    const int raytrace = (f->flags & VOLLY_FRAME_RAYTRACE) != 0;
    const int naive = raytrace || (f->flags & VOLLY_FRAME_NAIVE) != 0;
    for (int x = 0; x < f->x0 + f->tile_w; ++x) {
        if (x >= f->x0) {
            float dir_hi[3], dir_lo[3];
            float traced[4] = { 0, 0, 0, 0 };
            vo_report lo, hi;
            normalize3_w0(cx_hi, dir_hi);
            memset(&lo, 0, sizeof lo);
            if (raytrace) {                               /* main.cpp:274-278 (commented call site) */
                float inv_w;                              /* w of normalize(vec4(x,y,z,0)) = 0 * inversesqrt */
                {
                    float xx = cx_hi[0] * cx_hi[0], yy = cx_hi[1] * cx_hi[1], zz = cx_hi[2] * cx_hi[2];
                    inv_w = 0.0f * (1.0f / sqrtf((xx + yy) + (zz + 0.0f)));
                }
                const float dir4[4] = { dir_hi[0], dir_hi[1], dir_hi[2], inv_w };
                vo_raytrace(3, f->origin_hi, dir4, target, traced, &hi);
            } else if (naive) {
                if (target) vo_raycast_dense(f->origin_hi, dir_hi, target, 0, &hi);
                else        vo_raycast_svo(f->origin_hi, dir_hi, target_svo, &hi);
            } else {
                normalize3_w0(cx_lo, dir_lo);
                vo_raycast_beam(f->origin_lo, dir_hi, dir_lo, f->scale_lo_to_hi, low, target, target_svo, &lo, &hi);
            }
            size_t p = (size_t)(x - f->x0) + (size_t)(y - f->y0) * (size_t)f->tile_w;
            if (o->rgba8) {
                if (raytrace) {                           /* clamp to [0,1], u8vec4(255, rgb * 255.f) */
                    uint32_t px = 255u;
                    for (int c = 0; c < 3; ++c) {
                        float v = fminf(fmaxf(traced[c], 0.0f), 1.0f);
                        px |= (uint32_t)(uint8_t)(v * 255.f) << (8 * (c + 1));
                    }
                    o->rgba8[p] = px;
                } else {
                    o->rgba8[p] = hi.found ? vo_shade(hi.vox) : 255u;   /* u8vec4(255,0,0,0) */
                }
            }
            if (o->pt_end) memcpy(o->pt_end + 3 * p, hi.pt_end, 12);          /* RaycastReport::ptEnd of the target pass */
            if (o->hit_index) o->hit_index[p] = hi.hit_index;
            if (o->steps_lo)  o->steps_lo[p] = naive ? 0u : lo.steps;
            if (o->steps_hi)  o->steps_hi[p] = hi.steps;
            if (o->depth) {
                if (hi.found) {
                    float dx = hi.pt_end[0] - f->origin_hi[0];
                    float dy = hi.pt_end[1] - f->origin_hi[1];
                    float dz = hi.pt_end[2] - f->origin_hi[2];
                    float a = dx * dir_hi[0];
                    float b = dy * dir_hi[1];
                    float c = dz * dir_hi[2];
                    o->depth[p] = (a + b) + c;
                } else {
                    o->depth[p] = INFINITY;
                }
            }
        }
        for (int c = 0; c < 3; ++c) {          /* main.cpp:282-283 */
            cx_hi[c] = cx_hi[c] + f->xdiff_hi[c];
            cx_lo[c] = cx_lo[c] + f->xdiff_lo[c];
        }
    }
}

int vo_render_frame(const volly_frame_desc *f, const volly_dense_desc *low,
                    const volly_dense_desc *target, const volly_svo_desc *target_svo,
                    const volly_outputs *outs, int n_threads)
{
    if (!f || !outs || (!target && !target_svo) || (target && target_svo)) return VOLLY_E_ARG;
    if ((f->flags & VOLLY_FRAME_RAYTRACE) && !target) return VOLLY_E_UNSUPPORTED;
    if (!(f->flags & (VOLLY_FRAME_NAIVE | VOLLY_FRAME_RAYTRACE)) && !low) return VOLLY_E_UNSUPPORTED;
    if (f->x0 < 0 || f->y0 < 0 || f->tile_w < 0 || f->tile_h < 0 ||
        f->x0 + f->tile_w > f->width || f->y0 + f->tile_h > f->height) return VOLLY_E_ARG;

    const int y_end = f->y0 + f->tile_h;
    /* serial accumulation of the per-row start vectors, main.cpp:285-286 */
    float *rows = (float *)malloc(sizeof(float) * 6 * (size_t)(y_end > 0 ? y_end : 1));
    if (!rows) return VOLLY_E_NOMEM;
    float cy_hi[3] = { f->row0_hi[0], f->row0_hi[1], f->row0_hi[2] };
    float cy_lo[3] = { f->row0_lo[0], f->row0_lo[1], f->row0_lo[2] };
    for (int y = 0; y < y_end; ++y) {
        memcpy(rows + 6 * (size_t)y, cy_hi, 12);
        memcpy(rows + 6 * (size_t)y + 3, cy_lo, 12);
        for (int c = 0; c < 3; ++c) {
            cy_hi[c] = cy_hi[c] + f->ydiff_hi[c];
            cy_lo[c] = cy_lo[c] + f->ydiff_lo[c];
        }
    }
#ifdef _OPENMP
    if (n_threads > 1) {
        #pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
        for (int y = f->y0; y < y_end; ++y)
            render_row(f, y, rows + 6 * (size_t)y, rows + 6 * (size_t)y + 3, low, target, target_svo, outs);
    } else
#endif
    {
        (void)n_threads;
        for (int y = f->y0; y < y_end; ++y)
            render_row(f, y, rows + 6 * (size_t)y, rows + 6 * (size_t)y + 3, low, target, target_svo, outs);
    }
    free(rows);
    return VOLLY_OK;
}

/* ------------------------------------------------------------ mat4 (GLM) --- */
/* column major: m[col*4 + row] */

static void mat4_mul(const float a[16], const float b[16], float o[16])
{
    /* glm operator*(mat4,mat4): Result[j] = A0*B[j][0] + A1*B[j][1] + A2*B[j][2] + A3*B[j][3] */
    float r[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float s = a[0 * 4 + i] * b[j * 4 + 0];
            s = s + a[1 * 4 + i] * b[j * 4 + 1];
            s = s + a[2 * 4 + i] * b[j * 4 + 2];
            s = s + a[3 * 4 + i] * b[j * 4 + 3];
            r[j * 4 + i] = s;
        }
    memcpy(o, r, sizeof r);
}

static void mat4_mul_vec4(const float m[16], const float v[4], float o[4])
{
    /* glm operator*(mat4,vec4): (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w) */
    float r[4];
    for (int i = 0; i < 4; ++i) {
        float a = m[0 * 4 + i] * v[0] + m[1 * 4 + i] * v[1];
        float b = m[2 * 4 + i] * v[2] + m[3 * 4 + i] * v[3];
        r[i] = a + b;
    }
    memcpy(o, r, sizeof r);
}

/* glm::inverse(mat4), GLM's cofactor formulation (detail::compute_inverse) */
static void mat4_inverse(const float m[16], float o[16])
{
#define M(c, r) m[(c) * 4 + (r)]
    float c00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
    float c02 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
    float c03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
    float c04 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
    float c06 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    float c07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
    float c08 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
    float c10 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
    float c11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
    float c12 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
    float c14 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
    float c15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
    float c16 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
    float c18 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
    float c19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
    float c20 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
    float c22 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
    float c23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
    const float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 };
    const float f2[4] = { c08, c08, c10, c11 }, f3[4] = { c12, c12, c14, c15 };
    const float f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
    const float v0[4] = { M(1,0), M(0,0), M(0,0), M(0,0) };
    const float v1[4] = { M(1,1), M(0,1), M(0,1), M(0,1) };
    const float v2[4] = { M(1,2), M(0,2), M(0,2), M(0,2) };
    const float v3[4] = { M(1,3), M(0,3), M(0,3), M(0,3) };
    const float sa[4] = { +1, -1, +1, -1 }, sb[4] = { -1, +1, -1, +1 };
    float inv[16];
    for (int i = 0; i < 4; ++i) {
        float i0 = (v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i];
        float i1 = (v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i];
        float i2 = (v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i];
        float i3 = (v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i];
        inv[0 * 4 + i] = i0 * sa[i];
        inv[1 * 4 + i] = i1 * sb[i];
        inv[2 * 4 + i] = i2 * sa[i];
        inv[3 * 4 + i] = i3 * sb[i];
    }
    float d0 = M(0,0) * inv[0 * 4 + 0], d1 = M(0,1) * inv[1 * 4 + 0];
    float d2 = M(0,2) * inv[2 * 4 + 0], d3 = M(0,3) * inv[3 * 4 + 0];
    float det = (d0 + d1) + (d2 + d3);
    float ood = 1.0f / det;
    for (int k = 0; k < 16; ++k) o[k] = inv[k] * ood;
#undef M
}

static void mat4_scale_identity(const float v[3], float o[16])
{
    memset(o, 0, sizeof(float) * 16);
    o[0] = 1.0f * v[0]; o[5] = 1.0f * v[1]; o[10] = 1.0f * v[2]; o[15] = 1.0f;
}

void vo_look_at(const float eye[3], const float center[3], const float up[3], float view[16])
{
    float d[3] = { center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] };
    float f[3], s0[3], s[3], u[3];
    normalize3(d, f);
    cross3(f, up, s0);
    normalize3(s0, s);
    cross3(s, f, u);
    memset(view, 0, sizeof(float) * 16);
    view[15] = 1.0f;
    view[0 * 4 + 0] = s[0]; view[1 * 4 + 0] = s[1]; view[2 * 4 + 0] = s[2];
    view[0 * 4 + 1] = u[0]; view[1 * 4 + 1] = u[1]; view[2 * 4 + 1] = u[2];
    view[0 * 4 + 2] = -f[0]; view[1 * 4 + 2] = -f[1]; view[2 * 4 + 2] = -f[2];
    view[3 * 4 + 0] = -dot3(s, eye);
    view[3 * 4 + 1] = -dot3(u, eye);
    view[3 * 4 + 2] = dot3(f, eye);
}

void vo_cam_dir_from_mouse(int mouse_x, int mouse_y, int win_x, int win_y, float cam_dir[3])
{
    /* main.cpp:393-395: the angle is evaluated in double (M_PI), cos/sin in double */
    double ang = 2.5f * (float)(-mouse_x) * M_PI / win_x;
    float v[3];
    v[0] = (float)cos(ang);
    v[1] = (float)sin(ang);
    v[2] = -4.f * (float)mouse_y / (float)win_y + 2.f;
    for (int c = 0; c < 3; ++c) v[c] = v[c] + 0.000003f;
    normalize3(v, cam_dir);
}

static void local_cam_pos(const float model[16], const float cam_pos[3], float o[4])
{
    /* main.cpp:228-230 */
    float im[16], p[4] = { cam_pos[0], cam_pos[1], cam_pos[2], 1.0f }, r[4];
    mat4_inverse(model, im);
    mat4_mul_vec4(im, p, r);
    float w = r[3];
    o[0] = r[0] / w; o[1] = r[1] / w; o[2] = r[2] / w; o[3] = 0.0f;
}

static void frame_rays(const float view[16], const int32_t size[3], int32_t W, int32_t H,
                       float model[16], float row0[4], float xdiff[4], float ydiff[4])
{
    /* main.cpp:204-221,232 */
    float sc[3] = { 1.f / (float)size[0], 1.f / (float)size[1], 1.f / (float)size[2] };
    float mv[16], imv[16];
    mat4_scale_identity(sc, model);
    mat4_mul(view, model, mv);
    float fov = (float)(90.0 * M_PI / 180.0);
    float d = 1 / tanf(fov / 2);
    mat4_inverse(mv, imv);
    float aspect_n = (-1 * ((float)W)) / H;
    float aspect_p = (2 * ((float)W)) / H;
    float a[4] = { aspect_n, -1, d, 0 }, b[4] = { 0, 2, 0, 0 }, c[4] = { aspect_p, 0, 0, 0 };
    float yb[4], xb[4];
    mat4_mul_vec4(imv, a, row0);
    mat4_mul_vec4(imv, b, yb);
    mat4_mul_vec4(imv, c, xb);
    for (int k = 0; k < 4; ++k) { ydiff[k] = yb[k] / (float)H; xdiff[k] = xb[k] / (float)W; }
    row0[3] = ydiff[3] = xdiff[3] = 0.0f;
}

void vo_frame_from_camera(const float view[16], const float cam_pos[3],
                          const int32_t size_hi[3], const int32_t size_lo[3],
                          int32_t W, int32_t H, volly_frame_desc *out)
{
    float model[16], lowmodel[16];
    memset(out, 0, sizeof *out);
    frame_rays(view, size_hi, W, H, model, out->row0_hi, out->xdiff_hi, out->ydiff_hi);
    local_cam_pos(model, cam_pos, out->origin_hi);
    frame_rays(view, size_lo, W, H, lowmodel, out->row0_lo, out->xdiff_lo, out->ydiff_lo);
    local_cam_pos(lowmodel, cam_pos, out->origin_lo);
    /* main.cpp:239: vec4(size_hi,0) / vec4(size_lo,1) */
    for (int c = 0; c < 3; ++c) out->scale_lo_to_hi[c] = (float)size_hi[c] / (float)size_lo[c];
    out->scale_lo_to_hi[3] = 0.0f / 1.0f;
    out->width = W; out->height = H;
    out->x0 = 0; out->y0 = 0; out->tile_w = W; out->tile_h = H;
    out->flags = VOLLY_FRAME_BEAM;
}

/* ------------------------------------- VolumeStore::fillSphere volume.h:95 */

void vo_fill_sphere(int32_t sx, int32_t sy, int32_t sz, const uint8_t fill[8], uint8_t *voxels)
{
    const float size[3] = { (float)sx, (float)sy, (float)sz };
    float mid[3];
    for (int c = 0; c < 3; ++c) mid[c] = size[c] / 2.f;
    size_t ind = 0;
    for (int z = 0; z < sz; ++z)
        for (int y = 0; y < sy; ++y)
            for (int x = 0; x < sx; ++x) {
                const float pt[3] = { (float)x, (float)y, (float)z };
                float rad[3];
                for (int c = 0; c < 3; ++c) rad[c] = 4.0f * (pt[c] - mid[c]) / size[c];
                float r2 = dot3(rad, rad);
                if (r2 < 1) memcpy(voxels + ind * 8, fill, 8);
                else        memset(voxels + ind * 8, 0, 8);
                ++ind;
            }
}
