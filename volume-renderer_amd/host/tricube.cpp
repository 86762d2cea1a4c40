// tricube.cpp -- see tricube.h.  Build with -ffp-contract=off: the reference is
// compiled without FMA (Makefile:3), and several tests below sit on exact
// float comparisons.
#include "tricube.h"

#include <cmath>
#include <cstdint>

namespace volly {
namespace {

// tolerance of the reference's sign test; a double, so floats are widened
// before they are compared with it (triangleCube.c:8-12)
constexpr double kSignTol = 10e-5;

// six face planes, triangleCube.c:47-59 (strict comparisons: touching is inside)
uint32_t face_code(P3 p)
{
    uint32_t c = 0;
    if (p.x > .5) c |= 0x01;
    if (p.x < -.5) c |= 0x02;
    if (p.y > .5) c |= 0x04;
    if (p.y < -.5) c |= 0x08;
    if (p.z > .5) c |= 0x10;
    if (p.z < -.5) c |= 0x20;
    return c;
}

// twelve edge planes, triangleCube.c:65-83
uint32_t edge_code(P3 p)
{
    uint32_t c = 0;
    if (p.x + p.y > 1.0) c |= 0x001;
    if (p.x - p.y > 1.0) c |= 0x002;
    if (-p.x + p.y > 1.0) c |= 0x004;
    if (-p.x - p.y > 1.0) c |= 0x008;
    if (p.x + p.z > 1.0) c |= 0x010;
    if (p.x - p.z > 1.0) c |= 0x020;
    if (-p.x + p.z > 1.0) c |= 0x040;
    if (-p.x - p.z > 1.0) c |= 0x080;
    if (p.y + p.z > 1.0) c |= 0x100;
    if (p.y - p.z > 1.0) c |= 0x200;
    if (-p.y + p.z > 1.0) c |= 0x400;
    if (-p.y - p.z > 1.0) c |= 0x800;
    return c;
}

// eight corner planes, triangleCube.c:89-103
uint32_t corner_code(P3 p)
{
    uint32_t c = 0;
    if ((p.x + p.y + p.z) > 1.5) c |= 0x01;
    if ((p.x + p.y - p.z) > 1.5) c |= 0x02;
    if ((p.x - p.y + p.z) > 1.5) c |= 0x04;
    if ((p.x - p.y - p.z) > 1.5) c |= 0x08;
    if ((-p.x + p.y + p.z) > 1.5) c |= 0x10;
    if ((-p.x + p.y - p.z) > 1.5) c |= 0x20;
    if ((-p.x - p.y + p.z) > 1.5) c |= 0x40;
    if ((-p.x - p.y - p.z) > 1.5) c |= 0x80;
    return c;
}

// point at parameter t on p1->p2 lies on the cube, looking only at `mask`
// faces (triangleCube.c:111-119)
bool on_cube_at(P3 p1, P3 p2, float t, uint32_t mask)
{
    P3 q;
    q.x = p1.x + t * (p2.x - p1.x);
    q.y = p1.y + t * (p2.y - p1.y);
    q.z = p1.z + t * (p2.z - p1.z);
    return (face_code(q) & mask) == 0;
}

// segment p1->p2 against the face planes named in `diff` (triangleCube.c:128-144)
bool segment_hits_cube(P3 p1, P3 p2, uint32_t diff)
{
    if ((diff & 0x01) && on_cube_at(p1, p2, (0.5f - p1.x) / (p2.x - p1.x), 0x3e)) return true;
    if ((diff & 0x02) && on_cube_at(p1, p2, (-0.5f - p1.x) / (p2.x - p1.x), 0x3d)) return true;
    if ((diff & 0x04) && on_cube_at(p1, p2, (0.5f - p1.y) / (p2.y - p1.y), 0x3b)) return true;
    if ((diff & 0x08) && on_cube_at(p1, p2, (-0.5f - p1.y) / (p2.y - p1.y), 0x37)) return true;
    if ((diff & 0x10) && on_cube_at(p1, p2, (0.5f - p1.z) / (p2.z - p1.z), 0x2f)) return true;
    if ((diff & 0x20) && on_cube_at(p1, p2, (-0.5f - p1.z) / (p2.z - p1.z), 0x1f)) return true;
    return false;
}

inline float max3(float a, float b, float c) { return ((a > b) && (a > c)) ? a : ((b > c) ? b : c); }
inline float min3(float a, float b, float c) { return ((a < b) && (a < c)) ? a : ((b < c) ? b : c); }

inline P3 sub(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 crs(P3 a, P3 b)
{
    P3 c;
    c.x = a.y * b.z - a.z * b.y;
    c.y = -a.x * b.z + a.z * b.x;
    c.z = a.x * b.y - a.y * b.x;
    return c;
}

// The reference's SIGN3 macro (triangleCube.c:9-12) mixes ?: and | without
// parentheses, so it parses as one chain of conditionals: the first test that
// holds decides the value.  Reproduced as that chain.
uint32_t sign_code(P3 a)
{
    if (a.x < kSignTol) return 4;
    if (a.x > -kSignTol) return 32;
    if (a.y < kSignTol) return 2;
    if (a.y > -kSignTol) return 16;
    if (a.z < kSignTol) return 1;
    if (a.z > -kSignTol) return 8;
    return 0;
}

// triangleCube.c:150-201
bool point_in_triangle(P3 p, P3 v1, P3 v2, P3 v3)
{
    if (p.x > max3(v1.x, v2.x, v3.x)) return false;
    if (p.y > max3(v1.y, v2.y, v3.y)) return false;
    if (p.z > max3(v1.z, v2.z, v3.z)) return false;
    if (p.x < min3(v1.x, v2.x, v3.x)) return false;
    if (p.y < min3(v1.y, v2.y, v3.y)) return false;
    if (p.z < min3(v1.z, v2.z, v3.z)) return false;

    uint32_t s12 = sign_code(crs(sub(v1, v2), sub(v1, p)));
    uint32_t s23 = sign_code(crs(sub(v2, v3), sub(v2, p)));
    uint32_t s31 = sign_code(crs(sub(v3, v1), sub(v3, p)));
    return (s12 & s23 & s31) != 0;
}

} // namespace

// triangleCube.c:213-319
bool triangle_touches_unit_cube(P3 v1, P3 v2, P3 v3)
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

    // cube diagonals against the triangle's plane
    P3 n = crs(sub(v1, v2), sub(v1, v3));
    float d = n.x * v1.x + n.y * v1.y + n.z * v1.z;
    float den;
    den = n.x + n.y + n.z;
    if (std::fabs(den) > kSignTol) {
        float h = d / den;
        if (std::fabs(h) <= 0.5 && point_in_triangle({h, h, h}, v1, v2, v3)) return true;
    }
    den = n.x + n.y - n.z;
    if (std::fabs(den) > kSignTol) {
        float h = d / den;
        if (std::fabs(h) <= 0.5 && point_in_triangle({h, h, -h}, v1, v2, v3)) return true;
    }
    den = n.x - n.y + n.z;
    if (std::fabs(den) > kSignTol) {
        float h = d / den;
        if (std::fabs(h) <= 0.5 && point_in_triangle({h, -h, h}, v1, v2, v3)) return true;
    }
    den = n.x - n.y - n.z;
    if (std::fabs(den) > kSignTol) {
        float h = d / den;
        if (std::fabs(h) <= 0.5 && point_in_triangle({h, -h, -h}, v1, v2, v3)) return true;
    }
    return false;
}

} // namespace volly
