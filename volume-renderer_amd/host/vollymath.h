// vollymath.h -- the handful of GLM operations the reference uses, GLM-free.
//
// The reference pulls these from GLM (vollyglm.h:9-11; GLM is not vendored and
// its version is not pinned).  Semantics here are GLM's pure IEEE code paths:
// binary32, no fused multiply-add (build with -ffp-contract=off), column-major
// matrices m[col][row], normalize(v) = v * (1/sqrt(dot(v,v))).
#ifndef VOLLY_MATH_H
#define VOLLY_MATH_H

#include <cmath>
#include <cstdint>

namespace volly {

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default;
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    vec4(const vec3 &v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

struct ivec3 {
    int32_t x = 0, y = 0, z = 0;
    ivec3() = default;
    ivec3(int32_t x_, int32_t y_, int32_t z_) : x(x_), y(y_), z(z_) {}
    int32_t &operator[](int i) { return (&x)[i]; }
    const int32_t &operator[](int i) const { return (&x)[i]; }
};

struct mat4 {
    vec4 c[4];   // columns
    mat4() = default;
    explicit mat4(float d) { c[0].x = d; c[1].y = d; c[2].z = d; c[3].w = d; }
    vec4 &operator[](int i) { return c[i]; }
    const vec4 &operator[](int i) const { return c[i]; }
};

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator/(vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }

// glm::dot(vec3): x*x + y*y + z*z, left to right
inline float dot(vec3 a, vec3 b)
{
    float d = a.x * b.x + a.y * b.y;
    return d + a.z * b.z;
}
// glm::dot(vec4): (x + y) + (z + w) of the products
inline float dot(vec4 a, vec4 b)
{
    return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
}
inline vec3 cross(vec3 a, vec3 b)
{
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline vec4 normalize(vec4 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float clampf(float v, float lo, float hi)
{
    // glm::clamp = min(max(x, lo), hi) with min(a,b) = (b<a)?b:a, max(a,b) = (a<b)?b:a
    float m = (v < lo) ? lo : v;
    return (hi < m) ? hi : m;
}

// glm::scale(mat4(1), v)
inline mat4 scale_identity(vec3 v)
{
    mat4 m(1.0f);
    m[0] = m[0] * v.x;
    m[1] = m[1] * v.y;
    m[2] = m[2] * v.z;
    return m;
}

// glm operator*(mat4, vec4)
inline vec4 operator*(const mat4 &m, vec4 v)
{
    vec4 a = m[0] * v.x + m[1] * v.y;
    vec4 b = m[2] * v.z + m[3] * v.w;
    return a + b;
}

// glm operator*(mat4, mat4)
inline mat4 operator*(const mat4 &a, const mat4 &b)
{
    mat4 r;
    for (int j = 0; j < 4; ++j)
        r[j] = ((a[0] * b[j].x + a[1] * b[j].y) + a[2] * b[j].z) + a[3] * b[j].w;
    return r;
}

// glm::inverse(mat4): cofactor expansion in GLM's operation order
inline mat4 inverse(const mat4 &m)
{
    float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    vec4 f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11);
    vec4 f3(c12, c12, c14, c15), f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    vec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]);
    vec4 v1(m[1][1], m[0][1], m[0][1], m[0][1]);
    vec4 v2(m[1][2], m[0][2], m[0][2], m[0][2]);
    vec4 v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    vec4 i0 = (v1 * f0 - v2 * f1) + v3 * f2;
    vec4 i1 = (v0 * f0 - v2 * f3) + v3 * f4;
    vec4 i2 = (v0 * f1 - v1 * f3) + v3 * f5;
    vec4 i3 = (v0 * f2 - v1 * f4) + v2 * f5;
    vec4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
    mat4 inv;
    inv[0] = i0 * sa; inv[1] = i1 * sb; inv[2] = i2 * sa; inv[3] = i3 * sb;
    vec4 row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    vec4 d = m[0] * row0;
    float det = (d.x + d.y) + (d.z + d.w);
    float ood = 1.0f / det;
    for (int k = 0; k < 4; ++k) inv[k] = inv[k] * ood;
    return inv;
}

// glm::lookAt, right handed
inline mat4 look_at(vec3 eye, vec3 center, vec3 up)
{
    vec3 f = normalize(center - eye);
    vec3 s = normalize(cross(f, up));
    vec3 u = cross(s, f);
    mat4 r(1.0f);
    r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
    r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
    r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
    r[3][0] = -dot(s, eye);
    r[3][1] = -dot(u, eye);
    r[3][2] = dot(f, eye);
    return r;
}

} // namespace volly
#endif
