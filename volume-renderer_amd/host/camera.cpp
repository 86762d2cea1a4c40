// camera.cpp -- per-frame constants and headless camera generation (see volly_scene.h).
#include "volly_scene.h"

namespace volly {
namespace {

struct RaySetup { mat4 model; vec4 row0, xdiff, ydiff; };

// main.cpp:204-221,232 for one grid size
RaySetup ray_setup(const mat4 &view, ivec3 size, int W, int H)
{
    RaySetup r;
    r.model = scale_identity(vec3(1.f / (float)size.x, 1.f / (float)size.y, 1.f / (float)size.z));
    mat4 MV = view * r.model;
    float FOV = (float)(90.0 * M_PI / 180.0);
    float d = 1 / std::tan(FOV / 2);
    mat4 iMV = inverse(MV);
    r.row0 = iMV * vec4((-1 * ((float)W)) / H, -1, d, 0);
    r.ydiff = (iMV * vec4(0, 2, 0, 0)) / (float)H;
    r.xdiff = (iMV * vec4((2 * ((float)W)) / H, 0, 0, 0)) / (float)W;
    r.row0.w = r.ydiff.w = r.xdiff.w = 0;
    return r;
}

// main.cpp:228-230
vec4 to_local(const mat4 &model, vec3 camPos)
{
    vec4 p = inverse(model) * vec4(camPos, 1);
    p = p / p.w;
    p.w = 0;
    return p;
}

void store(float dst[4], vec4 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }

} // namespace

void frame_from_camera(const mat4 &view, vec3 camPos, ivec3 size_hi, ivec3 size_lo, int W, int H,
                       volly_frame_desc *out)
{
    std::memset(out, 0, sizeof *out);
    RaySetup hi = ray_setup(view, size_hi, W, H);
    RaySetup lo = ray_setup(view, size_lo, W, H);
    store(out->row0_hi, hi.row0); store(out->xdiff_hi, hi.xdiff); store(out->ydiff_hi, hi.ydiff);
    store(out->row0_lo, lo.row0); store(out->xdiff_lo, lo.xdiff); store(out->ydiff_lo, lo.ydiff);
    store(out->origin_hi, to_local(hi.model, camPos));
    store(out->origin_lo, to_local(lo.model, camPos));
    // main.cpp:239: vec4(size_hi, 0) / vec4(size_lo, 1)
    out->scale_lo_to_hi[0] = (float)size_hi.x / (float)size_lo.x;
    out->scale_lo_to_hi[1] = (float)size_hi.y / (float)size_lo.y;
    out->scale_lo_to_hi[2] = (float)size_hi.z / (float)size_lo.z;
    out->scale_lo_to_hi[3] = 0.0f;
    out->width = W; out->height = H;
    out->x0 = 0; out->y0 = 0; out->tile_w = W; out->tile_h = H;
    out->flags = VOLLY_FRAME_BEAM;
}

vec3 cam_dir_from_mouse(int mouseX, int mouseY, int winX, int winY)
{
    // main.cpp:393-395
    double ang = 2.5f * (float)(-mouseX) * M_PI / winX;
    vec3 d((float)std::cos(ang), (float)std::sin(ang), -4.f * (float)mouseY / (float)winY + 2.f);
    d = d + vec3(0.000003f);
    return normalize(d);
}

void orbit_camera(int k, int n, float radius, vec3 *camPos, mat4 *view)
{
    double theta = 2.0 * M_PI * (double)k / (double)(n > 0 ? n : 1);
    vec3 centre(0.5f, 0.5f, 0.5f);
    vec3 p(0.5f + radius * (float)std::cos(theta), 0.5f + radius * (float)std::sin(theta),
           0.5f + 0.3f * (float)std::sin(3.0 * theta));
    vec3 dir = normalize(p - centre);
    dir = normalize(dir + vec3(0.000003f));                // main.cpp:394-395
    *camPos = p;
    *view = look_at(vec3(0.f), dir, vec3(0, 0, -1));       // main.cpp:392,397
}

} // namespace volly
