// volly_host_c.cpp -- extern "C" wrappers of the host scene pipeline (include/volly_host.h).
#include "../../include/volly_host.h"
#include "volly_scene.h"
#include "tricube.h"

#include <cstdio>
#include <sys/types.h>
#include <memory>
#include <string>
#include <vector>

using namespace volly;

struct volly_host_scene {
    std::unique_ptr<VoxelMap> map;
    std::unique_ptr<VolumeStore<Voxel>> target, low;
    std::unique_ptr<SparseVoxelOctree> svo;
};

struct volly_host_lod_set {
    std::unique_ptr<Vol_LOD_Set> set;
};

struct volly_host_mesh {
    std::unique_ptr<Polyhedron> poly;
};
static_assert(sizeof(vec4) == 16 && sizeof(ivec3) == 12, "mesh arrays are handed out as plain float / int32 arrays");

namespace {
thread_local std::string g_err;

VolumeStore<Voxel> *reduce_any(const VolumeStore<Voxel> *hi, int f)
{
    ivec3 ls((hi->size.x + f - 1) / f, (hi->size.y + f - 1) / f, (hi->size.z + f - 1) / f);
    auto *lo = new VolumeStore<Voxel>(ls);
    size_t i = 0;
    for (int z = 0; z < hi->size.z; ++z)
        for (int y = 0; y < hi->size.y; ++y)
            for (int x = 0; x < hi->size.x; ++x, ++i)
                if (hi->data[i].rgba[3] > 0) {
                    Voxel &d = lo->sample(x / f, y / f, z / f);
                    if (d.rgba[3] == 0) d = hi->data[i];
                }
    return lo;
}

VolumeStore<Voxel> *copy_grid(const volly_dense_desc *d)
{
    auto *v = new VolumeStore<Voxel>(ivec3(d->sx, d->sy, d->sz));
    std::memcpy((void *)v->data, d->voxels, v->count() * sizeof(Voxel));
    return v;
}
} // namespace

extern "C" {

const char *volly_host_last_error(void) { return g_err.c_str(); }

volly_host_scene *volly_host_scene_from_mesh(const char *path, int32_t res, int32_t flags, int32_t n_threads)
try {
    g_err.clear();
    if (!path || res < 8 || (res % 4) != 0) { g_err = "volly_host_scene_from_mesh: need a path and res >= 8, multiple of 4"; return nullptr; }
    std::string err;
    std::unique_ptr<Polyhedron> poly(readMeshFile(path, &err));
    if (!err.empty() || poly->verts.empty() || poly->inds.empty()) {
        g_err = err.empty() ? std::string("empty mesh: ") + path : err;
        return nullptr;
    }
    auto s = std::make_unique<volly_host_scene>();
    normalizePoly(poly.get());                                                 // main.cpp:27
    s->map.reset(rasterizeVoxelMapFromPoly(poly.get(), res, n_threads));       // main.cpp:28
    s->low.reset(volumeFromVoxelMap(s->map.get(), res / 4, 2));                // main.cpp:33
    if (flags & VOLLY_SCENE_DENSE) s->target.reset(volumeFromVoxelMap(s->map.get(), res, 0));   // main.cpp:32
    if (flags & VOLLY_SCENE_SVO) s->svo = std::make_unique<SparseVoxelOctree>(s->map.get(), res);
    return s.release();
} catch (const std::exception &e) {          // e.g. std::bad_alloc: must not leave an extern "C" function
    g_err = std::string("volly_host_scene_from_mesh: ") + e.what();
    return nullptr;
}

volly_host_mesh *volly_host_mesh_load(const char *path, int32_t n_normalize)
try {
    g_err.clear();
    if (!path || n_normalize < 0) { g_err = "volly_host_mesh_load: bad arguments"; return nullptr; }
    std::string err;
    std::unique_ptr<Polyhedron> poly(readMeshFile(path, &err));
    if (!err.empty() || poly->verts.empty() || poly->inds.empty()) {
        g_err = err.empty() ? std::string("empty mesh: ") + path : err;
        return nullptr;
    }
    for (int i = 0; i < n_normalize; ++i) normalizePoly(poly.get());
    auto m = std::make_unique<volly_host_mesh>();
    m->poly = std::move(poly);
    return m.release();
} catch (const std::exception &e) {          // e.g. std::bad_alloc: must not leave an extern "C" function
    g_err = std::string("volly_host_mesh_load: ") + e.what();
    return nullptr;
}

int volly_host_mesh_view(const volly_host_mesh *m, volly_mesh_desc *out)
{
    if (!m || !out) return VOLLY_E_ARG;
    out->n_verts = (int64_t)m->poly->verts.size();
    out->verts = &m->poly->verts[0].x;
    out->n_tris = (int64_t)m->poly->inds.size();
    out->tris = &m->poly->inds[0].x;
    return VOLLY_OK;
}

void volly_host_mesh_free(volly_host_mesh *m) { delete m; }

volly_host_scene *volly_host_scene_sphere(int32_t res, const uint8_t fill[8], int32_t flags)
try {
    g_err.clear();
    if (res < 4 || !fill) { g_err = "volly_host_scene_sphere: bad arguments"; return nullptr; }
    auto s = std::make_unique<volly_host_scene>();
    Voxel v;
    std::memcpy(&v, fill, 8);
    s->target = std::make_unique<VolumeStore<Voxel>>(ivec3(res, res, res));
    s->target->fillSphere(v);
    s->low.reset(reduce_any(s->target.get(), 4));
    if (flags & VOLLY_SCENE_SVO) s->svo = std::make_unique<SparseVoxelOctree>(s->target.get());
    return s.release();
} catch (const std::exception &e) {          // e.g. std::bad_alloc: must not leave an extern "C" function
    g_err = std::string("volly_host_scene_sphere: ") + e.what();
    return nullptr;
}

volly_host_scene *volly_host_scene_from_dense(const volly_dense_desc *target, const volly_dense_desc *low, int32_t flags)
try {
    g_err.clear();
    if (!target || !target->voxels || target->sx < 1 || target->sy < 1 || target->sz < 1) {
        g_err = "volly_host_scene_from_dense: bad target";
        return nullptr;
    }
    auto s = std::make_unique<volly_host_scene>();
    s->target.reset(copy_grid(target));
    if (low && low->voxels) s->low.reset(copy_grid(low));
    if (flags & VOLLY_SCENE_SVO) s->svo = std::make_unique<SparseVoxelOctree>(s->target.get());
    return s.release();
} catch (const std::exception &e) {          // e.g. std::bad_alloc: must not leave an extern "C" function
    g_err = std::string("volly_host_scene_from_dense: ") + e.what();
    return nullptr;
}

int volly_host_scene_target(const volly_host_scene *s, volly_dense_desc *out)
{
    if (!s || !out) return VOLLY_E_ARG;
    if (!s->target) return VOLLY_E_UNSUPPORTED;
    *out = s->target->desc();
    return VOLLY_OK;
}

int volly_host_scene_low(const volly_host_scene *s, volly_dense_desc *out)
{
    if (!s || !out) return VOLLY_E_ARG;
    if (!s->low) return VOLLY_E_UNSUPPORTED;
    *out = s->low->desc();
    return VOLLY_OK;
}

int volly_host_scene_svo(const volly_host_scene *s, volly_svo_desc *out)
{
    if (!s || !out) return VOLLY_E_ARG;
    if (!s->svo) return VOLLY_E_UNSUPPORTED;
    *out = s->svo->desc();
    return VOLLY_OK;
}

int64_t volly_host_scene_lod_count(const volly_host_scene *s, int32_t lod)
{
    if (!s || !s->map) return -1;
    if (lod < 0 || (size_t)lod >= s->map->lod.size()) return 0;
    return (int64_t)s->map->lod[(size_t)lod].cell.size();
}

void volly_host_scene_free(volly_host_scene *s) { delete s; }

// ---- scene cache ------------------------------------------------------------
// layout: magic[8] | u32 version | u32 flags | i32 low[3] | i32 target[3] | u32 n_nodes | u32 n_leaves |
//         i32 depth | i32 res | u64 fnv1a of the payload | payload (low, target, nodes, leaves)
namespace {
const char kMagic[8] = {'V', 'O', 'L', 'L', 'Y', 'S', 'C', '1'};
struct CacheHeader {
    char magic[8];
    uint32_t version, flags;
    int32_t low[3], target[3];
    uint32_t n_nodes, n_leaves;
    int32_t depth, res;
    uint64_t checksum;
};
uint64_t fnv1a(uint64_t h, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
}
struct File {
    FILE *f;
    explicit File(FILE *f_) : f(f_) {}
    ~File() { if (f) std::fclose(f); }
};
} // namespace

int volly_host_scene_save(const volly_host_scene *s, const char *path)
{
    g_err.clear();
    if (!s || !path || !s->low) { g_err = "volly_host_scene_save: need a scene with a low grid and a path"; return VOLLY_E_ARG; }
    CacheHeader h{};
    std::memcpy(h.magic, kMagic, 8);
    h.version = 1;
    h.flags = (s->target ? VOLLY_SCENE_DENSE : 0) | (s->svo ? VOLLY_SCENE_SVO : 0);
    h.low[0] = s->low->size.x; h.low[1] = s->low->size.y; h.low[2] = s->low->size.z;
    if (s->target) { h.target[0] = s->target->size.x; h.target[1] = s->target->size.y; h.target[2] = s->target->size.z; }
    if (s->svo) {
        h.n_nodes = (uint32_t)s->svo->data.size(); h.n_leaves = (uint32_t)s->svo->leaves.size();
        h.depth = s->svo->maxScale; h.res = s->svo->res;
    }
    struct Part { const void *p; size_t n; };
    std::vector<Part> parts;
    parts.push_back({s->low->data, s->low->count() * sizeof(Voxel)});
    if (s->target) parts.push_back({s->target->data, s->target->count() * sizeof(Voxel)});
    if (s->svo) {
        parts.push_back({s->svo->data.data(), s->svo->data.size() * sizeof(volly_svo_node)});
        parts.push_back({s->svo->leaves.data(), s->svo->leaves.size() * sizeof(Voxel)});
    }
    uint64_t sum = 0xCBF29CE484222325ull;
    for (const Part &pt : parts) sum = fnv1a(sum, pt.p, pt.n);
    h.checksum = sum;
    File out(std::fopen(path, "wb"));
    if (!out.f) { g_err = std::string("cannot write ") + path; return VOLLY_E_ARG; }
    bool ok = std::fwrite(&h, sizeof h, 1, out.f) == 1;
    for (const Part &pt : parts) ok = ok && (pt.n == 0 || std::fwrite(pt.p, 1, pt.n, out.f) == pt.n);
    if (!ok) { g_err = std::string("short write to ") + path; return VOLLY_E_ARG; }
    return VOLLY_OK;
}

volly_host_scene *volly_host_scene_load(const char *path)
{
    g_err.clear();
    if (!path) { g_err = "volly_host_scene_load: null path"; return nullptr; }
    File in(std::fopen(path, "rb"));
    if (!in.f) { g_err = std::string("cannot open ") + path; return nullptr; }
    CacheHeader h{};
    if (std::fread(&h, sizeof h, 1, in.f) != 1 || std::memcmp(h.magic, kMagic, 8) != 0 || h.version != 1) {
        g_err = std::string("not a volly scene cache: ") + path;
        return nullptr;
    }
    auto bad_dim = [](const int32_t d[3]) { return d[0] < 1 || d[1] < 1 || d[2] < 1 || d[0] > 4096 || d[1] > 4096 || d[2] > 4096; };
    if (bad_dim(h.low) || ((h.flags & VOLLY_SCENE_DENSE) && bad_dim(h.target)) ||
        ((h.flags & VOLLY_SCENE_SVO) && (h.n_nodes < 1 || h.depth < 1 || h.depth > 13 || h.res < 1 || h.res > (1 << h.depth)))) {
        g_err = std::string("implausible header in ") + path;
        return nullptr;
    }
    // the header sizes every allocation below: hold it against the file's real size first, so that a
    // truncated or hostile file cannot ask for hundreds of gigabytes (4096^3 voxels, 2^32 leaves)
    auto cells = [](const int32_t d[3]) { return (uint64_t)d[0] * (uint64_t)d[1] * (uint64_t)d[2]; };
    uint64_t payload = cells(h.low) * sizeof(Voxel);
    if (h.flags & VOLLY_SCENE_DENSE) payload += cells(h.target) * sizeof(Voxel);
    if (h.flags & VOLLY_SCENE_SVO) payload += (uint64_t)h.n_nodes * sizeof(volly_svo_node) + (uint64_t)h.n_leaves * sizeof(Voxel);
    long long file_size = -1;
    if (fseeko(in.f, 0, SEEK_END) == 0) file_size = (long long)ftello(in.f);
    if (file_size < 0 || fseeko(in.f, (off_t)sizeof h, SEEK_SET) != 0 || (uint64_t)file_size != sizeof h + payload) {
        g_err = std::string("truncated or corrupted scene cache (size does not match its header): ") + path;
        return nullptr;
    }
    try {
    auto s = std::make_unique<volly_host_scene>();
    uint64_t sum = 0xCBF29CE484222325ull;
    auto read = [&](void *p, size_t n) {
        if (n && std::fread(p, 1, n, in.f) != n) return false;
        sum = fnv1a(sum, p, n);
        return true;
    };
    bool ok = true;
    s->low = std::make_unique<VolumeStore<Voxel>>(ivec3(h.low[0], h.low[1], h.low[2]));
    ok = ok && read((void *)s->low->data, s->low->count() * sizeof(Voxel));
    if (ok && (h.flags & VOLLY_SCENE_DENSE)) {
        s->target = std::make_unique<VolumeStore<Voxel>>(ivec3(h.target[0], h.target[1], h.target[2]));
        ok = read((void *)s->target->data, s->target->count() * sizeof(Voxel));
    }
    if (ok && (h.flags & VOLLY_SCENE_SVO)) {
        s->svo = std::make_unique<SparseVoxelOctree>();
        s->svo->maxScale = h.depth; s->svo->res = h.res;
        s->svo->data.resize(h.n_nodes);
        s->svo->leaves.resize(h.n_leaves);
        ok = read(s->svo->data.data(), (size_t)h.n_nodes * sizeof(volly_svo_node)) &&
             read(s->svo->leaves.data(), (size_t)h.n_leaves * sizeof(Voxel));
    }
    if (ok && std::fgetc(in.f) != EOF) ok = false;                    // trailing bytes
    if (!ok || sum != h.checksum) {
        g_err = std::string("truncated or corrupted scene cache: ") + path;
        return nullptr;
    }
    return s.release();
    } catch (const std::exception &e) {                    // std::bad_alloc must not leave an extern "C" function
        g_err = std::string("volly_host_scene_load: ") + e.what() + ": " + path;
        return nullptr;
    }
}

volly_host_lod_set *volly_host_lod_set_load(const char *path, int32_t sLowR, int32_t lowR, int32_t medR, int32_t highR)
try {
    g_err.clear();
    if (!path || sLowR < 1 || lowR < 1 || medR < 1) { g_err = "volly_host_lod_set_load: bad arguments"; return nullptr; }
    auto s = std::make_unique<volly_host_lod_set>();
    s->set = std::make_unique<Vol_LOD_Set>(path, 0, sLowR, lowR, medR, highR);
    if (!s->set->error.empty()) { g_err = s->set->error; return nullptr; }
    return s.release();
} catch (const std::exception &e) {          // e.g. std::bad_alloc: must not leave an extern "C" function
    g_err = std::string("volly_host_lod_set_load: ") + e.what();
    return nullptr;
}

int volly_host_lod_set_grid(const volly_host_lod_set *s, int32_t which, volly_dense_desc *out)
{
    if (!s || !out || which < 0 || which > 3) return VOLLY_E_ARG;
    if (!s->set->arr[which]) return VOLLY_E_UNSUPPORTED;
    *out = s->set->arr[which]->desc();
    return VOLLY_OK;
}

void volly_host_lod_set_free(volly_host_lod_set *s) { delete s; }

void volly_host_frame_from_camera(const float view[16], const float cam_pos[3], const int32_t size_hi[3],
                                  const int32_t size_lo[3], int32_t width, int32_t height, volly_frame_desc *out)
{
    mat4 v;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) v[c][r] = view[c * 4 + r];
    frame_from_camera(v, vec3(cam_pos[0], cam_pos[1], cam_pos[2]), ivec3(size_hi[0], size_hi[1], size_hi[2]),
                      ivec3(size_lo[0], size_lo[1], size_lo[2]), width, height, out);
}

void volly_host_look_at(const float eye[3], const float center[3], const float up[3], float view[16])
{
    mat4 v = look_at(vec3(eye[0], eye[1], eye[2]), vec3(center[0], center[1], center[2]), vec3(up[0], up[1], up[2]));
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) view[c * 4 + r] = v[c][r];
}

void volly_host_cam_dir_from_mouse(int32_t mouse_x, int32_t mouse_y, int32_t win_x, int32_t win_y, float cam_dir[3])
{
    vec3 d = cam_dir_from_mouse(mouse_x, mouse_y, win_x, win_y);
    cam_dir[0] = d.x; cam_dir[1] = d.y; cam_dir[2] = d.z;
}

void volly_host_orbit_camera(int32_t k, int32_t n, float radius, float cam_pos[3], float view[16])
{
    vec3 p;
    mat4 v;
    orbit_camera(k, n, radius, &p, &v);
    cam_pos[0] = p.x; cam_pos[1] = p.y; cam_pos[2] = p.z;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) view[c * 4 + r] = v[c][r];
}

int volly_host_tricube(const float t[9])
{
    return triangle_touches_unit_cube({t[0], t[1], t[2]}, {t[3], t[4], t[5]}, {t[6], t[7], t[8]}) ? 0 : 1;
}

} // extern "C"
