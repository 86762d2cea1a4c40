// host_sanitize_driver.cpp -- walks every entry point of the host library (include/volly_host.h)
// on small scenes.  tests/test_host_sanitize.py compiles it together with the
// host sources under -fsanitize=address,undefined and runs it: any heap overrun, use after free,
// leak or undefined arithmetic in the host pipeline fails the test.  No GPU.
//
//   usage: host_sanitize_driver <assets dir> <scratch dir>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "volly_host.h"

#define CHECK(cond)                                                                              \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            std::fprintf(stderr, "%s:%d: check failed: %s (%s)\n", __FILE__, __LINE__, #cond,   \
                         volly_host_last_error());                                               \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

static uint64_t fold(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

static int walk_scene(volly_host_scene *s, bool dense, bool svo)
{
    volly_dense_desc d{};
    volly_svo_desc t{};
    CHECK(volly_host_scene_low(s, &d) == VOLLY_OK);
    (void)fold(d.voxels, (size_t)d.sx * d.sy * d.sz * 8);
    if (dense) {
        CHECK(volly_host_scene_target(s, &d) == VOLLY_OK);
        (void)fold(d.voxels, (size_t)d.sx * d.sy * d.sz * 8);
    } else {
        CHECK(volly_host_scene_target(s, &d) == VOLLY_E_UNSUPPORTED);
    }
    if (svo) {
        CHECK(volly_host_scene_svo(s, &t) == VOLLY_OK);
        (void)fold(t.nodes, (size_t)t.n_nodes * 8);
        if (t.n_leaves) (void)fold(t.leaves, (size_t)t.n_leaves * 8);
    } else {
        CHECK(volly_host_scene_svo(s, &t) == VOLLY_E_UNSUPPORTED);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string assets = argv[1], tmp = argv[2];

    // meshes: both readers, every normalisation count the pipeline uses
    for (const char *name : {"icosahedron.ply", "icosahedron.off", "teapot.off", "cube.ply"}) {
        for (int n = 0; n <= 3; ++n) {
            volly_host_mesh *m = volly_host_mesh_load((assets + "/" + name).c_str(), n);
            if (!m) continue;                             // cube.ply holds quads: the reader may refuse it
            volly_mesh_desc d{};
            CHECK(volly_host_mesh_view(m, &d) == VOLLY_OK);
            CHECK(d.n_verts > 0 && d.n_tris > 0);
            (void)fold(d.verts, (size_t)d.n_verts * 16);
            (void)fold(d.tris, (size_t)d.n_tris * 12);
            volly_host_mesh_free(m);
        }
    }
    CHECK(volly_host_mesh_load((assets + "/no_such_file.ply").c_str(), 2) == nullptr);
    volly_host_mesh_free(nullptr);

    // scenes from meshes: dense, octree, both; one and several threads; odd resolutions
    for (int res : {8, 20, 32, 64}) {
        for (int flags : {VOLLY_SCENE_DENSE, VOLLY_SCENE_SVO, VOLLY_SCENE_DENSE | VOLLY_SCENE_SVO}) {
            for (int threads : {1, 4}) {
                volly_host_scene *s = volly_host_scene_from_mesh((assets + "/icosahedron.ply").c_str(), res, flags, threads);
                CHECK(s != nullptr);
                if (walk_scene(s, flags & VOLLY_SCENE_DENSE, flags & VOLLY_SCENE_SVO)) return 1;
                CHECK(volly_host_scene_lod_count(s, 0) > 0);
                (void)volly_host_scene_lod_count(s, 9);
                volly_host_scene_free(s);
            }
        }
    }
    {
        volly_host_scene *s = volly_host_scene_from_mesh((assets + "/teapot.off").c_str(), 64, VOLLY_SCENE_DENSE | VOLLY_SCENE_SVO, 4);
        CHECK(s != nullptr);
        if (walk_scene(s, true, true)) return 1;

        // the scene cache: round trip, then a truncated and a corrupted copy
        const std::string path = tmp + "/teapot64.vsc";
        CHECK(volly_host_scene_save(s, path.c_str()) == VOLLY_OK);
        volly_host_scene *l = volly_host_scene_load(path.c_str());
        CHECK(l != nullptr);
        volly_dense_desc a{}, b{};
        CHECK(volly_host_scene_target(s, &a) == VOLLY_OK && volly_host_scene_target(l, &b) == VOLLY_OK);
        CHECK(a.sx == b.sx && std::memcmp(a.voxels, b.voxels, (size_t)a.sx * a.sy * a.sz * 8) == 0);
        volly_host_scene_free(l);

        std::vector<unsigned char> blob;
        {
            FILE *f = std::fopen(path.c_str(), "rb");
            CHECK(f != nullptr);
            unsigned char buf[65536];
            size_t n;
            while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + n);
            std::fclose(f);
        }
        for (size_t cut : {(size_t)0, (size_t)7, blob.size() / 3, blob.size() - 1}) {
            FILE *f = std::fopen((tmp + "/cut.vsc").c_str(), "wb");
            CHECK(f != nullptr);
            std::fwrite(blob.data(), 1, cut, f);
            std::fclose(f);
            CHECK(volly_host_scene_load((tmp + "/cut.vsc").c_str()) == nullptr);
        }
        for (size_t at : {(size_t)5, (size_t)40, blob.size() / 2, blob.size() - 3}) {
            std::vector<unsigned char> bad = blob;
            bad[at] ^= 0x5A;
            FILE *f = std::fopen((tmp + "/bad.vsc").c_str(), "wb");
            CHECK(f != nullptr);
            std::fwrite(bad.data(), 1, bad.size(), f);
            std::fclose(f);
            volly_host_scene *x = volly_host_scene_load((tmp + "/bad.vsc").c_str());
            if (x) volly_host_scene_free(x);              // a flipped bit in padding may go unnoticed; it must not crash
        }
        volly_host_scene_free(s);
    }
    CHECK(volly_host_scene_from_mesh((assets + "/no_such_file.off").c_str(), 32, VOLLY_SCENE_DENSE, 1) == nullptr);

    // spheres and caller grids
    {
        const uint8_t fill[8] = {255, 255, 255, 255, 10, 200, 30, 0};
        for (int res : {4, 12, 33}) {
            volly_host_scene *s = volly_host_scene_sphere(res, fill, VOLLY_SCENE_DENSE | VOLLY_SCENE_SVO);
            CHECK(s != nullptr);
            if (walk_scene(s, true, true)) return 1;
            volly_dense_desc t{}, lo{};
            CHECK(volly_host_scene_target(s, &t) == VOLLY_OK && volly_host_scene_low(s, &lo) == VOLLY_OK);
            volly_host_scene *c = volly_host_scene_from_dense(&t, &lo, VOLLY_SCENE_DENSE | VOLLY_SCENE_SVO);
            CHECK(c != nullptr);
            if (walk_scene(c, true, true)) return 1;
            volly_host_scene_free(c);
            c = volly_host_scene_from_dense(&t, nullptr, VOLLY_SCENE_SVO);
            if (c) volly_host_scene_free(c);
            volly_host_scene_free(s);
        }
    }

    // the four-grid set, default and odd sizes
    for (int k = 0; k < 2; ++k) {
        volly_host_lod_set *ls = k == 0 ? volly_host_lod_set_load((assets + "/icosahedron.ply").c_str(), 16, 32, 128, 0)
                                        : volly_host_lod_set_load((assets + "/icosahedron.ply").c_str(), 8, 16, 64, 96);
        CHECK(ls != nullptr);
        for (int which = 0; which < 4; ++which) {
            volly_dense_desc d{};
            int rc = volly_host_lod_set_grid(ls, which, &d);
            if (rc == VOLLY_OK) (void)fold(d.voxels, (size_t)d.sx * d.sy * d.sz * 8);
        }
        volly_host_lod_set_free(ls);
    }

    // cameras and frame constants
    {
        float view[16], dir[3], pos[3] = {0.5f, -1.2f, 0.5f};
        const float eye[3] = {0, 0, 0}, up[3] = {0, 0, -1};
        for (int mx = -50; mx <= 1200; mx += 250) {
            volly_host_cam_dir_from_mouse(mx, 431, 1152, 864, dir);
            volly_host_look_at(eye, dir, up, view);
            const int32_t hi[3] = {128, 128, 128}, lo[3] = {32, 32, 32};
            volly_frame_desc fd;
            volly_host_frame_from_camera(view, pos, hi, lo, 64, 48, &fd);
            (void)fold(&fd, sizeof fd);
        }
        for (int k = 0; k < 16; ++k) {
            volly_host_orbit_camera(k, 16, 1.2f, pos, view);
            (void)fold(view, sizeof view);
        }
    }

    // the cube test on a handful of triangles, degenerate ones included
    {
        const float tris[][9] = {
            {0.1f, 0.1f, 0.1f, 0.2f, 0.1f, 0.1f, 0.1f, 0.2f, 0.1f},
            {2, 2, 2, 3, 2, 2, 2, 3, 2},
            {-1, -1, 0, 1, -1, 0, 0, 2, 0},
            {0, 0, 0, 0, 0, 0, 0, 0, 0},
            {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 9, 9, 9},
        };
        for (const auto &t : tris) (void)volly_host_tricube(t);
    }
    std::puts("host sanitize driver: ok");
    return 0;
}
