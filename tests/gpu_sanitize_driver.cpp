// gpu_sanitize_driver.cpp -- walks the C-ABI of libvolly_gpu.so (include/volly_gpu.h) with the
// HOST side of the library compiled under AddressSanitizer (device code is not instrumented:
// GPU sanitizers are not available on this pool).  Built and run by hand, see DESIGN.md section 3:
//
//   make sanitize-gpu            # builds build/san/{lib/*.so, gpu_sanitize_driver}
//   build/san/gpu_sanitize_driver assets        # on a box with an MI355X
//
// Exercises scene life cycles (upload / voxelize / noise / free), the synchronous entry point
// with pageable and pinned buffers, the device entry point, the readers and the error paths.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "volly_gpu.h"
#include "volly_host.h"

#define CHECK(cond)                                                                                      \
    do {                                                                                                 \
        if (!(cond)) {                                                                                   \
            std::fprintf(stderr, "%s:%d: check failed: %s (gpu: %s | host: %s)\n", __FILE__, __LINE__,  \
                         #cond, volly_gpu_last_error(), volly_host_last_error());                        \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)

struct Planes {
    std::vector<uint32_t> rgba, hit, slo, shi;
    std::vector<float> depth, end;
    explicit Planes(size_t n) : rgba(n), hit(n), slo(n), shi(n), depth(n), end(3 * n) {}
    volly_outputs outs() { return volly_outputs{rgba.data(), hit.data(), slo.data(), shi.data(), depth.data(), end.data()}; }
};

static void frames_for(int n, int res_hi, int res_lo, int w, int h, std::vector<volly_frame_desc> &out)
{
    out.resize((size_t)n);
    for (int k = 0; k < n; ++k) {
        float pos[3], view[16];
        volly_host_orbit_camera(k, n, 1.2f, pos, view);
        const int32_t hi[3] = {res_hi, res_hi, res_hi}, lo[3] = {res_lo, res_lo, res_lo};
        volly_host_frame_from_camera(view, pos, hi, lo, w, h, &out[(size_t)k]);
    }
}

static int render_all_ways(volly_gpu_scene *sc, int res_hi, int res_lo)
{
    const int W = 72, H = 40, N = 35;                      // more frames than one launch chunk holds
    std::vector<volly_frame_desc> fr;
    frames_for(N, res_hi, res_lo, W, H, fr);
    const size_t px = (size_t)W * H * N;

    Planes a(px), b(px);
    volly_outputs oa = a.outs();
    CHECK(volume_gpu_render(sc, fr.data(), N, &oa) == VOLLY_OK);          // pageable buffers

    void *pin[6];                                                          // pinned buffers: the overlapped path
    const size_t words[6] = {1, 1, 1, 1, 1, 3};
    for (int i = 0; i < 6; ++i) CHECK((pin[i] = volly_gpu_host_alloc(px * 4 * words[i])) != nullptr);
    volly_outputs op{(uint32_t *)pin[0], (uint32_t *)pin[1], (uint32_t *)pin[2], (uint32_t *)pin[3], (float *)pin[4], (float *)pin[5]};
    CHECK(volume_gpu_render(sc, fr.data(), N, &op) == VOLLY_OK);
    CHECK(std::memcmp(pin[0], a.rgba.data(), px * 4) == 0 && std::memcmp(pin[1], a.hit.data(), px * 4) == 0);
    CHECK(std::memcmp(pin[5], a.end.data(), px * 12) == 0);
    for (void *p : pin) volly_gpu_host_free(p);
    {   // (the call above is small enough to store straight into the pinned buffers;) a larger one takes the staged, overlapped path
        const int W2 = 320, H2 = 200;
        std::vector<volly_frame_desc> big;
        frames_for(N, res_hi, res_lo, W2, H2, big);
        const size_t px2 = (size_t)W2 * H2 * N;
        void *p0 = volly_gpu_host_alloc(px2 * 4), *p1 = volly_gpu_host_alloc(px2 * 4);
        CHECK(p0 && p1);
        volly_outputs o2{(uint32_t *)p0, (uint32_t *)p1, nullptr, nullptr, nullptr, nullptr};
        CHECK(volume_gpu_render(sc, big.data(), N, &o2) == VOLLY_OK);
        std::vector<uint32_t> ref(px2);
        volly_outputs o3{ref.data(), nullptr, nullptr, nullptr, nullptr, nullptr};
        CHECK(volume_gpu_render(sc, big.data(), N, &o3) == VOLLY_OK);
        CHECK(std::memcmp(p0, ref.data(), px2 * 4) == 0);
        volly_gpu_host_free(p0); volly_gpu_host_free(p1);
    }

    volly_outputs only_rgba{b.rgba.data(), nullptr, nullptr, nullptr, nullptr, nullptr};   // one plane, naive frames, a sub-rectangle
    for (auto &f : fr) { f.flags = VOLLY_FRAME_NAIVE; f.x0 = 8; f.y0 = 3; f.tile_w = 40; f.tile_h = 21; }
    CHECK(volume_gpu_render(sc, fr.data(), N, &only_rgba) == VOLLY_OK);

    // device entry point on a stream of the caller's, with timing
    uint32_t *d_rgba = nullptr;
    hipStream_t st;
    CHECK(hipMalloc(&d_rgba, (size_t)40 * 21 * N * 4) == hipSuccess && hipStreamCreate(&st) == hipSuccess);
    volly_outputs od{d_rgba, nullptr, nullptr, nullptr, nullptr, nullptr};
    CHECK(volly_gpu_set_profiling(sc, 1) == VOLLY_OK);
    CHECK(volume_gpu_render_device(sc, fr.data(), N, &od, st) == VOLLY_OK);
    CHECK(hipStreamSynchronize(st) == hipSuccess);
    double ms = 0;
    int launches = 0;
    CHECK(volly_gpu_kernel_time(sc, &ms, &launches) == VOLLY_OK && launches == (N + 63) / 64);   // one launch per 64 frames
    CHECK(volly_gpu_set_profiling(sc, 0) == VOLLY_OK);
    std::vector<uint32_t> back((size_t)40 * 21 * N);
    CHECK(hipMemcpy(back.data(), d_rgba, back.size() * 4, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(std::memcmp(back.data(), b.rgba.data(), back.size() * 4) == 0);
    // the entry point that allocates nothing, then the ordinary one on ANOTHER stream (the scene's
    // scratch is handed from stream to stream), then the no-alloc size limit
    hipStream_t st2;
    CHECK(hipStreamCreate(&st2) == hipSuccess);
    CHECK(hipMemset(d_rgba, 0, back.size() * 4) == hipSuccess);
    CHECK(volume_gpu_render_device_noalloc(sc, fr.data(), N, &od, st) == VOLLY_OK);
    CHECK(hipStreamSynchronize(st) == hipSuccess);
    CHECK(hipMemcpy(back.data(), d_rgba, back.size() * 4, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(std::memcmp(back.data(), b.rgba.data(), back.size() * 4) == 0);
    CHECK(volume_gpu_render_device(sc, fr.data(), N, &od, st) == VOLLY_OK);
    CHECK(volume_gpu_render_device(sc, fr.data(), N, &od, st2) == VOLLY_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    CHECK(hipMemcpy(back.data(), d_rgba, back.size() * 4, hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(std::memcmp(back.data(), b.rgba.data(), back.size() * 4) == 0);
    volly_frame_desc huge = fr[0];
    huge.width = huge.tile_w = 7680; huge.height = huge.tile_h = 4320; huge.x0 = huge.y0 = 0;
    CHECK(volume_gpu_render_device_noalloc(sc, &huge, 1, &od, st) == VOLLY_E_ARG);
    (void)hipStreamDestroy(st2);
    (void)hipFree(d_rgba);
    (void)hipStreamDestroy(st);

    // error paths leave the scene usable
    volly_frame_desc bad = fr[0];
    bad.tile_w = 4000;
    CHECK(volume_gpu_render(sc, &bad, 1, &only_rgba) == VOLLY_E_ARG);
    bad = fr[0];
    bad.flags = 0x40;
    CHECK(volume_gpu_render(sc, &bad, 1, &only_rgba) == VOLLY_E_ARG);
    CHECK(volume_gpu_render(sc, fr.data(), 0, &only_rgba) == VOLLY_E_ARG);
    CHECK(volume_gpu_render(sc, nullptr, 1, &only_rgba) == VOLLY_E_ARG);
    CHECK(volume_gpu_render(sc, fr.data(), 1, nullptr) == VOLLY_E_ARG);
    CHECK(volume_gpu_render(sc, fr.data(), 1, &only_rgba) == VOLLY_OK);
    for (int what = -1; what < 18; ++what) (void)volly_gpu_scene_info(sc, what);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const std::string assets = argv[1];
    CHECK(volly_gpu_abi_version() == 3);
    CHECK(std::strstr(volly_gpu_build_info(), "src_sha16=") != nullptr);

    // host scene -> dense upload and octree upload
    volly_host_scene *hs = volly_host_scene_from_mesh((assets + "/teapot.off").c_str(), 64, VOLLY_SCENE_DENSE | VOLLY_SCENE_SVO, 4);
    CHECK(hs != nullptr);
    volly_dense_desc t{}, lo{};
    volly_svo_desc sv{};
    CHECK(volly_host_scene_target(hs, &t) == VOLLY_OK && volly_host_scene_low(hs, &lo) == VOLLY_OK && volly_host_scene_svo(hs, &sv) == VOLLY_OK);
    {
        volly_gpu_scene *sc = nullptr;
        CHECK(volume_gpu_upload(0, &t, &lo, &sc) == VOLLY_OK);
        if (render_all_ways(sc, 64, 16)) return 1;
        std::vector<uint64_t> vox((size_t)64 * 64 * 64);
        CHECK(volly_gpu_read_voxels(sc, 0, vox.size(), vox.data()) == VOLLY_OK);
        CHECK(std::memcmp(vox.data(), t.voxels, vox.size() * 8) == 0);
        CHECK(volly_gpu_read_voxels(sc, vox.size(), 1, vox.data()) == VOLLY_E_ARG);
        CHECK(volly_gpu_read_low_voxels(sc, 5, 100, vox.data()) == VOLLY_OK);
        CHECK(volly_gpu_read_svo(sc, nullptr, nullptr) == VOLLY_E_UNSUPPORTED);
        volly_gpu_free(sc);
    }
    {
        volly_gpu_scene *sc = nullptr;
        CHECK(svo_gpu_upload(0, &sv, &lo, &sc) == VOLLY_OK);
        if (render_all_ways(sc, 64, 16)) return 1;
        std::vector<volly_svo_node> nodes(sv.n_nodes);
        std::vector<uint64_t> leaves(sv.n_leaves);
        CHECK(volly_gpu_read_svo(sc, nodes.data(), leaves.data()) == VOLLY_OK);
        CHECK(std::memcmp(nodes.data(), sv.nodes, nodes.size() * sizeof(volly_svo_node)) == 0);
        volly_gpu_free(sc);

        // a damaged tree is refused, nothing is left behind
        std::vector<volly_svo_node> broken(sv.nodes, sv.nodes + sv.n_nodes);
        broken[0].child_offset = sv.n_nodes + 7;
        volly_svo_desc bd = sv;
        bd.nodes = broken.data();
        CHECK(svo_gpu_upload(0, &bd, &lo, &sc) == VOLLY_E_ARG && sc == nullptr);
        CHECK(svo_gpu_upload(0, &sv, &lo, nullptr) == VOLLY_E_ARG);
        CHECK(svo_gpu_upload(99, &sv, &lo, &sc) == VOLLY_E_ARG);
    }
    volly_host_scene_free(hs);

    // scenes prepared on the device
    {
        volly_host_mesh *m = volly_host_mesh_load((assets + "/teapot.off").c_str(), 2);
        CHECK(m != nullptr);
        volly_mesh_desc md{};
        CHECK(volly_host_mesh_view(m, &md) == VOLLY_OK);
        for (int flags : {VOLLY_VOX_DENSE, VOLLY_VOX_SVO, VOLLY_VOX_DENSE | VOLLY_VOX_SVO}) {
            volly_gpu_scene *sc = nullptr;
            int64_t stats[8];
            CHECK(volume_gpu_voxelize(0, &md, 64, flags, &sc, stats) == VOLLY_OK);
            if (render_all_ways(sc, 64, 16)) return 1;
            volly_gpu_free(sc);
        }
        volly_gpu_scene *sc = nullptr;
        CHECK(volume_gpu_voxelize(0, &md, 63, VOLLY_VOX_SVO, &sc, nullptr) == VOLLY_E_ARG);
        CHECK(volume_gpu_voxelize(0, &md, 64, 0, &sc, nullptr) == VOLLY_E_ARG);
        volly_mesh_desc empty = md;
        empty.n_tris = 0;
        CHECK(volume_gpu_voxelize(0, &empty, 64, VOLLY_VOX_SVO, &sc, nullptr) == VOLLY_E_ARG);

        volly_host_mesh *m3 = volly_host_mesh_load((assets + "/teapot.off").c_str(), 3);
        CHECK(m3 != nullptr);
        volly_mesh_desc md3{};
        CHECK(volly_host_mesh_view(m3, &md3) == VOLLY_OK);
        volly_gpu_scene *set[4];
        CHECK(volume_gpu_voxelize_lod_set(0, &md, &md3, 16, 32, 128, 256, set) == VOLLY_OK);
        for (int i = 0; i < 4; ++i) {
            CHECK(set[i] != nullptr);
            if (i == 2 && render_all_ways(set[i], 128, 32)) return 1;
            volly_gpu_free(set[i]);
        }
        CHECK(volume_gpu_voxelize_lod_set(0, &md, nullptr, 16, 32, 128, 0, set) == VOLLY_OK);
        CHECK(set[3] == nullptr);
        for (int i = 0; i < 3; ++i) volly_gpu_free(set[i]);
        volly_host_mesh_free(m3);
        volly_host_mesh_free(m);
    }

    // the synthetic noise volume and the batch cube test
    {
        volly_gpu_scene *sc = nullptr;
        CHECK(volume_gpu_generate_noise(0, 64, 1u, 1u << 24, &sc) == VOLLY_OK);
        CHECK(volly_gpu_scene_info(sc, 8) == 3);           // the packed frame kernel and its queue of parked rays
        if (render_all_ways(sc, 64, 16)) return 1;
        volly_gpu_free(sc);
        // a side that is no multiple of 16: occupancy levels padded to whole regions, a 25^3 beam grid
        CHECK(volume_gpu_generate_noise(0, 100, 2u, 1u << 24, &sc) == VOLLY_OK);
        CHECK(volly_gpu_scene_info(sc, 8) == 3 && volly_gpu_scene_info(sc, 14) == 2 && volly_gpu_scene_info(sc, 15) == 2);
        CHECK(volly_gpu_scene_info(sc, 17) > 0 && volly_gpu_scene_info(sc, 17) < 100 * 100 * 100);
        if (render_all_ways(sc, 100, 25)) return 1;
        volly_gpu_free(sc);
        CHECK(volume_gpu_generate_noise(0, 30, 1u, 1u << 24, &sc) == VOLLY_E_ARG);
        std::vector<float> tri(9 * 1000);
        for (size_t i = 0; i < tri.size(); ++i) tri[i] = (float)((i * 2654435761u) % 1000u) / 500.0f - 1.0f;
        std::vector<int32_t> out(1000);
        CHECK(volly_gpu_tricube(0, tri.data(), 1000, out.data()) == VOLLY_OK);
        for (int i = 0; i < 1000; i += 37) CHECK(out[(size_t)i] == volly_host_tricube(&tri[(size_t)i * 9]));
        CHECK(volly_gpu_tricube(0, nullptr, 10, out.data()) == VOLLY_E_ARG);
    }
    volly_gpu_free(nullptr);
    std::puts("gpu sanitize driver: ok");
    std::fflush(stdout);
    return 0;
}
