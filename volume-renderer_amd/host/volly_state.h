// volly_state.h -- C++ adapter that keeps the reference's class API on top of the C-ABI.
//
// volly::State mirrors the reference's State (main.h:89-121): same fields, same
// accessors, and raycastOntoScreen() with the same contract -- fill
// data_stream[x + y*W] with u8vec4(255, r, g, b) pixels for the current volume,
// LoD and camera (main.cpp:197-291).  Its body no longer loops over pixels: it
// computes the per-frame constants exactly as main.cpp:201-249 does
// (volly::frame_from_camera) and hands them to volume_gpu_render.  SDL is gone:
// the caller owns the pixel buffer that Texture_SDL::data_stream used to point at
// (main.h:33), and feeds mouse / key state directly to resolveUserInput.
//
// Header only; link with libvolly_host.so and libvolly_gpu.so.
#ifndef VOLLY_STATE_H
#define VOLLY_STATE_H

#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "volly_scene.h"
#include "../../include/volly_gpu.h"
#include "../../include/volly_host.h"

namespace volly {
namespace gpu {

// RAII handle of one uploaded (target grid, low grid) pair
class Scene {
public:
    Scene() = default;
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    Scene(Scene &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Scene &operator=(Scene &&o) noexcept { std::swap(h_, o.h_); return *this; }
    ~Scene() { volly_gpu_free(h_); }
    volly_gpu_scene *get() const { return h_; }
    explicit operator bool() const { return h_ != nullptr; }

    static Scene upload(const VolumeStore<Voxel> &target, const VolumeStore<Voxel> *low, int device = 0)
    {
        Scene s;
        volly_dense_desc t = target.desc(), l{};
        if (low) l = low->desc();
        check(volume_gpu_upload(device, &t, low ? &l : nullptr, &s.h_), "volume_gpu_upload");
        return s;
    }
    static Scene upload(const SparseVoxelOctree &svo, const VolumeStore<Voxel> *low, int device = 0)
    {
        Scene s;
        volly_svo_desc t = svo.desc();
        volly_dense_desc l{};
        if (low) l = low->desc();
        check(svo_gpu_upload(device, &t, low ? &l : nullptr, &s.h_), "svo_gpu_upload");
        return s;
    }
    static Scene adopt(volly_gpu_scene *h)
    {
        Scene s;
        s.h_ = h;
        return s;
    }
    static void check(int rc, const char *where)
    {
        if (rc != VOLLY_OK) throw std::runtime_error(std::string(where) + ": " + volly_gpu_last_error());
    }
private:
    volly_gpu_scene *h_ = nullptr;
};

// The pixel memory Texture_SDL::lock() used to hand out (main.h:27-49): page-locked, so that
// volume_gpu_render copies frames back while it still traverses (volly_gpu_host_alloc).
class Texture {
public:
    Texture(int w, int h) : w_(w), h_(h), data_stream((uint32_t *)volly_gpu_host_alloc((size_t)w * (size_t)h * 4))
    {
        if (!data_stream) throw std::runtime_error(std::string("volly_gpu_host_alloc: ") + volly_gpu_last_error());
    }
    Texture(const Texture &) = delete;
    Texture &operator=(const Texture &) = delete;
    ~Texture() { volly_gpu_host_free(data_stream); }
    int width() const { return w_; }
    int height() const { return h_; }
private:
    int w_, h_;
public:
    uint32_t *data_stream;          // u8vec4(255, r, g, b) at [x + y*W], main.cpp:269-271
};

} // namespace gpu

// main.h:58-72 without SDL: scancodes are plain ints chosen by the caller
struct KeyHandler {
    std::map<int, bool> keysDown;
    bool getIfKeyDown(int scan) { return keysDown[scan]; }
    void setKeyState(int scan, bool state = true) { keysDown[scan] = state; }
};

enum Key { KEY_W, KEY_A, KEY_S, KEY_D, KEY_LCTRL, KEY_LSHIFT, KEY_M, KEY_COMMA, KEY_PERIOD, KEY_SLASH, KEY_0 /* .. KEY_0+8 */ };

// main.h:89-121
struct State {
    std::vector<Vol_LOD_Set *> volumes;
    int curVolume = 0, curLoD = 0;
    int frame = 0;

    vec3 camPos = vec3(0);
    vec3 camDir = vec3(0);
    vec3 camUp = vec3(0, 0, -1);
    mat4 view;
    KeyHandler keys;
    int mouseX = 0, mouseY = 0;
    int winSizeX = 1152, winSizeY = 864;
    int renderTargetSizeX = 400, renderTargetSizeY = 300;
    int device = 0;

    VolumeStore<Voxel> *getCurVol() { return volumes[(size_t)curVolume]->arr[curLoD]; }
    VolumeStore<Voxel> *getCurVol_LowLOD() { return volumes[(size_t)curVolume]->arr[1]; }
    VolumeStore<Voxel> *getCurVol_SLowLOD() { return volumes[(size_t)curVolume]->arr[0]; }

    ~State() { for (auto *v : volumes) delete v; }

    // main.cpp:369-387 with the asset paths passed in
    void initAllVolumes(const std::vector<std::string> &meshPaths)
    {
        for (const std::string &p : meshPaths) {
            auto *set = new Vol_LOD_Set(p);
            if (!set->error.empty()) { std::string e = set->error; delete set; throw std::runtime_error(e); }
            volumes.push_back(set);
        }
    }

    // The same with the four grids built on the GPU (volume_gpu_voxelize_lod_set) and never held
    // on the host: getCurVol() and friends are not available for such a volume, curVolSize() is.
    void initAllVolumesOnDevice(const std::vector<std::string> &meshPaths, int sLowR = 16, int lowR = 32, int medR = 128, int highR = 256)
    {
        for (const std::string &p : meshPaths) {
            volly_host_mesh *m2 = volly_host_mesh_load(p.c_str(), 2), *m3 = highR > 0 ? volly_host_mesh_load(p.c_str(), 3) : nullptr;
            if (!m2 || (highR > 0 && !m3)) {
                std::string e = volly_host_last_error();
                volly_host_mesh_free(m2); volly_host_mesh_free(m3);
                throw std::runtime_error(e);
            }
            volly_mesh_desc d2{}, d3{};
            volly_host_mesh_view(m2, &d2);
            if (m3) volly_host_mesh_view(m3, &d3);
            volly_gpu_scene *sc[4] = {nullptr, nullptr, nullptr, nullptr};
            int rc = volume_gpu_voxelize_lod_set(device, &d2, m3 ? &d3 : nullptr, sLowR, lowR, medR, highR, sc);
            volly_host_mesh_free(m2); volly_host_mesh_free(m3);
            gpu::Scene::check(rc, "volume_gpu_voxelize_lod_set");
            const int vol = (int)volumes.size();
            volumes.push_back(nullptr);                     // no host grids for this volume
            const int res[4] = {sLowR, lowR, medR, highR};
            for (int lod = 0; lod < 4; ++lod) {
                if (!sc[lod]) continue;
                uploaded_.emplace(std::make_pair(vol, lod), gpu::Scene::adopt(sc[lod]));
                deviceRes_[std::make_pair(vol, lod)] = res[lod];
            }
        }
    }

    // grid sizes of the current volume / LoD and of its beam grid, for either kind of volume
    ivec3 curVolSize()
    {
        if (volumes[(size_t)curVolume]) return getCurVol()->size;
        const int r = deviceRes_.at(std::make_pair(curVolume, curLoD));
        return ivec3(r, r, r);
    }
    ivec3 curLowSize()
    {
        if (volumes[(size_t)curVolume]) return getCurVol_LowLOD()->size;
        const int r = deviceRes_.at(std::make_pair(curVolume, 1));
        return ivec3(r, r, r);
    }

    // main.cpp:389-437 (mouse -> camDir -> view; keys -> camPos, LoD, model)
    void resolveUserInput()
    {
        camUp = vec3(0, 0, -1);
        camDir = cam_dir_from_mouse(mouseX, mouseY, winSizeX, winSizeY);
        view = look_at(vec3(0), camDir, camUp);
        float dy = 0, dx = 0, dz = 0, speed = 0.01f;
        if (keys.getIfKeyDown(KEY_W)) dy += 1;
        if (keys.getIfKeyDown(KEY_A)) dx -= 1;
        if (keys.getIfKeyDown(KEY_S)) dy -= 1;
        if (keys.getIfKeyDown(KEY_D)) dx += 1;
        if (keys.getIfKeyDown(KEY_LCTRL)) dz -= 1;
        if (keys.getIfKeyDown(KEY_LSHIFT)) dz += 1;
        camPos = camPos + (-camDir * dy) * speed;
        vec3 right = cross(camUp, camDir);
        camPos = camPos + (-right * dx) * speed;
        camPos = camPos + (-camUp * dz) * speed;
        for (int k = 0; k <= 8; ++k)
            if (keys.getIfKeyDown(KEY_0 + k) && (size_t)k < volumes.size()) curVolume = k;
        if (keys.getIfKeyDown(KEY_M)) curLoD = 0;
        if (keys.getIfKeyDown(KEY_COMMA)) curLoD = 1;
        if (keys.getIfKeyDown(KEY_PERIOD)) curLoD = 2;
        if (keys.getIfKeyDown(KEY_SLASH)) curLoD = 3;
    }

    // main.cpp:197-291.  data_stream: renderTargetSizeX*renderTargetSizeY pixels of 4
    // bytes, what Texture_SDL::lock() handed out (main.h:38-40).
    void raycastOntoScreen(uint32_t *data_stream)
    {
        gpu::Scene &scene = sceneFor(curVolume, curLoD);
        volly_frame_desc fd;
        frame_from_camera(view, camPos, curVolSize(), curLowSize(), renderTargetSizeX, renderTargetSizeY, &fd);
        volly_outputs out{};
        out.rgba8 = data_stream;
        gpu::Scene::check(volume_gpu_render(scene.get(), &fd, 1, &out), "volume_gpu_render");
        ++frame;
    }

private:
    std::map<std::pair<int, int>, gpu::Scene> uploaded_;
    std::map<std::pair<int, int>, int> deviceRes_;          // resolution of the grids built on the device
    gpu::Scene &sceneFor(int vol, int lod)
    {
        auto key = std::make_pair(vol, lod);
        auto it = uploaded_.find(key);
        if (it == uploaded_.end() && !volumes[(size_t)vol]) throw std::runtime_error("this level of detail was not built on the device");
        if (it == uploaded_.end())
            it = uploaded_.emplace(key, gpu::Scene::upload(*volumes[(size_t)vol]->arr[lod], volumes[(size_t)vol]->arr[1], device)).first;
        return it->second;
    }
};

} // namespace volly
#endif
