// volly_internal.h -- shared by the translation units of libvolly_gpu.so (not installed).
#ifndef VOLLY_INTERNAL_H
#define VOLLY_INTERNAL_H

#include <hip/hip_runtime.h>
#include <string>
#include <utility>
#include <vector>
#include "volly_gpu.h"
#include "volly_kernels.h"

namespace vimpl {

extern thread_local std::string g_err;

inline int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return vimpl::fail(e_ == hipErrorOutOfMemory ? VOLLY_E_NOMEM : VOLLY_E_HIP,           \
                               std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

// remembers the caller's current device and puts it back
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

} // namespace vimpl

struct volly_gpu_scene {
    int device = 0;
    bool is_svo = false, has_low = false;
    bool packed = false;            // dense target without the region level: frames run through vk::render_frames_packed
    uint32_t pack_budget = 48;      // iterations per round of that kernel
    int pack_passes = 6;            // passes of a launch: rays still under way at a pass's end are parked and pooled for the next
    int pack_lookahead_from = 3;    // passes from this index on use the look-ahead loop (bricks requested ahead of each ray)
    uint32_t pack_pass_len = 96;    // iterations of the first passes (later ones double)
    void *d_queue = nullptr;        // the two queues of parked rays + their counters
    size_t queue_rays = 0;
    vk::DenseDev target{}, low{};
    vk::SvoDev svo{};
    uint32_t n_leaves = 0;
    // owned device allocations
    std::vector<void *> owned;
    size_t bytes = 0;
    // per-call scratch
    volly_frame_desc *d_frames = nullptr;
    float *d_ckpt = nullptr;
    size_t ckpt_floats = 0;
    int2 *d_spans = nullptr;                // sky spans, one per frame and tile row of a launch
    size_t span_count = 0;
    hipStream_t scratch_stream = nullptr;   // the stream the last call queued its use of the scratch on
    hipEvent_t scratch_event = nullptr;     // orders a call on another stream behind it (render_chunk)
    bool scratch_in_use = false;
    int sky_cull = 2;               // tiles whose rays provably miss both grids write their miss values without ray set-up: 0 off, 1 bounding rectangle, 2 rectangle + hull edges
    void *d_planes = nullptr;       // staging of the synchronous entry point
    size_t plane_bytes = 0;
    hipStream_t s_render = nullptr, s_copy = nullptr;   // pinned-output path: render and copy-back overlap
    void *noalloc_scratch = nullptr; // this device's copy of the static scratch (volume_gpu_render_device_noalloc), not owned
    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};

namespace vimpl {

int new_scene(int device, volly_gpu_scene **out);
void destroy(volly_gpu_scene *s);
int dev_alloc(volly_gpu_scene *s, void **p, size_t bytes);
// hand an existing device allocation over to the scene (freed with it)
void adopt(volly_gpu_scene *s, void *p, size_t bytes);
// upload (host_vox) or adopt (dev_vox, already owned by the scene) the 8-byte voxels of one grid and derive its occupancy levels
int make_dense(volly_gpu_scene *s, int sx, int sy, int sz, const void *host_vox, uint2 *dev_vox, vk::DenseDev *out);
// describe a device-resident octree (arrays already owned by the scene), build its direct-index table and, when the
// level-3 nodes are the index range [l3_first, l3_first + l3_count) (l3_first < 0: unknown), its occupancy blocks
int finish_svo(volly_gpu_scene *s, uint2 *nodes, uint2 *leaves, uint32_t n_nodes, uint32_t n_leaves, int depth, int res,
               int64_t l3_first, int64_t l3_count);
// which frame kernel a finished scene's launches take (after target / low are known)
void choose_frame_kernel(volly_gpu_scene *s);

} // namespace vimpl

#endif
