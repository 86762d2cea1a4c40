// svo.cpp -- SparseVoxelOctree builder (see volly_scene.h, volly_gpu.h).
//
// The reference declares the node (svo.h:22-30) and a constructor that only
// computes maxScale = ceil(log2(res)) (svo.cpp:14-19).  This builder fills the
// flattened, pointer-free form the GPU path traverses: breadth-first node array,
// children of a node contiguous, leaf payloads in their own array, both in
// Morton order (octant bit = x | y<<1 | z<<2).
#include "volly_scene.h"

#include <algorithm>
#include <numeric>

namespace volly {
namespace {

inline uint64_t spread3(uint64_t v)
{
    // 21 bits -> every third bit
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
inline uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) { return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2); }

int ceil_log2(int v)
{
    int d = 0;
    while ((1 << d) < v) ++d;
    return d;
}

} // namespace

void SparseVoxelOctree::build(std::vector<uint64_t> &morton, std::vector<Voxel> &vox)
{
    maxScale = std::max(1, ceil_log2(res));                // svo.cpp:16
    // sort voxels by Morton code
    std::vector<uint32_t> order(morton.size());
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return morton[a] < morton[b]; });
    std::vector<uint64_t> child(morton.size());
    leaves.resize(morton.size());
    for (size_t i = 0; i < order.size(); ++i) { child[i] = morton[order[i]]; leaves[i] = vox[order[i]]; }

    // level[L-1] = nodes whose cell is 2^L voxels wide, in Morton order
    struct Proto { uint32_t first_child; uint8_t mask; };
    std::vector<std::vector<Proto>> level((size_t)maxScale);
    for (int L = 1; L <= maxScale; ++L) {
        std::vector<Proto> &nodes = level[(size_t)L - 1];
        std::vector<uint64_t> parent;
        for (size_t i = 0; i < child.size(); ++i) {
            uint64_t p = child[i] >> 3;
            if (parent.empty() || parent.back() != p) {
                parent.push_back(p);
                nodes.push_back(Proto{(uint32_t)i, 0});
            }
            nodes.back().mask |= (uint8_t)(1u << (child[i] & 7));
        }
        child.swap(parent);
    }
    if (level[(size_t)maxScale - 1].empty()) level[(size_t)maxScale - 1].push_back(Proto{0, 0});   // empty scene: bare root

    // breadth first: root level first
    std::vector<uint32_t> base((size_t)maxScale + 1, 0);   // base[L] = index of the first node of level L
    uint32_t at = 0;
    for (int L = maxScale; L >= 1; --L) { base[(size_t)L] = at; at += (uint32_t)level[(size_t)L - 1].size(); }
    data.assign(at, volly_svo_node{0, 0, 0, 0});
    for (int L = maxScale; L >= 1; --L) {
        const std::vector<Proto> &nodes = level[(size_t)L - 1];
        for (size_t i = 0; i < nodes.size(); ++i) {
            volly_svo_node &n = data[base[(size_t)L] + i];
            n.valid_mask = nodes[i].mask;
            n.leaf_mask = (L == 1) ? nodes[i].mask : 0;
            n.child_offset = (L == 1) ? nodes[i].first_child : base[(size_t)L - 1] + nodes[i].first_child;
            n.pad = 0;
        }
    }
}

SparseVoxelOctree::SparseVoxelOctree(const VoxelMap *coords, int res_) : res(res_)
{
    std::vector<uint64_t> morton;
    std::vector<Voxel> vox;
    if (coords && !coords->lod.empty()) {
        const VoxelMap::Level &l = coords->lod[0];
        const uint64_t r = (uint64_t)l.res;
        morton.reserve(l.cell.size());
        vox = l.vox;
        for (uint64_t c : l.cell)
            morton.push_back(morton3((uint32_t)(c % r), (uint32_t)((c / r) % r), (uint32_t)(c / (r * r))));
    }
    build(morton, vox);
}

SparseVoxelOctree::SparseVoxelOctree(const VolumeStore<Voxel> *vol)
{
    res = std::max(vol->size.x, std::max(vol->size.y, vol->size.z));
    std::vector<uint64_t> morton;
    std::vector<Voxel> vox;
    size_t i = 0;
    for (int z = 0; z < vol->size.z; ++z)
        for (int y = 0; y < vol->size.y; ++y)
            for (int x = 0; x < vol->size.x; ++x, ++i)
                if (vol->data[i].rgba[3] > 0) {
                    morton.push_back(morton3((uint32_t)x, (uint32_t)y, (uint32_t)z));
                    vox.push_back(vol->data[i]);
                }
    build(morton, vox);
}

} // namespace volly
