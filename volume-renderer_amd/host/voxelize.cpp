// voxelize.cpp -- surface voxelizer and dense-grid scatter (see volly_scene.h).
//
// Same content as the reference's rasterizeVoxelMapFromPoly
// (PLYfileReader.cpp:218-321) with different machinery: the per-triangle
// std::set / std::stack flood fill becomes a flat hash set, triangles are
// spread over threads, and the std::map<ivec4,Voxel> becomes sorted (cell,
// triangle) lists reduced with "highest triangle index wins" -- which is what
// "the later insert overwrites" (PLYfileReader.cpp:283) means, because every
// cell a triangle writes gets the same value.
#include "volly_scene.h"
#include "tricube.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <thread>

namespace volly {
namespace {

constexpr int kCoordBits = 13;               // coordinates are stored +2 (flood fill) or +1 (lists)
constexpr int kTriBits = 25;
constexpr uint64_t kCoordMask = (1ull << kCoordBits) - 1;
constexpr uint64_t kTriMask = (1ull << kTriBits) - 1;

inline uint64_t pack_cell(int x, int y, int z, int off)
{
    return ((uint64_t)(z + off) << (2 * kCoordBits)) | ((uint64_t)(y + off) << kCoordBits) | (uint64_t)(x + off);
}
inline void unpack_cell(uint64_t k, int off, int &x, int &y, int &z)
{
    x = (int)(k & kCoordMask) - off;
    y = (int)((k >> kCoordBits) & kCoordMask) - off;
    z = (int)((k >> (2 * kCoordBits)) & kCoordMask) - off;
}

// open-addressing set of packed cells, cleared by bumping a generation stamp
class CellSet {
public:
    CellSet() { resize(1u << 12); }
    void clear()
    {
        used_ = 0;
        if (++gen_ == 0) { std::fill(stamp_.begin(), stamp_.end(), 0u); gen_ = 1; }
    }
    // true if newly inserted
    bool insert(uint64_t k)
    {
        if ((used_ + 1) * 2 > keys_.size()) grow();
        size_t m = keys_.size() - 1;
        size_t h = (size_t)((k * 0x9E3779B97F4A7C15ull) >> 20) & m;
        while (stamp_[h] == gen_) {
            if (keys_[h] == k) return false;
            h = (h + 1) & m;
        }
        stamp_[h] = gen_;
        keys_[h] = k;
        ++used_;
        return true;
    }
private:
    void resize(size_t n) { keys_.assign(n, 0); stamp_.assign(n, 0u); gen_ = 1; used_ = 0; }
    void grow()
    {
        std::vector<uint64_t> old;
        for (size_t i = 0; i < keys_.size(); ++i)
            if (stamp_[i] == gen_) old.push_back(keys_[i]);
        resize(keys_.size() * 2);
        for (uint64_t k : old) insert(k);
    }
    std::vector<uint64_t> keys_;
    std::vector<uint32_t> stamp_;
    uint32_t gen_ = 1;
    size_t used_ = 0;
};

// ascending sort of 64-bit words, spread over threads by value range
void sort_words(std::vector<uint64_t> &v, int n_threads)
{
    if (v.size() < (1u << 18) || n_threads <= 1) { std::sort(v.begin(), v.end()); return; }
    uint64_t hi = *std::max_element(v.begin(), v.end());
    const int nb = 256;
    uint64_t width = hi / nb + 1;
    std::vector<size_t> cnt(nb + 1, 0);
    for (uint64_t w : v) ++cnt[w / width + 1];
    for (int b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
    std::vector<uint64_t> tmp(v.size());
    {
        std::vector<size_t> pos(cnt.begin(), cnt.end() - 1);
        for (uint64_t w : v) tmp[pos[w / width]++] = w;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t)
        pool.emplace_back([&] {
            for (int b; (b = next.fetch_add(1)) < nb;)
                std::sort(tmp.begin() + (ptrdiff_t)cnt[b], tmp.begin() + (ptrdiff_t)cnt[b + 1]);
        });
    for (auto &th : pool) th.join();
    v.swap(tmp);
}

// sorted (cell<<kTriBits | tri) words -> one word per cell, the highest triangle
void keep_last_per_cell(std::vector<uint64_t> &v)
{
    size_t o = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        if (i + 1 < v.size() && (v[i + 1] >> kTriBits) == (v[i] >> kTriBits)) continue;
        v[o++] = v[i];
    }
    v.resize(o);
}

int pick_threads(int n)
{
    if (n > 0) return n;
    unsigned hc = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(hc ? hc : 1u, 16u);
}

} // namespace

VoxelMap *rasterizeVoxelMapFromPoly(Polyhedron *poly, int res, int n_threads)
{
    normalizePoly(poly);                                   // PLYfileReader.cpp:219
    n_threads = pick_threads(n_threads);
    VoxelMap *ret = new VoxelMap();
    ret->res = res;
    if (res < 1 || res > (1 << kCoordBits) - 4) return ret;

    const float resF = (float)(res - 1);                   // PLYfileReader.cpp:222
    const size_t n_tri = poly->inds.size();
    const size_t n_vert = poly->verts.size();
    std::vector<std::array<uint8_t, 4>> tri_norm(n_tri);

    std::atomic<size_t> next{0};
    std::vector<std::vector<uint64_t>> found((size_t)n_threads);
    auto worker = [&](int tid) {
        CellSet seen;
        std::vector<uint64_t> todo;
        std::vector<uint64_t> &out = found[(size_t)tid];
        for (;;) {
            size_t first = next.fetch_add(32);
            if (first >= n_tri) break;
            size_t last = std::min(first + 32, n_tri);
            for (size_t i = first; i < last; ++i) {
                ivec3 g = poly->inds[i];
                tri_norm[i] = {0, 0, 0, 0};
                if (g.x < 0 || g.y < 0 || g.z < 0 || (size_t)g.x >= n_vert || (size_t)g.y >= n_vert ||
                    (size_t)g.z >= n_vert)
                    continue;                              // the reference would read out of bounds
                vec4 va = poly->verts[(size_t)g.x], vb = poly->verts[(size_t)g.y], vc = poly->verts[(size_t)g.z];
                vec3 a(va.x * resF, va.y * resF, va.z * resF);
                vec3 b(vb.x * resF, vb.y * resF, vb.z * resF);
                vec3 c(vc.x * resF, vc.y * resF, vc.z * resF);
                bool finite = true;
                for (const vec3 &k : {a, b, c})
                    for (int q = 0; q < 3; ++q) finite = finite && std::isfinite(k[q]);
                if (!finite) continue;                     // flat mesh axis: 1/(max-min) = inf in normalizePoly

                // PLYfileReader.cpp:233-237
                vec3 n = normalize(cross(b - a, b - c));
                for (int k = 0; k < 3; ++k) {
                    float v = clampf(n[k], 0.f, 1.f) * 255.f;
                    tri_norm[i][(size_t)k] = (v == v) ? (uint8_t)v : (uint8_t)0;   // NaN (degenerate face) -> 0
                }

                seen.clear();
                todo.clear();
                for (const vec3 &k : {a, b, c}) {          // PLYfileReader.cpp:242-248
                    uint64_t key = pack_cell((int)k.x, (int)k.y, (int)k.z, 2);
                    todo.push_back(key);
                    seen.insert(key);
                }
                while (!todo.empty()) {
                    uint64_t key = todo.back();
                    todo.pop_back();
                    int cx, cy, cz;
                    unpack_cell(key, 2, cx, cy, cz);
                    float ox = (float)cx + 0.5f, oy = (float)cy + 0.5f, oz = (float)cz + 0.5f;
                    bool touches = triangle_touches_unit_cube({a.x - ox, a.y - oy, a.z - oz},
                                                              {b.x - ox, b.y - oy, b.z - oz},
                                                              {c.x - ox, c.y - oy, c.z - oz});
                    if (!touches) continue;
                    if (cx > 1)                            // PLYfileReader.cpp:278
                        out.push_back((pack_cell(cx, cy, cz, 1) << kTriBits) | (uint64_t)i);
                    for (int dz = -1; dz <= 1; ++dz)
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                if (!(dx | dy | dz)) continue;
                                int nx = cx + dx, ny = cy + dy, nz = cz + dz;
                                if (nx < -2 || ny < -2 || nz < -2 || nx > res + 1 || ny > res + 1 || nz > res + 1)
                                    continue;              // cannot touch: the mesh lies in [0, res-1]^3
                                uint64_t nk = pack_cell(nx, ny, nz, 2);
                                if (seen.insert(nk)) todo.push_back(nk);
                            }
                }
            }
        }
    };
    if (n_tri >= (1u << kTriBits)) return ret;
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker, t);
        for (auto &th : pool) th.join();
    }

    // chain[k]: every (cell at LoD k, last triangle) the reference's while loop
    // visits (PLYfileReader.cpp:278-286), in range or not
    std::vector<uint64_t> chain;
    {
        size_t total = 0;
        for (auto &f : found) total += f.size();
        chain.reserve(total);
        for (auto &f : found) { chain.insert(chain.end(), f.begin(), f.end()); std::vector<uint64_t>().swap(f); }
    }
    sort_words(chain, n_threads);
    keep_last_per_cell(chain);

    for (int lod = 0; !chain.empty(); ++lod) {
        const int r = res / (1 << lod);
        VoxelMap::Level lvl;
        lvl.res = r;
        for (uint64_t w : chain) {
            int x, y, z;
            unpack_cell(w >> kTriBits, 1, x, y, z);
            if ((x < 0) | (y < 0) | (z < 0) | (x >= r) | (y >= r) | (z >= r)) continue;   // PLYfileReader.cpp:279
            size_t tri = (size_t)(w & kTriMask);
            Voxel v = Voxel::fromNormalizedFloats(1.f, 1.f, 1.f, 1.f);                    // PLYfileReader.cpp:276-277
            std::memcpy(v.norm, tri_norm[tri].data(), 4);
            lvl.cell.push_back((uint64_t)x + (uint64_t)r * ((uint64_t)y + (uint64_t)r * (uint64_t)z));
            lvl.vox.push_back(v);
        }
        ret->lod.push_back(std::move(lvl));

        // next LoD: toPlace /= 2 (C++ division: -1/2 == 0), kept while x > 1
        std::vector<uint64_t> up;
        up.reserve(chain.size() / 2);
        for (uint64_t w : chain) {
            int x, y, z;
            unpack_cell(w >> kTriBits, 1, x, y, z);
            x /= 2; y /= 2; z /= 2;
            if (x > 1) up.push_back((pack_cell(x, y, z, 1) << kTriBits) | (w & kTriMask));
        }
        sort_words(up, n_threads);
        keep_last_per_cell(up);
        chain.swap(up);
    }
    return ret;
}

VolumeStore<Voxel> *volumeFromVoxelMap(const VoxelMap *mp, int res, int scale)
{
    VolumeStore<Voxel> *ret = new VolumeStore<Voxel>(ivec3(res, res, res));
    if (scale < 0 || (size_t)scale >= mp->lod.size()) return ret;
    const VoxelMap::Level &l = mp->lod[(size_t)scale];
    const uint64_t r = (uint64_t)l.res;
    for (size_t i = 0; i < l.cell.size(); ++i) {
        uint64_t c = l.cell[i];
        int x = (int)(c % r), y = (int)((c / r) % r), z = (int)(c / (r * r));
        // the reference puts without a bounds check (volume.h:117-120); entries that
        // would land outside a smaller store are dropped here instead of corrupting memory
        if (x < res && y < res && z < res) ret->put(l.vox[i], x, y, z);
    }
    return ret;
}

Vol_LOD_Set::Vol_LOD_Set(const std::string &path, int /*coloringMode*/, int sLowR_, int lowR_, int medR_, int highR_)
    : sLowR(sLowR_), lowR(lowR_), medR(medR_), highR(highR_)
{
    // main.cpp:23-50
    std::unique_ptr<Polyhedron> asPoly(readMeshFile(path, &error));
    if (asPoly->verts.empty()) {
        if (error.empty()) error = "empty mesh: " + path;
    }
    normalizePoly(asPoly.get());
    std::unique_ptr<VoxelMap> m1(rasterizeVoxelMapFromPoly(asPoly.get(), medR));
    med = volumeFromVoxelMap(m1.get(), medR, 0);
    low = volumeFromVoxelMap(m1.get(), lowR, 2);
    sLow = volumeFromVoxelMap(m1.get(), sLowR, 3);
    if (highR > 0) {
        std::unique_ptr<VoxelMap> m2(rasterizeVoxelMapFromPoly(asPoly.get(), highR));
        high = volumeFromVoxelMap(m2.get(), highR, 0);
    }
    arr[0] = sLow; arr[1] = low; arr[2] = med; arr[3] = high;
}

} // namespace volly
