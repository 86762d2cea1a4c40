// volly_scene.h -- host-side scene pipeline (C++17, no GLM / SDL / DGtal).
//
// Mirrors the reference's class API for everything that feeds the traversal:
//   Voxel              voxel.h:13-37
//   VolumeStore<T>     volume.h:56-112
//   Polyhedron         PLYfileReader.h:18-21
//   readPLYFile / normalizePoly / rasterizeVoxelMapFromPoly   PLYfileReader.h:23-29
//   volumeFromVoxelMap volume.h:114-123
//   Vol_LOD_Set        main.h:74-87, main.cpp:23-50
//   SparseVoxelOctree  svo.h:33-39 (the reference constructor is an empty stub,
//                      svo.cpp:14-19; the builder here is new)
// The (x,y,z,LoD) -> Voxel std::map of the reference is replaced by VoxelMap, a
// per-LoD sorted sparse list with the same content.
#ifndef VOLLY_SCENE_H
#define VOLLY_SCENE_H

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <memory>

#include "vollymath.h"
#include "../../include/volly_gpu.h"

namespace volly {

// voxel.h:13-37.  8 bytes: rgba then norm; empty <=> rgba[3] == 0.
struct Voxel {
    uint8_t rgba[4] = {0, 0, 0, 0};
    uint8_t norm[4] = {0, 0, 0, 0};
    Voxel() = default;                                    // voxel.h:36
    Voxel(uint8_t r, uint8_t g, uint8_t b, uint8_t a = 255) { rgba[0] = r; rgba[1] = g; rgba[2] = b; rgba[3] = a; }
    static Voxel fromNormalizedFloats(float r, float g, float b, float a = 1.f)
    {   // voxel.h:26-33
        return Voxel((uint8_t)std::floor(r * 255.f), (uint8_t)std::floor(g * 255.f),
                     (uint8_t)std::floor(b * 255.f), (uint8_t)std::floor(a * 255.f));
    }
};
static_assert(sizeof(Voxel) == 8, "Voxel must stay 8 bytes (voxel.h:14-15)");

// volume.h:56-112
template <typename Data_T>
class VolumeStore {
public:
    ivec3 size;
    Data_T *data;

    bool inBounds(ivec3 p) const
    {
        return (p.x < size.x) & (p.y < size.y) & (p.z < size.z) & (p.x >= 0) & (p.y >= 0) & (p.z >= 0);
    }
    size_t index(int x, int y, int z) const { return (size_t)x + (size_t)size.x * ((size_t)y + (size_t)size.y * (size_t)z); }
    Data_T &sample(int x, int y, int z) { return data[index(x, y, z)]; }
    Data_T &sample(ivec3 p) { return data[index(p.x, p.y, p.z)]; }
    void put(Data_T d, int x, int y, int z) { data[index(x, y, z)] = d; }
    void put(Data_T d, ivec3 p) { data[index(p.x, p.y, p.z)] = d; }
    size_t count() const { return (size_t)size.x * size.y * size.z; }

    explicit VolumeStore(ivec3 size_) : size(size_) { data = new Data_T[count()]; }
    ~VolumeStore() { delete[] data; }
    VolumeStore(const VolumeStore &) = delete;
    VolumeStore &operator=(const VolumeStore &) = delete;

    // volume.h:95-111
    void fillSphere(Data_T fillWith);

    volly_dense_desc desc() const { return volly_dense_desc{size.x, size.y, size.z, data}; }
};

// PLYfileReader.h:18-21
struct Polyhedron {
    std::vector<vec4> verts;
    std::vector<ivec3> inds;
};

// PLYfileReader.cpp:15-61.  `path` is a file path (the reference hard-codes
// "assets/" + name + ".ply").  Carriage returns are stripped (the bundled assets
// are CRLF and the reference's header scan never terminates on them on Linux).
// Returns an empty Polyhedron and sets *err when the file cannot be read.
Polyhedron *readPLYFile(const std::string &path, std::string *err = nullptr);
// ASCII OFF (the reference's only OFF reader needs DGtal, volume.cpp:47-127, and
// is never called; this one feeds the same Polyhedron).  Triangles only.
Polyhedron *readOFFFile(const std::string &path, std::string *err = nullptr);
// by extension
Polyhedron *readMeshFile(const std::string &path, std::string *err = nullptr);

// PLYfileReader.cpp:63-139
int stringToInt(const std::string &s);

// PLYfileReader.cpp:144-171
void normalizePoly(Polyhedron *poly);

// Sparse replacement for std::map<ivec4, Voxel, ivec4_cmp> (volume.h:15-19).
// lod[k] holds the entries whose key has w == k: linear cell index
// x + r*(y + r*z) with r = res / (1 << k), ascending, unique, plus the voxel.
struct VoxelMap {
    struct Level {
        int32_t res = 0;
        std::vector<uint64_t> cell;
        std::vector<Voxel> vox;
    };
    int32_t res = 0;
    std::vector<Level> lod;
    size_t size() const { size_t n = 0; for (auto &l : lod) n += l.cell.size(); return n; }
};

// PLYfileReader.cpp:218-321 (same content; flood fill per triangle, last
// triangle in file order wins a cell).  n_threads <= 0: hardware concurrency.
VoxelMap *rasterizeVoxelMapFromPoly(Polyhedron *poly, int res, int n_threads = 0);

// volume.h:114-123
VolumeStore<Voxel> *volumeFromVoxelMap(const VoxelMap *mp, int res, int scale);

// svo.h:33-39.  Flattened octree, breadth first (root first), see volly_svo_node.
class SparseVoxelOctree {
public:
    std::vector<volly_svo_node> data;    // nodes
    std::vector<Voxel> leaves;           // payload of size-1 children
    int maxScale = 0;                    // svo.cpp:16
    int res = 0;

    SparseVoxelOctree() = default;      // filled by the scene cache loader
    // from the LoD-0 entries of a voxel map (the reference signature, svo.h:38)
    SparseVoxelOctree(const VoxelMap *coords, int res);
    // from a dense grid (leaves = its non-empty voxels)
    explicit SparseVoxelOctree(const VolumeStore<Voxel> *vol);

    volly_svo_desc desc() const
    {
        return volly_svo_desc{(uint32_t)data.size(), (uint32_t)leaves.size(), maxScale, res, data.data(), leaves.data()};
    }
private:
    void build(std::vector<uint64_t> &morton, std::vector<Voxel> &vox);
};

// main.h:74-87, main.cpp:23-50
struct Vol_LOD_Set {
    int sLowR, lowR, medR, highR;
    VolumeStore<Voxel> *sLow = nullptr, *low = nullptr, *med = nullptr, *high = nullptr;
    VolumeStore<Voxel> *arr[4] = {nullptr, nullptr, nullptr, nullptr};
    std::string error;

    // `path` = mesh file.  highR <= 0 skips the second rasterization (arr[3] = nullptr).
    Vol_LOD_Set(const std::string &path, int coloringMode = 0, int sLowR = 16, int lowR = 32, int medR = 128,
                int highR = 256);
    ~Vol_LOD_Set() { delete sLow; delete low; delete med; delete high; }
    Vol_LOD_Set(const Vol_LOD_Set &) = delete;
};

// ---- camera / frame constants -------------------------------------------
// main.cpp:201-249
void frame_from_camera(const mat4 &view, vec3 camPos, ivec3 size_hi, ivec3 size_lo, int W, int H,
                       volly_frame_desc *out);
// main.cpp:393-395
vec3 cam_dir_from_mouse(int mouseX, int mouseY, int winX, int winY);
// Orbit view k of n around the unit-cube volume (headless replacement of the
// mouse/WASD camera): position on a circle of `radius`
// around (0.5,0.5,0.5) with elevation 0.3*sin(3*theta); camDir as the reference
// builds it: normalize(normalize(camPos - centre) + 3e-6); camUp = (0,0,-1).
void orbit_camera(int k, int n, float radius, vec3 *camPos, mat4 *view);

template <typename Data_T>
void VolumeStore<Data_T>::fillSphere(Data_T fillWith)
{
    size_t ind = 0;
    vec3 sz((float)size.x, (float)size.y, (float)size.z);
    vec3 mid(sz.x / 2.f, sz.y / 2.f, sz.z / 2.f);
    for (int z = 0; z < size.z; ++z)
        for (int y = 0; y < size.y; ++y)
            for (int x = 0; x < size.x; ++x) {
                vec3 pt((float)x, (float)y, (float)z);
                vec3 rad(4.0f * (pt.x - mid.x) / sz.x, 4.0f * (pt.y - mid.y) / sz.y, 4.0f * (pt.z - mid.z) / sz.z);
                data[ind] = (dot(rad, rad) < 1) ? fillWith : Data_T();
                ++ind;
            }
}

} // namespace volly
#endif
