// mesh.cpp -- ASCII PLY / OFF loaders and normalizePoly (see volly_scene.h).
#include "volly_scene.h"

#include <cstdlib>
#include <fstream>
#include <sstream>

namespace volly {
namespace {

bool next_line(std::istream &in, std::string &line)
{
    if (!std::getline(in, line)) return false;
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
    return true;
}

// PLYfileReader.cpp:88-111: "x y z ..." -> three strtof reads split on single
// spaces (the third field keeps any trailing text; stof stops at it).
vec4 parse_vertex(const std::string &line)
{
    size_t s1 = line.find(' ');
    std::string a = line.substr(0, s1);
    std::string rest = (s1 == std::string::npos) ? std::string() : line.substr(s1 + 1);
    size_t s2 = rest.find(' ');
    std::string b = rest.substr(0, s2);
    std::string c = (s2 == std::string::npos) ? std::string() : rest.substr(s2 + 1);
    vec4 v;
    v.x = std::strtof(a.c_str(), nullptr);
    v.y = std::strtof(b.c_str(), nullptr);
    v.z = std::strtof(c.c_str(), nullptr);
    v.w = 0.f;   // left uninitialised by the reference; never read (normalizePoly zeroes its scale)
    return v;
}

// PLYfileReader.cpp:113-139: drop the leading count, then three fields; every
// digit found in a field is concatenated (stringToInt), so "2 3 " reads as 23.
ivec3 parse_face(std::string line)
{
    size_t s0 = line.find(' ');
    line = (s0 == std::string::npos) ? std::string() : line.substr(s0 + 1);
    size_t s1 = line.find(' ');
    std::string a = line.substr(0, s1);
    std::string rest = (s1 == std::string::npos) ? std::string() : line.substr(s1 + 1);
    size_t s2 = rest.find(' ');
    std::string b = rest.substr(0, s2);
    std::string c = (s2 == std::string::npos) ? std::string() : rest.substr(s2 + 1);
    return ivec3(stringToInt(a), stringToInt(b), stringToInt(c));
}

} // namespace

int stringToInt(const std::string &s)
{
    // PLYfileReader.cpp:63-86: all digits of the string, in order, as one number
    long long v = 0;
    for (char ch : s)
        if (ch >= '0' && ch <= '9') v = (v * 10 + (ch - '0')) & 0xFFFFFFFFll;
    return (int)(uint32_t)v;
}

Polyhedron *readPLYFile(const std::string &path, std::string *err)
{
    Polyhedron *product = new Polyhedron{};
    std::ifstream file(path);
    if (!file.is_open()) {
        if (err) *err = "cannot open " + path;
        return product;
    }
    int numVert = 0, numFaces = 0;
    std::string line;
    bool header = true;
    while (header) {
        if (!next_line(file, line)) {
            if (err) *err = "no end_header in " + path;
            return product;
        }
        if (!line.empty() && line[0] == 'e') {
            if (line == "end_header") header = false;
            else if (line.compare(0, 7, "element") == 0) {
                if (line.compare(7, 8, " vertex ") == 0) numVert = stringToInt(line);
                else if (line.compare(7, 6, " face ") == 0) numFaces = stringToInt(line);
            }
        }
    }
    for (int i = 0; i < numVert; ++i) {
        if (!next_line(file, line)) { if (err) *err = "truncated vertex list in " + path; return product; }
        product->verts.push_back(parse_vertex(line));
    }
    for (int i = 0; i < numFaces; ++i) {
        if (!next_line(file, line)) { if (err) *err = "truncated face list in " + path; return product; }
        product->inds.push_back(parse_face(line));
    }
    return product;
}

Polyhedron *readOFFFile(const std::string &path, std::string *err)
{
    Polyhedron *product = new Polyhedron{};
    std::ifstream file(path);
    if (!file.is_open()) {
        if (err) *err = "cannot open " + path;
        return product;
    }
    std::string line;
    // first non-empty, non-comment line must start with OFF
    auto content_line = [&](std::string &out) {
        while (next_line(file, out)) {
            size_t p = out.find_first_not_of(" \t");
            if (p == std::string::npos || out[p] == '#') continue;
            return true;
        }
        return false;
    };
    if (!content_line(line) || line.compare(0, 3, "OFF") != 0) {
        if (err) *err = "not an OFF file: " + path;
        return product;
    }
    std::string counts = line.substr(3);
    if (counts.find_first_not_of(" \t") == std::string::npos && !content_line(counts)) {
        if (err) *err = "no counts in " + path;
        return product;
    }
    long nv = 0, nf = 0, ne = 0;
    {
        std::istringstream cs(counts);
        cs >> nv >> nf >> ne;
    }
    if (nv <= 0 || nf < 0) { if (err) *err = "bad counts in " + path; return product; }
    product->verts.reserve((size_t)nv);
    for (long i = 0; i < nv; ++i) {
        if (!content_line(line)) { if (err) *err = "truncated vertex list in " + path; return product; }
        const char *p = line.c_str();
        char *e = nullptr;
        vec4 v;
        v.x = std::strtof(p, &e); p = e;
        v.y = std::strtof(p, &e); p = e;
        v.z = std::strtof(p, &e);
        product->verts.push_back(v);
    }
    product->inds.reserve((size_t)nf);
    for (long i = 0; i < nf; ++i) {
        if (!content_line(line)) { if (err) *err = "truncated face list in " + path; return product; }
        const char *p = line.c_str();
        char *e = nullptr;
        long k = std::strtol(p, &e, 10); p = e;
        if (k < 3) continue;
        long a = std::strtol(p, &e, 10); p = e;
        long b = std::strtol(p, &e, 10); p = e;
        long c = std::strtol(p, &e, 10);
        product->inds.push_back(ivec3((int)a, (int)b, (int)c));   // only the first three, like the PLY reader
    }
    return product;
}

Polyhedron *readMeshFile(const std::string &path, std::string *err)
{
    size_t dot = path.rfind('.');
    std::string ext = (dot == std::string::npos) ? std::string() : path.substr(dot);
    if (ext == ".off" || ext == ".OFF") return readOFFFile(path, err);
    return readPLYFile(path, err);
}

void normalizePoly(Polyhedron *poly)
{
    // PLYfileReader.cpp:144-171
    vec3 mn(100000000000.f), mx(-100000000000.f);
    for (const vec4 &g : poly->verts) {
        mn.x = std::fmin(mn.x, g.x); mn.y = std::fmin(mn.y, g.y); mn.z = std::fmin(mn.z, g.z);
        mx.x = std::fmax(mx.x, g.x); mx.y = std::fmax(mx.y, g.y); mx.z = std::fmax(mx.z, g.z);
    }
    vec3 mul(1.0f / (mx.x - mn.x), 1.0f / (mx.y - mn.y), 1.0f / (mx.z - mn.z));
    vec3 add(-mn.x * mul.x, -mn.y * mul.y, -mn.z * mul.z);
    for (vec4 &g : poly->verts) {
        for (int c = 0; c < 3; ++c) {
            float p = g[c] * mul[c];
            float v = p + add[c];               // glm::fma, unfused
            g[c] = clampf(v, 0.f, 1.f);
        }
        g.w = 0.f;                              // w*0 + 0
    }
}

} // namespace volly
