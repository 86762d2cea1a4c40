// tricube_ref_export.cpp -- ORACLE / test infrastructure.
//
// Compiles the reference's own triangleCube.c from where it lies
// (-I/root/reference; it must be built as C++ because it includes <cmath>) and
// exports its entry point under a C name so tests can call it with ctypes.
// Nothing of the reference is copied into this repository; the resulting
// library goes to oracle/_ref/ (git-ignored).
#include "triangleCube.c"

extern "C" int tricube_ref(const float t[9])
{
    Triangle3 tri{Point3{t[0], t[1], t[2]}, Point3{t[3], t[4], t[5]}, Point3{t[6], t[7], t[8]}};
    return (int)t_c_intersection(tri);
}
