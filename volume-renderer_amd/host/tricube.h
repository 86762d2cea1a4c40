// tricube.h -- triangle / unit-cube overlap test used by the host voxelizer.
//
// Restates the behaviour of the reference's triangleCube.c (Voorhies,
// Graphics Gems III, as modified there): t_c_intersection (triangleCube.c:213-319)
// and its helpers.  Returns true when the triangle touches the unit cube centred
// on the origin (the reference's INSIDE == 0).
#ifndef VOLLY_TRICUBE_H
#define VOLLY_TRICUBE_H

namespace volly {

struct P3 { float x, y, z; };

bool triangle_touches_unit_cube(P3 v1, P3 v2, P3 v3);

} // namespace volly
#endif
