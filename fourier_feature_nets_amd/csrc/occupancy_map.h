// Occupancy-grid addressing shared by the K9 kernels (occupancy.hip) and the fused render
// kernel (mlp.hip): G^3 bits over a box, cell (ix,iy,iz) = bit ((iz*G + iy)*G + ix).
#pragma once
#include "common.h"

namespace ffn {

struct GridMap {
    float min0, min1, min2;       // bounding-box corner
    float inv0, inv1, inv2;       // cells per world unit
    int G;
};

// A sample outside the box takes the occupancy of the nearest cell (clamped indices): rays are
// sampled between their entry and exit points of this same box, so "outside" only happens by
// rounding on a face -- and a rule that kept such samples would make every ray evaluate its two
// end points (one whole 32-sample block per ray in the fused render kernel).  NaN positions
// (rays that miss the volume) count as occupied: they are filtered by ray id, not by value.
__device__ __forceinline__ bool occupied_at(const GridMap& m, const uint32_t* __restrict__ bits,
                                            float x, float y, float z) {
    const float fx = (x - m.min0) * m.inv0, fy = (y - m.min1) * m.inv1, fz = (z - m.min2) * m.inv2;
    if (!(fx == fx && fy == fy && fz == fz)) return true;
    const float top = (float)(m.G - 1);
    const int ix = (int)fminf(fmaxf(fx, 0.0f), top), iy = (int)fminf(fmaxf(fy, 0.0f), top),
              iz = (int)fminf(fmaxf(fz, 0.0f), top);
    const int64_t cell = ((int64_t)iz * m.G + iy) * m.G + ix;
    return (bits[cell >> 5] >> (cell & 31)) & 1u;
}

static inline GridMap make_map(const float* box_min, const float* box_size, int G) {
    GridMap m;
    m.min0 = box_min[0]; m.min1 = box_min[1]; m.min2 = box_min[2];
    m.inv0 = (float)G / box_size[0]; m.inv1 = (float)G / box_size[1]; m.inv2 = (float)G / box_size[2];
    m.G = G;
    return m;
}

}  // namespace ffn
