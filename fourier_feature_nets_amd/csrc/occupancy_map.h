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

// samples outside the box are kept (the reference evaluates them too)
__device__ __forceinline__ bool occupied_at(const GridMap& m, const uint32_t* __restrict__ bits,
                                            float x, float y, float z) {
    const float fx = (x - m.min0) * m.inv0, fy = (y - m.min1) * m.inv1, fz = (z - m.min2) * m.inv2;
    const float g = (float)m.G;
    if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < g && fy < g && fz < g)) return true;
    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
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
