// Empty-space skipping for inference (the reference has only a CPU octree ray
// walker, octree.py:418-501, never used by its renderer -- this is new behaviour, opt-in, with
// PSNR-level parity).  An occupancy grid is a G^3 bit mask over the sampler's bounding box:
// bit set = the density model reported sigma above a threshold at (or next to) that cell.
// Rendering then evaluates the MLP only on the samples that fall into occupied cells:
//
//   K9a occupancy_build      sigma logits of the G^3 cell centres -> bits
//   K9b occupancy_dilate     26-neighbourhood dilation (a cell centre can miss a thin surface)
//   K9c occupancy_count      per 256-sample block: how many samples are occupied
//   K9d occupancy_scan       exclusive scan of the block counts (one workgroup), total
//   K9e occupancy_compact    positions / views of the occupied samples, packed, + source index
//   K9f scatter_logits       packed logits back to (N,4); skipped samples get sigma logit -100
//
// All HBM-streaming, one thread per sample (ballot + popcount for the in-block ranks).
#include "common.h"
#include "occupancy_map.h"

namespace ffn {

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// ---------------------------------------------------------------------------------- K9a
// one thread per 32 cells (one output word): deterministic, no atomics
__global__ void __launch_bounds__(256)
occupancy_build_kernel(const float4* __restrict__ logits, int64_t cells, float threshold,
                       uint32_t* __restrict__ bits) {
    const int64_t words = (cells + 31) >> 5;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words;
         w += (int64_t)gridDim.x * blockDim.x) {
        uint32_t word = 0u;
        for (int b = 0; b < 32; ++b) {
            const int64_t c = w * 32 + b;
            if (c < cells && softplus_f(logits[c].w) > threshold) word |= 1u << b;
        }
        bits[w] = word;
    }
}

// ---------------------------------------------------------------------------------- K9b
__global__ void __launch_bounds__(256)
occupancy_dilate_kernel(const uint32_t* __restrict__ src, int G, uint32_t* __restrict__ dst) {
    const int64_t cells = (int64_t)G * G * G;
    const int64_t words = (cells + 31) >> 5;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words;
         w += (int64_t)gridDim.x * blockDim.x) {
        uint32_t word = 0u;
        for (int b = 0; b < 32; ++b) {
            const int64_t c = w * 32 + b;
            if (c >= cells) break;
            const int ix = (int)(c % G), iy = (int)((c / G) % G), iz = (int)(c / ((int64_t)G * G));
            bool any = false;
            for (int dz = -1; dz <= 1 && !any; ++dz)
                for (int dy = -1; dy <= 1 && !any; ++dy)
                    for (int dx = -1; dx <= 1 && !any; ++dx) {
                        const int x = ix + dx, y = iy + dy, z = iz + dz;
                        if (x < 0 || y < 0 || z < 0 || x >= G || y >= G || z >= G) continue;
                        const int64_t n = ((int64_t)z * G + y) * G + x;
                        any = (src[n >> 5] >> (n & 31)) & 1u;
                    }
            if (any) word |= 1u << b;
        }
        dst[w] = word;
    }
}

// ---------------------------------------------------------------------------------- K9c
__global__ void __launch_bounds__(256)
occupancy_count_kernel(const float* __restrict__ positions, int64_t n, GridMap map,
                       const uint32_t* __restrict__ bits, int32_t* __restrict__ block_counts) {
    __shared__ int wave_counts[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool occ = false;
    if (i < n) occ = occupied_at(map, bits, positions[i * 3 + 0], positions[i * 3 + 1], positions[i * 3 + 2]);
    const uint64_t ballot = __ballot(occ);
    if ((threadIdx.x & 63) == 0) wave_counts[threadIdx.x >> 6] = __popcll(ballot);
    __syncthreads();
    if (threadIdx.x == 0)
        block_counts[blockIdx.x] = (wave_counts[0] + wave_counts[1]) + (wave_counts[2] + wave_counts[3]);
}

// ---------------------------------------------------------------------------------- K9d
// exclusive scan of `blocks` counts in place (single workgroup of 1024 threads), total -> *total
__global__ void __launch_bounds__(1024)
occupancy_scan_kernel(int32_t* __restrict__ counts, int blocks, int64_t* __restrict__ total) {
    __shared__ int64_t partial[1024];
    const int per = (blocks + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, blocks);
    int64_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[i];
    partial[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele over 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t add = threadIdx.x >= off ? partial[threadIdx.x - off] : 0;
        __syncthreads();
        partial[threadIdx.x] += add;
        __syncthreads();
    }
    int64_t run = threadIdx.x == 0 ? 0 : partial[threadIdx.x - 1];
    for (int i = lo; i < hi; ++i) {
        const int c = counts[i];
        counts[i] = (int32_t)run;
        run += c;
    }
    if (threadIdx.x == 1023) *total = partial[1023];
}

// ---------------------------------------------------------------------------------- K9e
__global__ void __launch_bounds__(256)
occupancy_compact_kernel(const float* __restrict__ positions, const float* __restrict__ views,
                         int64_t n, GridMap map, const uint32_t* __restrict__ bits,
                         const int32_t* __restrict__ block_offsets, float* __restrict__ out_pos,
                         float* __restrict__ out_view, int32_t* __restrict__ out_index) {
    __shared__ int wave_counts[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float x = 0.f, y = 0.f, z = 0.f;
    bool occ = false;
    if (i < n) {
        x = positions[i * 3 + 0]; y = positions[i * 3 + 1]; z = positions[i * 3 + 2];
        occ = occupied_at(map, bits, x, y, z);
    }
    const uint64_t ballot = __ballot(occ);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_counts[wave] = __popcll(ballot);
    __syncthreads();
    if (!occ) return;
    int rank = __popcll(ballot & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rank += wave_counts[w];
    const int64_t dst = (int64_t)block_offsets[blockIdx.x] + rank;
    out_pos[dst * 3 + 0] = x; out_pos[dst * 3 + 1] = y; out_pos[dst * 3 + 2] = z;
    if (views != nullptr) {
        out_view[dst * 3 + 0] = views[i * 3 + 0];
        out_view[dst * 3 + 1] = views[i * 3 + 1];
        out_view[dst * 3 + 2] = views[i * 3 + 2];
    }
    out_index[dst] = (int32_t)i;
}

// ---------------------------------------------------------------------------------- K9f
__global__ void __launch_bounds__(256)
fill_logits_kernel(float4* __restrict__ out, int64_t n, float4 fill) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = fill;
}
__global__ void __launch_bounds__(256)
scatter_logits_kernel(const float4* __restrict__ packed, const int32_t* __restrict__ index,
                      int64_t m, float4* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (int64_t)gridDim.x * blockDim.x)
        out[index[i]] = packed[i];
}

// K9g: the inverse of the scatter for the backward pass -- d_logits rows of the evaluated samples
__global__ void __launch_bounds__(256)
gather_logits_kernel(const float4* __restrict__ full, const int32_t* __restrict__ index, int64_t m,
                     float4* __restrict__ packed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (int64_t)gridDim.x * blockDim.x)
        packed[i] = full[index[i]];
}

}  // namespace ffn

using namespace ffn;

static inline int stream_grid(int64_t work, int per_block = 256) {
    int64_t g = (work + per_block - 1) / per_block;
    if (g > 8192) g = 8192;
    return (int)(g < 1 ? 1 : g);
}

extern "C" int ffn_occupancy_build(const float* logits, int resolution, float sigma_threshold,
                                   int dilate, uint32_t* scratch_bits, uint32_t* bits, void* stream) {
    if (resolution < 1 || resolution > 1024) return fail_arg("ffn_occupancy_build: resolution");
    const int64_t cells = (int64_t)resolution * resolution * resolution;
    const int64_t words = (cells + 31) >> 5;
    hipStream_t st = (hipStream_t)stream;
    uint32_t* first = dilate ? scratch_bits : bits;
    if (dilate && scratch_bits == nullptr) return fail_arg("ffn_occupancy_build: dilation needs scratch");
    hipLaunchKernelGGL(occupancy_build_kernel, dim3(stream_grid(words)), dim3(256), 0, st,
                       (const float4*)logits, cells, sigma_threshold, first);
    if (dilate)
        hipLaunchKernelGGL(occupancy_dilate_kernel, dim3(stream_grid(words)), dim3(256), 0, st,
                           scratch_bits, resolution, bits);
    return check_launch("ffn_occupancy_build");
}

extern "C" int ffn_occupancy_count(const float* positions, int64_t n, const float* box_min,
                                   const float* box_size, int resolution, const uint32_t* bits,
                                   int32_t* block_offsets, int64_t* total, void* stream) {
    if (n <= 0 || resolution < 1) return fail_arg("ffn_occupancy_count: shape");
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffff) return fail_arg("ffn_occupancy_count: too many samples");
    hipStream_t st = (hipStream_t)stream;
    const GridMap map = make_map(box_min, box_size, resolution);
    hipLaunchKernelGGL(occupancy_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, positions, n,
                       map, bits, block_offsets);
    hipLaunchKernelGGL(occupancy_scan_kernel, dim3(1), dim3(1024), 0, st, block_offsets, (int)blocks,
                       total);
    return check_launch("ffn_occupancy_count");
}

extern "C" int ffn_occupancy_compact(const float* positions, const float* views, int64_t n,
                                     const float* box_min, const float* box_size, int resolution,
                                     const uint32_t* bits, const int32_t* block_offsets,
                                     float* out_positions, float* out_views, int32_t* out_index,
                                     void* stream) {
    if (n <= 0 || resolution < 1) return fail_arg("ffn_occupancy_compact: shape");
    if ((views == nullptr) != (out_views == nullptr)) return fail_arg("ffn_occupancy_compact: views");
    const int64_t blocks = (n + 255) / 256;
    const GridMap map = make_map(box_min, box_size, resolution);
    hipLaunchKernelGGL(occupancy_compact_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, positions, views, n, map, bits, block_offsets,
                       out_positions, out_views, out_index);
    return check_launch("ffn_occupancy_compact");
}

extern "C" int ffn_scatter_logits(const float* packed, const int32_t* index, int64_t m, int64_t n,
                                  float empty_sigma_logit, float* out, void* stream) {
    if (n <= 0 || m < 0 || m > n) return fail_arg("ffn_scatter_logits: shape");
    hipStream_t st = (hipStream_t)stream;
    const float4 fill = make_float4(0.0f, 0.0f, 0.0f, empty_sigma_logit);
    hipLaunchKernelGGL(fill_logits_kernel, dim3(stream_grid(n)), dim3(256), 0, st, (float4*)out, n, fill);
    if (m > 0)
        hipLaunchKernelGGL(scatter_logits_kernel, dim3(stream_grid(m)), dim3(256), 0, st,
                           (const float4*)packed, index, m, (float4*)out);
    return check_launch("ffn_scatter_logits");
}

extern "C" int ffn_gather_logits(const float* full, const int32_t* index, int64_t m, float* packed,
                                 void* stream) {
    if (m < 0) return fail_arg("ffn_gather_logits: shape");
    if (m == 0) return 0;
    hipLaunchKernelGGL(gather_logits_kernel, dim3(stream_grid(m)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)full, index, m, (float4*)packed);
    return check_launch("ffn_gather_logits");
}

// ---------------------------------------------------------------------------------- K10
// Dense voxel radiance field lookup (voxels_model.py:35-45 of the reference: grid_sample of a
// (1,4,S,S,S) volume, trilinear, padding_mode="border", align_corners=False, plus a bias).
// Used as the opacity model of the focus sampler.  A position p maps to the continuous voxel
// coordinate ((p/scale + 1) * S - 1) / 2 per axis, clamped to [0, S-1]; x indexes the last
// (fastest) volume axis.  One thread per sample, four channels each; HBM/L2-gather bound.
namespace ffn {
__global__ void __launch_bounds__(256)
voxels_forward_kernel(const float* __restrict__ volume, const float* __restrict__ bias,
                      const float* __restrict__ positions, int64_t n, int side, float inv_scale,
                      float4* __restrict__ out) {
    const int64_t plane = (int64_t)side * side, chan = plane * side;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int lo[3];
        float frac[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float c = ((positions[i * 3 + d] * inv_scale + 1.0f) * (float)side - 1.0f) * 0.5f;
            c = fminf(fmaxf(c, 0.0f), (float)(side - 1));          // border padding
            const float f = floorf(c);
            lo[d] = (int)f;
            frac[d] = c - f;
        }
        const int hi0 = min(lo[0] + 1, side - 1), hi1 = min(lo[1] + 1, side - 1), hi2 = min(lo[2] + 1, side - 1);
        float acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* v = volume + c * chan;
            auto at = [&](int z, int y, int x) { return v[(int64_t)z * plane + (int64_t)y * side + x]; };
            // same association as ATen's trilinear kernel: eight corner weights, summed
            const float wx1 = frac[0], wx0 = 1.0f - frac[0];
            const float wy1 = frac[1], wy0 = 1.0f - frac[1];
            const float wz1 = frac[2], wz0 = 1.0f - frac[2];
            float s = at(lo[2], lo[1], lo[0]) * (wx0 * wy0 * wz0);
            s += at(lo[2], lo[1], hi0) * (wx1 * wy0 * wz0);
            s += at(lo[2], hi1, lo[0]) * (wx0 * wy1 * wz0);
            s += at(lo[2], hi1, hi0) * (wx1 * wy1 * wz0);
            s += at(hi2, lo[1], lo[0]) * (wx0 * wy0 * wz1);
            s += at(hi2, lo[1], hi0) * (wx1 * wy0 * wz1);
            s += at(hi2, hi1, lo[0]) * (wx0 * wy1 * wz1);
            s += at(hi2, hi1, hi0) * (wx1 * wy1 * wz1);
            acc[c] = s + bias[c];
        }
        out[i] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}
}  // namespace ffn

extern "C" int ffn_voxels_forward(const float* volume, const float* bias, const float* positions,
                                  int64_t n, int side, float scale, float* out, void* stream) {
    if (n == 0) return 0;
    if (n < 0 || side < 1 || !(scale > 0.0f)) return fail_arg("ffn_voxels_forward: shape");
    hipLaunchKernelGGL(ffn::voxels_forward_kernel, dim3(stream_grid(n)), dim3(256), 0,
                       (hipStream_t)stream, volume, bias, positions, n, side, 1.0f / scale,
                       (float4*)out);
    return check_launch("ffn_voxels_forward");
}
