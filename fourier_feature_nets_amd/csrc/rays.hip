// Ray generation, AABB slab test, t-sampling, sample materialisation, image assembly.
// HBM-bound streaming kernels; compiled with -ffp-contract=off so that every multiply
// and add is rounded on its own, which is what makes ffn_sample_t / materialise
// bit-identical to the reference's op-by-op ATen sequence.
#include "common.h"

namespace ffn {

// ---------------------------------------------------------------------------------- K1
// One thread per (camera, pixel).  33 B written per ray, nothing but 76 B/camera read.
__global__ void __launch_bounds__(256)
raygen_nearfar_kernel(const float* __restrict__ unproj, const float* __restrict__ cam_pos,
                      const float* __restrict__ points, int num_cameras, int width, int height,
                      float3 lo, float3 hi,
                      float* __restrict__ starts, float* __restrict__ dirs,
                      float* __restrict__ near_far, uint8_t* __restrict__ valid) {
    const int64_t per_cam = (int64_t)width * height;
    const int64_t total = per_cam * num_cameras;
    for (int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ray < total;
         ray += (int64_t)gridDim.x * blockDim.x) {
        const int cam = (int)(ray / per_cam);
        const int pix = (int)(ray - cam * per_cam);
        // explicit (P,2) pixel coordinates (CameraInfo.raycast) or the integer grid
        const float px = points != nullptr ? points[2 * pix + 0] : (float)(pix % width);
        const float py = points != nullptr ? points[2 * pix + 1] : (float)(pix / width);
        const float* u = unproj + cam * 16;
        const float cx = cam_pos[cam * 3 + 0], cy = cam_pos[cam * 3 + 1], cz = cam_pos[cam * 3 + 2];
        // world = U @ [x, y, 1, 1]; k-ordered fused chain like a 4-deep SGEMM micro-kernel
        float w[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float acc = u[d * 4 + 0] * px;
            acc = __builtin_fmaf(u[d * 4 + 1], py, acc);
            acc = __builtin_fmaf(u[d * 4 + 2], 1.0f, acc);
            acc = __builtin_fmaf(u[d * 4 + 3], 1.0f, acc);
            w[d] = acc;
        }
        const float dx = w[0] - cx, dy = w[1] - cy, dz = w[2] - cz;
        const float norm = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float rx = dx / norm, ry = dy / norm, rz = dz / norm;
        const float sx = cx + 0.0f * rx, sy = cy + 0.0f * ry, sz = cz + 0.0f * rz;
        starts[ray * 3 + 0] = sx; starts[ray * 3 + 1] = sy; starts[ray * 3 + 2] = sz;
        dirs[ray * 3 + 0] = rx; dirs[ray * 3 + 1] = ry; dirs[ray * 3 + 2] = rz;
        // slab test; NaN/inf from a zero direction component are tolerated on purpose
        const float ax = (lo.x - sx) / rx, bx = (hi.x - sx) / rx;
        const float ay = (lo.y - sy) / ry, by = (hi.y - sy) / ry;
        const float az = (lo.z - sz) / rz, bz = (hi.z - sz) / rz;
        const float nx = ax < bx ? ax : bx, fx = ax > bx ? ax : bx;
        const float ny = ay < by ? ay : by, fy = ay > by ? ay : by;
        const float nz = az < bz ? az : bz, fz = az > bz ? az : bz;
        float near = np_max(np_max(nx, ny), nz);
        const float far = np_min(np_min(fx, fy), fz);
        const bool ok = near < far;
        if (ok) near = near > 0.1f ? near : 0.1f;
        near_far[ray] = near;
        near_far[total + ray] = far;
        valid[ray] = ok ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------- K2a
// One thread per (ray, sample).  Reads 8 B/ray (+4 B/sample of noise), writes 4 B/sample.
__global__ void __launch_bounds__(256)
sample_t_kernel(const float* __restrict__ near_far, int64_t total_rays,
                const int64_t* __restrict__ ray_index, int num_rays, int count,
                const float* __restrict__ unit, const float* __restrict__ noise, float anneal,
                float* __restrict__ t_out, int t_stride) {
    const int64_t n = (int64_t)num_rays * count;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / count);
        const int s = (int)(e - (int64_t)r * count);
        const int64_t ray = ray_index[r];
        float near = near_far[ray];
        float far = near_far[total_rays + ray];
        if (anneal >= 0.0f) {
            const float mid = (near + far) * 0.5f;
            near = mid + (near - mid) * anneal;
            far = mid + (far - mid) * anneal;
        }
        const float span = far - near;
        float t = near + unit[s] * span;
        if (noise != nullptr) {
            const float scale = span / (float)count;
            t = t + noise[e] * scale;
        }
        t_out[(int64_t)r * t_stride + s] = t;
    }
}

// ---------------------------------------------------------------------------------- K2b
// One thread per output float of the (R,S,3) arrays: fully coalesced dword stores.
__global__ void __launch_bounds__(256)
materialise_kernel(const float* __restrict__ starts, const float* __restrict__ dirs,
                   const int64_t* __restrict__ ray_index, const float* __restrict__ t_values,
                   int num_rays, int num_samples, float* __restrict__ positions,
                   float* __restrict__ views) {
    const int64_t n = (int64_t)num_rays * num_samples * 3;
    const int row = num_samples * 3;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / row);
        const int rem = (int)(e - (int64_t)r * row);
        const int s = rem / 3;
        const int d = rem - s * 3;
        const int64_t ray = ray_index[r];
        const float dir = dirs[ray * 3 + d];
        const float t = t_values[(int64_t)r * num_samples + s];
        positions[e] = starts[ray * 3 + d] + t * dir;
        if (views != nullptr) views[e] = dir;
    }
}

// ---------------------------------------------------------------------------------- K2a + K2b
// sample_t_kernel and materialise_kernel in ONE launch (samplers without an opacity model, where
// nothing merges into t between the two): t with the K2a operations, then the sample's position
// and view direction with the K2b operations -- the same bits as the two launches (this file is
// compiled without FMA contraction).
// A pure 28 S B / ray output stream, so the kernel is shaped by its STORES (round 6; one thread per
// (ray, sample) with three 4-byte stores at a 12-byte lane stride per array moved 2.6 TB/s): a
// workgroup owns 1024 consecutive samples = 4 KiB of t, 12 KiB of positions, 12 KiB of views, all
// three 16-byte aligned; a thread computes the t-values of ITS four samples (one float4 store) and
// leaves them with their ray ids in LDS; then every thread writes three float4 of positions and
// three of views -- lane i the bytes 16 i .. 16 i + 15 of a 4 KiB run, whole cache lines per wave --
// from the LDS copy and the (L1-resident) ray state.
constexpr int kSmChunk = 1024;

__global__ void __launch_bounds__(256)
sample_materialise_kernel(const float* __restrict__ near_far, int64_t total_rays,
                          const float* __restrict__ starts, const float* __restrict__ dirs,
                          const int64_t* __restrict__ ray_index, int num_rays, int count,
                          const float* __restrict__ unit, const float* __restrict__ noise, float anneal,
                          float* __restrict__ t_out, float* __restrict__ positions,
                          float* __restrict__ views) {
    __shared__ float t_s[kSmChunk];
    __shared__ int64_t ray_s[kSmChunk];
    const int64_t n = (int64_t)num_rays * count;
    const int64_t e0 = (int64_t)blockIdx.x * kSmChunk;
    const int live = (int)(n - e0 < kSmChunk ? n - e0 : kSmChunk);       // samples of this chunk
    const int64_t r0 = e0 / count;                                       // (uniform)
    const unsigned rem0 = (unsigned)(e0 - r0 * count);
    float4 tq;
    float* tv = reinterpret_cast<float*>(&tq);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int li = 4 * (int)threadIdx.x + j;
        float t = 0.0f;
        int64_t ray = 0;
        if (li < live) {
            const unsigned q = (rem0 + (unsigned)li) / (unsigned)count;
            const int s = (int)(rem0 + (unsigned)li - q * (unsigned)count);
            ray = ray_index[r0 + q];
            float near = near_far[ray];
            float far = near_far[total_rays + ray];
            if (anneal >= 0.0f) {
                const float mid = (near + far) * 0.5f;
                near = mid + (near - mid) * anneal;
                far = mid + (far - mid) * anneal;
            }
            const float span = far - near;
            t = near + unit[s] * span;
            if (noise != nullptr) {
                const float scale = span / (float)count;
                t = t + noise[e0 + li] * scale;
            }
        }
        tv[j] = t;
        t_s[li] = t;
        ray_s[li] = ray;
    }
    if (4 * (int)threadIdx.x + 3 < live) {
        reinterpret_cast<float4*>(t_out + e0)[threadIdx.x] = tq;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * (int)threadIdx.x + j < live) t_out[e0 + 4 * threadIdx.x + j] = tv[j];
    }
    __syncthreads();
    float* pos_chunk = positions + e0 * 3;
    float* view_chunk = views == nullptr ? nullptr : views + e0 * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int j0 = 4 * (q * 256 + (int)threadIdx.x);         // first float of this thread's float4
        if (j0 >= 3 * live) continue;
        float4 pq, vq;
        float* pv = reinterpret_cast<float*>(&pq);
        float* vv = reinterpret_cast<float*>(&vq);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int el = j0 + i;
            const int smp = el < 3 * live ? el / 3 : live - 1;
            const int d = el - 3 * (el / 3);
            const int64_t ray = ray_s[smp];
            const float dir = dirs[ray * 3 + d];
            pv[i] = starts[ray * 3 + d] + t_s[smp] * dir;
            vv[i] = dir;
        }
        if (j0 + 3 < 3 * live) {
            reinterpret_cast<float4*>(pos_chunk)[q * 256 + threadIdx.x] = pq;
            if (view_chunk != nullptr) reinterpret_cast<float4*>(view_chunk)[q * 256 + threadIdx.x] = vq;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (j0 + i < 3 * live) {
                    pos_chunk[j0 + i] = pv[i];
                    if (view_chunk != nullptr) view_chunk[j0 + i] = vv[i];
                }
        }
    }
}

// ---------------------------------------------------------------------------------- K8
__global__ void __launch_bounds__(256)
to_image_kernel(const float* __restrict__ colors, const int64_t* __restrict__ pixel_index,
                int64_t n, uint8_t* __restrict__ image) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * 3;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / 3;
        const int c = (int)(e - i * 3);
        const float v = colors[e] * 255.0f;
        image[pixel_index[i] * 3 + c] = (uint8_t)(int)v;  // truncation, no clip / round
    }
}

// ---------------------------------------------------------------------------------- K8b
// 8-bit YCrCb -> RGB in place, one thread per pixel (ray_sampler.py:197-198, ray_dataset.py:180-181
// call cv2.cvtColor(pixels, COLOR_YCrCb2RGB) on the truncated u8 frame).  OpenCV's 8-bit path is
// 14-bit fixed point: R = Y + round(1.403 (Cr - 128)), G = Y + round(-0.714 (Cr - 128) - 0.344
// (Cb - 128)), B = Y + round(1.773 (Cb - 128)) with the coefficients scaled by 2^14 and rounded
// (22987, -11698, -5636, 29049), "round" = add 2^13, arithmetic shift right by 14, and a
// saturating cast to u8.  cv2 is not in this image: restated from OpenCV's documented
// constants, parity unpinned (like the Dilate-mode ellipse).
__device__ __forceinline__ int descale14(int v) { return (v + (1 << 13)) >> 14; }
__device__ __forceinline__ uint8_t saturate_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ void __launch_bounds__(256)
ycrcb_to_rgb_kernel(uint8_t* __restrict__ image, int64_t pixels) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pixels;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int y = image[i * 3 + 0];
        const int cr = (int)image[i * 3 + 1] - 128;
        const int cb = (int)image[i * 3 + 2] - 128;
        image[i * 3 + 0] = saturate_u8(y + descale14(cr * 22987));
        image[i * 3 + 1] = saturate_u8(y + descale14(cb * -5636 + cr * -11698));
        image[i * 3 + 2] = saturate_u8(y + descale14(cb * 29049));
    }
}

static inline int grid_for(int64_t n, int block = 256, int cap = 256 * 8) {
    int64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace ffn

using namespace ffn;

extern "C" int ffn_raygen_nearfar(const float* unproj, const float* cam_pos,
                                  const float* points, int num_cameras, int width, int height,
                                  const float* box_lo,
                                  const float* box_hi, float* starts, float* directions,
                                  float* near_far, uint8_t* valid, void* stream) {
    if (num_cameras <= 0 || width <= 0 || height <= 0) return fail_arg("ffn_raygen_nearfar: empty");
    const int64_t total = (int64_t)num_cameras * width * height;
    const float3 lo = make_float3(box_lo[0], box_lo[1], box_lo[2]);
    const float3 hi = make_float3(box_hi[0], box_hi[1], box_hi[2]);
    hipLaunchKernelGGL(raygen_nearfar_kernel, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, unproj, cam_pos, points, num_cameras, width, height, lo, hi,
                       starts, directions, near_far, valid);
    return check_launch("ffn_raygen_nearfar");
}

extern "C" int ffn_sample_t(const float* near_far, int64_t num_rays_total,
                            const int64_t* ray_index, int num_rays, int count,
                            const float* unit, const float* noise, float anneal, float* t_out,
                            int t_stride, void* stream) {
    if (num_rays == 0) return 0;
    if (num_rays < 0 || count <= 0 || t_stride < count) return fail_arg("ffn_sample_t: shape");
    hipLaunchKernelGGL(sample_t_kernel, dim3(grid_for((int64_t)num_rays * count)), dim3(256), 0,
                       (hipStream_t)stream, near_far, num_rays_total, ray_index, num_rays, count,
                       unit, noise, anneal, t_out, t_stride);
    return check_launch("ffn_sample_t");
}

extern "C" int ffn_sample_materialise(const float* near_far, int64_t num_rays_total, const float* starts,
                                      const float* directions, const int64_t* ray_index, int num_rays,
                                      int count, const float* unit, const float* noise, float anneal,
                                      float* t_out, float* positions, float* views, void* stream) {
    if (num_rays == 0) return 0;
    if (num_rays < 0 || count <= 0) return fail_arg("ffn_sample_materialise: shape");
    // (the kernel stores whole float4: every chunk of 1024 samples starts 16-byte aligned in all three outputs)
    if ((((uintptr_t)t_out) | ((uintptr_t)positions) | ((uintptr_t)views)) & 15)
        return fail_arg("ffn_sample_materialise: t_out, positions and views must be 16-byte aligned");
    const int64_t chunks = ((int64_t)num_rays * count + kSmChunk - 1) / kSmChunk;
    hipLaunchKernelGGL(sample_materialise_kernel, dim3((unsigned)chunks), dim3(256), 0,
                       (hipStream_t)stream, near_far, num_rays_total, starts, directions, ray_index,
                       num_rays, count, unit, noise, anneal, t_out, positions, views);
    return check_launch("ffn_sample_materialise");
}

extern "C" int ffn_materialise_samples(const float* starts, const float* directions,
                                       const int64_t* ray_index, const float* t_values,
                                       int num_rays, int num_samples, float* positions,
                                       float* views, void* stream) {
    if (num_rays == 0) return 0;
    if (num_rays < 0 || num_samples <= 0) return fail_arg("ffn_materialise_samples: shape");
    hipLaunchKernelGGL(materialise_kernel, dim3(grid_for((int64_t)num_rays * num_samples * 3)),
                       dim3(256), 0, (hipStream_t)stream, starts, directions, ray_index, t_values,
                       num_rays, num_samples, positions, views);
    return check_launch("ffn_materialise_samples");
}

extern "C" int ffn_to_image(const float* colors, const int64_t* pixel_index, int64_t n,
                            int width, int height, uint8_t* image, void* stream) {
    hipError_t err = hipMemsetAsync(image, 0, (size_t)width * height * 3, (hipStream_t)stream);
    if (err != hipSuccess) { set_error("ffn_to_image: memset", err); return (int)err; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(to_image_kernel, dim3(grid_for(n * 3)), dim3(256), 0, (hipStream_t)stream,
                       colors, pixel_index, n, image);
    return check_launch("ffn_to_image");
}

extern "C" int ffn_ycrcb_to_rgb_u8(uint8_t* image, int64_t pixels, void* stream) {
    if (pixels == 0) return 0;
    if (pixels < 0 || image == nullptr) return fail_arg("ffn_ycrcb_to_rgb_u8: shape");
    hipLaunchKernelGGL(ycrcb_to_rgb_kernel, dim3(grid_for(pixels)), dim3(256), 0,
                       (hipStream_t)stream, image, pixels);
    return check_launch("ffn_ycrcb_to_rgb_u8");
}
