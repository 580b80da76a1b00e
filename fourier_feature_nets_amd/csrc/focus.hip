// Opacity-guided ("focus") sampling: per-ray CDF build, inverse-transform sampling, and the
// merge + sort with the uniform half.  One wavefront per ray; the per-ray arithmetic lives in
// focus_terms.h with floating-point contraction switched off per function (the lerp arithmetic
// is bit-identical to the reference's op sequence, the CDF itself differs from torch.cumsum
// only by summation order).
#include "common.h"
#include "focus_terms.h"

namespace ffn {

// ---------------------------------------------------------------------------------- K2c
// LOGITS: `opacity` holds the coarse model's raw (P,n,4) outputs; sigma = softplus of the last
// channel is taken here (ray_sampler.py:261-265) instead of in a separate pass.
template <int ROWS, bool LOGITS>
__global__ void __launch_bounds__(256)
cdf_build_kernel(const float* __restrict__ t_probe, const float* __restrict__ opacity,
                 int64_t num_rays, int n, float* __restrict__ cdf) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t ray = wave; ray < num_rays; ray += waves) {
        const float* tr = t_probe + ray * n;
        const float* op = opacity + ray * n * (LOGITS ? 4 : 1);
        float sigma[ROWS], delta[ROWS];
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
            const int s = row * 64 + lane;
            sigma[row] = 0.0f;
            delta[row] = 0.0f;
            if (s < n) {
                sigma[row] = LOGITS ? softplus_probe(op[4 * s + 3]) : op[s];
                if (s < n - 1) delta[row] = tr[s + 1] - tr[s];
            }
        }
        cdf_of_probe<ROWS>(sigma, delta, n, lane, cdf + ray * (n - 1));
    }
}

// ---------------------------------------------------------------------------------- K2d
// LDS per wave: cdf row (<=255), merged t row (<=256).
// ROWS_LOCAL: `cdfs` holds one row per BATCH ray (built on the fly from a live coarse model)
// instead of the per-sampler table indexed by global ray id.
template <bool ROWS_LOCAL>
__global__ void __launch_bounds__(256)
focus_merge_kernel(const float* __restrict__ near_far, int64_t total_rays,
                   const float* __restrict__ cdfs, const int64_t* __restrict__ ray_index,
                   const float* __restrict__ u, const float* __restrict__ unit_focus, int R, int S,
                   int n_focus, float* __restrict__ t_io) {
    __shared__ float lds_cdf[4][256];
    __shared__ float lds_t[4][256];
    const int lane = lane_id();
    const int wslot = threadIdx.x >> 6;
    float* c = lds_cdf[wslot];
    float* tv = lds_t[wslot];
    const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
    const int width = n_focus - 1;
    const int n_uniform = S - n_focus;
    for (int r = wave; r < R; r += waves) {
        const int64_t ray = ray_index[r];
        const float near = near_far[ray];
        const float far = near_far[total_rays + ray];
        const float span = far - near;
        const int64_t cdf_row = ROWS_LOCAL ? (int64_t)r : ray;
        for (int i = lane; i < width; i += 64) c[i] = cdfs[cdf_row * width + i];
        float* row = t_io + (int64_t)r * S;
        for (int i = lane; i < n_uniform; i += 64) tv[i] = row[i];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave are done
        __builtin_amdgcn_wave_barrier();
        focus_merge_ray(near, span, c, tv, u + (int64_t)r * n_focus, unit_focus, S, n_focus, lane, row);
    }
}

}  // namespace ffn

using namespace ffn;

template <bool LOGITS>
static int launch_cdf_build(const float* t_probe, const float* opacity, int64_t num_rays, int n,
                            float* cdf, void* stream, const char* what) {
    if (num_rays == 0) return 0;
    if (n < 3 || n > 256) return fail_arg("ffn_cdf_build: probe length must be in [3,256]");
    int64_t blocks = (num_rays + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    const dim3 grid((int)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int rows = (n + 63) / 64;
    switch (rows) {
        case 1: hipLaunchKernelGGL((cdf_build_kernel<1, LOGITS>), grid, block, 0, st, t_probe, opacity, num_rays, n, cdf); break;
        case 2: hipLaunchKernelGGL((cdf_build_kernel<2, LOGITS>), grid, block, 0, st, t_probe, opacity, num_rays, n, cdf); break;
        default: hipLaunchKernelGGL((cdf_build_kernel<4, LOGITS>), grid, block, 0, st, t_probe, opacity, num_rays, n, cdf); break;
    }
    return check_launch(what);
}

extern "C" int ffn_cdf_build(const float* t_probe, const float* opacity, int64_t num_rays, int n,
                             float* cdf, void* stream) {
    return launch_cdf_build<false>(t_probe, opacity, num_rays, n, cdf, stream, "ffn_cdf_build");
}

extern "C" int ffn_cdf_build_logits(const float* t_probe, const float* logits, int64_t num_rays,
                                    int n, float* cdf, void* stream) {
    return launch_cdf_build<true>(t_probe, logits, num_rays, n, cdf, stream, "ffn_cdf_build_logits");
}

static int launch_focus_merge(bool rows_local, const float* near_far, int64_t num_rays_total,
                              const float* cdfs, const int64_t* ray_index, const float* u,
                              const float* unit_focus, int num_rays, int num_samples, int n_focus,
                              float* t_io, void* stream) {
    if (num_rays == 0) return 0;
    if (num_samples > 256 || n_focus < 2 || n_focus > num_samples)
        return fail_arg("ffn_focus_sample_merge: need 2 <= n_focus <= S <= 256");
    int64_t blocks = ((int64_t)num_rays + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (rows_local)
        hipLaunchKernelGGL(focus_merge_kernel<true>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                           near_far, num_rays_total, cdfs, ray_index, u, unit_focus, num_rays,
                           num_samples, n_focus, t_io);
    else
        hipLaunchKernelGGL(focus_merge_kernel<false>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                           near_far, num_rays_total, cdfs, ray_index, u, unit_focus, num_rays,
                           num_samples, n_focus, t_io);
    return check_launch("ffn_focus_sample_merge");
}

extern "C" int ffn_focus_sample_merge(const float* near_far, int64_t num_rays_total,
                                      const float* cdfs, const int64_t* ray_index, const float* u,
                                      const float* unit_focus, int num_rays, int num_samples,
                                      int n_focus, float* t_io, void* stream) {
    return launch_focus_merge(false, near_far, num_rays_total, cdfs, ray_index, u, unit_focus,
                              num_rays, num_samples, n_focus, t_io, stream);
}

extern "C" int ffn_focus_sample_merge_rows(const float* near_far, int64_t num_rays_total,
                                           const float* cdf_rows, const int64_t* ray_index,
                                           const float* u, const float* unit_focus, int num_rays,
                                           int num_samples, int n_focus, float* t_io,
                                           void* stream) {
    return launch_focus_merge(true, near_far, num_rays_total, cdf_rows, ray_index, u, unit_focus,
                              num_rays, num_samples, n_focus, t_io, stream);
}
