// Per-ray pieces of opacity-guided ("focus") sampling, shared by the standalone kernels
// (focus.hip) and the fused coarse-pass kernel (mlp.hip): blend-weight CDF of a probe
// (ray_sampler.py:59-67) and inverse-transform sampling + merge + sort (ray_sampler.py:301-357,
// :388-392).  Every function body switches floating-point contraction OFF: each multiply and add
// rounds separately like the reference's op-by-op ATen sequence, whatever the including
// translation unit is compiled with -- the two users must produce identical bits.
#pragma once
#include "common.h"

namespace ffn {

__device__ __forceinline__ float scan_mul(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float up = __shfl_up(v, off, 64);
        if (lane >= off) v *= up;
    }
    return v;
}
__device__ __forceinline__ float scan_add(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float up = __shfl_up(v, off, 64);
        if (lane >= off) v += up;
    }
    return v;
}

// sigma = F.softplus(logit), beta = 1, threshold = 20 (ray_sampler.py:261-265)
__device__ __forceinline__ float softplus_probe(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// cdf = [0, cumsum(w[1:-1] + 1e-5) / sum], w = blend weights of a probe of n samples held as
// ROWS x 64 lanes (sample s on lane s & 63 of row s >> 6): sigma[row] and delta[row] =
// t[s+1] - t[s] (anything for s >= n-1: the last delta is 1e10).  Writes n-1 floats to `out`
// (LDS or global).
template <int ROWS>
__device__ __forceinline__ void cdf_of_probe(const float (&sigma)[ROWS], const float (&delta_in)[ROWS],
                                             int n, int lane, float* out) {
#pragma clang fp contract(off)
    float w[ROWS];
    float carry = 1.0f;
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
        const int s = row * 64 + lane;
        float alpha = 0.0f, tau = 1.0f;
        if (s < n) {
            const float delta = (s == n - 1) ? 1e10f : delta_in[row];
            alpha = 1.0f - expf(-(sigma[row] * delta));
            const float u = (1.0f - alpha) + 1e-10f;
            tau = u < 1.0f ? u : 1.0f;
        }
        const float incl = scan_mul(tau, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        w[row] = alpha * (carry * excl);
        carry *= __shfl(incl, 63, 64);
    }
    // interior weights + 1e-5, running sum
    float run[ROWS];
    float base = 0.0f;
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
        const int s = row * 64 + lane;
        const float v = (s >= 1 && s <= n - 2) ? w[row] + 1e-5f : 0.0f;
        const float incl = scan_add(v, lane);
        run[row] = base + incl;
        base += __shfl(incl, 63, 64);
    }
    const float total = base;
    if (lane == 0) out[0] = 0.0f;
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
        const int s = row * 64 + lane;
        if (s >= 1 && s <= n - 2) out[s] = run[row] / total;
    }
}

// Inverse-transform sampling of n_focus values from the CDF `c` (LDS, n_focus-1 entries), merge
// with the S - n_focus uniform samples already in `tv` (LDS), rank sort, write the S sorted
// values to `row` (global).  `u_row`: the ray's n_focus uniforms; `unit_focus` =
// linspace(0,1,n_focus).  The caller has made the LDS contents visible to the wave.
__device__ __forceinline__ void focus_merge_ray(float near, float span, const float* c, float* tv,
                                                const float* __restrict__ u_row,
                                                const float* __restrict__ unit_focus, int S,
                                                int n_focus, int lane, float* __restrict__ row) {
#pragma clang fp contract(off)
    const int width = n_focus - 1;
    const int n_uniform = S - n_focus;
    for (int i = lane; i < n_focus; i += 64) {
        const float uu = u_row[i];
        // searchsorted(right=True): number of cdf entries <= u (cdf is ascending)
        int lo_b = 0, hi_b = width;
        while (lo_b < hi_b) {
            const int mid = (lo_b + hi_b) >> 1;
            if (c[mid] <= uu) lo_b = mid + 1; else hi_b = mid;
        }
        const int k = lo_b;
        const int lo = k - 1 > 0 ? k - 1 : 0;
        const int hi = k < width - 1 ? k : width - 1;
        const float c_lo = c[lo], c_hi = c[hi];
        // bin centres of linspace(near, far, n_focus)
        const float g_lo0 = near + unit_focus[lo] * span, g_lo1 = near + unit_focus[lo + 1] * span;
        const float g_hi0 = near + unit_focus[hi] * span, g_hi1 = near + unit_focus[hi + 1] * span;
        const float t_lo = 0.5f * (g_lo0 + g_lo1);
        const float t_hi = 0.5f * (g_hi0 + g_hi1);
        float denom = c_hi - c_lo;
        if (denom < 1e-5f) denom = 1.0f;
        const float frac = (uu - c_lo) / denom;
        tv[n_uniform + i] = t_lo + frac * (t_hi - t_lo);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // rank sort: position = #smaller + #equal-with-lower-index
    for (int i = lane; i < S; i += 64) {
        const float v = tv[i];
        int rank = 0;
        for (int j = 0; j < S; ++j) {
            const float o = tv[j];
            rank += (o < v || (o == v && j < i)) ? 1 : 0;
        }
        row[rank] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}

}  // namespace ffn
