// Activations + front-to-back alpha compositing, forward and backward, and the MSE loss.
//
// One wavefront per ray.  Sample s of a ray lives on lane (s & 63) of "row" (s >> 6), so
// every load is a coalesced 1 KiB (float4 logits) or 256 B (t) wave access, and the
// exclusive transmittance product is a 6-step cross-lane scan per row with a scalar
// carry between rows.  HBM-bound: 20*S bytes read + 20 bytes written per ray forward.
#include "common.h"
#include "composite_terms.h"

namespace ffn {

// ---------------------------------------------------------------------------------- K5
__global__ void __launch_bounds__(256)
composite_fwd_kernel(const float4* __restrict__ logits, const float* __restrict__ t, int R, int S,
                     float* __restrict__ color, float* __restrict__ alpha_out,
                     float* __restrict__ depth, int32_t* nan_flag) {
    const int lane = lane_id();
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int waves = (int)((gridDim.x * (int64_t)blockDim.x) >> 6);
    const int rows = (S + 63) >> 6;
    for (int ray = wave; ray < R; ray += waves) {
        const float4* lg = logits + (int64_t)ray * S;
        const float* tr = t + (int64_t)ray * S;
        RayAccum acc;
        acc.reset();
        for (int row = 0; row < rows; ++row) {
            const int s = row * 64 + lane;
            const bool active = s < S;
            const SampleTerms q = load_terms(lg, tr, s, S, active, nan_flag);
            acc.row(q, lane, s, active && s < S - 1);
        }
        acc.finish();
        if (lane == 0) {
            color[ray * 3 + 0] = acc.cr; color[ray * 3 + 1] = acc.cg; color[ray * 3 + 2] = acc.cb;
            alpha_out[ray] = acc.asum;
            if (depth != nullptr) depth[ray] = tr[acc.depth_pick(S)];
        }
    }
}

// ---------------------------------------------------------------------------------- K5w
__global__ void __launch_bounds__(256)
blend_weights_kernel(const float* __restrict__ t, const float* __restrict__ sigma, int R, int S,
                     float* __restrict__ weights) {
    const int lane = lane_id();
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int waves = (int)((gridDim.x * (int64_t)blockDim.x) >> 6);
    const int rows = (S + 63) >> 6;
    for (int ray = wave; ray < R; ray += waves) {
        const float* tr = t + (int64_t)ray * S;
        const float* sg = sigma + (int64_t)ray * S;
        float carry = 1.0f;
        for (int row = 0; row < rows; ++row) {
            const int s = row * 64 + lane;
            float alpha = 0.0f, tau = 1.0f;
            if (s < S) {
                const float delta = (s == S - 1) ? 1e10f : tr[s + 1] - tr[s];
                alpha = 1.0f - expf(-(sg[s] * delta));
                const float u = (1.0f - alpha) + 1e-10f;
                tau = u < 1.0f ? u : 1.0f;
            }
            const float incl = wave_scan_mul(tau, lane);
            const float excl = wave_shift_up(incl, 1.0f);
            if (s < S) weights[(int64_t)ray * S + s] = alpha * (carry * excl);
            carry *= wave_last(incl);
        }
    }
}

// K5w backward: d(loss)/d(sigma) and d(loss)/d(t) from d(loss)/d(weights) -- the autograd of
// utils.py:72-97 (exp, minimum with its 1/2-1/2 tie rule, cumprod) in closed form:
//   dL/dalpha_s = g_s T_s - dtau/du * Q_s / tau_s,   Q_s = sum_{k>s} g_k w_k
//   dL/dsigma_s = dL/dalpha_s * e_s * delta_s,       dL/ddelta_s = dL/dalpha_s * e_s * sigma_s
//   dL/dt_s     = dL/ddelta_{s-1} - dL/ddelta_s      (delta_{S-1} = 1e10 is a constant)
template <int ROWS>
__global__ void __launch_bounds__(256)
blend_weights_bwd_kernel(const float* __restrict__ t, const float* __restrict__ sigma,
                         const float* __restrict__ d_weights, int R, int S,
                         float* __restrict__ d_sigma, float* __restrict__ d_t) {
    const int lane = lane_id();
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int waves = (int)((gridDim.x * (int64_t)blockDim.x) >> 6);
    for (int ray = wave; ray < R; ray += waves) {
        const float* tr = t + (int64_t)ray * S;
        const float* sg = sigma + (int64_t)ray * S;
        const float* gw = d_weights + (int64_t)ray * S;
        float sig[ROWS], delta[ROWS], e[ROWS], alpha[ROWS], u[ROWS], tau[ROWS], T[ROWS], dd[ROWS];
        float carry = 1.0f;
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
            const int s = row * 64 + lane;
            sig[row] = 0.0f; delta[row] = 0.0f; e[row] = 1.0f; alpha[row] = 0.0f; u[row] = 1.0f; tau[row] = 1.0f;
            if (s < S) {
                sig[row] = sg[s];
                delta[row] = (s == S - 1) ? 1e10f : tr[s + 1] - tr[s];
                e[row] = expf(-(sig[row] * delta[row]));
                alpha[row] = 1.0f - e[row];
                u[row] = (1.0f - alpha[row]) + 1e-10f;
                tau[row] = u[row] < 1.0f ? u[row] : 1.0f;
            }
            const float incl = wave_scan_mul(tau[row], lane);
            const float excl = wave_shift_up(incl, 1.0f);
            T[row] = carry * excl;
            carry *= wave_last(incl);
        }
        float tail = 0.0f;
#pragma unroll
        for (int row = ROWS - 1; row >= 0; --row) {
            const int s = row * 64 + lane;
            const bool active = s < S;
            const float g = active ? gw[s] : 0.0f;
            const float gwv = g * (alpha[row] * T[row]);
            const float incl = wave_suffix_add(gwv, lane);
            const float Q = (incl - gwv) + tail;
            tail += __shfl(incl, 0, 64);
            dd[row] = 0.0f;
            if (active) {
                const float dtau_du = u[row] < 1.0f ? 1.0f : (u[row] == 1.0f ? 0.5f : 0.0f);
                const float dL_dtau = (s < S - 1) ? Q / tau[row] : 0.0f;
                const float dL_dalpha = g * T[row] - dtau_du * dL_dtau;
                d_sigma[(int64_t)ray * S + s] = dL_dalpha * e[row] * delta[row];
                dd[row] = (s < S - 1) ? dL_dalpha * e[row] * sig[row] : 0.0f;
            }
        }
        if (d_t != nullptr) {
#pragma unroll
            for (int row = 0; row < ROWS; ++row) {
                const int s = row * 64 + lane;
                const float first = row > 0 ? wave_last(dd[row > 0 ? row - 1 : 0]) : 0.0f;
                const float prev = wave_shift_up(dd[row], first);
                if (s < S) d_t[(int64_t)ray * S + s] = prev - dd[row];
            }
        }
    }
}

// ---------------------------------------------------------------------------------- K5b
// One row of the backward walk (shared by K5b and the training composite K5t): d(loss)/d(logits)
// of the row's sample from the ray's d(loss)/d(colour), d(loss)/d(alpha); `tail` carries
// sum of g*w over all later rows.
__device__ __forceinline__ float4 composite_bwd_row(const SampleTerms& p, float T, float dcr, float dcg,
                                                    float dcb, float da, int s, int S, int lane,
                                                    float& tail) {
    // every product and sum rounded on its own (ATen's autograd evaluates them one op at a time;
    // and the two kernels that inline this row must not contract it differently)
#pragma clang fp contract(off)
    const bool active = s < S;
    const float w = p.alpha * T;
    const float g = active ? (dcr * p.r + dcg * p.g + dcb * p.b) + (s < S - 1 ? da : 0.0f)
                           : 0.0f;
    const float gw = g * w;
    const float incl = wave_suffix_add(gw, lane);
    const float Q = (incl - gw) + tail;          // strictly-after sum
    tail += __shfl(incl, 0, 64);
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        // tau = min(1, u): gradient 1 below the clamp, 1/2 on a tie, 0 above it
        const float dtau_du = p.u < 1.0f ? 1.0f : (p.u == 1.0f ? 0.5f : 0.0f);
        const float dL_dtau = (s < S - 1) ? Q / p.tau : 0.0f;
        const float dL_dalpha = g * T - dtau_du * dL_dtau;
        const float dL_dsigma = dL_dalpha * p.e * p.delta;
        const float x = p.sigma_logit;
        const float z = expf(x);
        const float dsig = x > 20.0f ? 1.0f : z / (z + 1.0f);   // softplus' as torch: z/(z+1)
        out.x = w * dcr * p.r * (1.0f - p.r);
        out.y = w * dcg * p.g * (1.0f - p.g);
        out.z = w * dcb * p.b * (1.0f - p.b);
        out.w = dL_dsigma * dsig;
    }
    return out;
}

// Recomputes the forward terms (cheaper than storing 12 B/sample) and walks the rows in
// reverse to form Q_s = sum_{j>s} g_j w_j, the quantity cumprod's backward needs.
template <int ROWS>
__global__ void __launch_bounds__(256)
composite_bwd_kernel(const float4* __restrict__ logits, const float* __restrict__ t,
                     const float* __restrict__ d_color, const float* __restrict__ d_alpha,
                     int R, int S, float4* __restrict__ d_logits) {
    const int lane = lane_id();
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int waves = (int)((gridDim.x * (int64_t)blockDim.x) >> 6);
    for (int ray = wave; ray < R; ray += waves) {
        const float4* lg = logits + (int64_t)ray * S;
        const float* tr = t + (int64_t)ray * S;
        const float dcr = d_color[ray * 3 + 0], dcg = d_color[ray * 3 + 1], dcb = d_color[ray * 3 + 2];
        const float da = d_alpha[ray];
        SampleTerms q[ROWS];
        float T[ROWS];
        float carry = 1.0f;
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
            const int s = row * 64 + lane;
            q[row] = load_terms(lg, tr, s, S, s < S, nullptr);
            const float incl = wave_scan_mul(q[row].tau, lane);
            const float excl = wave_shift_up(incl, 1.0f);
            T[row] = carry * excl;
            carry *= wave_last(incl);
        }
        float tail = 0.0f;  // sum of g*w over all later rows
#pragma unroll
        for (int row = ROWS - 1; row >= 0; --row) {
            const int s = row * 64 + lane;
            const float4 out = composite_bwd_row(q[row], T[row], dcr, dcg, dcb, da, s, S, lane, tail);
            if (s < S) d_logits[(int64_t)ray * S + s] = out;
        }
    }
}

// ---------------------------------------------------------------------------------- K5t
// The training step's composite: K5 (forward), K6 (ground-truth gather, residuals, loss sums)
// and K5b (backward) of a batch in ONE launch -- one wave per ray keeps the ray's sample terms in
// registers between the forward reduction and the backward walk, the colour / alpha / d_colour /
// d_alpha round trip through HBM and three launches go away (14 us of the 1.2 ms step at the
// reference's default batch).  Same expressions in the same order as the three kernels above:
// d_logits is bit-identical to theirs.  Loss sums leave as one pair per workgroup
// (`partials`), summed in a fixed order by ffn_loss_from_partials.
template <int ROWS>
__global__ void __launch_bounds__(256)
composite_train_kernel(const float4* __restrict__ logits, const float* __restrict__ t,
                       const float* __restrict__ gt_colors, const float* __restrict__ gt_alphas,
                       const int64_t* __restrict__ ray_index, int R, int S, float color_scale,
                       float alpha_scale, float4* __restrict__ d_logits,
                       float* __restrict__ partials, int32_t* nan_flag) {
    __shared__ float red[2][4];
    const int lane = lane_id();
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int waves = (int)((gridDim.x * (int64_t)blockDim.x) >> 6);
    float ec = 0.0f, ea = 0.0f;
    for (int ray = wave; ray < R; ray += waves) {
        const float4* lg = logits + (int64_t)ray * S;
        const float* tr = t + (int64_t)ray * S;
        SampleTerms q[ROWS];
        float T[ROWS];
        RayAccum acc;
        acc.reset();
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
            const int s = row * 64 + lane;
            q[row] = load_terms(lg, tr, s, S, s < S, nan_flag);
            T[row] = acc.row(q[row], lane, s, s < S - 1);
        }
        acc.finish();
        const float cr = acc.cr, cg = acc.cg, cb = acc.cb, asum = acc.asum;
        // K6 on this ray
        const int64_t gt = ray_index[ray];
        float gr = gt_colors[gt * 3 + 0], gg = gt_colors[gt * 3 + 1], gb = gt_colors[gt * 3 + 2];
        float da = 0.0f;
        if (gt_alphas != nullptr) {
            const float ga = gt_alphas[gt];
            if (!(ga > 0.0f)) { gr = 0.f; gg = 0.f; gb = 0.f; }
            const float diff = asum - ga;
            ea += diff * diff;
            da = alpha_scale * 2.0f * diff;
        }
        const float d0 = cr - gr, d1 = cg - gg, d2 = cb - gb;
        ec += (d0 * d0 + d1 * d1) + d2 * d2;
        float dcr = color_scale * 2.0f * d0, dcg = color_scale * 2.0f * d1, dcb = color_scale * 2.0f * d2;
        // (opaque to the optimiser, like K5b's loads of d_colour / d_alpha: the backward rows see
        // plain values in both kernels)
        asm volatile("" : "+v"(dcr), "+v"(dcg), "+v"(dcb), "+v"(da));
        // K5b on the terms still in registers
        float tail = 0.0f;
#pragma unroll
        for (int row = ROWS - 1; row >= 0; --row) {
            const int s = row * 64 + lane;
            const float4 out = composite_bwd_row(q[row], T[row], dcr, dcg, dcb, da, s, S, lane, tail);
            if (s < S) d_logits[(int64_t)ray * S + s] = out;
        }
    }
    const int w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = ec; red[1][w] = ea; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// sums of K5t's / K6's per-workgroup partials, and (optionally) the scalar loss from them:
//   loss = sums[0] / colour_count + alpha_weight * (sums[1] / alpha_count)   (image_dataset.py:237-242)
__global__ void __launch_bounds__(64)
loss_from_partials_kernel(const float* __restrict__ partials, int blocks, float colour_count,
                          float alpha_count, float alpha_weight, float* __restrict__ sums,
                          float* __restrict__ loss) {
    float ec = 0.0f, ea = 0.0f;
    for (int i = threadIdx.x; i < blocks; i += 64) { ec += partials[i * 2]; ea += partials[i * 2 + 1]; }
    ec = wave_sum(ec); ea = wave_sum(ea);
    if (threadIdx.x == 0) {
        if (sums != nullptr) { sums[0] = ec; sums[1] = ea; }
        if (loss != nullptr) {
            const float colour = ec / colour_count;
            const float alpha = alpha_weight != 0.0f ? alpha_weight * (ea / alpha_count) : 0.0f;
            loss[0] = colour + alpha;
        }
    }
}

// ---------------------------------------------------------------------------------- K6
__global__ void __launch_bounds__(256)
mse_partial_kernel(const float* __restrict__ color, const float* __restrict__ alpha,
                   const float* __restrict__ gt_colors, const float* __restrict__ gt_alphas,
                   const int64_t* __restrict__ ray_index, int R, float color_scale,
                   float alpha_scale, float* __restrict__ d_color, float* __restrict__ d_alpha,
                   float* __restrict__ scratch) {
    __shared__ float red[2][4];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float ec = 0.0f, ea = 0.0f;
    if (r < R) {
        const int64_t ray = ray_index[r];
        float gr = gt_colors[ray * 3 + 0], gg = gt_colors[ray * 3 + 1], gb = gt_colors[ray * 3 + 2];
        if (gt_alphas != nullptr) {
            const float ga = gt_alphas[ray];
            if (!(ga > 0.0f)) { gr = 0.f; gg = 0.f; gb = 0.f; }
            const float diff = alpha[r] - ga;
            ea = diff * diff;
            if (d_alpha != nullptr) d_alpha[r] = alpha_scale * 2.0f * diff;
        } else if (d_alpha != nullptr) {
            d_alpha[r] = 0.0f;
        }
        const float d0 = color[r * 3 + 0] - gr, d1 = color[r * 3 + 1] - gg, d2 = color[r * 3 + 2] - gb;
        ec = (d0 * d0 + d1 * d1) + d2 * d2;
        if (d_color != nullptr) {
            d_color[r * 3 + 0] = color_scale * 2.0f * d0;
            d_color[r * 3 + 1] = color_scale * 2.0f * d1;
            d_color[r * 3 + 2] = color_scale * 2.0f * d2;
        }
    }
    ec = wave_sum(ec); ea = wave_sum(ea);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) { red[0][w] = ec; red[1][w] = ea; }
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[blockIdx.x * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        scratch[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(64)
mse_final_kernel(const float* __restrict__ scratch, int blocks, float* __restrict__ sums) {
    float ec = 0.0f, ea = 0.0f;
    for (int i = threadIdx.x; i < blocks; i += 64) { ec += scratch[i * 2]; ea += scratch[i * 2 + 1]; }
    ec = wave_sum(ec); ea = wave_sum(ea);
    if (threadIdx.x == 0) { sums[0] = ec; sums[1] = ea; }
}

}  // namespace ffn

using namespace ffn;

static inline int ray_grid(int R) {
    int64_t blocks = ((int64_t)R + 3) / 4;  // 4 waves (rays) per 256-thread block
    if (blocks > 256 * 16) blocks = 256 * 16;
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" int ffn_composite_fwd(const float* logits, const float* t, int num_rays,
                                 int num_samples, float* color, float* alpha, float* depth,
                                 int32_t* nan_flag, void* stream) {
    if (num_rays == 0) return 0;
    if (num_rays < 0 || num_samples < 1) return fail_arg("ffn_composite_fwd: shape");
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(ray_grid(num_rays)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)logits, t, num_rays, num_samples, color,
                       alpha, depth, nan_flag);
    return check_launch("ffn_composite_fwd");
}

extern "C" int ffn_blend_weights(const float* t, const float* sigma, int num_rays,
                                 int num_samples, float* weights, void* stream) {
    if (num_rays == 0) return 0;
    if (num_rays < 0 || num_samples < 1) return fail_arg("ffn_blend_weights: shape");
    hipLaunchKernelGGL(blend_weights_kernel, dim3(ray_grid(num_rays)), dim3(256), 0,
                       (hipStream_t)stream, t, sigma, num_rays, num_samples, weights);
    return check_launch("ffn_blend_weights");
}

extern "C" int ffn_blend_weights_bwd(const float* t, const float* sigma, const float* d_weights,
                                     int num_rays, int num_samples, float* d_sigma, float* d_t,
                                     void* stream) {
    if (num_rays == 0) return 0;
    if (num_rays < 0 || num_samples < 1) return fail_arg("ffn_blend_weights_bwd: shape");
    const int rows = (num_samples + 63) / 64;
    const dim3 grid(ray_grid(num_rays)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define FFN_BWB(ROWS)                                                                          \
    hipLaunchKernelGGL(blend_weights_bwd_kernel<ROWS>, grid, block, 0, st, t, sigma, d_weights, \
                       num_rays, num_samples, d_sigma, d_t)
    switch (rows) {
        case 1: FFN_BWB(1); break;
        case 2: FFN_BWB(2); break;
        case 3: case 4: FFN_BWB(4); break;
        default: return fail_arg("ffn_blend_weights_bwd: num_samples > 256");
    }
#undef FFN_BWB
    return check_launch("ffn_blend_weights_bwd");
}

extern "C" int ffn_composite_bwd(const float* logits, const float* t, const float* d_color,
                                 const float* d_alpha, int num_rays, int num_samples,
                                 float* d_logits, void* stream) {
    if (num_rays == 0) return 0;
    if (num_rays < 0 || num_samples < 1) return fail_arg("ffn_composite_bwd: shape");
    const int rows = (num_samples + 63) / 64;
    const dim3 grid(ray_grid(num_rays)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define FFN_BWD(ROWS)                                                                          \
    hipLaunchKernelGGL(composite_bwd_kernel<ROWS>, grid, block, 0, st, (const float4*)logits, t, \
                       d_color, d_alpha, num_rays, num_samples, (float4*)d_logits)
    switch (rows) {
        case 1: FFN_BWD(1); break;
        case 2: FFN_BWD(2); break;
        case 3: FFN_BWD(3); break;
        case 4: FFN_BWD(4); break;
        case 5: case 6: case 7: case 8: FFN_BWD(8); break;
        default: return fail_arg("ffn_composite_bwd: num_samples > 512");
    }
#undef FFN_BWD
    return check_launch("ffn_composite_bwd");
}

extern "C" int ffn_mse_loss(const float* color, const float* alpha, const float* gt_colors,
                            const float* gt_alphas, const int64_t* ray_index, int num_rays,
                            float color_scale, float alpha_scale, float* sums, float* d_color,
                            float* d_alpha, float* scratch, void* stream) {
    if (num_rays <= 0) return fail_arg("ffn_mse_loss: empty batch");
    const int blocks = (num_rays + 255) / 256;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, color,
                       alpha, gt_colors, gt_alphas, ray_index, num_rays, color_scale, alpha_scale,
                       d_color, d_alpha, scratch);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, blocks,
                       sums);
    return check_launch("ffn_mse_loss");
}

extern "C" int ffn_composite_train_blocks(int num_rays) { return num_rays > 0 ? ray_grid(num_rays) : 0; }

extern "C" int ffn_composite_train(const float* logits, const float* t, int num_rays, int num_samples,
                                   const float* gt_colors, const float* gt_alphas,
                                   const int64_t* ray_index, float color_scale, float alpha_scale,
                                   float* d_logits, float* partials, int32_t* nan_flag, void* stream) {
    if (num_rays <= 0 || num_samples < 1) return fail_arg("ffn_composite_train: shape");
    if (logits == nullptr || t == nullptr || gt_colors == nullptr || ray_index == nullptr ||
        d_logits == nullptr || partials == nullptr)
        return fail_arg("ffn_composite_train: null argument");
    const int rows = (num_samples + 63) / 64;
    const dim3 grid(ray_grid(num_rays)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define FFN_CT(ROWS)                                                                           \
    hipLaunchKernelGGL(composite_train_kernel<ROWS>, grid, block, 0, st, (const float4*)logits, t, \
                       gt_colors, gt_alphas, ray_index, num_rays, num_samples, color_scale,    \
                       alpha_scale, (float4*)d_logits, partials, nan_flag)
    switch (rows) {
        case 1: FFN_CT(1); break;
        case 2: FFN_CT(2); break;
        case 3: FFN_CT(3); break;
        case 4: FFN_CT(4); break;
        case 5: case 6: case 7: case 8: FFN_CT(8); break;
        default: return fail_arg("ffn_composite_train: num_samples > 512");
    }
#undef FFN_CT
    return check_launch("ffn_composite_train");
}

extern "C" int ffn_loss_from_partials(const float* partials, int num_blocks, float colour_count,
                                      float alpha_count, float alpha_weight, float* sums,
                                      float* loss_out, void* stream) {
    if (partials == nullptr || num_blocks < 1 || (sums == nullptr && loss_out == nullptr))
        return fail_arg("ffn_loss_from_partials: arguments");
    hipLaunchKernelGGL(loss_from_partials_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials,
                       num_blocks, colour_count, alpha_count, alpha_weight, sums, loss_out);
    return check_launch("ffn_loss_from_partials");
}
