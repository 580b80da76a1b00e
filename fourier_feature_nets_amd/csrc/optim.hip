// clip_grad_value_ -> clip_grad_norm_ -> Adam (L2 weight decay) over one flat fp32 buffer.
// Two launches: (1) clamp + per-block sum of squares, (2) every block re-reduces the few
// hundred partials (deterministic order), scales, and applies the Adam update.
#include "common.h"

namespace ffn {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float total = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return total;
}

__global__ void __launch_bounds__(256)
clip_value_sumsq_kernel(float* __restrict__ grads, int64_t n, float clip_value,
                        float* __restrict__ partial) {
    __shared__ float red[4];
    const int64_t base = (int64_t)blockIdx.x * 1024;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < n) {
            float g = grads[i];
            g = g < -clip_value ? -clip_value : (g > clip_value ? clip_value : g);
            grads[i] = g;
            acc += g * g;
        }
    }
    const float total = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256)
norm_adam_kernel(float* __restrict__ params, float* __restrict__ grads, float* __restrict__ m,
                 float* __restrict__ v, int64_t n, int num_partials,
                 const float* __restrict__ partial, float max_norm, float step_size,
                 float inv_sqrt_bc2, float beta1, float beta2, float eps, float weight_decay,
                 float* __restrict__ grad_norm_out) {
    __shared__ float red[4];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < num_partials; i += 256) acc += partial[i];
    const float norm = sqrtf(block_sum_256(acc, red));
    float coef = max_norm / (norm + 1e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
    if (blockIdx.x == 0 && threadIdx.x == 0 && grad_norm_out != nullptr) grad_norm_out[0] = norm;
    const int64_t base = (int64_t)blockIdx.x * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < n) {
            float g = grads[i] * coef;
            grads[i] = g;
            const float p = params[i];
            if (weight_decay != 0.0f) g = g + weight_decay * p;
            float mi = m[i];
            mi = mi + (g - mi) * (1.0f - beta1);           // lerp_
            float vi = v[i] * beta2 + (1.0f - beta2) * g * g;
            m[i] = mi;
            v[i] = vi;
            const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
            params[i] = p - step_size * (mi / denom);
        }
    }
}

// loss = sum_colour / colour_count + alpha_weight * (sum_alpha / alpha_count): the four scalar ATen
// launches of `sums[0] / (3 n) + w * (sums[1] / n)` as one (image_dataset.py:237-242)
__global__ void loss_value_kernel(const float* __restrict__ sums, float colour_count,
                                  float alpha_count, float alpha_weight, float* __restrict__ out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float colour = sums[0] / colour_count;
        const float alpha = alpha_weight != 0.0f ? alpha_weight * (sums[1] / alpha_count) : 0.0f;
        out[0] = colour + alpha;
    }
}

}  // namespace ffn

using namespace ffn;

extern "C" int ffn_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                             int64_t n, float clip_value, float max_norm, float step_size,
                             float inv_sqrt_bc2, float beta1, float beta2, float eps,
                             float weight_decay, float* scratch, float* grad_norm_out,
                             void* stream) {
    if (n <= 0) return fail_arg("ffn_clip_adam: empty");
    const int blocks = (int)((n + 1023) / 1024);
    hipLaunchKernelGGL(clip_value_sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       grads, n, clip_value, scratch);
    hipLaunchKernelGGL(norm_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params,
                       grads, exp_avg, exp_avg_sq, n, blocks, scratch, max_norm, step_size,
                       inv_sqrt_bc2, beta1, beta2, eps, weight_decay, grad_norm_out);
    return check_launch("ffn_clip_adam");
}

extern "C" int ffn_loss_value(const float* sums, float colour_count, float alpha_count,
                              float alpha_weight, float* loss_out, void* stream) {
    if (sums == nullptr || loss_out == nullptr) return fail_arg("ffn_loss_value: null argument");
    hipLaunchKernelGGL(loss_value_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums,
                       colour_count, alpha_count, alpha_weight, loss_out);
    return check_launch("ffn_loss_value");
}
