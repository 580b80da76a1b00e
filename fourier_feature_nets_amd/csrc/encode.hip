// Standalone Fourier feature encoding (API / testing; the training path fuses this into
// the first MLP layer).  One thread per (sample, frequency): writes the cos and the sin
// entry.  HBM-write-bound: 12 B read + 4*(2F[+3]) B written per sample.
#include "common.h"

namespace ffn {

__global__ void __launch_bounds__(256)
fourier_encode_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ b,
                      const float* __restrict__ a, int F, float scale, int include_input,
                      float* __restrict__ out) {
    const int width = 2 * F + (include_input ? 3 : 0);
    const int64_t total = n * (int64_t)(F + (include_input ? 3 : 0));
    const int per = F + (include_input ? 3 : 0);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / per;
        const int k = (int)(e - i * per);
        const float x0 = x[i * 3 + 0], x1 = x[i * 3 + 1], x2 = x[i * 3 + 2];
        float* row = out + i * width;
        if (k < F) {
            // (scale*x) @ B as a k-ordered chain; exact for block-diagonal B
            const float s0 = scale * x0, s1 = scale * x1, s2 = scale * x2;
            float ang = s0 * b[k];
            ang = __builtin_fmaf(s1, b[F + k], ang);
            ang = __builtin_fmaf(s2, b[2 * F + k], ang);
            float sn, cs;
            fast_sincos(ang, sn, cs);
            const float amp = a != nullptr ? a[k] : 1.0f;
            row[k] = amp * cs;
            row[F + k] = amp * sn;
        } else {
            const int d = k - F;
            row[2 * F + d] = d == 0 ? x0 : (d == 1 ? x1 : x2);
        }
    }
}

}  // namespace ffn

using namespace ffn;

extern "C" int ffn_fourier_encode(const float* x, int64_t n, const float* b, const float* a,
                                  int num_freq, float scale, int include_input, float* out,
                                  void* stream) {
    if (n == 0) return 0;
    if (n < 0 || num_freq < 0) return fail_arg("ffn_fourier_encode: shape");
    if (num_freq == 0) include_input = 1;
    const int64_t total = n * (int64_t)(num_freq + (include_input ? 3 : 0));
    int64_t grid = (total + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(fourier_encode_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream,
                       x, n, b, a, num_freq, scale, include_input, out);
    return check_launch("ffn_fourier_encode");
}
