// Standalone Fourier feature encoding (API / testing; the training path fuses this into
// the first MLP layer).  HBM-write-bound by bytes: 12 B read + 4*(2F[+3]) B written per sample.
//
// What limits a write-bound kernel here is the shape of its stores.  Rows have an arbitrary
// pitch (2040 B for the tiny model, 252 B for NeRF), so stores issued straight from the
// computing lanes are 4-byte-per-lane runs that start and end inside cache lines: every
// variant of that (one thread per element, per pair, per pair with the frequencies owned by
// the thread, batched / double-buffered point loads) measured 2.7-3.9 TB/s, and non-temporal
// stores halved it (the partial lines then reach memory unmerged).  So a block assembles a
// GROUP of rows (a multiple of four: the group then starts on a 16-byte boundary whatever the
// pitch) in LDS and streams it out as aligned 16-byte-per-lane stores of whole lines:
// 4.4 TB/s (tiny, 510 columns) / 4.3 TB/s (NeRF, 63 columns), where a device-to-device copy of
// the same bytes moves 4.1 TB/s -- profiles/r02_hbm_microbench.json.
//
// Compute side: a thread owns the pair of frequencies (k, k + ceil(F/2)), so its b / a values
// are registers, there is no per-element index arithmetic, and the body is 3 packed
// multiply-adds, one packed sincos and four conflict-free LDS writes.
#include "common.h"

namespace ffn {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int ENCODE_TILE_BYTES = 64 * 1024;                  // most LDS a block may take
constexpr int ENCODE_GROUP_BYTES = 32 * 1024;                 // what a block normally assembles
constexpr int ENCODE_MAX_GROUP = 128;                         // rows per block pass

template <bool INC>                                           // F >= 1
__global__ void __launch_bounds__(256)
fourier_encode_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ b,
                      const float* __restrict__ a, int F, float scale,
                      float* __restrict__ out, int group) {
    extern __shared__ float4 tile4[];
    float* tile = reinterpret_cast<float*>(tile4);            // group x width floats
    __shared__ float points[3 * ENCODE_MAX_GROUP];
    const int width = 2 * F + (INC ? 3 : 0);
    const int pairs = (F + 1) >> 1;
    const int passthrough = INC ? 3 : 0;                      // every thread also copies x
    const int per = pairs > passthrough ? pairs : passthrough;
    const int lanes = per < 256 ? per : 256;                  // threads that share one sample
    const int spb = 256 / lanes;                              // samples in flight per block
    const int sub = (int)threadIdx.x / lanes;
    const int lane = (int)threadIdx.x - sub * lanes;
    const int64_t groups = (n + group - 1) / group;
    for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const int64_t row0 = g * group;
        const int rows = n - row0 < group ? (int)(n - row0) : group;
        __syncthreads();                                       // the previous group has left the tile
        for (int e = threadIdx.x; e < 3 * rows; e += 256) points[e] = x[row0 * 3 + e];
        __syncthreads();
        if (sub < spb) {
            for (int slot = lane; slot < per; slot += lanes) {
                // a slot beyond the pairs (F < 5 with pass-through columns) repeats pair 0, a
                // slot beyond the three pass-through columns repeats column 2: same values to
                // the same addresses, no branches
                const int k = slot < pairs ? slot : 0;
                const int k1 = k + pairs < F ? k + pairs : k;
                const int col = slot < 2 ? slot : 2;
                f32x2 b0, b1, b2, amp;
                b0[0] = b[k]; b0[1] = b[k1];
                b1[0] = b[F + k]; b1[1] = b[F + k1];
                b2[0] = b[2 * F + k]; b2[1] = b[2 * F + k1];
                amp[0] = a != nullptr ? a[k] : 1.0f; amp[1] = a != nullptr ? a[k1] : 1.0f;
                for (int r = sub; r < rows; r += spb) {
                    const float p0 = points[3 * r], p1 = points[3 * r + 1], p2 = points[3 * r + 2];
                    float* row = tile + r * width;
                    if (INC) row[2 * F + col] = col == 0 ? p0 : (col == 1 ? p1 : p2);
                    // (scale*x) @ B as a k-ordered chain; exact for block-diagonal B
                    const float s0 = scale * p0, s1 = scale * p1, s2 = scale * p2;
                    f32x2 ang;
                    ang[0] = s0 * b0[0]; ang[1] = s0 * b0[1];
                    ang[0] = __builtin_fmaf(s1, b1[0], ang[0]); ang[1] = __builtin_fmaf(s1, b1[1], ang[1]);
                    ang[0] = __builtin_fmaf(s2, b2[0], ang[0]); ang[1] = __builtin_fmaf(s2, b2[1], ang[1]);
                    f32x2 sn, cs;
                    fast_sincos_n<f32x2, 2>(ang, sn, cs);
                    row[k] = amp[0] * cs[0]; row[F + k] = amp[0] * sn[0];
                    row[k1] = amp[1] * cs[1]; row[F + k1] = amp[1] * sn[1];
                }
            }
        }
        __syncthreads();
        // stream the group out: row0 is a multiple of four rows, so row0 * width * 4 B is a
        // multiple of 16 B and (with a 16-byte aligned `out`) every store is an aligned float4
        const int total = rows * width;
        float* dst = out + row0 * width;
        float4* dst4 = reinterpret_cast<float4*>(dst);
        for (int v = threadIdx.x; v < total / 4; v += 256) dst4[v] = tile4[v];
        for (int e = (total & ~3) + threadIdx.x; e < total; e += 256) dst[e] = tile[e];
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
    if (num_freq == 0) {                                       // the encoding of no frequencies is x
        if (hipMemcpyAsync(out, x, (size_t)n * 12, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return fail_arg("ffn_fourier_encode: copy");
        return 0;
    }
    if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return fail_arg("ffn_fourier_encode: out must be 16-byte aligned");
    const int width = 2 * num_freq + (include_input ? 3 : 0);
    // 32 KB groups measured best (8 KB: 3.3 TB/s, 16 KB: 4.1, 32 KB: 4.4, 48 KB: 4.4)
    int group = ENCODE_GROUP_BYTES / (4 * width) / 4 * 4;     // rows per block pass, a multiple of 4
    if (group > ENCODE_MAX_GROUP) group = ENCODE_MAX_GROUP;
    if (group < 4) group = ENCODE_TILE_BYTES / (4 * width) / 4 * 4 >= 4 ? 4 : 0;
    if (group < 4) return fail_arg("ffn_fourier_encode: more than 2046 frequencies");
    const size_t lds = (size_t)group * width * 4;
    int64_t grid = (n + group - 1) / group;
    if (grid > 256 * 16) grid = 256 * 16;
    if (include_input)
        hipLaunchKernelGGL(fourier_encode_kernel<true>, dim3((int)grid), dim3(256), lds, (hipStream_t)stream,
                           x, n, b, a, num_freq, scale, out, group);
    else
        hipLaunchKernelGGL(fourier_encode_kernel<false>, dim3((int)grid), dim3(256), lds, (hipStream_t)stream,
                           x, n, b, a, num_freq, scale, out, group);
    return check_launch("ffn_fourier_encode");
}
