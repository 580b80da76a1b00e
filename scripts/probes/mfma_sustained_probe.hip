// What the bf16 matrix pipe SUSTAINS on the whole chip (the power-limited rate, not the 2.5 PFLOP/s a
// 2.4 GHz clock would give): every SIMD of every CU issues back-to-back v_mfma_f32_32x32x16_bf16 from
// registers -- no memory instruction, no vector instruction, no barrier in the loop -- on operands with
// the statistics of the bf16x6 chain kernels: three-part splits (hi, mid, lo) of normal weights and of
// ReLU'd normal activations, the six partial products in the kernels' order.  Arms:
//   zero     operands all zero (what the pipe does when nothing toggles)
//   split    the three-part splits described above (the chain kernels' data)
//   hi_only  only the hi x hi product, six times (dense random bf16 data, no small parts)
// Per arm: wall-clock TFLOP/s over the whole chip, cycles per matrix instruction from s_memtime of one
// wave, and the effective clock (cycles of that wave / wall time).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_sustained_probe.hip -o scripts/probes/mfma_sustained_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kW[6] = {0, 2, 1, 0, 1, 0};           // (weight part, operand part), smallest product first
constexpr int kX[6] = {2, 0, 1, 1, 0, 0};

template <bool HI_ONLY>
__global__ void __launch_bounds__(256, 1) stream(const f32x4* __restrict__ wparts, const f32x4* __restrict__ xparts,
                                                 float* out, int reps, long long* cycles) {
    // four weight sets and two operand sets of three parts each, per lane (different per wave)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 w[4][3], x[2][3];
    for (int s = 0; s < 4; ++s)
        for (int p = 0; p < 3; ++p) w[s][p] = __builtin_bit_cast(bf16x8, wparts[((wave * 4 + s) * 3 + p) * 64 + lane]);
    for (int s = 0; s < 2; ++s)
        for (int p = 0; p < 3; ++p) x[s][p] = __builtin_bit_cast(bf16x8, xparts[((wave * 2 + s) * 3 + p) * 64 + lane]);
    f32x16 acc[4];
    for (int o = 0; o < 4; ++o) acc[o] = (f32x16)(0.0f);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int s = 0; s < 4; ++s)                 // a "unit": one weight set against both operand sets, six products
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[2 * (s & 1) + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        w[s][HI_ONLY ? 0 : kW[q]], x[b][HI_ONLY ? 0 : kX[q]], acc[2 * (s & 1) + b], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int o = 0; o < 4; ++o)
        for (int r = 0; r < 16; ++r) sum += acc[o][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

static unsigned short bf16_rne(float v) {
    unsigned u;
    memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_f32(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float v;
    memcpy(&v, &u, 4);
    return v;
}

// n sets of three parts x 64 lanes x 8 values, from `draw`
template <class F>
static std::vector<unsigned short> parts(int sets, F draw) {
    std::vector<unsigned short> out((size_t)sets * 3 * 64 * 8);
    for (int s = 0; s < sets; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                float v = draw();
                for (int p = 0; p < 3; ++p) {
                    const unsigned short h = bf16_rne(v);
                    out[(((size_t)s * 3 + p) * 64 + lane) * 8 + j] = h;
                    v -= bf16_f32(h);
                }
            }
    return out;
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    std::mt19937 rng(7);
    std::normal_distribution<float> nw(0.f, 0.0625f), nx(0.f, 1.f);
    const std::vector<unsigned short> w_split = parts(16, [&] { return nw(rng); });
    const std::vector<unsigned short> x_split = parts(8, [&] { const float v = nx(rng); return v > 0.f ? v : 0.f; });
    const std::vector<unsigned short> w_zero(w_split.size(), 0), x_zero(x_split.size(), 0);
    void *dw, *dx;
    float* out;
    long long* cyc;
    hipMalloc(&dw, w_split.size() * 2);
    hipMalloc(&dx, x_split.size() * 2);
    hipMalloc(&out, (size_t)cus * 256 * 4);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 40000;                          // x 48 matrix instructions per wave: ~30 ms per launch
    const double flop = (double)cus * 4 * reps * 48.0 * 2.0 * 32 * 32 * 16;
    struct Arm { const char* name; const std::vector<unsigned short>*w, *x; bool hi_only; };
    const Arm arms[] = {{"zero", &w_zero, &x_zero, false}, {"split", &w_split, &x_split, false},
                        {"hi_only", &w_split, &x_split, true}, {"split", &w_split, &x_split, false},
                        {"zero", &w_zero, &x_zero, false}};
    printf("{\"probe\": \"back-to-back v_mfma_f32_32x32x16_bf16 from registers, one wave per SIMD, %d CUs, %d instructions per wave and launch\", \"arms\": [", cus, reps * 48);
    bool first = true;
    for (const Arm& a : arms) {
        hipMemcpy(dw, a.w->data(), a.w->size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dx, a.x->data(), a.x->size() * 2, hipMemcpyHostToDevice);
        double best_ms = 1e30, sum_ms = 0;
        long long c = 0;
        const int launches = 6;
        for (int it = 0; it < launches + 2; ++it) {  // two warm-up launches, then back to back (the clock settles)
            hipEventRecord(e0, 0);
            if (a.hi_only) hipLaunchKernelGGL(stream<true>, dim3(cus), dim3(256), 0, 0, (const f32x4*)dw, (const f32x4*)dx, out, reps, cyc);
            else hipLaunchKernelGGL(stream<false>, dim3(cus), dim3(256), 0, 0, (const f32x4*)dw, (const f32x4*)dx, out, reps, cyc);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (it >= 2) { sum_ms += ms; best_ms = ms < best_ms ? ms : best_ms; }
        }
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double avg_ms = sum_ms / launches;
        printf("%s{\"arm\": \"%s\", \"avg_ms\": %.3f, \"tflops\": %.1f, \"of_2500\": %.4f, \"cycles_per_matrix_instruction\": %.2f, \"effective_ghz\": %.3f}",
               first ? "" : ", ", a.name, avg_ms, flop / avg_ms * 1e-9, flop / avg_ms * 1e-9 / 2500.0,
               (double)c / (reps * 48.0), (double)c / (avg_ms * 1e6));
        first = false;
    }
    printf("]}\n");
    return 0;
}
