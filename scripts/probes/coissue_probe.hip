// Does independent VALU work issued between v_mfma_f32_32x32x2_f32 of ONE wave per SIMD hide
// under the 64-cycle matrix op?  N fma per MFMA, independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NVALU, int DEP>
__global__ void __launch_bounds__(256, 1) probe(float* out, int reps, float seed) {
    f32x16 acc[8];
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    float a = seed, b = seed * 2;
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[o], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NVALU; ++j) {
                    const int idx = DEP ? 0 : ((o * NVALU + j) & 7);
                    v[idx] = __builtin_fmaf(v[idx], 1.0001f, 0.5f);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
            }
    }
    float s = 0.f;
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) s += acc[o][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NVALU, int DEP>
void run(float* out) {
    const int reps = 2000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NVALU, DEP><<<grid, 256>>>(out, 10, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NVALU, DEP><<<grid, 256>>>(out, reps, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * reps * 32;
    const double cyc = ms * 1e-3 * 2.4e9 / (reps * 32.0);
    printf("%2d %s VALU per MFMA: %7.3f ms  %6.1f TFLOP/s  ~%5.1f cycles/MFMA @2.4GHz\n", NVALU, DEP ? "dependent  " : "independent", ms, mfma * 4096 / ms / 1e9, cyc);
}
int main() {
    float* out; hipMalloc(&out, 1 << 20);
    run<0, 0>(out); run<1, 0>(out); run<2, 0>(out); run<4, 0>(out); run<8, 0>(out); run<12, 0>(out); run<16, 0>(out);
    run<4, 1>(out); run<8, 1>(out);
    return 0;
}
