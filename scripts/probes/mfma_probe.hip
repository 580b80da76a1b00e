// Calibration probe: what does v_mfma_f32_32x32x2_f32 sustain on this chip with one wave per
// SIMD under the access patterns of the fused MLP loop?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT, int OT>
__global__ void __launch_bounds__(256, 1) probe(const float* __restrict__ w, float* out, int groups, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    f32x4* act = reinterpret_cast<f32x4*>(smem + wave * 32768);
    for (int i = 0; i < 32; ++i) { f32x4 v; v[0] = v[1] = v[2] = v[3] = 0.001f * lane; act[i * 64 + lane] = v; }
    f32x16 acc[OT];
    for (int o = 0; o < OT; ++o) for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;
    const f32x4* wp = reinterpret_cast<const f32x4*>(w) + lane;
    f32x4 a0[OT], a1[OT];
    f32x4 x; x[0] = 1.f; x[1] = 2.f; x[2] = 3.f; x[3] = 4.f;
    for (int o = 0; o < OT; ++o) a0[o] = wp[o * 64];
    for (int o = 0; o < OT; ++o) a1[o] = wp[(OT + o) * 64];
    for (int rep = 0; rep < reps; ++rep) {
        for (int g = 0; g < groups; g += 2) {
            if (VARIANT >= 1) for (int o = 0; o < OT; ++o) a1[o] = wp[((g + 1) * OT + o) * 64];
            f32x4 x0 = x, x1 = x;
            if (VARIANT >= 2) { x0 = act[(g & 31) * 64 + lane]; x1 = act[((g + 1) & 31) * 64 + lane]; }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int o = 0; o < OT; ++o) acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[o][p], x0[p], acc[o], 0, 0, 0);
            if (VARIANT >= 1) { const int gn = g + 2 < groups ? g + 2 : g; for (int o = 0; o < OT; ++o) a0[o] = wp[(gn * OT + o) * 64]; }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int o = 0; o < OT; ++o) acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[o][p], x1[p], acc[o], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int o = 0; o < OT; ++o) for (int r = 0; r < 16; ++r) s += acc[o][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VARIANT, int OT>
void run(const char* name, const float* w, float* out, int grid) {
    const int groups = 32, reps = 96;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<VARIANT, OT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<VARIANT, OT><<<grid, 256, 131072>>>(w, out, groups, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<VARIANT, OT><<<grid, 256, 131072>>>(w, out, groups, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * reps * groups * 4 * OT;
    printf("%-34s grid %5d  %8.3f ms  %7.1f TFLOP/s (%.1f%% of 157.3)\n", name, grid, ms, mfma * 4096 / ms / 1e9, mfma * 4096 / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *w, *out;
    hipMalloc(&w, 64 << 20); hipMalloc(&out, 64 << 20);
    std::vector<float> h(16 << 20, 0.5f);
    hipMemcpy(w, h.data(), 64 << 20, hipMemcpyHostToDevice);
    run<0, 8>("regs only, 8 acc", w, out, 256);
    run<0, 4>("regs only, 4 acc", w, out, 256);
    run<1, 8>("+ weights from L2 (8 x b128/grp)", w, out, 256);
    run<2, 8>("+ weights + LDS x reads", w, out, 256);
    run<2, 8>("same, grid 2048 (8 WG/CU queued)", w, out, 2048);
    run<0, 8>("regs only, grid 128 (half chip)", w, out, 128);
    return 0;
}
