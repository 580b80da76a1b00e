// What is the f32-MFMA cadence in CORE cycles, and what core clock does the chip sustain while
// every SIMD issues v_mfma_f32_32x32x2_f32 back to back?  clock64() = s_memtime (core clock
// domain), wall_clock64() = s_memrealtime (constant 100 MHz).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACCS>
__global__ void __launch_bounds__(256, 1) probe(float* out, long long* stamps, int reps, float seed) {
    f32x16 acc[ACCS];
    for (int o = 0; o < ACCS; ++o) for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;
    float a = seed * (threadIdx.x & 63) + 0.37f, b = 1.0f + seed * (threadIdx.x >> 3);
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int o = 0; o < ACCS; ++o) acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[o], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int o = 0; o < ACCS; ++o) for (int r = 0; r < 16; ++r) s += acc[o][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 0] = c1 - c0;
        stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = w1 - w0;
    }
}

template <int ACCS>
void run(const char* name, int grid, float* out, long long* stamps, float seed) {
    const int reps = 40000;
    probe<ACCS><<<grid, 256>>>(out, stamps, 200, seed);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<ACCS><<<grid, 256>>>(out, stamps, reps, seed);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double core = 0, wall = 0;
    for (int i = 0; i < grid * 4; ++i) { core += h[2 * i]; wall += h[2 * i + 1]; }
    core /= grid * 4; wall /= grid * 4;
    const double mfmas = (double)reps * 8 * ACCS;
    printf("%-40s grid %4d: %7.2f ms  core cycles/MFMA %.2f  core clock %.0f MHz  wall ns/MFMA %.2f  -> %.1f TFLOP/s (%.1f%% of 157.3)\n",
           name, grid, ms, core / mfmas, core / wall * 100.0, wall * 10.0 / mfmas,
           mfmas * 4096 * grid * 4 / (wall * 10e-9) / 1e12, mfmas * 4096 * grid * 4 / (wall * 10e-9) / 1e12 / 157.3 * 100);
}

// VARIANT 1: 32 different A registers; 2: 4 different B registers; 3: both (the K loop's pattern:
// tile o, component p -> A register a[o][p], B register x[p])
template <int VARIANT>
__global__ void __launch_bounds__(256, 1) probe_operands(float* out, long long* stamps, int reps, float seed) {
    f32x16 acc[8];
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;
    float a[8][4], x[4];
    for (int o = 0; o < 8; ++o) for (int p = 0; p < 4; ++p) a[o][p] = seed * ((threadIdx.x & 63) + 3 * o + p) + 0.37f;
    for (int p = 0; p < 4; ++p) x[p] = 1.0f + seed * ((threadIdx.x >> 3) + p);
    for (int o = 0; o < 8; ++o) for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(a[o][p]));
    for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(x[p]));
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(VARIANT == 2 ? a[0][0] : a[o][p],
                                                                  VARIANT == 1 ? x[0] : x[p], acc[o], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) s += acc[o][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 0] = c1 - c0;
        stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = w1 - w0;
    }
}

template <int VARIANT>
void run_operands(const char* name, float* out, long long* stamps) {
    const int reps = 5000, grid = 256;
    probe_operands<VARIANT><<<grid, 256>>>(out, stamps, 100, 0.013f);
    hipDeviceSynchronize();
    probe_operands<VARIANT><<<grid, 256>>>(out, stamps, reps, 0.013f);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double core = 0, wall = 0;
    for (int i = 0; i < grid * 4; ++i) { core += h[2 * i]; wall += h[2 * i + 1]; }
    const double mfmas = (double)reps * 64 * grid * 4;
    printf("%-40s core cycles/MFMA %.2f  core clock %.0f MHz\n", name, core / mfmas, core / wall * 100.0);
}

int main() {
    float* out; long long* stamps;
    hipMalloc(&out, 1 << 22); hipMalloc(&stamps, 1 << 20);
    run<8>("8 accumulators, random data, full chip", 256, out, stamps, 0.013f);
    run<8>("8 accumulators, zeros, full chip", 256, out, stamps, 0.0f);
    run<8>("8 accumulators, random data, 32 CUs", 32, out, stamps, 0.013f);
    run<4>("4 accumulators, random data, full chip", 256, out, stamps, 0.013f);
    run<16>("16 accumulators, random data, full chip", 256, out, stamps, 0.013f);
    run_operands<1>("32 A registers, one B register", out, stamps);
    run_operands<2>("one A register, 4 B registers", out, stamps);
    run_operands<3>("32 A registers, 4 B registers", out, stamps);
    return 0;
}
