// Rate of v_mfma_f32_32x32x16_bf16 for ONE wave per SIMD, alone and with independent VALU work
// interleaved (does the vector ALU overlap the bf16 matrix pipe, unlike the f32 MFMA?), and with
// the A operand re-read from LDS per MFMA (the weight-sharing scheme of a split-bf16 kernel).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_bf16_probe.hip -o scripts/probes/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NVALU, int LDS_A>
__global__ void __launch_bounds__(256, 1) probe(float* out, int reps, float seed, long long* cycles) {
    __shared__ f32x4 lds[4 * 64 * 8];
    f32x16 acc[8];
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;
    f32x4 araw, braw;
    for (int i = 0; i < 4; ++i) { araw[i] = seed + i + threadIdx.x; braw[i] = seed * 3 + i; }
    for (int i = 0; i < 8; ++i) lds[(threadIdx.x >> 6) * 512 + i * 64 + (threadIdx.x & 63)] = araw;
    __syncthreads();
    const f32x4* mine = lds + (threadIdx.x >> 6) * 512 + (threadIdx.x & 63);
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                f32x4 a4 = araw;
                if (LDS_A) a4 = mine[o * 64];
                const bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, braw);
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[o], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NVALU; ++j) {
                    const int idx = (o * NVALU + j) & 7;
                    v[idx] = __builtin_fmaf(v[idx], 1.0001f, 0.5f);
                }
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int o = 0; o < 8; ++o) for (int r = 0; r < 16; ++r) s += acc[o][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int NVALU, int LDS_A>
void run(float* out, long long* cyc) {
    const int reps = 2000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NVALU, LDS_A><<<grid, 256>>>(out, 10, 1.f, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NVALU, LDS_A><<<grid, 256>>>(out, reps, 1.f, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mfma = (double)grid * 4 * reps * 32;
    printf("%2d VALU per MFMA, A from %s: %7.3f ms  %7.1f TFLOP/s (bf16 dense)  %5.1f ticks/MFMA\n", NVALU,
           LDS_A ? "LDS " : "regs", ms, mfma * 32768 / ms / 1e9, (double)h / (reps * 32.0));
}

int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    run<0, 0>(out, cyc); run<1, 0>(out, cyc); run<2, 0>(out, cyc); run<4, 0>(out, cyc); run<6, 0>(out, cyc); run<8, 0>(out, cyc);
    run<0, 1>(out, cyc); run<2, 1>(out, cyc); run<4, 1>(out, cyc);
    return 0;
}
