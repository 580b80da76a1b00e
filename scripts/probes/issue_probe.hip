// How long does one wave (alone on its SIMD, nothing in the MFMA pipe) take per LDS / VMEM
// instruction?  Explains the epilogue and head-unit costs of the fused kernels.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/issue_probe.hip -o scripts/probes/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(256, 1) probe(const f32x4* __restrict__ g, f32x4* out, int reps, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4* slab = reinterpret_cast<f32x4*>(smem + wave * 32768);
    f32x4 v[8];
    for (int i = 0; i < 8; ++i) { v[i][0] = lane + i; v[i][1] = 1.f; v[i][2] = 2.f; v[i][3] = 3.f; }
    for (int i = 0; i < 32; ++i) slab[i * 64 + lane] = v[0];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    f32x4 acc = v[0];
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0) {            // 32 x ds_write_b128, lane-linear, then wait
#pragma unroll
            for (int i = 0; i < 32; ++i) slab[i * 64 + lane] = v[i & 7];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 1) {     // 32 x ds_read_b128, then wait
            f32x4 t[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) t[i] = slab[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < 32; ++i) acc += t[i];
        } else if (KIND == 2) {     // 16 x global_load_dwordx4 (L2 hits), then wait
            f32x4 t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = g[(r & 7) * 4096 + i * 64 + lane];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += t[i];
        } else {                    // 32 x global_store_dwordx4
#pragma unroll
            for (int i = 0; i < 32; ++i) out[((blockIdx.x * 4 + wave) * 32 + i) * 64 + lane] = v[i & 7];
        }
        asm volatile("" : "+v"(acc));
    }
    const long long t1 = __builtin_readcyclecounter();
    out[(gridDim.x * 4 * 32 + blockIdx.x * 4 + wave) * 64 + lane] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[KIND] = t1 - t0;
}

int main() {
    f32x4 *g, *out; long long* cyc;
    hipMalloc(&g, 64 << 20); hipMalloc(&out, 256 << 20); hipMalloc(&cyc, 64);
    hipMemset(g, 0, 64 << 20);
    const int reps = 2000;
    const char* names[4] = {"32 x ds_write_b128 + wait", "32 x ds_read_b128 + use", "16 x global_load_dwordx4 (L2) + use", "32 x global_store_dwordx4"};
    const int per[4] = {32, 32, 16, 32};
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    probe<0><<<256, 256, 131072>>>(g, out, reps, cyc);
    probe<1><<<256, 256, 131072>>>(g, out, reps, cyc);
    probe<2><<<256, 256, 131072>>>(g, out, reps, cyc);
    probe<3><<<256, 256, 131072>>>(g, out, reps, cyc);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    for (int k = 0; k < 4; ++k)
        printf("%-40s %8.1f cycles per batch, %6.1f per instruction (s_memtime ticks)\n", names[k], (double)h[k] / reps, (double)h[k] / reps / per[k]);
    return 0;
}
