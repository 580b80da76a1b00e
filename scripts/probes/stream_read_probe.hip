// How fast can ONE 256-thread workgroup per CU stream slabs HBM -> LDS on gfx950, as a function of
// how the bytes are kept in flight?  (a) register-staged, one 64 KiB block ahead (the staging of
// wgrad_bf16.hip: 16 global_load_dwordx4 per lane requested a block ahead, deposited with ds_write)
// (b) LDS-DMA (global_load_lds_dwordx4, nt) into a ring of STAGES x 32 KiB stages, STAGES-1 in flight.
// Each "consumer" step only touches the stage with a few ds_reads, so this is the staging ceiling.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int STAGES>
__global__ void __launch_bounds__(256, 1)
dma_ring(const char* __restrict__ src, int64_t bytes_per_wg, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (int64_t)blockIdx.x * bytes_per_wg;
    const int64_t stages = bytes_per_wg / 32768;
    // a stage = 32 pieces of 1 KiB; wave w issues pieces w, w+4, ...: 8 per stage
    auto issue = [&](int64_t st) {
        const char* g = base + st * 32768 + lane * 16;
        char* l = smem + (st % STAGES) * 32768;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int piece = wave + 4 * p;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 1024),
                                             (__attribute__((address_space(3))) void*)(l + piece * 1024), 16, 0, 2);
        }
    };
    for (int s = 0; s < STAGES - 1 && s < stages; ++s) issue(s);
    float acc = 0.0f;
    for (int64_t st = 0; st < stages; ++st) {
        // my pieces of stage st have landed when at most 8 * (stages issued after it) are outstanding
        const int64_t younger = (stages - 1 - st) < (STAGES - 2) ? (stages - 1 - st) : (STAGES - 2);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<const float*>(smem + (st % STAGES) * 32768)[tid * 7 % 8192];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // everyone is done with the stage the next issue overwrites
        if (st + STAGES - 1 < stages) issue(st + STAGES - 1);
    }
    if (acc == 12345.678f) sink[blockIdx.x * 256 + tid] = acc;
}

__global__ void __launch_bounds__(256, 1)
reg_staged(const char* __restrict__ src, int64_t bytes_per_wg, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const f32x4* base = reinterpret_cast<const f32x4*>(src + (int64_t)blockIdx.x * bytes_per_wg) + tid;
    const int64_t blocks = bytes_per_wg / 65536;
    f32x4 R[16];
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) R[j] = __builtin_nontemporal_load(base + j * 256);
    for (int64_t b = 0; b < blocks; ++b) {
        char* l = smem + (b & 1) * 65536;
#pragma unroll
        for (int j = 0; j < 16; ++j) reinterpret_cast<f32x4*>(l)[j * 256 + tid] = R[j];
        if (b + 1 < blocks) {
#pragma unroll
            for (int j = 0; j < 16; ++j) R[j] = __builtin_nontemporal_load(base + (b + 1) * 4096 + j * 256);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<const float*>(l)[tid * 7 % 16384];
    }
    if (acc == 12345.678f) sink[blockIdx.x * 256 + tid] = acc;
}

__global__ void fill_random(uint32_t* p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + 12345u;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        p[i] = (x & 0x007fffffu) | 0x3f000000u;           // floats in [0.5, 1)
    }
}

// (c) the weight-gradient kernel's pattern: TWO slabs (A and B), per block 32 KiB from each, block
// stride STRIDE bytes inside each slab (a 256-channel window of a 512-channel slab has 64 KiB), the
// workgroup's blocks contiguous
template <int STRIDE>
__global__ void __launch_bounds__(256, 1)
reg_staged_two(const char* __restrict__ a, const char* __restrict__ b, int64_t blocks_per_wg, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const char* pa = a + (int64_t)blockIdx.x * blocks_per_wg * STRIDE + tid * 16;
    const char* pb = b + (int64_t)blockIdx.x * blocks_per_wg * STRIDE + tid * 16;
    f32x4 R[16];
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        R[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa + j * 4096));
        R[8 + j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pb + j * 4096));
    }
    for (int64_t blk = 0; blk < blocks_per_wg; ++blk) {
        char* l = smem + (blk & 1) * 65536;
#pragma unroll
        for (int j = 0; j < 16; ++j) reinterpret_cast<f32x4*>(l)[j * 256 + tid] = R[j];
        if (blk + 1 < blocks_per_wg) {
            pa += STRIDE; pb += STRIDE;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                R[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa + j * 4096));
                R[8 + j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pb + j * 4096));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<const float*>(l)[tid * 7 % 16384];
    }
    if (acc == 12345.678f) sink[blockIdx.x * 256 + tid] = acc;
}

int main() {
    const int64_t total = 16ll << 30;               // 16 GiB
    const int wgs = 256;
    const int64_t per = total / wgs;
    char* src; float* sink;
    hipMalloc(&src, total); hipMalloc(&sink, wgs * 256 * 4);
    hipMemset(src, 1, total);
    const bool random = getenv("PROBE_RANDOM") != nullptr;
    if (random) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (uint32_t*)src, total / 4); hipDeviceSynchronize(); }
    printf("data: %s\n", random ? "random floats" : "constant bytes");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, const char* what) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 3; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-44s %7.3f ms  %6.2f TB/s\n", what, ms, total / (ms * 1e-3) / 1e12);
    };
    hipFuncSetAttribute((const void*)reg_staged, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    time([&] { hipLaunchKernelGGL(reg_staged, dim3(wgs), dim3(256), 131072, 0, src, per, sink); }, "register-staged, one 64 KiB block ahead");
#define RING(S) hipFuncSetAttribute((const void*)dma_ring<S>, hipFuncAttributeMaxDynamicSharedMemorySize, S * 32768); \
    time([&] { hipLaunchKernelGGL(dma_ring<S>, dim3(wgs), dim3(256), S * 32768, 0, src, per, sink); }, "LDS-DMA ring, " #S " x 32 KiB stages");
    RING(2) RING(3) RING(4) RING(5)
    {
        const char* a = src; const char* b = src + total / 2;
        hipFuncSetAttribute((const void*)reg_staged_two<32768>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipFuncSetAttribute((const void*)reg_staged_two<65536>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        const int64_t blocks32 = total / 2 / wgs / 32768, blocks64 = total / 2 / wgs / 65536;
        auto time2 = [&](auto launch, const char* what, double bytes) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int i = 0; i < 3; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
            printf("%-44s %7.3f ms  %6.2f TB/s\n", what, ms, bytes / (ms * 1e-3) / 1e12);
        };
        time2([&] { hipLaunchKernelGGL(reg_staged_two<32768>, dim3(wgs), dim3(256), 131072, 0, a, b, blocks32, sink); },
              "two slabs, 32 KiB + 32 KiB per block, dense", (double)total);
        time2([&] { hipLaunchKernelGGL(reg_staged_two<65536>, dim3(wgs), dim3(256), 131072, 0, a, b, blocks64, sink); },
              "two slabs, 32 KiB of every 64 KiB", (double)total / 2);
    }
    return 0;
}
