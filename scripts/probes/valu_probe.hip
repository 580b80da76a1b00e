// Issue cost of plain VALU work for ONE wave per SIMD (the fused MLP kernels' situation): cycles per
// instruction for 1 / 2 / 4 / 8 independent dependency chains of v_fma_f32, v_pk_fma_f32 and
// v_alignbit_b32.  If a dependent instruction costs two issue slots, bursts and epilogues need
// explicit instruction-level parallelism.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/valu_probe.hip -o scripts/probes/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int CHAINS>
__global__ void __launch_bounds__(256, 1) probe(float* out, int reps, float seed, long long* cycles) {
    float v[8]; f32x2 p[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed + i + threadIdx.x; p[i] = (f32x2)(v[i]); u[i] = threadIdx.x * 7 + i; }
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const int c = k % CHAINS;
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(seed), "v"(0.5f));
            else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(p[7]), "v"(p[6]));
            else if (KIND == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[c]) : "v"(u[7]));
            else asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[c]) : "v"(u[7]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND, int CHAINS>
void run(float* out, long long* cyc, const char* name) {
    const int reps = 2000;
    probe<KIND, CHAINS><<<256, 256>>>(out, reps, 1.0001f, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-16s %d chain(s): %6.2f cycles per instruction\n", name, CHAINS, (double)h / reps / 64);
}

int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    run<0, 1>(out, cyc, "v_fma_f32"); run<0, 2>(out, cyc, "v_fma_f32"); run<0, 4>(out, cyc, "v_fma_f32"); run<0, 8>(out, cyc, "v_fma_f32");
    run<1, 1>(out, cyc, "v_pk_fma_f32"); run<1, 2>(out, cyc, "v_pk_fma_f32"); run<1, 4>(out, cyc, "v_pk_fma_f32"); run<1, 6>(out, cyc, "v_pk_fma_f32");
    run<2, 1>(out, cyc, "v_alignbit_b32"); run<2, 2>(out, cyc, "v_alignbit_b32"); run<2, 4>(out, cyc, "v_alignbit_b32");
    run<3, 1>(out, cyc, "v_max_i32"); run<3, 4>(out, cyc, "v_max_i32");
    return 0;
}
