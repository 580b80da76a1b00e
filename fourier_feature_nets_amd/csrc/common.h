// Shared helpers for the gfx950 kernels of libffn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ffn_hip.h"

namespace ffn {

void set_error(const char* what, hipError_t code);

inline int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error(what, err);
        return (int)err;
    }
    return 0;
}

inline int fail_arg(const char* what) {
    set_error(what, hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
}

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// numpy-style max/min: NaN if either side is NaN (np.max / np.min over an axis).
__device__ __forceinline__ float np_max(float a, float b) {
    return (a != a || b != b) ? __builtin_nanf("") : (a > b ? a : b);
}
__device__ __forceinline__ float np_min(float a, float b) {
    return (a != a || b != b) ? __builtin_nanf("") : (a < b ? a : b);
}

// a*b + c with the product and the sum rounded separately, whatever -ffp-contract the including
// file is compiled with (HIP's __fmul_rn / __fadd_rn are plain operators and DO get fused into
// an fma after inlining).  The sampling arithmetic of the reference is a sequence of separate
// ATen ops (mul, then add), so bit-exact t-values and positions need exactly this.
__device__ __forceinline__ float mul_add_rn(float a, float b, float c) {
#pragma clang fp contract(off)
    const float p = a * b;
    return p + c;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// Encoding tables live in LDS while a kernel runs: per encoding four rows [b row 0 | b row 1 |
// b row 2 | a] of 256 floats (F <= 256; the tail of a row is zero), so that the entries of
// two consecutive frequencies are one aligned 8-byte read.
constexpr int kEncRowPitch = 256;
constexpr int kEncTablePitch = 4 * kEncRowPitch;   // floats per encoding
constexpr int kEncTableBytes = 2 * kEncTablePitch * 4;

__device__ __forceinline__ void stage_encoding_tables(const ffn_encoding* enc, float* table,
                                                      int tid, int nthreads) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int f = enc[e].num_freq;
        float* dst = table + e * kEncTablePitch;
        for (int i = tid; i < kEncTablePitch; i += nthreads) {
            const int row = i / kEncRowPitch, k = i % kEncRowPitch;
            float v = 0.0f;
            if (k < f) v = row < 3 ? enc[e].b[row * f + k] : enc[e].a[k];
            dst[i] = v;
        }
    }
}

// Branch-free sin/cos: 3-constant Cody-Waite reduction by pi/2 (fused multiply-adds) and
// degree-7/8 minimax polynomials on [-pi/4, pi/4].  Max abs error 8.8e-8 for |x| <= 5000
// (checked against a 200-bit reference), i.e. the class of a 1-ulp libm; the encodings on
// this path stay below ~900 rad.  ~30 VALU instructions for both values, no branches, so
// the compiler can interleave it with MFMAs.
__device__ __forceinline__ void fast_sincos(float x, float& sn, float& cs) {
    const float k = __builtin_rintf(x * 0.6366197466850281f);
    float r = __builtin_fmaf(-k, 1.5707963705062866f, x);
    r = __builtin_fmaf(-k, -4.371138828673793e-08f, r);
    r = __builtin_fmaf(-k, -1.7151245100058819e-15f, r);
    const float z = r * r;
    float sp = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    sp = __builtin_fmaf(sp * z, r, r);
    float cp = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    cp = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const bool swap = (q & 1) != 0;
    const float s0 = swap ? cp : sp;
    const float c0 = swap ? sp : cp;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

// N angles at once (N = 2 or 4) on the packed-f32 pipe (v_pk_fma_f32 / v_pk_mul_f32: identical
// rounding per component, half the instructions for the polynomial part).  With N = 4 every
// source line becomes two independent packed instructions, which is the instruction-level
// parallelism a lone in-order wave needs to cover the packed-FMA latency.
typedef float ffn_f32x2 __attribute__((ext_vector_type(2)));
typedef float ffn_f32x4 __attribute__((ext_vector_type(4)));

template <typename V, int N>
__device__ __forceinline__ void fast_sincos_n(V x, V& sn, V& cs) {
    const V t = x * 0.6366197466850281f;
    V k;
#pragma unroll
    for (int j = 0; j < N; ++j) k[j] = __builtin_rintf(t[j]);
    const V nk = -k;
    V r = __builtin_elementwise_fma(nk, (V)(1.5707963705062866f), x);
    r = __builtin_elementwise_fma(nk, (V)(-4.371138828673793e-08f), r);
    r = __builtin_elementwise_fma(nk, (V)(-1.7151245100058819e-15f), r);
    const V z = r * r;
    V sp = __builtin_elementwise_fma((V)(-1.9515295891e-4f), z, (V)(8.3321608736e-3f));
    V cp = __builtin_elementwise_fma((V)(2.443315711809948e-5f), z, (V)(-1.388731625493765e-3f));
    sp = __builtin_elementwise_fma(sp, z, (V)(-1.6666654611e-1f));
    cp = __builtin_elementwise_fma(cp, z, (V)(4.166664568298827e-2f));
    const V half = __builtin_elementwise_fma((V)(-0.5f), z, (V)(1.0f));
    sp = __builtin_elementwise_fma(sp * z, r, r);
    cp = __builtin_elementwise_fma(cp * z, z, half);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int q = (int)k[j];
        const bool swap = (q & 1) != 0;
        const float s0 = swap ? cp[j] : sp[j];
        const float c0 = swap ? sp[j] : cp[j];
        // sign flips as integer XORs of the sign bit
        sn[j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s0) ^ ((unsigned)(q & 2) << 30));
        cs[j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c0) ^ ((unsigned)((q + 1) & 2) << 30));
    }
}

}  // namespace ffn
