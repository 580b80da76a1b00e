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

// Encoding tables live in LDS while a kernel runs: per encoding [b row 0 | b row 1 | b row 2 |
// a], each max(F,1) floats, at a fixed 1024-float pitch.  Feature arithmetic then waits on
// lgkmcnt only and never drains the weight / operand prefetches that sit on vmcnt.
constexpr int kEncTablePitch = 1024;            // floats per encoding (F <= 256)
constexpr int kEncTableBytes = 2 * kEncTablePitch * 4;

__device__ __forceinline__ void stage_encoding_tables(const ffn_encoding* enc, float* table,
                                                      int tid, int nthreads) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int fi = enc[e].num_freq > 0 ? enc[e].num_freq : 1;
        float* dst = table + e * kEncTablePitch;
        for (int i = tid; i < 3 * fi; i += nthreads) dst[i] = enc[e].b[i];
        for (int i = tid; i < fi; i += nthreads) dst[3 * fi + i] = enc[e].a[i];
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

}  // namespace ffn
