// Fourier-feature generation shared by the fused MLP kernels (mlp.hip) and the split-bf16
// weight-gradient kernel (wgrad_bf16.hip), which regenerates the features instead of reading a
// saved slab: the same instructions in the same order, so both produce the same bits.
#pragma once
#include "common.h"

namespace ffn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- features
// Internal feature order of an encoding: channel c' = 2k + trig (trig 0 = cos, 1 = sin)
// for k < F, then the raw inputs at c' = 2F..2F+2, then zero padding.  The float4 a lane
// needs for K group g is channels 8g + 4h + {0,1,2,3}, i.e. frequencies 4g+2h and 4g+2h+1.
// Branch-free: out-of-range frequencies are clamped for the table reads and masked after.
struct EncRegs {
    const float* tab;  // LDS copy: rows b0 | b1 | b2 | a, kEncRowPitch floats each
    int F;             // number of frequencies
    int raw;           // raw inputs follow the trig block
    float scale;
};

__device__ __forceinline__ EncRegs load_enc(const ffn_encoding& e, const float* table) {
    EncRegs r;
    r.F = e.num_freq;
    r.tab = table;
    r.raw = (e.include_input != 0 || e.num_freq == 0) ? 1 : 0;
    r.scale = e.scale;
    return r;
}

typedef ffn_f32x2 f32x2;

// Features of K group g for this lane's sample: two frequencies on the packed-f32 pipe.
// f32 VALU work does not overlap f32 MFMA on gfx950 (it is paid in full, and costs about
// twice as much when interleaved with matrix instructions), so features are produced in a
// burst between K loops -- and every instruction counts.  TRIG_ONLY: all four frequencies
// of the group are real ones (no raw-input / padding selects, no branches).
template <bool TRIG_ONLY>
__device__ __forceinline__ f32x4 feature_quad(const EncRegs& enc, int g, int h, float x0, float x1,
                                              float x2, f32x2 s0, f32x2 s1, f32x2 s2) {
    const int k = 4 * g + 2 * h;                                     // even
    const int kk = TRIG_ONLY ? k : (k < kEncRowPitch - 2 ? k : kEncRowPitch - 2);
    const f32x2 b0 = *reinterpret_cast<const f32x2*>(enc.tab + kk);
    const f32x2 b1 = *reinterpret_cast<const f32x2*>(enc.tab + kEncRowPitch + kk);
    const f32x2 b2 = *reinterpret_cast<const f32x2*>(enc.tab + 2 * kEncRowPitch + kk);
    const f32x2 amp = *reinterpret_cast<const f32x2*>(enc.tab + 3 * kEncRowPitch + kk);
    // same operation order as the reference's (scale * x) @ B row: mul, fma, fma
    f32x2 ang = b0 * s0;
    ang = __builtin_elementwise_fma(s1, b1, ang);
    ang = __builtin_elementwise_fma(s2, b2, ang);
    f32x2 sn, cs;
    fast_sincos_n<f32x2, 2>(ang, sn, cs);
    const f32x2 c = amp * cs, s = amp * sn;
    f32x4 v;
    if (TRIG_ONLY) {
        v[0] = c[0]; v[1] = s[0]; v[2] = c[1]; v[3] = s[1];
        return v;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int off = 2 * (k + j - enc.F);   // offset past the trig block
        const float raw_even = (enc.raw && off == 0) ? x0 : ((enc.raw && off == 2) ? x2 : 0.0f);
        const float raw_odd = (enc.raw && off == 0) ? x1 : 0.0f;
        const bool trig = k + j < enc.F;
        v[2 * j] = trig ? c[j] : raw_even;
        v[2 * j + 1] = trig ? s[j] : raw_odd;
    }
    return v;
}

// K groups g and g+1 at once, all four frequencies real: 4-wide source = pairs of independent
// packed instructions.
__device__ __forceinline__ void feature_oct(const EncRegs& enc, int g, int h, f32x4 s0, f32x4 s1,
                                            f32x4 s2, f32x4& va, f32x4& vb) {
    const float* t = enc.tab + 4 * g + 2 * h;
    auto two = [](const float* p) {
        const f32x2 lo = *reinterpret_cast<const f32x2*>(p), hi = *reinterpret_cast<const f32x2*>(p + 4);
        f32x4 v; v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
        return v;
    };
    const f32x4 b0 = two(t), b1 = two(t + kEncRowPitch), b2 = two(t + 2 * kEncRowPitch);
    const f32x4 amp = two(t + 3 * kEncRowPitch);
    f32x4 ang = b0 * s0;
    ang = __builtin_elementwise_fma(s1, b1, ang);
    ang = __builtin_elementwise_fma(s2, b2, ang);
    f32x4 sn, cs;
    fast_sincos_n<f32x4, 4>(ang, sn, cs);
    const f32x4 c = amp * cs, s = amp * sn;
    va[0] = c[0]; va[1] = s[0]; va[2] = c[1]; va[3] = s[1];
    vb[0] = c[2]; vb[1] = s[2]; vb[2] = c[3]; vb[3] = s[3];
}

}  // namespace ffn
