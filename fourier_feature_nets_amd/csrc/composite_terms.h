// Per-sample activation terms and the wavefront scans of front-to-back compositing
// (ray_caster.py:66-93, utils.py:72-97), shared by the standalone composite kernels
// (composite.hip) and the fused render kernel (mlp.hip).
#pragma once
#include "common.h"

namespace ffn {

__device__ __forceinline__ float softplus_torch(float x) {
    // F.softplus, beta = 1, threshold = 20
    return x > 20.0f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Cross-lane scans on the DPP path (row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast15 / row_bcast31 across rows): six VALU instructions with a DPP modifier instead of
// six ds_bpermute round trips through the LDS pipeline (~100+ cycles each, serially dependent).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                  __builtin_bit_cast(int, src), CTRL,
                                                                  ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u(unsigned old, unsigned src) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false);
}
constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118;
constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143, kWaveShr1 = 0x138;

// inclusive multiplicative scan over the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_mul(float v, int) {
    v *= dpp_f<kRowShr1, 0xf>(1.0f, v);
    v *= dpp_f<kRowShr2, 0xf>(1.0f, v);
    v *= dpp_f<kRowShr4, 0xf>(1.0f, v);
    v *= dpp_f<kRowShr8, 0xf>(1.0f, v);
    v *= dpp_f<kRowBcast15, 0xa>(1.0f, v);
    v *= dpp_f<kRowBcast31, 0xc>(1.0f, v);
    return v;
}
// value of lane i-1 (lane 0 gets `first`)
__device__ __forceinline__ float wave_shift_up(float v, float first) { return dpp_f<kWaveShr1, 0xf>(first, v); }
__device__ __forceinline__ float wave_last(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// inclusive additive suffix scan (lane i gets sum over lanes >= i)
__device__ __forceinline__ float wave_suffix_add(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float dn = __shfl_down(v, off, 64);
        if (lane + off < 64) v += dn;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<kRowShr1, 0xf>(0.0f, v);
    v += dpp_f<kRowShr2, 0xf>(0.0f, v);
    v += dpp_f<kRowShr4, 0xf>(0.0f, v);
    v += dpp_f<kRowShr8, 0xf>(0.0f, v);
    v += dpp_f<kRowBcast15, 0xa>(0.0f, v);
    v += dpp_f<kRowBcast31, 0xc>(0.0f, v);
    return wave_last(v);
}
// max over the wave of the 64-bit key (hi, lo), returned in every lane
__device__ __forceinline__ void wave_max_key(unsigned& hi, unsigned& lo) {
#define FFN_KEY_STEP(CTRL, MASK)                                                               \
    {                                                                                          \
        const unsigned oh = dpp_u<CTRL, MASK>(0u, hi), ol = dpp_u<CTRL, MASK>(0u, lo);         \
        const bool take = oh > hi || (oh == hi && ol > lo);                                    \
        hi = take ? oh : hi;                                                                   \
        lo = take ? ol : lo;                                                                   \
    }
    FFN_KEY_STEP(kRowShr1, 0xf)
    FFN_KEY_STEP(kRowShr2, 0xf)
    FFN_KEY_STEP(kRowShr4, 0xf)
    FFN_KEY_STEP(kRowShr8, 0xf)
    FFN_KEY_STEP(kRowBcast15, 0xa)
    FFN_KEY_STEP(kRowBcast31, 0xc)
#undef FFN_KEY_STEP
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}

struct SampleTerms {
    float r, g, b;      // sigmoid(rgb logits)
    float sigma_logit;  // raw
    float delta, e, alpha, u, tau;
};

// Terms of one sample from its raw logits, its t and the t of the NEXT sample of the ray
// (`last` = it is the ray's final sample: delta = 1e10).
__device__ __forceinline__ SampleTerms make_terms(float4 l, float t_here, float t_next, bool last,
                                                   bool active, int32_t* nan_flag) {
    SampleTerms o;
    if (!active) {
        o.r = o.g = o.b = 0.f; o.sigma_logit = 0.f; o.delta = 0.f; o.e = 1.f; o.alpha = 0.f;
        o.u = 1.f; o.tau = 1.f;
        return o;
    }
    o.r = sigmoid_f(l.x); o.g = sigmoid_f(l.y); o.b = sigmoid_f(l.z);
    o.sigma_logit = l.w;
    const float sigma = softplus_torch(l.w);
    if (nan_flag != nullptr && (o.r != o.r || o.g != o.g || o.b != o.b || sigma != sigma))
        atomicOr(nan_flag, 1);
    o.delta = last ? 1e10f : t_next - t_here;
    o.e = expf(-(sigma * o.delta));
    o.alpha = 1.0f - o.e;
    o.u = (1.0f - o.alpha) + 1e-10f;
    o.tau = o.u < 1.0f ? o.u : 1.0f;
    return o;
}

__device__ __forceinline__ SampleTerms load_terms(const float4* __restrict__ logits,
                                                   const float* __restrict__ t, int s, int S,
                                                   bool active, int32_t* nan_flag) {
    if (!active) return make_terms(make_float4(0.f, 0.f, 0.f, 0.f), 0.f, 0.f, false, false, nullptr);
    const bool last = s == S - 1;
    return make_terms(logits[s], t[s], last ? 0.0f : t[s + 1], last, true, nan_flag);
}

// Running state of one ray's front-to-back reduction: `row` consumes 64 lane-ordered samples
// (lane i = the i-th of them; `index` = the sample's position on the ray, `inner` = it counts
// towards alpha / depth, i.e. it is not the ray's last sample), `finish` reduces over the wave.
struct RayAccum {
    float carry, cr, cg, cb, asum, best_w;
    int best_s;
    __device__ __forceinline__ void reset() {
        carry = 1.0f; cr = cg = cb = asum = 0.0f;
        best_w = -1.0f;       // weights are >= 0, so -1 means "none yet"
        best_s = 0;
    }
    // Returns the sample's transmittance T.  The roundings are spelled out (no contraction left to
    // the compiler's context-dependent choice): the weight w = alpha * T is a rounded product --
    // `asum + alpha * T` as one fused operation differs in the last bit once a ray has more than
    // one row -- and the colour sums take w * c with one rounding.  Every kernel that composites
    // (K5, the fused render, the training composite K5t) therefore produces the same bits.
    __device__ __forceinline__ float row(const SampleTerms& q, int lane, int index, bool inner) {
#pragma clang fp contract(off)
        const float incl = wave_scan_mul(q.tau, lane);
        const float excl = wave_shift_up(incl, 1.0f);
        const float T = carry * excl;
        const float w = q.alpha * T;
        cr = __builtin_fmaf(w, q.r, cr); cg = __builtin_fmaf(w, q.g, cg); cb = __builtin_fmaf(w, q.b, cb);
        if (inner) {
            asum = asum + w;
            if (w > best_w) { best_w = w; best_s = index; }
        }
        carry *= wave_last(incl);
        return T;
    }
    // after this every lane holds the ray's colour / alpha; best_s / best_w the depth pick
    __device__ __forceinline__ void finish() {
        cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb); asum = wave_sum(asum);
        // argmax with first-occurrence tie break: weights are >= 0, so their bit patterns order
        // like unsigned integers; the low word prefers the smaller sample index, 0 = "none"
        unsigned key_hi = best_w < 0.0f ? 0u : __builtin_bit_cast(unsigned, best_w);
        unsigned key_lo = best_w < 0.0f ? 0u : 0xffffffffu - (unsigned)best_s;
        wave_max_key(key_hi, key_lo);
        best_w = key_lo == 0u ? -1.0f : __builtin_bit_cast(float, key_hi);
        best_s = key_lo == 0u ? 0 : (int)(0xffffffffu - key_lo);
    }
    // index of the sample whose t is the ray's depth (ray_caster.py:85-89)
    __device__ __forceinline__ int depth_pick(int S) const {
        return (asum < 0.1f || best_w < 0.0f) ? S - 1 : best_s;
    }
};

}  // namespace ffn
