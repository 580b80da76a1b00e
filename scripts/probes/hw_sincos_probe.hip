// RESULT (MI355X): the transcendental unit is good to 1.25e-7 of the exact function of its f32
// argument; a Cody-Waite reduction by pi + v_sin / v_cos gives 2.2e-7 max abs error for |x| <= 1000
// (polynomial fast_sincos_n: 9.2e-8) at 10 instead of ~20 instructions per angle -- and, built into
// the fused forward kernel, only -0.75 % of its time (17.88 -> 17.75 ms: the quarter-rate
// transcendentals give most of the saving back).  Not adopted.
// How accurate are the hardware v_sin_f32 / v_cos_f32 (input in revolutions) on gfx950?  Max / mean
// abs error against double-precision sin/cos over [-pi, pi] (uniform grid + points near the axes),
// next to the library's polynomial fast_sincos.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include "common.h"
namespace ffn {
void set_error(const char*, hipError_t) {}
// The same N angles through the hardware's v_sin_f32 / v_cos_f32 (argument in revolutions):
// Cody-Waite reduction by pi (3 constants, fused multiply-adds: exact to ~1e-12 for |x| <= 5000),
// r / 2pi in [-1/4, 1/4] to the transcendental unit, sign = parity of the multiple of pi.
// 10 instructions per angle where the polynomial version needs ~20 (no quadrant selects, no
// polynomials).  Max abs error 2.2e-7 for |x| <= 1000 (scripts/probes/hw_sincos_probe.hip: the
// unit itself is good to 1.25e-7 of the exact function of its f32 argument, the rounding of
// r / 2pi adds up to 9.4e-8 rad) against 9.2e-8 for the polynomial version.
template <typename V, int N>
__device__ __forceinline__ void hw_sincos_n(V x, V& sn, V& cs) {
    const V t = x * 0.3183098861837907f;
    V k;
#pragma unroll
    for (int j = 0; j < N; ++j) k[j] = __builtin_rintf(t[j]);
    const V nk = -k;
    V r = __builtin_elementwise_fma(nk, (V)(3.1415927410125732f), x);
    r = __builtin_elementwise_fma(nk, (V)(-8.742277657347586e-08f), r);
    r = __builtin_elementwise_fma(nk, (V)(-3.4302490200117637e-15f), r);
    const V rev = r * 0.15915494309189535f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const unsigned flip = (unsigned)(int)k[j] << 31;
        sn[j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, __builtin_amdgcn_sinf(rev[j])) ^ flip);
        cs[j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, __builtin_amdgcn_cosf(rev[j])) ^ flip);
    }
}

}  // namespace ffn

__global__ void probe(int n, double* err_hw, double* err_poly, double* err_hw2) {
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float r = (float)(-3.14159265358979 + 6.28318530717958 * ((double)i + 0.37) / n);
        const float rev = r * 0.15915494309189535f;
        const float s_hw = __builtin_amdgcn_sinf(rev), c_hw = __builtin_amdgcn_cosf(rev);
        float s_p, c_p;
        ffn::fast_sincos(r, s_p, c_p);
        const double s = sin((double)r), c = cos((double)r);
        m0 = fmax(m0, fmax(fabs(s_hw - s), fabs(c_hw - c)));
        m1 = fmax(m1, fmax(fabs(s_p - s), fabs(c_p - c)));
        // revolutions computed in higher precision (what a two-step reduction could give)
        const double revd = (double)r * 0.15915494309189535;
        const float rev_hi = (float)revd;
        m2 = fmax(m2, fmax(fabs(__builtin_amdgcn_sinf(rev_hi) - sin(6.283185307179586 * (double)rev_hi)),
                           fabs(__builtin_amdgcn_cosf(rev_hi) - cos(6.283185307179586 * (double)rev_hi))));
        // the whole reduction + unit pipeline on a wide range (the encodings stay below ~900 rad)
        const float big = r * (1000.0f / 3.14159265f);
        ffn::ffn_f32x2 xv, sv, cv;
        xv[0] = big; xv[1] = -0.37f * big;
        ffn::hw_sincos_n<ffn::ffn_f32x2, 2>(xv, sv, cv);
        for (int j = 0; j < 2; ++j)
            m3 = fmax(m3, fmax(fabs(sv[j] - sin((double)xv[j])), fabs(cv[j] - cos((double)xv[j]))));
        ffn::fast_sincos_n<ffn::ffn_f32x2, 2>(xv, sv, cv);
        for (int j = 0; j < 2; ++j)
            m4 = fmax(m4, fmax(fabs(sv[j] - sin((double)xv[j])), fabs(cv[j] - cos((double)xv[j]))));
    }
    err_hw[blockIdx.x * blockDim.x + threadIdx.x] = m0;
    err_poly[blockIdx.x * blockDim.x + threadIdx.x] = m1;
    err_hw2[blockIdx.x * blockDim.x + threadIdx.x] = m2;
    err_hw2[gridDim.x * blockDim.x + blockIdx.x * blockDim.x + threadIdx.x] = m3;
    err_hw2[2 * gridDim.x * blockDim.x + blockIdx.x * blockDim.x + threadIdx.x] = m4;
}

int main() {
    const int T = 256 * 1024;
    double *a, *b, *c;
    hipMalloc(&a, T * 8); hipMalloc(&b, T * 8); hipMalloc(&c, 3 * T * 8);
    probe<<<1024, 256>>>(1 << 28, a, b, c);
    static double ha[T], hb[T], hc[3 * T];
    hipMemcpy(ha, a, T * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, b, T * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hc, c, 3 * T * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < T; ++i) { m0 = fmax(m0, ha[i]); m1 = fmax(m1, hb[i]); m2 = fmax(m2, hc[i]); }
    double m3 = 0, m4 = 0;
    for (int i = 0; i < T; ++i) { m3 = fmax(m3, hc[T + i]); m4 = fmax(m4, hc[2 * T + i]); }
    printf("|x| <= 1000: hw_sincos_n %.3e   fast_sincos_n %.3e\n", m3, m4);
    printf("max abs error over [-pi, pi], 2^28 points: v_sin/v_cos(r/2pi) %.3e   polynomial fast_sincos %.3e   v_sin/v_cos vs exact function of their own f32 argument %.3e\n", m0, m1, m2);
    return 0;
}
