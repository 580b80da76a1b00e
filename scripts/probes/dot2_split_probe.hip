// Probe (gfx950): is  x - float(bf16_rne(x))  computed by v_dot2c_f32_bf16 (accumulator x, operands the packed
// pair of parts and the constant {-1, 0} / {0, -1}) BIT-IDENTICAL to the shift / mask + v_sub_f32 form, through
// all three stages of the three-way split, for every class of input (normal, tiny, denormal, huge, signed
// zeros, values whose parts are denormal)?  Prints mismatch counts per class.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/dot2_split_probe.hip -o scripts/probes/dot2_split_probe && scripts/probes/dot2_split_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__global__ void split_both(const float* in, uint32_t* ref, uint32_t* dot, int n_pairs, uint32_t even_bits, uint32_t odd_bits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    float a0 = in[2 * i], a1 = in[2 * i + 1];
    float b0 = a0, b1 = a1;
    // (kernel arguments, not literals: hipcc folds 0x0000bf80 into the INLINE constant -1.0, which this
    // instruction does not read as the packed pair {-1, 0} -- first run of this probe)
    const bf16x2 neg_even = __builtin_bit_cast(bf16x2, even_bits);
    const bf16x2 neg_odd = __builtin_bit_cast(bf16x2, odd_bits);
#pragma unroll
    for (int stage = 0; stage < 3; ++stage) {
        f32x2v pa, pb;
        pa[0] = a0; pa[1] = a1;
        pb[0] = b0; pb[1] = b1;
        const unsigned ha = __builtin_bit_cast(unsigned, __builtin_convertvector(pa, bf16x2));
        const bf16x2 hb2 = __builtin_convertvector(pb, bf16x2);
        const unsigned hb = __builtin_bit_cast(unsigned, hb2);
        ref[(int64_t)(3 * i + stage) * 3 + 0] = ha;
        dot[(int64_t)(3 * i + stage) * 3 + 0] = hb;
        a0 = a0 - __builtin_bit_cast(float, ha << 16);
        a1 = a1 - __builtin_bit_cast(float, ha & 0xffff0000u);
        b0 = __builtin_amdgcn_fdot2_f32_bf16(hb2, neg_even, b0, false);
        b1 = __builtin_amdgcn_fdot2_f32_bf16(hb2, neg_odd, b1, false);
        ref[(int64_t)(3 * i + stage) * 3 + 1] = __builtin_bit_cast(unsigned, a0);
        ref[(int64_t)(3 * i + stage) * 3 + 2] = __builtin_bit_cast(unsigned, a1);
        dot[(int64_t)(3 * i + stage) * 3 + 1] = __builtin_bit_cast(unsigned, b0);
        dot[(int64_t)(3 * i + stage) * 3 + 2] = __builtin_bit_cast(unsigned, b1);
    }
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

int main() {
    const int per_class = 1 << 20;
    const char* names[] = {"uniform bits (all exponents, no inf/nan)", "normal |x| in 2^-20..2^4", "tiny 2^-126..2^-100",
                           "denormal", "huge 2^100..2^127", "few mantissa bits (parts hit zero)", "zeros and signs"};
    const int classes = 7;
    std::vector<float> h((size_t)classes * per_class);
    for (int c = 0; c < classes; ++c)
        for (int k = 0; k < per_class; ++k) {
            uint32_t m = rnd() & 0x7fffffu, s = rnd() & 0x80000000u, e;
            switch (c) {
                case 0: e = rnd() % 255; break;
                case 1: e = 107 + rnd() % 25; break;
                case 2: e = 1 + rnd() % 27; break;
                case 3: e = 0; break;
                case 4: e = 227 + rnd() % 28; break;
                case 5: e = 100 + rnd() % 50; m &= 0x7f0000u >> (rnd() % 8); break;
                default: e = 0; m = (rnd() & 3) == 0 ? 1 : 0; break;
            }
            const uint32_t bits = s | (e << 23) | m;
            std::memcpy(&h[(size_t)c * per_class + k], &bits, 4);
        }
    const int n = classes * per_class, n_pairs = n / 2;
    float* d_in; uint32_t *d_ref, *d_dot;
    hipMalloc(&d_in, (size_t)n * 4);
    hipMalloc(&d_ref, (size_t)n_pairs * 9 * 4);
    hipMalloc(&d_dot, (size_t)n_pairs * 9 * 4);
    hipMemcpy(d_in, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_both, dim3((n_pairs + 255) / 256), dim3(256), 0, 0, d_in, d_ref, d_dot, n_pairs, 0x0000bf80u, 0xbf800000u);
    std::vector<uint32_t> r((size_t)n_pairs * 9), d((size_t)n_pairs * 9);
    hipMemcpy(r.data(), d_ref, r.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), d_dot, d.size() * 4, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    for (int c = 0; c < classes; ++c) {
        long bad_part = 0, bad_rem = 0, shown = 0;
        for (int p = c * (per_class / 2); p < (c + 1) * (per_class / 2); ++p)
            for (int st = 0; st < 3; ++st) {
                const size_t o = ((size_t)3 * p + st) * 3;
                if (r[o] != d[o]) ++bad_part;
                for (int k = 1; k < 3; ++k)
                    if (r[o + k] != d[o + k]) {
                        ++bad_rem;
                        if (shown++ < 3) {
                            uint32_t xb; std::memcpy(&xb, &h[2 * (size_t)p + k - 1], 4);
                            printf("   x=%08x stage %d: sub %08x dot2 %08x\n", xb, st, r[o + k], d[o + k]);
                        }
                    }
            }
        printf("%-45s parts differing %ld  remainders differing %ld  (of %d values x 3 stages)\n", names[c], bad_part, bad_rem, per_class);
    }
    return 0;
}
