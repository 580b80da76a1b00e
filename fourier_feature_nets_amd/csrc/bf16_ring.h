// Shared machinery of the split-bf16 kernels (mlp_bf16.hip forward, mlp_bf16_bwd.hip backward
// data): the bf16 (hi, lo) split, and the LDS weight ring the four lockstep waves of a workgroup
// read their K blocks from.  See mlp_bf16.hip for the organisation.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace ffn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBlockVecs16 = 1024;                  // one K block of weights: 8 tiles x (hi, lo) x 64 lanes x 16 B
constexpr int kRingBlocks16 = 8;                    // LDS ring: two chunks of four K blocks (128 KiB)
constexpr int OT16 = 8;                             // every step runs eight output tiles (zero-padded packs)

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

// Three-way split: x = hi + mid + lo EXACTLY (8 + 8 + 8 significand bits cover f32's 24; both
// remainders are exact f32 subtractions).  The operands of the f32-accurate "bf16x6" mode.
// On PAIRS of values: one v_cvt_pk_bf16_f32 converts two, and its packed result IS the operand layout
// (element 2 i in the low half of dword i) -- 5.5 vector instructions per value where a conversion per
// value, a shift back and the packing at the end compile to 8.5.  -DFFN_SPLIT_ONE_AT_A_TIME keeps that
// form (A/B; identical bits: the same round-to-nearest-even conversions, the same exact remainders).
__device__ __forceinline__ void split8x3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#ifdef FFN_SPLIT_ONE_AT_A_TIME
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        const float r = x[j] - (float)h;
        const __bf16 m = (__bf16)r;
        hi[j] = h;
        mid[j] = m;
        lo[j] = (__bf16)(r - (float)m);
    }
#else
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f2 v;
        v[0] = x[2 * i]; v[1] = x[2 * i + 1];
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
        f2 r;
        r[0] = v[0] - __builtin_bit_cast(float, h << 16);
        r[1] = v[1] - __builtin_bit_cast(float, h & 0xffff0000u);
        const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, b2));
        f2 q;
        q[0] = r[0] - __builtin_bit_cast(float, m << 16);
        q[1] = r[1] - __builtin_bit_cast(float, m & 0xffff0000u);
        ph[i] = h;
        pm[i] = m;
        pl[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(q, b2));
    }
    hi = __builtin_bit_cast(bf16x8, ph);
    mid = __builtin_bit_cast(bf16x8, pm);
    lo = __builtin_bit_cast(bf16x8, pl);
#endif
}

struct Ring16 {
    int lane, tid;
    f32x4* wbuf;              // LDS: ring of kRingBlocks16 weight K blocks
    const f32x4* gweights;    // all K blocks of the chain, back to back (16 KiB each)
    int total_kb;             // K blocks of the whole chain
    int flat;                 // next K block of the chain (0 .. total_kb-1)
    unsigned ring;            // running K-block counter: ring slot = ring & 7
};

__device__ __forceinline__ void lockstep_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// One K block (ring position g) against the (hi, lo) operand bh / bl.  Its weight operands were
// read from LDS into register set PAR one block ago; while its 24 matrix instructions run,
// (1) the operands of block g+1 stream from LDS into set 1-PAR -- four waves in lockstep read
// 64 KiB per block, ~500 cycles of LDS time that would otherwise sit in front of the matrix pipe
// -- and (2) the weights of block g+5 (cyclically: the next pass starts over) are deposited into
// ring slot (g+5) & 7 (requested from L2 one block earlier; the request for g+6 goes out now:
// two blocks of latency tolerance).  ONE workgroup barrier per four K blocks: a block deposited
// at g is behind a barrier by g+4, i.e. readable during g+4 for g+5, and slot (g+5) & 7 was last
// read during g-4.  Every layer has an even number of K blocks, so PAR is a compile-time
// property of the call site.
template <int PAR>
__device__ __forceinline__ void ring_kblock(Ring16& w, f32x16 (&acc)[OT16], const bf16x8& bh,
                                            const bf16x8& bl, f32x4 (&stage)[2][4],
                                            bf16x8 (&wh)[2][8], bf16x8 (&wl)[2][8]) {
    int ahead = w.flat + 6;
    while (ahead >= w.total_kb) ahead -= w.total_kb;   // (uniform; one pass except in toy chains)
    const f32x4* src = w.gweights + (int64_t)ahead * kBlockVecs16 + w.tid;
#pragma unroll
    for (int i = 0; i < 4; ++i) stage[PAR][i] = src[256 * i];
    const f32x4* nl = w.wbuf + ((w.ring + 1u) & 7u) * kBlockVecs16 + w.lane;
#pragma unroll
    for (int o = 0; o < OT16; ++o) {
        wh[1 - PAR][o] = __builtin_bit_cast(bf16x8, nl[(2 * o) * 64]);
        wl[1 - PAR][o] = __builtin_bit_cast(bf16x8, nl[(2 * o + 1) * 64]);
    }
    // three products, tile-major: consecutive matrix instructions hit different accumulators
#pragma unroll
    for (int o = 0; o < OT16; ++o) acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[PAR][o], bl, acc[o], 0, 0, 0);
#pragma unroll
    for (int o = 0; o < OT16; ++o) acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[PAR][o], bh, acc[o], 0, 0, 0);
#pragma unroll
    for (int o = 0; o < OT16; ++o) acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[PAR][o], bh, acc[o], 0, 0, 0);
    f32x4* dst = w.wbuf + ((w.ring + 5u) & 7u) * kBlockVecs16 + w.tid;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[256 * i] = stage[1 - PAR][i];
    // issue order: one memory instruction behind each matrix instruction -- 16 operand reads,
    // 4 L2 requests, 4 deposits spread over the block's 24 MFMAs (the bf16 matrix pipe and the
    // vector ALU run side by side, but an in-order wave only overlaps what is interleaved in
    // program order).  Left to itself hipcc clusters the 16 ds_read_b128 in front of the MFMAs,
    // and a wave that spends ~200 cycles issuing LDS reads lets the matrix pipe run dry.
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (i >= 16 && i < 20) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    if ((w.ring & 3u) == 3u) lockstep_barrier();
    w.ring += 1u;
    w.flat = w.flat + 1 < w.total_kb ? w.flat + 1 : 0;
}

// Fills ring slots 0..4 with the chain's first five K blocks, requests the sixth and reads the
// operands of the first: the state ring_kblock expects at ring position 0.
__device__ __forceinline__ void ring_prime(Ring16& w, f32x4 (&stage)[2][4], bf16x8 (&wh)[2][8],
                                           bf16x8 (&wl)[2][8]) {
    w.flat = 0;
    w.ring = 0u;
    for (int b = 0; b < 5; ++b) {
        const int fb = b % w.total_kb;
        for (int i = 0; i < 4; ++i)
            w.wbuf[b * kBlockVecs16 + w.tid + 256 * i] = w.gweights[(int64_t)fb * kBlockVecs16 + w.tid + 256 * i];
    }
    __syncthreads();
    {   // the first block deposits what "the block before it" requested: block 5's weights
        const int fb = 5 % w.total_kb;
        for (int i = 0; i < 4; ++i) stage[1][i] = w.gweights[(int64_t)fb * kBlockVecs16 + w.tid + 256 * i];
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        wh[0][o] = __builtin_bit_cast(bf16x8, w.wbuf[(2 * o) * 64 + w.lane]);
        wl[0][o] = __builtin_bit_cast(bf16x8, w.wbuf[(2 * o + 1) * 64 + w.lane]);
    }
}

// ---------------------------------------------------------------------------------- encoding features (shared by mlp_bf16.hip and mlp_bf16_ws.hip)
struct Enc16 {
    const float* tab;   // LDS: rows b0 | b1 | b2 | a, kEncRowPitch floats each
    int F, raw;
    float scale;
};

// The 8 internal feature channels 16*G + 8*h + j (j = 0..7) of this lane's sample: frequencies
// 8G + 4h + {0,1,2,3}, (cos, sin) interleaved; channels 2F..2F+2 are the raw inputs.
template <bool TRIG_ONLY>
__device__ __forceinline__ void features16(const Enc16& enc, int G, int h, float x0, float x1,
                                           float x2, float (&v)[8]) {
    const int k0 = 8 * G + 4 * h;
    const int kk = k0 < kEncRowPitch - 4 ? k0 : kEncRowPitch - 4;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(enc.tab + kk);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(enc.tab + kEncRowPitch + kk);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(enc.tab + 2 * kEncRowPitch + kk);
    const f32x4 amp = *reinterpret_cast<const f32x4*>(enc.tab + 3 * kEncRowPitch + kk);
    const f32x4 s0 = (f32x4)(enc.scale * x0), s1 = (f32x4)(enc.scale * x1), s2 = (f32x4)(enc.scale * x2);
    // same operation order as the f32 kernel: mul, fma, fma
    f32x4 ang = b0 * s0;
    ang = __builtin_elementwise_fma(s1, b1, ang);
    ang = __builtin_elementwise_fma(s2, b2, ang);
    f32x4 sn, cs;
    fast_sincos_n<f32x4, 4>(ang, sn, cs);
    const f32x4 c = amp * cs, s = amp * sn;
    if (TRIG_ONLY) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = c[i]; v[2 * i + 1] = s[i]; }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + i;
        const int off = 2 * (k - enc.F);
        const float raw_even = (enc.raw && off == 0) ? x0 : ((enc.raw && off == 2) ? x2 : 0.0f);
        const float raw_odd = (enc.raw && off == 0) ? x1 : 0.0f;
        const bool trig = k < enc.F;
        v[2 * i] = trig ? c[i] : raw_even;
        v[2 * i + 1] = trig ? s[i] : raw_odd;
    }
}

// The same eight features with UNPACKED f32 arithmetic (one v_fma_f32 per component instead of
// v_pk_fma_f32; identical rounding, identical bits).  For kernels that run two waves per SIMD:
// a packed-f32 instruction occupies the datapath the matrix instructions of the co-resident wave
// run on (MI355X_MICROARCH.md: "+22 cycles per v_pk_fma_f32 beside MFMAs"), a plain one issues in
// the gaps between them.  Compile the including file with -fno-slp-vectorize, or the compiler
// re-packs the components.
template <bool TRIG_ONLY>
__device__ __forceinline__ void features16_unpacked(const Enc16& enc, int G, int h, float x0, float x1,
                                                    float x2, float (&v)[8]) {
    const int k0 = 8 * G + 4 * h;
    const int kk = k0 < kEncRowPitch - 4 ? k0 : kEncRowPitch - 4;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(enc.tab + kk);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(enc.tab + kEncRowPitch + kk);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(enc.tab + 2 * kEncRowPitch + kk);
    const f32x4 amp = *reinterpret_cast<const f32x4*>(enc.tab + 3 * kEncRowPitch + kk);
    const float s0 = enc.scale * x0, s1 = enc.scale * x1, s2 = enc.scale * x2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ang = b0[i] * s0;                    // same operation order as the f32 kernel
        ang = __builtin_fmaf(s1, b1[i], ang);
        ang = __builtin_fmaf(s2, b2[i], ang);
        float sn, cs;
        fast_sincos(ang, sn, cs);
        const float c = amp[i] * cs, s = amp[i] * sn;
        if (TRIG_ONLY) {
            v[2 * i] = c;
            v[2 * i + 1] = s;
        } else {
            const int k = k0 + i;
            const int off = 2 * (k - enc.F);
            const float raw_even = (enc.raw && off == 0) ? x0 : ((enc.raw && off == 2) ? x2 : 0.0f);
            const float raw_odd = (enc.raw && off == 0) ? x1 : 0.0f;
            const bool trig = k < enc.F;
            v[2 * i] = trig ? c : raw_even;
            v[2 * i + 1] = trig ? s : raw_odd;
        }
    }
}

// features16_unpacked with the four angles walked LEVEL BY LEVEL (the same operations in the same order per
// angle: identical bits): four independent dependency chains in flight instead of one after the other.  A
// wave that is alone with its vector work (the vector waves of mlp_bf16_mv.hip) issues a lone dependent
// chain every ~8.5 cycles, independent instructions every ~5.
template <bool TRIG_ONLY>
__device__ __forceinline__ void features16_lockstep(const Enc16& enc, int G, int h, float x0, float x1,
                                                    float x2, float (&v)[8]) {
    const int k0 = 8 * G + 4 * h;
    const int kk = k0 < kEncRowPitch - 4 ? k0 : kEncRowPitch - 4;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(enc.tab + kk);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(enc.tab + kEncRowPitch + kk);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(enc.tab + 2 * kEncRowPitch + kk);
    const f32x4 amp = *reinterpret_cast<const f32x4*>(enc.tab + 3 * kEncRowPitch + kk);
    const float s0 = enc.scale * x0, s1 = enc.scale * x1, s2 = enc.scale * x2;
    float ang[4], k[4], r[4], z[4], sp[4], cp[4], hf[4], sn[4], cs[4];
#define FFN_L4 _Pragma("unroll") for (int i = 0; i < 4; ++i)
    FFN_L4 ang[i] = b0[i] * s0;                     // same operation order as the f32 kernel
    FFN_L4 ang[i] = __builtin_fmaf(s1, b1[i], ang[i]);
    FFN_L4 ang[i] = __builtin_fmaf(s2, b2[i], ang[i]);
    FFN_L4 k[i] = __builtin_rintf(ang[i] * 0.6366197466850281f);      // fast_sincos, level by level
    FFN_L4 r[i] = __builtin_fmaf(-k[i], 1.5707963705062866f, ang[i]);
    FFN_L4 r[i] = __builtin_fmaf(-k[i], -4.371138828673793e-08f, r[i]);
    FFN_L4 r[i] = __builtin_fmaf(-k[i], -1.7151245100058819e-15f, r[i]);
    FFN_L4 z[i] = r[i] * r[i];
    FFN_L4 sp[i] = __builtin_fmaf(-1.9515295891e-4f, z[i], 8.3321608736e-3f);
    FFN_L4 cp[i] = __builtin_fmaf(2.443315711809948e-5f, z[i], -1.388731625493765e-3f);
    FFN_L4 sp[i] = __builtin_fmaf(sp[i], z[i], -1.6666654611e-1f);
    FFN_L4 cp[i] = __builtin_fmaf(cp[i], z[i], 4.166664568298827e-2f);
    FFN_L4 hf[i] = __builtin_fmaf(-0.5f, z[i], 1.0f);
    FFN_L4 sp[i] = __builtin_fmaf(sp[i] * z[i], r[i], r[i]);
    FFN_L4 cp[i] = __builtin_fmaf(cp[i] * z[i], z[i], hf[i]);
    FFN_L4 {
        const int q = (int)k[i];
        const bool swap = (q & 1) != 0;
        const float s_0 = swap ? cp[i] : sp[i];
        const float c_0 = swap ? sp[i] : cp[i];
        sn[i] = (q & 2) ? -s_0 : s_0;
        cs[i] = ((q + 1) & 2) ? -c_0 : c_0;
    }
    FFN_L4 {
        const float c = amp[i] * cs[i], s = amp[i] * sn[i];
        if (TRIG_ONLY) {
            v[2 * i] = c;
            v[2 * i + 1] = s;
        } else {
            const int kf = k0 + i;
            const int off = 2 * (kf - enc.F);
            const float raw_even = (enc.raw && off == 0) ? x0 : ((enc.raw && off == 2) ? x2 : 0.0f);
            const float raw_odd = (enc.raw && off == 0) ? x1 : 0.0f;
            const bool trig = kf < enc.F;
            v[2 * i] = trig ? c : raw_even;
            v[2 * i + 1] = trig ? s : raw_odd;
        }
    }
#undef FFN_L4
}

// Hardware sin / cos for the split-bf16 kernels: exact two-constant reduction by 2 pi (the angle is
// the f32 kernels' angle, bit for bit), then v_sin_f32 / v_cos_f32 on the remainder in revolutions
// -- 7 instructions per angle where the polynomial pair with its quadrant logic takes ~25, on the
// transcendental unit instead of the FMA lanes.  Max abs error 2.4e-7 for |x| <= 5000 (the
// polynomials: 8.8e-8): far below the 1.5e-5 relative precision of a (hi, lo) bf16 pair, which is
// why only this opt-in mode uses it.
__device__ __forceinline__ void hw_sincos(float x, float& sn, float& cs) {
    const float k = __builtin_rintf(x * 0.15915494309189535f);
    float r = __builtin_fmaf(-k, 6.2831854820251465f, x);
    r = __builtin_fmaf(-k, -1.7484555314695172e-07f, r);
    const float t = r * 0.15915494309189535f;          // |t| <= 0.5 revolutions
    sn = __builtin_amdgcn_sinf(t);
    cs = __builtin_amdgcn_cosf(t);
}

template <bool TRIG_ONLY>
__device__ __forceinline__ void features16_hw(const Enc16& enc, int G, int h, float x0, float x1,
                                              float x2, float (&v)[8]) {
    const int k0 = 8 * G + 4 * h;
    const int kk = k0 < kEncRowPitch - 4 ? k0 : kEncRowPitch - 4;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(enc.tab + kk);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(enc.tab + kEncRowPitch + kk);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(enc.tab + 2 * kEncRowPitch + kk);
    const f32x4 amp = *reinterpret_cast<const f32x4*>(enc.tab + 3 * kEncRowPitch + kk);
    const float s0 = enc.scale * x0, s1 = enc.scale * x1, s2 = enc.scale * x2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ang = b0[i] * s0;                    // same operation order as the f32 kernel
        ang = __builtin_fmaf(s1, b1[i], ang);
        ang = __builtin_fmaf(s2, b2[i], ang);
        float sn, cs;
        hw_sincos(ang, sn, cs);
        const float c = amp[i] * cs, s = amp[i] * sn;
        if (TRIG_ONLY) {
            v[2 * i] = c;
            v[2 * i + 1] = s;
        } else {
            const int k = k0 + i;
            const int off = 2 * (k - enc.F);
            const float raw_even = (enc.raw && off == 0) ? x0 : ((enc.raw && off == 2) ? x2 : 0.0f);
            const float raw_odd = (enc.raw && off == 0) ? x1 : 0.0f;
            const bool trig = k < enc.F;
            v[2 * i] = trig ? c : raw_even;
            v[2 * i + 1] = trig ? s : raw_odd;
        }
    }
}

// The two-waves-per-SIMD organisation of the same chains (mlp_bf16_ws.hip): same packs, same
// slab / mask formats.  FFN_BF16_KERNELS=ring|ws overrides the per-chain default (A/B).
int launch_forward16_ws(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                        const float* positions, const float* views, int64_t n, float* logits,
                        float* saved, uint32_t* masks, void* stream);
int launch_backward16_ws(const ffn_mlp_chain* chain, const uint16_t* packed_wt, const float* d_logits,
                         int64_t n, const uint32_t* masks, float* dz, void* stream);
// The f32-accurate split mode ("bf16x6": three bf16 parts per operand, six partial products per
// f32 product, f32 accumulation; mlp_bf16_ws.hip).  products = 6 or 9 (9 = every partial product:
// measurement only).
int launch_forward_bf16x6(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                          const float* positions, const float* views, int64_t n, float* logits,
                          float* saved, uint32_t* masks, void* stream);
int launch_backward_bf16x6(const ffn_mlp_chain* chain, const uint16_t* packed_wt, const float* d_logits,
                           int64_t n, const uint32_t* masks, float* dz, void* stream);
// The matrix-waves / vector-waves organisation of the bf16x6 chains (mlp_bf16_mv.hip) for the chains it
// covers (mv_covers: a features-only first step, then 256 -> 256 steps); FFN_BF16X6_ORG=ws keeps the
// two-waves-per-SIMD kernels for them too (A/B).
bool mv_covers(const ffn_mlp_chain* chain, int* num_units);
int launch_forward_bf16x6_mv(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                             const float* positions, const float* views, int64_t n, float* logits,
                             float* saved, uint32_t* masks, int num_units, void* stream);
bool mv_covers_bwd(const ffn_mlp_chain* chain, int* num_units);
int launch_backward_bf16x6_mv(const ffn_mlp_chain* chain, const uint16_t* packed_wt, const float* d_logits,
                              int64_t n, const uint32_t* masks, float* dz, int num_units, void* stream);
inline bool bf16x6_prefers_mv() {
    const char* v = getenv("FFN_BF16X6_ORG");
    return !(v != nullptr && v[0] == 'w');
}
inline bool prefer_ws_kernels(bool by_default) {     // read per launch: tests flip it inside one process
    const char* v = getenv("FFN_BF16_KERNELS");
    if (v != nullptr && v[0] == 'r') return false;
    if (v != nullptr && v[0] == 'w') return true;
    return by_default;
}
// FFN_BF16_SINCOS=poly: the two-waves-per-SIMD kernels generate the encoding features with the
// f32 kernels' polynomials (bit-identical features; tests) instead of v_sin_f32 / v_cos_f32
inline bool use_poly_sincos() {
    const char* v = getenv("FFN_BF16_SINCOS");
    return v != nullptr && v[0] == 'p';
}

// float4 index of (channel quad cq, sample s) inside a saved-activation block: mlp.hip's layout
__device__ __forceinline__ int saved_index16(int cq, int s) { return cq * 32 + (s ^ (cq & 15)); }

}  // namespace ffn
