// Pieces shared by the exact-f32 weight-gradient kernel (wgrad.hip) and its split-bf16 variant
// (wgrad_bf16.hip): the LDS map of a unit, the partial format, and the logits-head unit.
#pragma once
#include "common.h"

namespace ffn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPartialFloats = 16 * 16 * 64 + 256;  // 16 tiles + bias strip

// fold of a narrow input window in the exact-f32 unit kernel (see wgrad.hip); the host passes the
// same value to the reducer in ffn_reduce_job.n_fold
__host__ __device__ __forceinline__ constexpr int ffn_wgrad_fold(int n_quads) {
    return n_quads <= 8 ? 4 : (n_quads <= 16 ? 2 : 1);
}

__device__ __forceinline__ f32x4 zero4() { f32x4 z; z[0] = z[1] = z[2] = z[3] = 0.0f; return z; }

// tells the compiler a pointer is wave-uniform (so that it lives in SGPRs and loads through it
// use scalar-base addressing)
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

// ---------------------------------------------------------------------------------- units
// A 256-thread workgroup owns one unit (dZ slab window of <=256 channels  x  input slab
// window of <=256 channels) over a range of sample blocks.  Per block the two operand images
// (<= 32 KiB each) are copied into LDS exactly as they sit in HBM, double buffered; each wave
// computes a 128x128 quadrant from LDS with two conflict-free ds_read_b128 per 16 MFMAs.
// HBM/L2 traffic = the unique operand bytes.
//
// Slab traffic is streamed ONCE (written by the forward / backward-data kernels, read here): the
// loads are non-temporal and so are the producers' stores, which keeps the operand packs of the
// whole network -- the only reused data -- in the XCDs' L2 (measured: exact-f32 step -1.0 %,
// split-bf16 step -2.4 %).
//
// The copy goes through registers: block b+2 is requested (16 global_load_dwordx4 per lane,
// spread over the middle steps of block b) into 64 otherwise idle VGPRs, and deposited into
// the free LDS buffer during the first steps of block b+1 -- more than a block of latency
// tolerance, and no issue slot that is not under a running MFMA.  (global_load_lds was
// measured at ~130 exposed cycles per instruction in this loop: 9 % of the kernel.)
//
// A window of <=128 channels has only one quadrant along that side; the waves that would own
// the missing quadrants split the block's 16 sample pairs with their siblings instead (so a
// 256x128 unit costs half, a 128x128 unit a quarter, of a full one).
// LDS map of the unit kernel: four operand images, each followed by a 512-B row of zeros --
//   [A even | 0][A odd | 0][B even | 0][B odd | 0],  kImageStride = 32 KiB + 512 B apart.
// Idle lanes of a narrow window read "their" image's zero row instead of branching or
// selecting; and because the even / odd block buffers of an operand are one constant apart,
// the buffer toggle of the double buffering rides in the ds_read immediate offset.
constexpr int kImageBytes = 32 * 1024;
constexpr int kImageStride = kImageBytes + 512;
constexpr int kUnitLdsBytes = 4 * kImageStride;
__device__ __forceinline__ constexpr int image_a(int cur) { return cur * kImageStride; }
__device__ __forceinline__ constexpr int image_b(int cur) { return (2 + cur) * kImageStride; }

// Logits-head rows inside the LDS-staged kernel: dW_head[j][k] = sum_s dl[s][j] * X[k][s] for
// the <= 4 head outputs j.  X (<= 256 channels) is staged like any slab (through registers, two
// blocks ahead).  The products run on v_mfma_f32_4x4x1_16b_f32 -- sixteen independent 4x4 outer
// products per instruction: block b = lane / 4 is a SAMPLE, A lane (b, i) = channel 4 quad + i of
// that sample (one ds_read_b32), B lane (b, j) = d_logits column j of it (a coalesced dword load),
// D[vgpr i][lane (b, j)] the partial dW[j][4 quad + i] of sample block b; the sixteen sample
// blocks are summed by the reducer.  Wave w owns quads 16 w .. 16 w + 15 of the window, every
// sample: 32 eight-cycle matrix instructions per 32-sample block where the 32x32x2 formulation
// (4 of 32 columns used) spent 32 sixty-four-cycle ones.  The unit is bound by the stream of X
// (32 KiB per block).  Partial of wave w (slot segment.slot + w): float (q * 4 + i) * 64 + lane
// for its quad q, channel i; the bias strip holds dl summed per lane (b, j).
__device__ __forceinline__ void head_segment(const ffn_mlp_chain& ch, const ffn_wgrad_unit& unit,
                                             const ffn_wgrad_segment& seg, char* smem,
                                             const float* __restrict__ saved,
                                             const float* __restrict__ d_logits, int64_t n,
                                             int64_t num_blocks, float* __restrict__ partials) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sb = lane >> 2;                 // sample block of this lane: sample 16 hs + sb
    const int li = lane & 3;                  // A: channel of the quad; B / D: logits column
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg_col = unit.m_slot, lg_n = unit.m_cq0;      // head units reuse the M fields
    const bool col_ok = li < lg_n;
    const int64_t x_stride = (int64_t)ch.slot_channels[unit.n_slot] * 128;    // bytes per block
    const char* x_s = reinterpret_cast<const char*>(saved + ch.slot_offset[unit.n_slot] * num_blocks * 32) +
                      unit.n_cq0 * 512 + seg.blk_begin * x_stride;
    x_s = uniform_ptr(x_s);
    const int c_last = (unit.n_quads >> 3) - 1;
    const int t16 = tid * 16;
    f32x4 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = zero4();
    float bsum = 0.0f;
    // d_logits of a block: lane (sb, li) takes column lg_col + li of samples sb and 16 + sb --
    // uniform block base + a per-lane constant; fetched one block ahead.  Only a ragged last
    // block needs per-sample clamping.
    const int dl_lane = 4 * sb + (col_ok ? lg_col + li : 0);
    auto load_dl = [&](int64_t blk, float (&dst)[2]) {
        typedef const float __attribute__((address_space(1)))* gfloat;
        if ((blk + 1) * 32 <= n) {
            gfloat base = (gfloat)uniform_ptr(d_logits + blk * 128);
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                const float v = base[dl_lane + 64 * hs];
                dst[hs] = col_ok ? v : 0.0f;
            }
        } else {
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                const int64_t sample = blk * 32 + 16 * hs + sb;
                const int64_t sc = sample < n ? sample : n - 1;
                const float v = d_logits[sc * 4 + (col_ok ? lg_col + li : 0)];
                dst[hs] = (col_ok && sample < n) ? v : 0.0f;
            }
        }
    };
    f32x4 R[8];      // staged chunks of X (4 KiB each; past the window: its last chunk again)
    typedef const f32x4 __attribute__((address_space(1)))* gptr;
#define FFN_REQUEST(j)                                                                         \
    do {                                                                                       \
        gptr chunk = (gptr)(x_s + ((j) < c_last ? (j) : c_last) * 4096);                       \
        asm volatile("" : "+s"(chunk));                                                        \
        R[j] = __builtin_nontemporal_load(&chunk[tid]);                                                                     \
    } while (0)
#define FFN_DEPOSIT(cur, j) *reinterpret_cast<f32x4*>(smem + image_a(cur) + (j) * 4096 + t16) = R[j]
    float dl[2], dl_next[2];
    load_dl(seg.blk_begin, dl);
#pragma unroll
    for (int j = 0; j < 8; ++j) FFN_REQUEST(j);
#pragma unroll
    for (int j = 0; j < 8; ++j) FFN_DEPOSIT(0, j);
    x_s += x_stride;
    if (seg.blk_begin + 1 < seg.blk_end) {
#pragma unroll
        for (int j = 0; j < 8; ++j) FFN_REQUEST(j);
    }
    x_s += x_stride;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // LDS byte address (inside an image) of this lane's float of quad q, sample half 0: quad row,
    // float4 slot (sample ^ (quad & 15)), component li; the second half is +256 B.  Quads past the
    // window read the zero row behind the image.
    unsigned a_at[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int quad = 16 * wave + q;
        a_at[q] = quad < unit.n_quads ? (unsigned)(quad * 512 + ((sb ^ q) << 4) + 4 * li)
                                      : (unsigned)(kImageBytes + 4 * li);
    }
    for (int64_t blk = seg.blk_begin; blk < seg.blk_end; ++blk) {
        const int cur = (int)((blk - seg.blk_begin) & 1);
        const bool has1 = blk + 1 < seg.blk_end, has2 = blk + 2 < seg.blk_end;
        load_dl(has1 ? blk + 1 : blk, dl_next);
        const char* img = smem + image_a(cur);
        bsum += dl[0] + dl[1];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float a0 = *reinterpret_cast<const float*>(img + a_at[q]);
            const float a1 = *reinterpret_cast<const float*>(img + a_at[q] + 256);   // (zero row: 512 B)
            acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, dl[0], acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, dl[1], acc[q], 0, 0, 0);
            if (q < 8) {
                if (q < 2) {
                    if (has1) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) FFN_DEPOSIT(cur ^ 1, q * 4 + jj);
                    }
                } else if (q < 6) {
                    if (has2) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) FFN_REQUEST((q - 2) * 2 + jj);
                    }
                }
            }
        }
        x_s += x_stride;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        dl[0] = dl_next[0];
        dl[1] = dl_next[1];
    }
#undef FFN_REQUEST
#undef FFN_DEPOSIT
    float* out = partials + (int64_t)(seg.slot + wave) * kPartialFloats;
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[(q * 4 + i) * 64 + lane] = acc[q][i];
    out[16 * 16 * 64 + lane] = bsum;
}

}  // namespace ffn
