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
// the <= 4 head outputs j.  X (<= 256 channels) is staged like any slab (through registers,
// two blocks ahead); wave w owns channel half (w & 1) and sample half (w >> 1) of every
// block: 8 steps x 4 MFMAs.  The unit is bound by the stream of X (32 KiB per ~2k cycles of
// MFMA work), which is why the copy is kept two blocks deep.
__device__ __forceinline__ void head_segment(const ffn_mlp_chain& ch, const ffn_wgrad_unit& unit,
                                             const ffn_wgrad_segment& seg, char* smem,
                                             const float* __restrict__ saved,
                                             const float* __restrict__ d_logits, int64_t n,
                                             int64_t num_blocks, float* __restrict__ partials) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int hh = lane >> 5;
    const int li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, sh = wave >> 1;
    const int lg_col = unit.m_slot, lg_n = unit.m_cq0;      // head units reuse the M fields
    const bool x_ok = li < unit.n_quads - 32 * half;
    const bool col_ok = li < lg_n;
    const int col = col_ok ? lg_col + li : 0;
    const int64_t x_stride = (int64_t)ch.slot_channels[unit.n_slot] * 128;    // bytes per block
    const char* x_s = reinterpret_cast<const char*>(saved + ch.slot_offset[unit.n_slot] * num_blocks * 32) +
                      unit.n_cq0 * 512 + seg.blk_begin * x_stride;
    x_s = uniform_ptr(x_s);
    const int c_last = (unit.n_quads >> 3) - 1;
    const int t16 = tid * 16;
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    float bsum = 0.0f;
    // this wave's 8 sample pairs of a block; d_logits straight from HBM (512 B / block),
    // fetched one block ahead so that their latency hides behind the previous block
    // a block's d_logits are 32 consecutive float4: uniform block base + a per-lane constant
    // + 32 B per step (immediate); only a ragged last block needs per-sample clamping
    const int dl_lane = (2 * (8 * sh) + hh) * 4 + col;
    auto load_dl = [&](int64_t blk, float (&dst)[8]) {
        typedef const float __attribute__((address_space(1)))* gfloat;
        if ((blk + 1) * 32 <= n) {
            gfloat base = (gfloat)uniform_ptr(d_logits + blk * 128);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = base[dl_lane + 8 * k];
                dst[k] = col_ok ? v : 0.0f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t sample = blk * 32 + 2 * (8 * sh + k) + hh;
                const int64_t sc = sample < n ? sample : n - 1;
                const float v = d_logits[sc * 4 + col];
                dst[k] = (col_ok && sample < n) ? v : 0.0f;
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
    float dl[8], dl_next[8];
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
    const int sw = li & 15;
    for (int64_t blk = seg.blk_begin; blk < seg.blk_end; ++blk) {
        const int cur = (int)((blk - seg.blk_begin) & 1);
        const bool has1 = blk + 1 < seg.blk_end, has2 = blk + 2 < seg.blk_end;
        load_dl(has1 ? blk + 1 : blk, dl_next);
        const f32x4* lx = reinterpret_cast<const f32x4*>(smem + image_a(cur)) +
                          (x_ok ? (32 * half + li) * 32 : kImageBytes / 16);
        f32x4 a = lx[(2 * (8 * sh) + hh) ^ sw];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // the next step's operand is read while this step's MFMAs run
            const f32x4 a_next = lx[(2 * (8 * sh + (k < 7 ? k + 1 : k)) + hh) ^ sw];
            bsum += dl[k];
#pragma unroll
            for (int p = 0; p < 4; ++p)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], dl[k], acc[p], 0, 0, 0);
            if (k < 2) {
                if (has1) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) FFN_DEPOSIT(cur ^ 1, k * 4 + jj);
                }
            } else if (k < 6) {
                if (has2) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) FFN_REQUEST((k - 2) * 2 + jj);
                }
            }
            a = a_next;
        }
        x_s += x_stride;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k) dl[k] = dl_next[k];
    }
#undef FFN_REQUEST
#undef FFN_DEPOSIT
    float* out = partials + (int64_t)(seg.slot + wave) * kPartialFloats;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(p * 16 + r) * 64 + lane] = acc[p][r];
    out[16 * 16 * 64 + lane] = bsum;
}

}  // namespace ffn
