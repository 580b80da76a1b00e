// Shared machinery of the split-bf16 kernels (mlp_bf16.hip forward, mlp_bf16_bwd.hip backward
// data): the bf16 (hi, lo) split, and the LDS weight ring the four lockstep waves of a workgroup
// read their K blocks from.  See mlp_bf16.hip for the organisation.
#pragma once
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

// float4 index of (channel quad cq, sample s) inside a saved-activation block: mlp.hip's layout
__device__ __forceinline__ int saved_index16(int cq, int s) { return cq * 32 + (s ^ (cq & 15)); }

}  // namespace ffn
