// Split-bf16 chain kernels, second organisation: TWO WAVES PER SIMD (OPT-IN mode, separately
// labelled; the exact-f32 kernels of mlp.hip stay the parity mode and the headline).
//
// The ring kernels (mlp_bf16.hip, mlp_bf16_bwd.hip) keep a 32-sample block's activations in the
// registers of ONE wave per SIMD (128 accumulators + 128 operand registers + 128 weight
// registers): every conversion, feature polynomial and epilogue instruction of that in-order wave
// is time the matrix pipe idles -- it ran a third busy.  A bf16 MFMA does not use the FP32 lanes,
// so a second resident wave per SIMD can do its vector work under the first one's matrix
// instructions; that needs waves of <= 256 registers.  Here
//
//   * a workgroup is EIGHT waves (two per SIMD) working on 4 / TPW blocks of 32 samples;
//   * wave w OWNS OUTPUT TILE w (32 channels; TPW = 2 for 512-wide chains: tiles w and w + 8) of
//     every step, for all blocks: 16 accumulator registers per (tile, block);
//   * its slice of the weights streams from L2 STRAIGHT INTO REGISTERS, two K blocks (a "chunk",
//     16 registers per tile) at a time, double buffered, requested a chunk (24 matrix instructions
//     of its own, ~1.5k cycles of wall time) ahead -- no LDS ring, no ring barriers; per workgroup
//     and pass every weight is still fetched once;
//   * the activations live in LDS as ready-made B operands -- X[block][K block][hi | lo][lane],
//     16 B per lane, 128 KiB -- written by the epilogues (a tile's accumulators ARE K blocks 2o and
//     2o + 1 of the next step in the hand-off order of mlp_bf16.hip, so the packs are shared with
//     the ring kernels) and by the feature generators, read by every wave (8 ds_read_b128 per
//     12 matrix instructions: a third of the LDS rate);
//   * two workgroup barriers bracket every refill of X ("all consumed" / "filled").
//
// Arithmetic per accumulator is the ring kernels': products w_hi x_lo, w_lo x_hi, w_hi x_hi per
// K block in K order, same feature code, same epilogue: hidden activations, saved slabs and sign
// masks are bit-identical between the two organisations; the fused heads' partial sums meet in a
// different order (per tile, then over the tiles: 1e-7).
#include "bf16_ring.h"

namespace ffn {

constexpr int kWsXBytes = 128 * 1024;
constexpr int kWsBiasFloats = 4096;
constexpr size_t kWsLdsBytes = (size_t)kWsXBytes + kEncTableBytes + kWsBiasFloats * 4;

// Geometry of a workgroup.  TPW_ tiles per wave (2 for 512-wide chains), SPLIT_ waves per output
// tile: with SPLIT_ = 2 the workgroup is SIXTEEN waves (four per SIMD, <= 128 registers each) and
// the two waves of a tile each take half of the pass's blocks -- twice the vector-instruction
// issue rate per SIMD and four instruction streams to fill the matrix pipe from; the duplicate
// weight requests of the two waves of a tile are served by the CU's L1.
//
// PARTS_ = 3 is the F32-ACCURATE mode ("bf16x6"): every f32 operand is THREE bf16 parts (hi, mid,
// lo: 24 significand bits, the operand exactly) and every f32 product SIX matrix instructions --
// all partial products down to 2^-16 of the leading one: h.l, l.h, m.m, h.m, m.h, h.h, smallest
// first (PRODUCTS_ = 9 adds m.l, l.m, l.l: measurement only) -- 12 matrix cycles per K where
// v_mfma_f32_32x32x2_f32 takes 32.  The X image then holds three parts per value, so a pass is TWO
// blocks (96 KiB): the same number of matrix instructions per pass and weight byte as the
// two-part kernels' four blocks.
template <int TPW_, int SPLIT_, int PARTS_ = 2, int PRODUCTS_ = (PARTS_ == 2 ? 3 : 6), int ACCS_ = (PARTS_ == 2 ? 1 : 2)>
struct WsShape {
    static constexpr int ACCS = ACCS_;            // 2: the small partial products on their own accumulator (WsOps)
    // (sixteen waves -- two per tile, a block each -- were measured for the three-part chains too:
    // backward data 6.45 -> 7.14 ms, and the forward does not fit 128 registers: 376 spilled)
    static_assert(PARTS_ == 2 || (PARTS_ == 3 && TPW_ == 1 && SPLIT_ == 1), "three-part chains: narrow, eight waves");
    static_assert(PARTS_ == 2 ? PRODUCTS_ == 3 : (PRODUCTS_ == 6 || PRODUCTS_ == 9), "partial products");
    static constexpr int TPW = TPW_;
    static constexpr int SPLIT = SPLIT_;
    static constexpr int PARTS = PARTS_;          // bf16 parts per f32 operand
    static constexpr int PRODUCTS = PRODUCTS_;    // matrix instructions per f32 product
    static constexpr int WAVES = 8 * SPLIT_;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int NB = (PARTS_ == 3 ? 2 : 4) / TPW_;      // 32-sample blocks per pass
    static constexpr int NBW = NB / SPLIT_;       // ... per wave
    static constexpr int TILES = 8 * TPW_;        // output tiles of the operand packs
    static constexpr int KBMAX = 16 * TPW_;       // K blocks of one block's X image
    static constexpr int kBlkVecs = 64 * PARTS_;  // float4 per (K block, block) of X: parts x 64 lanes
    static constexpr int kKbVecs = NB * kBlkVecs; // float4 per K block of X
    static constexpr int kTileVecs = 64 * PARTS_; // float4 per (K block, tile) of the weight packs
    // two alternating sets of hi operands (see ws_kblock)?  Not at 128 registers per wave: there
    // the hi operands are refilled late, and the other three waves of the SIMD cover the round trip
    static constexpr bool XH = SPLIT_ == 1;
};

// float4 index of (K block G, sample block b, part, lane 0) in the X image: K-block major, so that
// the reads of one K block (every block, both parts) are ONE base register + immediate offsets
template <class S>
__device__ __forceinline__ constexpr int ws_x_index(int G, int b, int part) {
    return (G * S::NB + b) * S::kBlkVecs + part * 64;
}

struct WsWave {
    int lane, h, s, wave;
    int tile;                  // wave % 8: the output tile(s) tile + 8 t of this wave
    int b0;                    // first block (inside the pass) of this wave
    f32x4* xbuf;               // LDS: the B-operand images
    const f32x4* gw;           // the chain's operand packs (uniform)
    int total_chunks;          // pairs of K blocks in the whole chain
    int cpos;                  // flat index of the next chunk to consume
    bool stale;                // the weight registers do not hold chunks cpos, cpos + 1
    const float* enc_table;    // LDS
    const float* bias_lds;     // LDS copy of the head of the bias buffer
    const float* bias_glb;
};

__device__ __forceinline__ void ws_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// the wave's slice of chunk c: K blocks 2c, 2c+1, its TPW tiles, (hi, lo): 1 KiB per load
// (K block k of chunk c; ws_load_chunk = both)
template <class S>
__device__ __forceinline__ void ws_load_kblock(const WsWave& w, bf16x8 (&dst)[S::TPW][S::PARTS], int c, int k) {
    constexpr int TILES = S::TILES, TPW = S::TPW;
    typedef const f32x4 __attribute__((address_space(1)))* gptr;
    // wave-uniform bases (one per K block and tile: the strides exceed the immediate offset) kept
    // in SGPRs; the lanes add 16 B each -- scalar-base addressing, no per-lane 64-bit pointers
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        gptr base = (gptr)(w.gw + ((int64_t)c * 2 + k) * (TILES * S::kTileVecs) + (w.tile + 8 * t) * S::kTileVecs);
        asm volatile("" : "+s"(base));
#pragma unroll
        for (int part = 0; part < S::PARTS; ++part)
            dst[t][part] = __builtin_bit_cast(bf16x8, base[part * 64 + w.lane]);
    }
}
template <class S>
__device__ __forceinline__ void ws_load_chunk(const WsWave& w, bf16x8 (&dst)[2][S::TPW][S::PARTS], int c) {
    ws_load_kblock<S>(w, dst[0], c, 0);
    ws_load_kblock<S>(w, dst[1], c, 1);
}

// The B-operand registers of a wave.  Two parts: the lo operands and two alternating sets of hi
// operands (see ws_kblock); three parts: two complete sets x[K block parity][block][part] -- with
// two blocks per wave there is room for a plain double buffer.
template <class S, int PARTS = S::PARTS>
struct WsOps;
template <class S>
struct WsOps<S, 2> {
    bf16x8 xb[S::NBW][2];
    bf16x8 xh[S::NBW];
};
template <class S>
struct WsOps<S, 3> {
    bf16x8 x[2][S::NBW][3];
    // The SMALL partial products (everything but h.h: <= 2^-8 of the leading term) accumulate
    // here, from zero, for the length of a K-loop segment (8 or 16 K blocks) and meet the main
    // accumulator at its end in one rounded f32 add per element (ws_segment).  The matrix unit's
    // f32 accumulation is not round-to-nearest: with all six products on one accumulator its
    // error has a systematic part that survives sums over samples (bias gradients 4-5x the exact
    // kernels' error against float64, with six and with nine products alike; measured:
    // profiles/r05_bf16x6_probe.json); on an accumulator 2^-8 the size it is 2^-8 of that, and the
    // main one sees one matrix-unit addition per K block instead of six.  (Live only inside a
    // segment: kept across the feature generators it costs the forward kernels 500 spilled
    // registers.)
    f32x16 lo[S::TPW][S::NBW];
};

// one part (0 = hi, .. PARTS-1 = lo) of the B operands of X K block G, every block of this wave
template <class S, int COLS>
__device__ __forceinline__ void ws_read_x(const WsWave& w, bf16x8 (&x)[S::NBW][COLS], int G, int part) {
    constexpr int NBW = S::NBW;
    const f32x4* p = w.xbuf + G * S::kKbVecs + w.b0 * S::kBlkVecs + part * 64 + w.lane;
#pragma unroll
    for (int b = 0; b < NBW; ++b) x[b][part] = __builtin_bit_cast(bf16x8, p[b * S::kBlkVecs]);
}

// One K block.  Per accumulator the products run w_hi x_lo, w_lo x_hi, w_hi x_hi (the ring
// kernels' order).  The lo operands x[b][1] are free after the first third of the matrix
// instructions and are REPLACED right there by those of K block `next` (consumed two thirds of a
// K block later); the hi operands are needed until the last matrix instruction, so they alternate
// between two sets -- K block parity HB reads hi from (HB ? xh : x[b][0]) and fills the other set
// for `next` early -- a late refill would leave a third of a K block (128 cycles) for an LDS
// round trip.  48 operand registers instead of 64.
// Issue order pinned: a ds_read behind each matrix instruction of the first third and of the
// second third, the NVM weight requests behind the last third.
template <class S, int NT, int HB>
__device__ __forceinline__ void ws_kblock2(const WsWave& w, f32x16 (&acc)[S::TPW][S::NBW],
                                           const bf16x8 (&wk)[S::TPW][2], bf16x8 (&x)[S::NBW][2],
                                           bf16x8 (&xh)[S::NBW], int next) {
    constexpr int NBW = S::NBW;
    const f32x4* p = w.xbuf + next * S::kKbVecs + w.b0 * 128 + w.lane;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NBW; ++b)
            acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[t][0], x[b][1], acc[t][b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < NBW; ++b) x[b][1] = __builtin_bit_cast(bf16x8, p[b * 128 + 64]);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NBW; ++b)
            acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[t][1], (S::XH && HB) ? xh[b] : x[b][0], acc[t][b], 0, 0, 0);
    if (S::XH) {
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            if (HB) x[b][0] = __builtin_bit_cast(bf16x8, p[b * 128]);
            else xh[b] = __builtin_bit_cast(bf16x8, p[b * 128]);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NBW; ++b)
            acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[t][0], (S::XH && HB) ? xh[b] : x[b][0], acc[t][b], 0, 0, 0);
    if (!S::XH) {
#pragma unroll
        for (int b = 0; b < NBW; ++b) x[b][0] = __builtin_bit_cast(bf16x8, p[b * 128]);
    }
}

// The partial products of the three-part mode, smallest first (the large terms meet an accumulator
// that already holds the small ones): (weight part, operand part), 0 = hi, 1 = mid, 2 = lo.  Six
// products drop m.l, l.m, l.l -- each below 2^-24 of the leading term.
template <int PRODUCTS> struct WsProducts;
template <> struct WsProducts<6> {
    static constexpr int W[6] = {0, 2, 1, 0, 1, 0};
    static constexpr int X[6] = {2, 0, 1, 1, 0, 0};
};
template <> struct WsProducts<9> {
    static constexpr int W[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
    static constexpr int X[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
};

// One K block of the three-part mode: K block parity HB multiplies out of operand set HB while
// the set of K block `next` streams from LDS into the other one.
template <class S, int NT, int HB>
__device__ __forceinline__ void ws_kblock3(const WsWave& w, f32x16 (&acc)[S::TPW][S::NBW], f32x16 (&lo)[S::TPW][S::NBW],
                                           const bf16x8 (&wk)[S::TPW][3], bf16x8 (&x)[2][S::NBW][3], int next) {
    constexpr int NBW = S::NBW;
    typedef WsProducts<S::PRODUCTS> P;
    const f32x4* p = w.xbuf + next * S::kKbVecs + w.b0 * S::kBlkVecs + w.lane;
#pragma unroll
    for (int q = 0; q < S::PRODUCTS; ++q) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                if (S::ACCS == 1 || (P::W[q] == 0 && P::X[q] == 0))
                    acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[t][P::W[q]], x[HB][b][P::X[q]], acc[t][b], 0, 0, 0);
                else
                    lo[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[t][P::W[q]], x[HB][b][P::X[q]], lo[t][b], 0, 0, 0);
            }
        if (q < 3) {
#pragma unroll
            for (int b = 0; b < NBW; ++b)
                x[HB ^ 1][b][2 - q] = __builtin_bit_cast(bf16x8, p[b * S::kBlkVecs + (2 - q) * 64]);
        }
    }
}

template <class S, int NT, int HB>
__device__ __forceinline__ void ws_kblock(const WsWave& w, f32x16 (&acc)[S::TPW][S::NBW],
                                          const bf16x8 (&wk)[S::TPW][S::PARTS], WsOps<S>& ops, int next) {
    if constexpr (S::PARTS == 2) ws_kblock2<S, NT, HB>(w, acc, wk, ops.xb, ops.xh, next);
    else ws_kblock3<S, NT, HB>(w, acc, ops.lo, wk, ops.x, next);
}

template <class S, int NT, int NVM>
__device__ __forceinline__ void ws_pin_kblock() {
    constexpr int NBW = S::NBW;
    if constexpr (S::PARTS == 3) {
        // a ds_read behind each of the first 3 NBW matrix instructions (the other operand set), the
        // NVM weight requests behind the last ones
        constexpr int total = S::PRODUCTS * NT * NBW;
#pragma unroll
        for (int i = 0; i < total; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 3 * NBW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i >= total - NVM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        return;
    }
    constexpr int third = NT * NBW;
#pragma unroll
    for (int i = 0; i < third; ++i) {              // w_hi x_lo: block b's lo operand is free after
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // the last tile's instruction on it
        if (i >= third - NBW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < third; ++i) {              // w_lo x_hi, the other hi set being refilled
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (S::XH && i >= third - NBW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < third; ++i) {              // w_hi x_hi, with the weight requests behind
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (!S::XH && i >= third - NBW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((NVM * (i + 1)) / third - (NVM * i) / third > k) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    }
}

// chunk in weight buffer P: K blocks g, g + 1 of the X image (operands of g already in x)
template <class S, int NT, int P>
__device__ __forceinline__ void ws_chunk(WsWave& w, f32x16 (&acc)[S::TPW][S::NBW],
                                         bf16x8 (&wreg)[2][2][S::TPW][S::PARTS], WsOps<S>& ops,
                                         int g, int g_last) {      // (X K-block indices)
    int c2 = w.cpos + 2;                                   // (uniform; branch-free: the chunk stays
    c2 -= c2 >= w.total_chunks ? w.total_chunks : 0;       // one basic block whose issue order is
    c2 -= c2 >= w.total_chunks ? w.total_chunks : 0;       // pinned; twice for one-chunk toy chains)
    // the registers of a K block's weights are refilled (chunk after next, same buffer) as soon as
    // its matrix instructions have issued: every request has a chunk and a half -- 36 matrix
    // instructions of this wave, >= 1.1k cycles -- to come back from L2
    ws_kblock<S, NT, 0>(w, acc, wreg[P][0], ops, g + 1);
    ws_load_kblock<S>(w, wreg[P][0], c2, 0);
    const int nxt = g + 2 <= g_last ? g + 2 : g_last;     // (the last chunk re-reads, never consumed)
    ws_kblock<S, NT, 1>(w, acc, wreg[P][1], ops, nxt);
    ws_load_kblock<S>(w, wreg[P][1], c2, 1);
    w.cpos = w.cpos + 1 < w.total_chunks ? w.cpos + 1 : 0;
    if constexpr (NT > 0) {
        // (a K block's pattern carries the requests issued BEHIND THE PREVIOUS K block, whose
        // registers were free from its last matrix instruction on)
        ws_pin_kblock<S, NT, 0>();
        ws_pin_kblock<S, NT, S::PARTS * S::TPW>();
    }
}

// `count` K blocks (even) of the X image, from its K block g0.  Returns true when the segment had
// an odd number of chunks: the two weight buffers then have to trade places before the next chunk.
template <class S, int NT>
__device__ __forceinline__ bool ws_segment(WsWave& w, f32x16 (&acc)[S::TPW][S::NBW],
                                           bf16x8 (&wreg)[2][2][S::TPW][S::PARTS], WsOps<S>& ops,
                                           int count, int g0 = 0) {
    if (NT == 0) {                  // a wave without a tile in this step only keeps count
        w.cpos = (w.cpos + (count >> 1)) % w.total_chunks;
        w.stale = true;
        return false;
    }
    if (w.stale) {
        ws_load_chunk<S>(w, wreg[0], w.cpos);
        ws_load_chunk<S>(w, wreg[1], w.cpos + 1 < w.total_chunks ? w.cpos + 1 : 0);
        w.stale = false;
        // "Use" the last request here: the compiler then waits for these loads on THIS
        // (rare) path.  Otherwise its wait-count analysis merges this state -- every weight
        // register pending, the newest last -- into the loop head, and the steady state of the K
        // loop waits with vmcnt(0) for requests it issued a few instructions earlier.
        asm volatile("" ::"v"(wreg[1][1][S::TPW - 1][S::PARTS - 1]));
    }
    if constexpr (S::PARTS == 2) {
        ws_read_x<S>(w, ops.xb, g0, 1);
        ws_read_x<S>(w, ops.xb, g0, 0);
    } else {
        ws_read_x<S>(w, ops.x[0], g0, 2);
        ws_read_x<S>(w, ops.x[0], g0, 1);
        ws_read_x<S>(w, ops.x[0], g0, 0);
        if constexpr (S::ACCS == 2) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int b = 0; b < S::NBW; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ops.lo[t][b][r] = 0.0f;
        }
    }
    // (pairs of chunks in ONE basic block per trip, the odd chunk outside the loop: with a
    // conditional second chunk inside it, hipcc builds a loop in which chunk<1> can follow
    // chunk<1>, and its wait-count analysis then makes every weight register wait for the four
    // newest requests -- the ones issued a few instructions earlier)
    int g = 0;
    for (; g + 4 <= count; g += 4) {
        ws_chunk<S, NT, 0>(w, acc, wreg, ops, g0 + g, g0 + count - 1);
        ws_chunk<S, NT, 1>(w, acc, wreg, ops, g0 + g + 2, g0 + count - 1);
    }
    const bool odd = g < count;
    if (odd) ws_chunk<S, NT, 0>(w, acc, wreg, ops, g0 + g, g0 + count - 1);
    if constexpr (S::PARTS == 3 && S::ACCS == 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < S::NBW; ++b) acc[t][b] += ops.lo[t][b];
    }
    return odd;
}

template <class S>
__device__ __forceinline__ void ws_swap(bf16x8 (&wreg)[2][2][S::TPW][S::PARTS]) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int t = 0; t < S::TPW; ++t)
#pragma unroll
            for (int part = 0; part < S::PARTS; ++part) {
                const bf16x8 tmp = wreg[0][k][t][part];
                wreg[0][k][t][part] = wreg[1][k][t][part];
                wreg[1][k][t][part] = tmp;
            }
}

// eight f32 values as the S::PARTS bf16 operand parts of the mode
template <class S>
__device__ __forceinline__ void ws_split(const float (&v)[8], bf16x8 (&part)[S::PARTS]) {
    if constexpr (S::PARTS == 2) split8(v, part[0], part[1]);
    else split8x3(v, part[0], part[1], part[2]);
}

// Sign masks (mlp.hip's format): per (slot, block) TPW records of 64 lanes x 16 bytes -- a narrow
// chain's tiles 0..7 in one record, a wide chain's first / second half of the tiles in two (the
// halves its two-waves-per-block kernels own).  Inside a record the 16 bits of the p-th tile of
// that half sit in word p / 2, an even p in the upper half-word; a half of ONE tile keeps its
// 16 bits right-aligned in word 0.
__device__ __forceinline__ int ws_mask_byte(int p, int per) {
    return per == 1 ? 0 : 4 * (p >> 1) + ((p & 1) ? 0 : 2);
}
template <class S>
__device__ __forceinline__ int64_t ws_mask_at(int slot, int64_t num_blocks, int64_t block, int lane, int o, int ot) {
    constexpr int TPW = S::TPW;
    const int per = TPW == 1 ? ot : ot >> 1;          // tiles per record (a power of two)
    const int hf = TPW == 1 ? 0 : (o >= per ? 1 : 0);
    const int p = o - hf * per;
    return ((((int64_t)slot * num_blocks + block) * TPW + hf) * 64 + lane) * 16 + ws_mask_byte(p, per);
}
// the half-word slot (record t, position `tile`) the wave of that tile zero-fills when no tile of the
// step uses it, so that whole records are defined like the ring / f32 kernels' (-1: in use)
template <class S>
__device__ __forceinline__ int64_t ws_mask_idle_at(int slot, int64_t num_blocks, int64_t block, int lane,
                                                   int t, int tile, int ot) {
    constexpr int TPW = S::TPW;
    const int per = TPW == 1 ? ot : ot >> 1;
    const int byte = 4 * (tile >> 1) + ((tile & 1) ? 0 : 2);
    const bool used = per == 1 ? byte == 0 : tile < per;
    if (used) return -1;
    return ((((int64_t)slot * num_blocks + block) * TPW + t) * 64 + lane) * 16 + byte;
}

// ---------------------------------------------------------------------------------- forward
template <class S>
struct WsFwd : WsWave {
    float x0, x1, x2, v0, v1, v2;       // inputs of this wave's feature block (wave % NB)
    float logit[S::NBW][4];            // fused heads: partial sums over this wave's tiles
    int64_t block0, num_blocks;         // first block of the pass; blocks of the launch
    float* saved;
    char* masks;
};

template <class S, bool TRAIN, bool HWSIN>
__device__ __forceinline__ void ws_step_fwd(const ffn_mlp_chain& ch, const ffn_step& L, bool last_step,
                                            WsFwd<S>& w, bf16x8 (&wreg)[2][2][S::TPW][S::PARTS]) {
    constexpr int NB = S::NB, NBW = S::NBW, TPW = S::TPW;
    constexpr int KBMAX = S::KBMAX;
    const int ot = L.out_tiles;
    const int kb_act = L.act_groups >> 1, kb_feat = L.aux_groups >> 1;
    constexpr int WAVES = S::WAVES;
    // tiles of this wave that exist in this step (uniform): tile + 8 t < ot
    const int nt = ot > w.tile + 8 ? (TPW > 1 ? 2 : 1) : (ot > w.tile ? 1 : 0);

    f32x16 acc[S::TPW][NBW];
    {
        const int room = L.b_off + 32 * ot <= kWsBiasFloats;
        const float* bsrc = room ? w.bias_lds : w.bias_glb;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int o = w.tile + 8 * t;
            const float* bv = bsrc + L.b_off + 32 * (o < ot ? o : 0) + 4 * w.h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + 8 * q);
                if (o >= ot) b4 = (f32x4)(0.0f);
#pragma unroll
                for (int b = 0; b < NBW; ++b)
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[t][b][4 * q + p] = b4[p];
            }
        }
    }
    WsOps<S> ops;
    bool swap_due = false;
    auto run = [&](int count, int g0) -> bool {
        if (nt == 0) return ws_segment<S, 0>(w, acc, wreg, ops, count, g0);
        if (TPW > 1 && nt == 1) return ws_segment<S, 1>(w, acc, wreg, ops, count, g0);
        return ws_segment<S, TPW>(w, acc, wreg, ops, count, g0);
    };
    if (kb_act > 0) swap_due = run(kb_act, 0);     // X holds the previous step's output (filled)
    if (kb_feat > 0) {
        Enc16 enc;
        const ffn_encoding& e = ch.enc[L.enc_id];
        enc.tab = w.enc_table + L.enc_id * kEncTablePitch;
        enc.F = e.num_freq;
        enc.raw = (e.include_input != 0 || e.num_freq == 0) ? 1 : 0;
        enc.scale = e.scale;
        const float p0 = L.enc_id == 0 ? w.x0 : w.v0;
        const float p1 = L.enc_id == 0 ? w.x1 : w.v1;
        const float p2 = L.enc_id == 0 ? w.x2 : w.v2;
        const int g_trig = e.num_freq >> 3;       // K blocks whose eight frequencies are all real
        const int fb = w.wave % NB;               // the block this wave generates features for
        f32x4* fsave = nullptr;
        if (TRAIN && w.saved != nullptr && L.save_enc_slot >= 0 && w.block0 + fb < w.num_blocks)
            fsave = reinterpret_cast<f32x4*>(w.saved + ch.slot_offset[L.save_enc_slot] * w.num_blocks * 32) +
                    (w.block0 + fb) * (int64_t)(ch.slot_channels[L.save_enc_slot] * 8);
        // feature K blocks c0 .. c0+count-1 of this wave's block into X K blocks x0 ..
        auto generate = [&](int c0, int count, int x0) {
            for (int G = w.wave / NB; G < count; G += WAVES / NB) {
                float f[8];
                if (HWSIN) {
                    if (c0 + G < g_trig) features16_hw<true>(enc, c0 + G, w.h, p0, p1, p2, f);
                    else features16_hw<false>(enc, c0 + G, w.h, p0, p1, p2, f);
                } else {
                    if (c0 + G < g_trig) features16_unpacked<true>(enc, c0 + G, w.h, p0, p1, p2, f);
                    else features16_unpacked<false>(enc, c0 + G, w.h, p0, p1, p2, f);
                }
                if (TRAIN && fsave != nullptr) {
                    const int cq = 4 * (c0 + G) + 2 * w.h;
                    f32x4 f0, f1;
#pragma unroll
                    for (int p = 0; p < 4; ++p) { f0[p] = f[p]; f1[p] = f[4 + p]; }
                    __builtin_nontemporal_store(f0, &fsave[saved_index16(cq, w.s)]);
                    __builtin_nontemporal_store(f1, &fsave[saved_index16(cq + 1, w.s)]);
                }
                bf16x8 fp[S::PARTS];
                ws_split<S>(f, fp);
                f32x4* dst = w.xbuf + ws_x_index<S>(x0 + G, fb, 0) + w.lane;
#pragma unroll
                for (int part = 0; part < S::PARTS; ++part) dst[64 * part] = __builtin_bit_cast(f32x4, fp[part]);
            }
        };
        if (kb_act == 0) {
            // A features-only step (the first layer: all the encoding work of the tiny NeRF) runs
            // as a PIPELINE over half-size segments ping-ponging between the two halves of X: while
            // segment k is multiplied, segment k+1 is generated -- by the same waves, in OPPOSITE
            // order on the two waves of a SIMD (waves w and w + 4): one streams matrix
            // instructions while the other runs the sin/cos polynomials, then they trade.  One
            // barrier per segment ("k consumed, k+1 filled").
            constexpr int HALF = KBMAX / 2;
            const bool k_first = ((w.wave >> 2) & 1) == 0;      // (waves i, i + 4, i + 8, .. share a SIMD)
            ws_barrier();                          // every wave has consumed what X held
            generate(0, kb_feat < HALF ? kb_feat : HALF, 0);
            ws_barrier();
            int side = 0;
            for (int c0 = 0; c0 < kb_feat; c0 += HALF, side ^= 1) {
                const int count = kb_feat - c0 < HALF ? kb_feat - c0 : HALF;
                const int c1 = c0 + HALF;
                const int next = c1 >= kb_feat ? 0 : (kb_feat - c1 < HALF ? kb_feat - c1 : HALF);
                if (swap_due) { ws_swap<S>(wreg); swap_due = false; }
                if (k_first) {
                    swap_due = run(count, side * HALF);
                    if (next > 0) generate(c1, next, (side ^ 1) * HALF);
                } else {
                    if (next > 0) generate(c1, next, (side ^ 1) * HALF);
                    swap_due = run(count, side * HALF);
                }
                if (next > 0) ws_barrier();
            }
        } else {
            for (int c0 = 0; c0 < kb_feat; c0 += KBMAX) {
                const int count = kb_feat - c0 < KBMAX ? kb_feat - c0 : KBMAX;
                if (swap_due) { ws_swap<S>(wreg); swap_due = false; }
                ws_barrier();                      // every wave has consumed what X held
                generate(c0, count, 0);
                ws_barrier();                      // filled
                swap_due = run(count, 0);
            }
        }
    }

    // ---- epilogue: ReLU, sign bits, saves, fused head, the next step's operands into X
    const bool fused_head = L.head_off >= 0;
    const float* hw = w.bias_lds + (fused_head ? L.head_off : 0) + 4 + 16 * w.h;
    if (fused_head && w.tile == 0 && w.h == 0) {
        const f32x4 hb = *reinterpret_cast<const f32x4*>(w.bias_lds + L.head_off);
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) w.logit[b][c] += hb[c];
    }
    const int relu_floor = L.relu ? 0 : (int)0x80000000;
    // Everything but the writes into X happens BEFORE the "all consumed" barrier: a wave that
    // leaves its K loops early (the older wave of a SIMD wins the matrix pipe) does its ReLU /
    // sign-bit / bf16-split / head arithmetic under the matrix instructions of the waves still
    // multiplying; after the barrier only the 16-byte LDS stores are left.
    bf16x8 res[S::TPW][NBW][2][S::PARTS];
    int save_s = w.s, save_h = w.h, e_lane = w.lane;
    asm volatile("" : "+v"(save_s), "+v"(save_h), "+v"(e_lane));     // (see mlp_bf16.hip: no hoisted address tables)
    const int64_t blk0 = w.block0 + w.b0;          // this wave's first block of the launch
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int o = w.tile + 8 * t;
        if (TRAIN && w.masks != nullptr && L.relu && L.mask_slot >= 0) {
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                const int64_t at = ws_mask_idle_at<S>(L.mask_slot, w.num_blocks, blk0 + b, e_lane, t, w.tile, ot);
                if (at >= 0 && blk0 + b < w.num_blocks) *reinterpret_cast<uint16_t*>(w.masks + at) = (uint16_t)0;
            }
        }
        if (o >= ot) continue;
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const bool live = blk0 + b < w.num_blocks;
            f32x4* save_out = nullptr;
            if (TRAIN && w.saved != nullptr && L.out_slot >= 0 && live)
                save_out = reinterpret_cast<f32x4*>(w.saved + ch.slot_offset[L.out_slot] * w.num_blocks * 32) +
                           (blk0 + b) * (int64_t)(ch.slot_channels[L.out_slot] * 8);
            unsigned sign_bits = 0u;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = acc[t][b][8 * half + j];
                    if (TRAIN) sign_bits = __builtin_amdgcn_alignbit(sign_bits, __builtin_bit_cast(unsigned, 0.0f - v), 31);
                    y[j] = __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, v), relu_floor));
                }
                if (TRAIN && save_out != nullptr) {
                    f32x4 y0, y1;
#pragma unroll
                    for (int p = 0; p < 4; ++p) { y0[p] = y[p]; y1[p] = y[4 + p]; }
                    const int cq = 2 * (4 * o + 2 * half) + save_h;
                    __builtin_nontemporal_store(y0, &save_out[saved_index16(cq, save_s)]);
                    __builtin_nontemporal_store(y1, &save_out[saved_index16(cq + 2, save_s)]);
                }
                if (fused_head) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int group = 4 * o + 2 * half + (j >> 2);
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(hw + group * 32 + (j & 3) * 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) w.logit[b][c] = __builtin_fmaf(y[j], w4[c], w.logit[b][c]);
                    }
                }
                if (!last_step) ws_split<S>(y, res[t][b][half]);
            }
            if (TRAIN && w.masks != nullptr && L.relu && L.mask_slot >= 0 && live) {
                *reinterpret_cast<uint16_t*>(w.masks + ws_mask_at<S>(L.mask_slot, w.num_blocks, blk0 + b, e_lane, o, ot)) =
                    (uint16_t)(sign_bits & 0xffffu);
            }
        }
    }
    if (swap_due) ws_swap<S>(wreg);
    ws_barrier();                                  // every K loop of this step has read X
    if (!last_step) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int o = w.tile + 8 * t;
            if (o >= ot) continue;
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    f32x4* dst = w.xbuf + ws_x_index<S>(2 * o + half, w.b0 + b, 0) + e_lane;
#pragma unroll
                    for (int part = 0; part < S::PARTS; ++part)
                        dst[64 * part] = __builtin_bit_cast(f32x4, res[t][b][half][part]);
                }
        }
        ws_barrier();                              // the step's output is in X
    }
}

template <class S, bool TRAIN, bool HWSIN>
__global__ void __launch_bounds__(S::THREADS)
mlp_forward_bf16_ws_kernel(const ffn_mlp_chain ch, const uint16_t* __restrict__ packed,
                           const float* __restrict__ bias, const float* __restrict__ positions,
                           const float* __restrict__ views, int64_t n, float* __restrict__ logits,
                           float* __restrict__ saved, uint32_t* __restrict__ masks) {
    constexpr int NB = S::NB, NBW = S::NBW, TPW = S::TPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* enc_table = reinterpret_cast<float*>(smem + kWsXBytes);
    float* bias_lds = reinterpret_cast<float*>(smem + kWsXBytes + kEncTableBytes);
    stage_encoding_tables(ch.enc, enc_table, threadIdx.x, S::THREADS);
    {
        const int staged = ch.bias_floats < kWsBiasFloats ? ch.bias_floats : kWsBiasFloats;
        for (int i = threadIdx.x; i < staged; i += S::THREADS) bias_lds[i] = bias[i];
    }
    WsFwd<S> w;
    w.lane = threadIdx.x & 63;
    w.h = w.lane >> 5;
    w.s = w.lane & 31;
    w.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.tile = w.wave & 7;
    w.b0 = (w.wave >> 3) * NBW;
    w.xbuf = reinterpret_cast<f32x4*>(smem);
    w.enc_table = enc_table;
    w.bias_lds = bias_lds;
    w.bias_glb = bias;
    w.gw = reinterpret_cast<const f32x4*>(packed + ch.step[0].w_off);
    int total_kb = 0;
    for (int li = 0; li < ch.num_steps; ++li) total_kb += (ch.step[li].act_groups + ch.step[li].aux_groups) >> 1;
    w.total_chunks = total_kb >> 1;
    w.cpos = 0;
    w.stale = true;
    w.saved = saved;
    w.masks = reinterpret_cast<char*>(masks);
    w.num_blocks = (n + 31) / 32;
    const int64_t passes = (w.num_blocks + NB - 1) / NB;
    bf16x8 wreg[2][2][S::TPW][S::PARTS];
    __syncthreads();                               // tables and biases are staged
    const int fb = w.wave % NB;
    float in_next[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto request_inputs = [&](int64_t pass) {
        int64_t block = pass * NB + fb;
        block = block < w.num_blocks ? block : w.num_blocks - 1;
        const int64_t sample = block * 32 + w.s;
        const int64_t src = sample < n ? sample : n - 1;
        in_next[0] = positions[src * 3 + 0]; in_next[1] = positions[src * 3 + 1]; in_next[2] = positions[src * 3 + 2];
        if (views != nullptr) {
            in_next[3] = views[src * 3 + 0]; in_next[4] = views[src * 3 + 1]; in_next[5] = views[src * 3 + 2];
        }
    };
    request_inputs(blockIdx.x);
    for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
        w.block0 = pass * NB;
        w.x0 = in_next[0]; w.x1 = in_next[1]; w.x2 = in_next[2];
        w.v0 = in_next[3]; w.v1 = in_next[4]; w.v2 = in_next[5];
        request_inputs(pass + gridDim.x < passes ? pass + gridDim.x : pass);
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) w.logit[b][c] = 0.0f;
        for (int li = 0; li < ch.num_steps; ++li)
            ws_step_fwd<S, TRAIN, HWSIN>(ch, ch.step[li], li + 1 == ch.num_steps, w, wreg);
        // the tiles' partial logits meet through X (free after the last step's K loops: the
        // barrier in front of its epilogue), in a fixed order: [block][tile][sample]
        f32x4* scratch = w.xbuf;
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            f32x4 part;
#pragma unroll
            for (int c = 0; c < 4; ++c) part[c] = w.logit[b][c] + __shfl_xor(w.logit[b][c], 32);
            if (w.h == 0) scratch[((w.b0 + b) * 8 + w.tile) * 32 + w.s] = part;
        }
        ws_barrier();
        if (w.wave < NB && w.h == 0) {
            const int64_t block = w.block0 + w.wave;
            const int64_t sample = block * 32 + w.s;
            f32x4 out = scratch[(w.wave * 8) * 32 + w.s];
#pragma unroll
            for (int k = 1; k < 8; ++k) out += scratch[(w.wave * 8 + k) * 32 + w.s];
            if (block < w.num_blocks && sample < n) reinterpret_cast<f32x4*>(logits)[sample] = out;
        }
        // (the next pass refills X only behind its own "all consumed" barrier)
    }
}

// ---------------------------------------------------------------------------------- backward data
template <class S>
struct WsBwd : WsWave {
    f32x4 dl;                           // d(loss)/d(logits) of this wave's block (wave % NB), lane's sample
    int64_t block0, num_blocks;
    float* dz;
    const char* masks;
};

template <class S>
__device__ __forceinline__ void ws_step_bwd(const ffn_mlp_chain& ch, const ffn_step& L, bool last_step,
                                            WsBwd<S>& w, bf16x8 (&wreg)[2][2][S::TPW][S::PARTS]) {
    constexpr int NB = S::NB, NBW = S::NBW, TPW = S::TPW;
    const int ot = L.out_tiles;
    const int kb_act = L.act_groups >> 1;
    const int nt = ot > w.tile + 8 ? (TPW > 1 ? 2 : 1) : (ot > w.tile ? 1 : 0);
    const int64_t blk0 = w.block0 + w.b0;          // this wave's first block of the launch
    // the sign masks of the layer being differentiated: requested now, used after the K loops
    unsigned mbits[S::TPW][NBW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            mbits[t][b] = 0xffffu;
            const int o = w.tile + 8 * t;
            if (L.mask_slot >= 0 && o < ot) {
                int64_t blk = blk0 + b;
                blk = blk < w.num_blocks ? blk : w.num_blocks - 1;
                mbits[t][b] = *reinterpret_cast<const uint16_t*>(
                    w.masks + ws_mask_at<S>(L.mask_slot, w.num_blocks, blk, w.lane, o, ot));
            }
        }
    f32x16 acc[S::TPW][NBW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][b][r] = 0.0f;
    WsOps<S> ops;
    bool swap_due = false;
    auto run = [&](int count) -> bool {
        if (nt == 0) return ws_segment<S, 0>(w, acc, wreg, ops, count, 0);
        if (TPW > 1 && nt == 1) return ws_segment<S, 1>(w, acc, wreg, ops, count, 0);
        return ws_segment<S, TPW>(w, acc, wreg, ops, count, 0);
    };
    if (kb_act > 0) swap_due = run(kb_act);
    if (L.aux_groups > 0) {
        // the d_logits term: one K block whose K rows 0..lg_n-1 (lane half 0) are logits columns
        // lg_col.., and a zero K block -- generated like a feature segment, one (block, K block)
        // item per wave (NB x 2 items)
        if (swap_due) { ws_swap<S>(wreg); swap_due = false; }
        ws_barrier();
        for (int item = w.wave; item < 2 * NB; item += S::WAVES) {
            const int G = item / NB;
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = L.lg_col + j;
                const float d = c == 0 ? w.dl[0] : (c == 1 ? w.dl[1] : (c == 2 ? w.dl[2] : w.dl[3]));
                v[j] = (G == 0 && w.h == 0 && j < L.lg_n && c < 4) ? d : 0.0f;
                v[4 + j] = 0.0f;
            }
            bf16x8 dp[S::PARTS];
            ws_split<S>(v, dp);
            f32x4* dst = w.xbuf + ws_x_index<S>(G, item % NB, 0) + w.lane;
#pragma unroll
            for (int part = 0; part < S::PARTS; ++part) dst[64 * part] = __builtin_bit_cast(f32x4, dp[part]);
        }
        ws_barrier();
        swap_due = run(2);
    }
    // ---- epilogue: mask, save dZ, the next step's operands into X (arithmetic and global stores
    // before the "all consumed" barrier, the LDS stores after it: see the forward step)
    bf16x8 res[S::TPW][NBW][2][S::PARTS];
    int save_s = w.s, save_h = w.h, e_lane = w.lane;
    asm volatile("" : "+v"(save_s), "+v"(save_h), "+v"(e_lane));
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int o = w.tile + 8 * t;
        if (o >= ot) continue;
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const bool live = blk0 + b < w.num_blocks;
            f32x4* save_out = nullptr;
            if (L.out_slot >= 0 && live)
                save_out = reinterpret_cast<f32x4*>(w.dz + ch.slot_offset[L.out_slot] * w.num_blocks * 32) +
                           (blk0 + b) * (int64_t)(ch.slot_channels[L.out_slot] * 8);
            const unsigned word = mbits[t][b];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int bit = 15 - (8 * half + j);
                    const int keep = ((int)(word << (31 - bit))) >> 31;
                    const float a = acc[t][b][8 * half + j];
                    y[j] = __builtin_bit_cast(float, __builtin_bit_cast(int, a) & keep);
                }
                if (save_out != nullptr) {
                    f32x4 y0, y1;
#pragma unroll
                    for (int p = 0; p < 4; ++p) { y0[p] = y[p]; y1[p] = y[4 + p]; }
                    const int cq = 2 * (4 * o + 2 * half) + save_h;
                    __builtin_nontemporal_store(y0, &save_out[saved_index16(cq, save_s)]);
                    __builtin_nontemporal_store(y1, &save_out[saved_index16(cq + 2, save_s)]);
                }
                if (!last_step) ws_split<S>(y, res[t][b][half]);
            }
        }
    }
    if (swap_due) ws_swap<S>(wreg);
    ws_barrier();
    if (!last_step) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int o = w.tile + 8 * t;
            if (o >= ot) continue;
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    f32x4* dst = w.xbuf + ws_x_index<S>(2 * o + half, w.b0 + b, 0) + e_lane;
#pragma unroll
                    for (int part = 0; part < S::PARTS; ++part)
                        dst[64 * part] = __builtin_bit_cast(f32x4, res[t][b][half][part]);
                }
        }
        ws_barrier();
    }
}

template <class S>
__global__ void __launch_bounds__(S::THREADS)
mlp_backward_bf16_ws_kernel(const ffn_mlp_chain ch, const uint16_t* __restrict__ packed,
                            const float* __restrict__ d_logits, int64_t n,
                            const uint32_t* __restrict__ masks, float* __restrict__ dz) {
    constexpr int NB = S::NB, NBW = S::NBW, TPW = S::TPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WsBwd<S> w;
    w.lane = threadIdx.x & 63;
    w.h = w.lane >> 5;
    w.s = w.lane & 31;
    w.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.tile = w.wave & 7;
    w.b0 = (w.wave >> 3) * NBW;
    w.xbuf = reinterpret_cast<f32x4*>(smem);
    w.enc_table = nullptr;
    w.bias_lds = nullptr;
    w.bias_glb = nullptr;
    w.gw = reinterpret_cast<const f32x4*>(packed + ch.step[0].w_off);
    int total_kb = 0;
    for (int li = 0; li < ch.num_steps; ++li)
        total_kb += (ch.step[li].act_groups >> 1) + (ch.step[li].aux_groups > 0 ? 2 : 0);
    w.total_chunks = total_kb >> 1;
    w.cpos = 0;
    w.stale = true;
    w.dz = dz;
    w.masks = reinterpret_cast<const char*>(masks);
    w.num_blocks = (n + 31) / 32;
    const int64_t passes = (w.num_blocks + NB - 1) / NB;
    bf16x8 wreg[2][2][S::TPW][S::PARTS];
    const int fb = w.wave % NB;
    f32x4 dl_next = (f32x4)(0.0f);
    auto request_inputs = [&](int64_t pass) {
        int64_t block = pass * NB + fb;
        block = block < w.num_blocks ? block : w.num_blocks - 1;
        const int64_t sample = block * 32 + w.s;
        // samples past n (the ragged tail of the last block) contribute zero
        dl_next = (pass * NB + fb < w.num_blocks && sample < n) ? reinterpret_cast<const f32x4*>(d_logits)[sample]
                                                                  : (f32x4)(0.0f);
    };
    request_inputs(blockIdx.x);
    for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
        w.block0 = pass * NB;
        w.dl = dl_next;
        request_inputs(pass + gridDim.x < passes ? pass + gridDim.x : pass);
        for (int li = 0; li < ch.num_steps; ++li)
            ws_step_bwd<S>(ch, ch.step[li], li + 1 == ch.num_steps, w, wreg);
        // (the next pass refills X only behind its own "all consumed" barrier)
    }
}

static int64_t ws_grid(int64_t passes) {
    int cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return passes < cus ? passes : cus;
}

typedef WsShape<1, 1> WsNarrow8;      // 8 waves, a tile each, 4 blocks per wave
typedef WsShape<1, 2> WsNarrow16;     // 16 waves, two per tile, 2 blocks per wave
typedef WsShape<2, 1> WsWide8;        // 512-wide chains: 8 waves, two tiles each, 2 blocks per pass

// FFN_BF16_WAVES=8|16: waves per workgroup of the narrow-chain kernels (A/B; default below)
inline bool ws_sixteen_waves() {
    const char* v = getenv("FFN_BF16_WAVES");
    if (v != nullptr && v[0] == '8') return false;
    if (v != nullptr && v[0] == '1') return true;
    return false;
}

template <class S, bool TRAIN, bool HWSIN>
static void ws_launch_fwd(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                          const float* positions, const float* views, int64_t n, float* logits,
                          float* saved, uint32_t* masks, void* stream) {
    const int64_t grid = ws_grid(((n + 31) / 32 + S::NB - 1) / S::NB);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_forward_bf16_ws_kernel<S, TRAIN, HWSIN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWsLdsBytes);
    hipLaunchKernelGGL((mlp_forward_bf16_ws_kernel<S, TRAIN, HWSIN>), dim3((unsigned)grid), dim3(S::THREADS),
                       kWsLdsBytes, (hipStream_t)stream, *chain, packed_w, bias, positions, views, n,
                       logits, saved, masks);
}

template <class S>
static void ws_launch_fwd_modes(bool hw, const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                                const float* positions, const float* views, int64_t n, float* logits,
                                float* saved, uint32_t* masks, void* stream) {
    // ONE instantiation serves training and inference launches: inference passes null slabs and
    // the saving code tests the pointers.  The instantiation without any saving code is the
    // SLOWER inference kernel -- hipcc spills 159 (bf16x3) / 186 (bf16x6) registers in it against
    // 54 / 52 in this one, and every reload in front of a K loop waits, through the in-order
    // vmcnt, for the weight requests in flight (interleaved on one box, scripts/gpu/r5_ab.sh,
    // r5_ab2.sh: bf16x6 tiny 13.8 -> 11.9 ms, full NeRF 16.0 -> 13.1 ms per 2^21; bf16x3 full
    // NeRF 14.5 -> 13.4 ms, tiny 6.51 -> 6.42 ms).
    if (hw) ws_launch_fwd<S, true, true>(chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
    else ws_launch_fwd<S, true, false>(chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
}

int launch_forward16_ws(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                        const float* positions, const float* views, int64_t n, float* logits,
                        float* saved, uint32_t* masks, void* stream) {
    const bool hw = !use_poly_sincos();
    if (chain->wide) {             // 512-wide chains: two tiles per wave, two blocks per pass
        if (saved != nullptr) ws_launch_fwd<WsWide8, true, true>(chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
        else ws_launch_fwd<WsWide8, false, true>(chain, packed_w, bias, positions, views, n, logits, nullptr, nullptr, stream);
        return 0;       // (the wide shape keeps its own inference instantiation: not measured the other way)
    }
    if (ws_sixteen_waves()) ws_launch_fwd_modes<WsNarrow16>(hw, chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
    else ws_launch_fwd_modes<WsNarrow8>(hw, chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
    return 0;
}

template <class S>
static void ws_launch_bwd(const ffn_mlp_chain* chain, const uint16_t* packed_wt, const float* d_logits,
                          int64_t n, const uint32_t* masks, float* dz, void* stream) {
    const int64_t grid = ws_grid(((n + 31) / 32 + S::NB - 1) / S::NB);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_backward_bf16_ws_kernel<S>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWsXBytes);
    hipLaunchKernelGGL((mlp_backward_bf16_ws_kernel<S>), dim3((unsigned)grid), dim3(S::THREADS), kWsXBytes,
                       (hipStream_t)stream, *chain, packed_wt, d_logits, n, masks, dz);
}

int launch_backward16_ws(const ffn_mlp_chain* chain, const uint16_t* packed_wt, const float* d_logits,
                         int64_t n, const uint32_t* masks, float* dz, void* stream) {
    if (chain->wide) ws_launch_bwd<WsWide8>(chain, packed_wt, d_logits, n, masks, dz, stream);
    else if (ws_sixteen_waves()) ws_launch_bwd<WsNarrow16>(chain, packed_wt, d_logits, n, masks, dz, stream);
    else ws_launch_bwd<WsNarrow8>(chain, packed_wt, d_logits, n, masks, dz, stream);
    return 0;
}

// ---------------------------------------------------------------------------------- bf16x6
typedef WsShape<1, 1, 3, 6> WsSplit6;       // three parts, six partial products: the f32-accurate mode
typedef WsShape<1, 1, 3, 9> WsSplit9;       // all nine partial products (FFN_BF16X6_PRODUCTS=9: measurement)
// The FORWARD kernels keep one accumulator: there the systematic part of the matrix unit's
// rounding stays inside the exact kernels' own error (logits 0.8-0.9x theirs against float64, 0.4x
// with two accumulators) and no sum over samples amplifies it, while the second accumulator costs
// the forward 40 more spilled registers and 8 % (training forward 12.8 -> 14.0 ms, interleaved on one
// box, profiles/r05_bf16x6_forward_accumulators_ab.txt).  BACKWARD DATA keeps two: its dZ feed the
// bias and weight gradients -- sums over every sample of the batch (and it is 2 % FASTER with
// them: shorter dependent chains, no spills).  FFN_BF16X6_FWD_ACCS=2 selects two in the forward too.
typedef WsShape<1, 1, 3, 6, 1> WsSplit6one;

inline bool bf16x6_nine_products() {        // read per launch: the probe flips it inside one process
    const char* v = getenv("FFN_BF16X6_PRODUCTS");
    return v != nullptr && v[0] == '9';
}

// (encoding features: the f32 kernels' polynomials, bit for bit -- the hardware sin / cos of the
// bf16x3 kernels is 2.4e-7 off, which an f32-accurate mode cannot afford)
int launch_forward_bf16x6(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                          const float* positions, const float* views, int64_t n, float* logits,
                          float* saved, uint32_t* masks, void* stream) {
    // (inference launches run the saving instantiation with null slabs: see ws_launch_fwd_modes)
    const char* two = getenv("FFN_BF16X6_FWD_ACCS");
    int mv_units = 0;
    if (bf16x6_prefers_mv() && !bf16x6_nine_products() && (two == nullptr || two[0] != '2') && mv_covers(chain, &mv_units))
        return launch_forward_bf16x6_mv(chain, packed_w, bias, positions, views, n, logits, saved, masks, mv_units, stream);
    if (bf16x6_nine_products()) ws_launch_fwd<WsSplit9, true, false>(chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
    else if (two == nullptr || two[0] != '2') ws_launch_fwd<WsSplit6one, true, false>(chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
    else ws_launch_fwd<WsSplit6, true, false>(chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
    return 0;
}

int launch_backward_bf16x6(const ffn_mlp_chain* chain, const uint16_t* packed_wt, const float* d_logits,
                           int64_t n, const uint32_t* masks, float* dz, void* stream) {
    int mv_units = 0;
    if (bf16x6_prefers_mv() && !bf16x6_nine_products() && mv_covers_bwd(chain, &mv_units))
        return launch_backward_bf16x6_mv(chain, packed_wt, d_logits, n, masks, dz, mv_units, stream);
    if (bf16x6_nine_products()) ws_launch_bwd<WsSplit9>(chain, packed_wt, d_logits, n, masks, dz, stream);
    else ws_launch_bwd<WsSplit6>(chain, packed_wt, d_logits, n, masks, dz, stream);
    return 0;
}

}  // namespace ffn

using namespace ffn;

static int check_chain_bf16x6(const char* what, const ffn_mlp_chain* chain, int64_t n, bool forward) {
    if (n < 0 || chain == nullptr || chain->num_steps < 1 || chain->num_steps > FFN_MAX_STEPS) return fail_arg(what);
    if (chain->wide != 0 || chain->bias_floats < 0) return fail_arg(what);      // narrow chains (<= 256 channels)
    for (int i = 0; i < chain->num_steps; ++i) {
        const ffn_step& L = chain->step[i];
        const int ot = L.out_tiles;
        if (!(ot == 1 || ot == 2 || ot == 4 || ot == 8) || (L.act_groups & 3) || L.act_groups < 0 ||
            L.act_groups > 32 || L.aux_groups < 0)
            return fail_arg(what);
        if (forward) {
            if (L.dst != 0 || (L.aux_groups & 3) || L.act_groups + L.aux_groups == 0 ||
                (L.aux_groups > 0 && (L.enc_id < 0 || L.enc_id > 1)) ||
                (L.head_off >= 0 && L.head_off + 4 + 128 * ot > 4096))
                return fail_arg(what);
        } else if ((L.act_groups == 0 && L.aux_groups == 0) ||
                   (L.aux_groups > 0 && (L.lg_col < 0 || L.lg_n < 1 || L.lg_col + L.lg_n > 4))) {
            return fail_arg(what);
        }
    }
    return 0;
}

extern "C" int ffn_mlp_bf16x6_organisation(const ffn_mlp_chain* chain, int backward) {
    int units = 0;
    if (chain == nullptr || !bf16x6_prefers_mv() || bf16x6_nine_products()) return 0;
    if (backward) return mv_covers_bwd(chain, &units) ? 1 : 0;
    const char* two = getenv("FFN_BF16X6_FWD_ACCS");
    return ((two == nullptr || two[0] != '2') && mv_covers(chain, &units)) ? 1 : 0;
}

extern "C" int ffn_mlp_forward_bf16x6(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                                      const float* positions, const float* views, int64_t n, float* logits,
                                      void* stream) {
    const char* what = "ffn_mlp_forward_bf16x6: unsupported chain or size";
    if (n == 0) return 0;
    if (const int rc = check_chain_bf16x6(what, chain, n, true)) return rc;
    launch_forward_bf16x6(chain, packed_w, bias, positions, views, n, logits, nullptr, nullptr, stream);
    return check_launch(what);
}

extern "C" int ffn_mlp_forward_bf16x6_train(const ffn_mlp_chain* chain, const uint16_t* packed_w,
                                            const float* bias, const float* positions, const float* views,
                                            int64_t n, float* logits, float* saved, uint32_t* masks,
                                            void* stream) {
    const char* what = "ffn_mlp_forward_bf16x6_train: unsupported chain or size";
    if (n == 0) return 0;
    if (saved == nullptr || masks == nullptr) return fail_arg("ffn_mlp_forward_bf16x6_train: saved and masks are required");
    if (const int rc = check_chain_bf16x6(what, chain, n, true)) return rc;
    launch_forward_bf16x6(chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
    return check_launch(what);
}

extern "C" int ffn_mlp_backward_data_bf16x6(const ffn_mlp_chain* chain, const uint16_t* packed_wt,
                                            const float* d_logits, int64_t n, const uint32_t* masks,
                                            float* dz, void* stream) {
    const char* what = "ffn_mlp_backward_data_bf16x6: unsupported chain or size";
    if (n == 0) return 0;
    if (const int rc = check_chain_bf16x6(what, chain, n, false)) return rc;
    launch_backward_bf16x6(chain, packed_wt, d_logits, n, masks, dz, stream);
    return check_launch(what);
}
