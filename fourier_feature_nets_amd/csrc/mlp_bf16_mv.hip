// bf16x6 chain kernels, third organisation: MATRIX WAVES AND VECTOR WAVES (OPT-IN arithmetic mode,
// separately labelled; the exact-f32 kernels of mlp.hip stay the parity mode and the headline).
//
// What bounds the two-waves-per-SIMD kernels of mlp_bf16_ws.hip (round 6, cycle stamps): every wave
// there is both -- it multiplies its tile, then runs its epilogue -- and the two waves of a SIMD do not
// share the matrix pipe evenly (issue is arbitrated by priority, then age), so a step ends with the
// younger wave's epilogue, the X refill and two barriers exposed (5.3 k of 18.8 k cycles per hidden
// step), and every K loop starts cold behind a barrier (0.9 k) -- eight times per features-only step.
// Here the waves of a SIMD have DIFFERENT JOBS for the whole kernel (twelve waves, three per SIMD):
//
//   * waves 0..3 (one per SIMD, the oldest) are MATRIX waves: wave m owns the output tiles m ("A") and
//     m + 4 ("B") of every step for the pass's two 32-sample blocks and issues nothing but matrix
//     instructions, LDS operand reads and weight requests -- ONE stream of "units" (one tile x one K
//     block: 12 matrix instructions, 6 operand reads, 3 weight requests) through tiles, steps and
//     passes; its weights stream L2 -> registers three units ahead out of a ring of four (the order of
//     the units is a table in LDS); the bias enters as the addend of a tile's first product;
//   * waves 4..11 (two per SIMD, a 32-sample block each) are VECTOR waves: they generate the encoding
//     features and run EVERY epilogue (ReLU, sign bits, fused heads, the three-way bf16 split, the X
//     stores, the slab / mask stores) -- a tile's accumulators reach them through a 32 KiB hand-over
//     buffer in LDS;
//   * a hidden step runs tile A then tile B: the epilogue of A (writing K blocks 0..7 of the next
//     X image IN PLACE, once every matrix wave is past K block 7 of tile B) runs under the K loop of B,
//     the epilogue of B (K blocks 8..15) under the first half of the NEXT step's K loop of A, which
//     reads K blocks 0..7 only: 96 KiB of X, 32 KiB of hand-over -- the LDS the ws kernels allocate;
//   * the barriers ("S4: B of the step before handed over", "S1: K blocks 8..15 are in X", "S2: A
//     handed over", "S3: K blocks 0..7 consumed", "S3b: K blocks 0..7 of the next image are in X") sit
//     INSIDE the matrix waves' stream, in front of the operand reads they guard: no K loop starts
//     cold, a hand-over's LDS stores run under the next tile's first unit;
//   * a features-only step multiplies a segment of eight K blocks tile by tile (A, then B over the
//     same operands) while the vector waves generate the next segment into the other half of X, one
//     barrier per segment; the first segment of the NEXT pass is generated under the last step's
//     tile B (K blocks 0..7 are free from S3 on), so that a pass starts multiplying at once;
//   * what nobody waits for -- the slab, mask and dZ stores to HBM -- goes behind the barrier that
//     releases the matrix waves.
//
// Arithmetic per accumulator is mlp_bf16_ws.hip's (the six partial products of a K block in the same
// order, K blocks ascending; one accumulator in the forward, the small products on their own in the
// backward): hidden activations, slabs, sign masks and dZ are bit-identical; the fused heads'
// partial sums meet in another order (1e-7).
//
// Covers chains whose first step is features-only (a multiple of sixteen K blocks) and whose other
// steps are 256 -> 256 (sixteen activation K blocks, eight output tiles), and their backward-data
// chains: the tiny NeRF / Fourier MLP family.  Everything else runs the ws kernels (mv_covers).
#include <type_traits>

#include "bf16_ring.h"

namespace ffn {

constexpr int kMvXBytes = 96 * 1024;               // X[K block 0..15][block 0..1][part 0..2][lane] x 16 B
constexpr int kMvHandBytes = 32 * 1024;            // hand[matrix wave][block][register quad][lane] x 16 B
constexpr int kMvBiasFloats = 4096;
constexpr int kMvMaxUnits = 1024;                  // units of a pass (tiny NeRF: 128)
constexpr int kMvLogitBytes = 4096;                // [vector wave][block][sample] x 16 B: the partial logits of a pass
constexpr size_t kMvLdsBytes = (size_t)kMvXBytes + kMvHandBytes + kEncTableBytes + kMvBiasFloats * 4 + kMvMaxUnits * 4 +
                               kMvLogitBytes;      // 160 KiB: all of a CU's LDS
constexpr int kMvKbVecs = 384;                     // float4 per K block of X: 2 blocks x 3 parts x 64 lanes
constexpr int kMvBlkVecs = 192;
constexpr int kMvTileVecs = 192;                   // float4 per (K block, tile) of the packs: 3 parts x 64 lanes
constexpr int kMvVecPerSimd = 2;                   // vector waves beside every matrix wave: a 32-sample block each
constexpr int kMvWaves = 4 + 4 * kMvVecPerSimd;
constexpr int kMvThreads = 64 * kMvWaves;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* mv_gptr;
typedef const char __attribute__((address_space(1)))* mv_gbytes;

// -DMV_STAMPS (scripts/probes/mv_stamps.py builds that variant library): cycle stamps of one pass of
// workgroup 0 -- (id, s_memtime) pairs of its matrix wave 0 and its vector wave 4
#ifdef MV_STAMPS
__device__ long long mv_stamp_buf[2][1024];
#ifdef MV_STAMPS_ENDS               // (only the two ends of a pass: a stamp is an s_memtime behind an lgkmcnt(0) -- ~50 of them move a pass)
#define MV_STAMP(w, id) do { if ((id) == 1 || (id) == 24 || (id) == 80) mv_stamp((w), (id)); } while (0)
#else
#define MV_STAMP(w, id) mv_stamp((w), (id))
#endif
#else
#define MV_STAMP(w, id) ((void)0)
#endif

struct MvCtx {
    int lane, h, s, wave, m;       // m = wave & 3: the SIMD; tiles m and m + 4
    int sub;                       // vector waves: (wave - 4) >> 2, the block of the pass whose epilogues this wave runs
    f32x4* xbuf;                   // LDS
    f32x4* hand;                   // LDS
    f32x4* logit_lds;              // LDS
    const float* enc_table;        // LDS
    const float* bias_lds;         // LDS copy of the head of the bias buffer
    const float* bias_glb;
    const i32x4* tbl4;             // LDS: refill table, entry t = units 4t+3 .. 4t+6 (mod U)
    int trips_total;               // U / 4
    const f32x4* gw;               // the chain's operand packs
    mv_gptr gwm;                   // ... from this wave's tile on (gw + m tiles): the SGPR base of every weight request
    int lane16;                    // lane * 16: the VGPR part of a request's address, next to the table entry (a byte offset)
    int b_step1, b_stride;         // bias offset of step li >= 1: b_step1 + (li - 1) * b_stride (mv_covers) -- no load in the stream
    int64_t block0, num_blocks;
    float* saved;
    char* masks;
#ifdef MV_STAMPS
    mutable int stamp_on, stamp_n;
#endif
};

#ifdef MV_STAMPS
__device__ __forceinline__ void mv_stamp(const MvCtx& w, int id) {
    if (w.stamp_on && w.stamp_n < 510) {
        const long long t = __builtin_readcyclecounter();
        if (w.lane == 0) {
            mv_stamp_buf[w.wave >> 2][2 * w.stamp_n] = id;
            mv_stamp_buf[w.wave >> 2][2 * w.stamp_n + 1] = t;
        }
        w.stamp_n += 1;
    }
}
#endif

__device__ __forceinline__ void mv_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// the matrix wave's registers that live across tiles, steps and passes
struct MvMat {
    bf16x8 wr[4][3];               // weight ring: slot = unit & 3, three parts
    // operands of the K block being multiplied, per 32-sample block: the hi and mid parts double-buffered
    // (the last products of a K block need them), the lo part REPLACED IN PLACE behind its only use (the
    // first product) by the next K block's -- 40 registers instead of 48: with three waves per SIMD a
    // wave has 168.  (The mid part in place too -- behind the fourth product, eight matrix instructions
    // before its next use -- is 8 registers less and measurably slower: the LDS round trip shows.)
    bf16x8 xh[2][2], xm[2][2], xl[2];
    // refill entries of the coming trip: four byte offsets, the same in every lane -- kept in SCALAR registers
    // (v_readfirstlane of the LDS read, once per trip): as four vector registers they were what tipped the
    // backward kernel's matrix wave over its 168 -- hipcc spilled a weight-ring slot at the head of every
    // hidden step, `s_waitcnt vmcnt(0)` + scratch store behind the request, the same again at the reload:
    // the whole ring's latency exposed twice per step
    int cur[4];
    i32x4 cur_read;                // ... between the LDS read and the readfirstlanes (two matrix instructions)
    int tq;                        // index of `cur` in the table
};

// (weight part, operand part) of the six partial products, smallest first (mlp_bf16_ws.hip)
struct MvProducts {
    static constexpr int W[6] = {0, 2, 1, 0, 1, 0};
    static constexpr int X[6] = {2, 0, 1, 1, 0, 0};
};

// One unit: the 12 matrix instructions of (one tile, one K block) out of ring slot J and hi-operand
// set HB.  READX: the operands of the next K block stream in behind the matrix instructions that free
// their registers (lo behind the first product; hi and mid -- other set -- behind the second and third);
// the ring slot of the unit before this one is requested again (three units ahead) behind
// the others.  bar (uniform): a workgroup barrier in front of the unit -- in front of the operand reads
// it guards.  Every group (one matrix instruction, at most one memory instruction) is fenced.
// FIRST = 1: the first unit of a tile's K loop in a FORWARD chain -- acc[0] holds the BIAS (read straight
// from LDS into its registers, mv_load_bias), and the first product of block 1 accumulates onto it
// before the first product of block 0 overwrites it: no accumulator is ever initialised by vector
// instructions.  FIRST = 2: a BACKWARD chain's -- the first products start from zero.
// TWO (backward data): the five small partial products accumulate on `lo`, h.h on `acc` (mlp_bf16_ws.hip:
// the matrix unit's accumulation is not round-to-nearest); they meet at the end of the K loop.
template <int J, int HB, int READX, int FIRST = 0, bool TWO = false>       // READX: bit 0 = the lo parts (in place), bit 1 = the hi and mid parts (other set)
__device__ __forceinline__ void mv_unit(const MvCtx& w, f32x16 (&acc)[2], f32x16 (&lo)[2], MvMat& r, const f32x4* xnext, bool bar) {
    typedef MvProducts P;
    // the weight requests' address: one SGPR base for the whole kernel (w.gwm) + a 32-bit VGPR offset = the
    // table entry of this unit (a byte offset, the same in every lane) + lane * 16 -- ONE vector add, issued
    // behind the fifth matrix instruction.  (Four 64-bit SGPR bases per trip -- readfirstlane, add, two
    // multiplies, add, add-with-carry each -- were 31 instructions in a row at the head of every trip: ~3 cycles
    // per matrix instruction of a stream whose pipe drains in 32.)
    unsigned voff = 0;
    if (bar) {
        MV_STAMP(w, 30);
        mv_barrier();
        MV_STAMP(w, 31);
    }
    __builtin_amdgcn_sched_barrier(0);
    auto group = [&](auto qc, auto bc) {
        // (two accumulators: h.h -- the only product on `acc` -- stays LAST in the unit.  The order matters per
        // accumulator only, but issued first it costs the backward kernel 18 %: 6.14 -> 7.25 ms, measured.)
        constexpr int q = decltype(qc)::value, b = decltype(bc)::value;
        constexpr int g = 2 * q + b;
        const bf16x8 operand = P::X[q] == 0 ? r.xh[HB][b] : (P::X[q] == 1 ? r.xm[HB][b] : r.xl[b]);
        if constexpr (TWO) {
            const f32x16 zero = (f32x16)(0.0f);
            if constexpr (q == 5)
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.wr[J][P::W[q]], operand, FIRST == 2 ? zero : acc[b], 0, 0, 0);
            else
                lo[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.wr[J][P::W[q]], operand, (FIRST == 2 && q == 0) ? zero : lo[b], 0, 0, 0);
        } else if constexpr (FIRST == 1 && q == 0) {
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.wr[J][P::W[q]], operand, acc[0], 0, 0, 0);
        } else if constexpr (FIRST == 2 && q == 0) {
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.wr[J][P::W[q]], operand, (f32x16)(0.0f), 0, 0, 0);
        } else {
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r.wr[J][P::W[q]], operand, acc[b], 0, 0, 0);
        }
#ifdef MV_KO_PAIR_X
        if constexpr (READX == 3) {
#else
        {
#endif
        if constexpr ((READX & 1) != 0 && q == 0) r.xl[b] = __builtin_bit_cast(bf16x8, xnext[b * kMvBlkVecs + 2 * 64]);
        if constexpr ((READX & 2) != 0 && q == 1) r.xh[HB ^ 1][b] = __builtin_bit_cast(bf16x8, xnext[b * kMvBlkVecs]);
        if constexpr ((READX & 2) != 0 && q == 2) r.xm[HB ^ 1][b] = __builtin_bit_cast(bf16x8, xnext[b * kMvBlkVecs + 64]);
        }
        if constexpr (g == 5) asm volatile("v_add_u32 %0, %1, %2" : "=v"(voff) : "s"(r.cur[J]), "v"(w.lane16));
        // the weight requests: behind groups 6, 7, 8
        constexpr int slot = g >= 6 && g < 9 ? g - 6 : -1;
#ifdef MV_KO_PAIR_W
        if constexpr (slot >= 0 && READX == 3)
#else
        if constexpr (slot >= 0)
#endif
            r.wr[(J + 3) & 3][slot] = __builtin_bit_cast(bf16x8, *reinterpret_cast<mv_gptr>(reinterpret_cast<mv_gbytes>(w.gwm) + (uint64_t)voff + slot * 1024));
        // the table entry of the NEXT trip: behind the last request of this one (five matrix instructions
        // ahead of its first use)
        if constexpr (J == 3 && g == 9) {
            r.tq = r.tq + 1 < w.trips_total ? r.tq + 1 : 0;
            r.cur_read = w.tbl4[r.tq];
        }
        if constexpr (J == 3 && g == 11) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r.cur[j] = __builtin_amdgcn_readfirstlane(r.cur_read[j]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> i0;
    typedef std::integral_constant<int, 1> i1;
    if constexpr (FIRST == 1) { group(i0{}, i1{}); group(i0{}, i0{}); }
    else { group(i0{}, i0{}); group(i0{}, i1{}); }
    group(i1{}, i0{}); group(i1{}, i1{});
    group(std::integral_constant<int, 2>{}, i0{}); group(std::integral_constant<int, 2>{}, i1{});
    group(std::integral_constant<int, 3>{}, i0{}); group(std::integral_constant<int, 3>{}, i1{});
    group(std::integral_constant<int, 4>{}, i0{}); group(std::integral_constant<int, 4>{}, i1{});
    group(std::integral_constant<int, 5>{}, i0{}); group(std::integral_constant<int, 5>{}, i1{});
}

// The K loop of ONE tile over `trips` x 4 K blocks of X from K block g0 (operands of g0 already in the
// registers); the last unit streams K block g_after in -- the first one of whatever comes next.
// bars: bit 4 t + j = barrier in front of unit j of trip t.  FIRST != 0: the tile starts here (1: acc[0]
// = bias, 2: from zero).  TWO: see mv_unit.
template <int FIRST, bool TWO>
__device__ __forceinline__ void mv_k_loop2(const MvCtx& w, MvMat& r, f32x16 (&acc)[2], f32x16 (&lo)[2], int g0, int trips,
                                           int g_after, unsigned bars) {
    const f32x4* xb = w.xbuf + w.lane;
    if constexpr (FIRST != 0) {
        mv_unit<0, 0, 3, FIRST, TWO>(w, acc, lo, r, xb + (g0 + 1) * kMvKbVecs, (bars & 1u) != 0);
        mv_unit<1, 1, 3, 0, TWO>(w, acc, lo, r, xb + (g0 + 2) * kMvKbVecs, (bars & 2u) != 0);
        mv_unit<2, 0, 3, 0, TWO>(w, acc, lo, r, xb + (g0 + 3) * kMvKbVecs, (bars & 4u) != 0);
        mv_unit<3, 1, 3, 0, TWO>(w, acc, lo, r, xb + (trips == 1 ? g_after : g0 + 4) * kMvKbVecs, (bars & 8u) != 0);
    }
    // (not unrolled: with constant trip counts hipcc unrolls, turns the operand addresses beyond the 64 KiB
    // immediate range into values it keeps -- and SPILLS them: scratch traffic inside the stream, in
    // order with the weight requests)
#pragma nounroll
    for (int t = FIRST != 0 ? 1 : 0; t < trips; ++t) {
        const int g = g0 + 4 * t;
        const int g4 = t + 1 == trips ? g_after : g + 4;
        const unsigned bt = bars >> (4 * t);
        mv_unit<0, 0, 3, 0, TWO>(w, acc, lo, r, xb + (g + 1) * kMvKbVecs, (bt & 1u) != 0);
        mv_unit<1, 1, 3, 0, TWO>(w, acc, lo, r, xb + (g + 2) * kMvKbVecs, (bt & 2u) != 0);
        mv_unit<2, 0, 3, 0, TWO>(w, acc, lo, r, xb + (g + 3) * kMvKbVecs, (bt & 4u) != 0);
        mv_unit<3, 1, 3, 0, TWO>(w, acc, lo, r, xb + g4 * kMvKbVecs, (bt & 8u) != 0);
    }
}
template <bool FIRST = false>
__device__ __forceinline__ void mv_k_loop(const MvCtx& w, MvMat& r, f32x16 (&acc)[2], int g0, int trips, int g_after,
                                          unsigned bars) {
    mv_k_loop2<FIRST ? 1 : 0, false>(w, r, acc, acc, g0, trips, g_after, bars);
}

// the operand registers <- X K block G, hi set 0 (a cold start: the first step of a pass)
__device__ __forceinline__ void mv_read_x0(const MvCtx& w, MvMat& r, int G) {
    const f32x4* p = w.xbuf + G * kMvKbVecs + w.lane;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        r.xh[0][b] = __builtin_bit_cast(bf16x8, p[b * kMvBlkVecs]);
        r.xm[0][b] = __builtin_bit_cast(bf16x8, p[b * kMvBlkVecs + 64]);
        r.xl[b] = __builtin_bit_cast(bf16x8, p[b * kMvBlkVecs + 128]);
    }
}

// the bias of output tile o in the accumulator layout of one block, straight into a tile's acc[0] (see FIRST)
__device__ __forceinline__ void mv_load_bias(const MvCtx& w, int b_off, int o, f32x16& acc0) {
    const float* bv = w.bias_lds + b_off + 32 * o + 4 * w.h;         // (mv_covers: the whole bias buffer is staged)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + 8 * q);
#pragma unroll
        for (int p = 0; p < 4; ++p) acc0[4 * q + p] = b4[p];
    }
}

// a tile's accumulators into the hand-over buffer of this SIMD pair
__device__ __forceinline__ void mv_hand_over(const MvCtx& w, const f32x16 (&acc)[2]) {
    f32x4* dst = w.hand + (w.m * 2) * 256 + w.lane;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int p = 0; p < 4; ++p) v[p] = acc[b][4 * q + p];
            dst[(b * 4 + q) * 64] = v;
        }
}
#ifdef MV_KO_VECTOR_IDLE             // (timing only: the vector waves keep the barriers and do nothing else)
#define MV_VECTOR_WORK 0
#else
#define MV_VECTOR_WORK 1
#endif
__device__ __forceinline__ void mv_take_over(const MvCtx& w, f32x16& acc) {
    if (!MV_VECTOR_WORK) return;
    const f32x4* src = w.hand + (w.m * 2 + w.sub) * 256 + w.lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = src[q * 64];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[4 * q + p] = v[p];
    }
}

// (X K block of feature K block k: k & 15 -- segments of eight K blocks alternate between the two halves
// of X; the number of segments is even (mv_covers), so the LAST segment of the step sits in K blocks
// 8..15: its epilogue of tile A writes K blocks 0..7 of the next image while tile B still reads it, and
// the FIRST one in K blocks 0..7: the vector waves generate it, for the next pass, while the last step's
// tile B reads K blocks 8..15)

// ---------------------------------------------------------------------------------- the matrix waves
// Barriers of a pass, in order (the vector waves' code has the same list):
//   features-only: F_0 .. F_{segments-2} (segment s + 1 is in X, segment s consumed), S2, S3, S3b
//   hidden:        S4 (of the step before), S1, S2, S3, S3b
//   end of a pass: S4 (of the last step)
// (the accumulators of the two tiles live across steps: the barrier that publishes a hand-over sits in
// front of the SECOND unit of whatever the matrix wave multiplies next -- the hand-over's LDS stores and
// the barrier's wait for them run under the first unit's matrix instructions)
struct MvAcc {
    f32x16 a[2], b[2];
};

__device__ __forceinline__ void mv_matrix_features(int segments, int b_off, const MvCtx& w, MvMat& r, MvAcc& c) {
    mv_load_bias(w, b_off, w.m, c.a[0]);
    mv_load_bias(w, b_off, w.m + 4, c.b[0]);
    mv_read_x0(w, r, 0);                           // (segment 0 is in X: the first pass's prologue, or S4 of the pass before)
    MV_STAMP(w, 9);
    // every segment tile by tile (eight K blocks of tile A, then the same eight of tile B): the barrier F
    // in front of the last unit of B -- in front of its reads of the next segment's first K block
    MV_STAMP(w, 32);
    mv_k_loop<true>(w, r, c.a, 0, 2, 0, 0u);
    mv_k_loop<true>(w, r, c.b, 0, 2, 8, 0x80u);                                              // F_0: unit 7
#pragma nounroll
    for (int seg = 1; seg + 1 < segments; ++seg) {
        const int g0 = (seg & 1) * 8;
        MV_STAMP(w, 32);
        mv_k_loop(w, r, c.a, g0, 2, g0, 0u);
        mv_k_loop(w, r, c.b, g0, 2, g0 ^ 8, 0x80u);                                          // F_seg: unit 7
    }
    MV_STAMP(w, 10);
    // the last segment (K blocks 8..15 of X)
    mv_k_loop(w, r, c.a, 8, 2, 8, 0u);
    MV_STAMP(w, 11);
    mv_hand_over(w, c.a);
    mv_load_bias(w, w.b_step1, w.m, c.a[0]);       // (tile A of the next step starts from it)
    mv_k_loop(w, r, c.b, 8, 2, 0, 0xA2u);                                                    // S2, S3, S3b: units 1, 5, 7
    MV_STAMP(w, 13);
    // (tile B is handed over in front of the next step's tile A)
}

// (b_off: this step's biases; the next step's are b_stride further -- scalar arithmetic: a load of a step's
// descriptor here, from the kernel's arguments, is an s_load and an lgkmcnt(0) in front of the tile's first
// matrix instruction -- with the hand-over's eight LDS stores in the queue)
__device__ __forceinline__ void mv_matrix_hidden(int b_off, bool last_step, const MvCtx& w, MvMat& r, MvAcc& c) {
    MV_STAMP(w, 20);
    mv_hand_over(w, c.b);                          // tile B of the step before
    mv_load_bias(w, b_off, w.m + 4, c.b[0]);
    mv_k_loop<true>(w, r, c.a, 0, 4, 0, 0x82u);                                              // S4, S1: units 1, 7
    MV_STAMP(w, 21);
    mv_hand_over(w, c.a);
    if (!last_step) mv_load_bias(w, b_off + w.b_stride, w.m, c.a[0]);
    // (the last step stores nothing into X: S3b right behind S3, and the vector waves have the rest of
    // this K loop for the next pass's first segment of features)
    mv_k_loop<true>(w, r, c.b, 0, 4, 0, last_step ? 0x182u : 0x882u);                        // S2, S3, S3b: units 1, 7, 11 (1, 7, 8)
    MV_STAMP(w, 23);
}

// ---------------------------------------------------------------------------------- the vector waves
struct MvVec {
    float x0, x1, x2, v0, v1, v2;  // inputs of this wave's feature block (wave & 1)
    float logit[4];                // fused heads: partial sums over the two tiles of this SIMD, this wave's block
};

// feature K blocks k0 .. k0 + count - 1 of this wave's block into X; wave `rank` of `stride` generators
__device__ __forceinline__ void mv_generate(const ffn_mlp_chain& ch, const ffn_step& L, const MvCtx& w, const MvVec& v,
                                            int64_t block0, int k0, int count, int rank, int stride) {
    Enc16 enc;
    const ffn_encoding& e = ch.enc[L.enc_id];
    enc.tab = w.enc_table + L.enc_id * kEncTablePitch;
    enc.F = e.num_freq;
    enc.raw = (e.include_input != 0 || e.num_freq == 0) ? 1 : 0;
    enc.scale = e.scale;
    const float p0 = L.enc_id == 0 ? v.x0 : v.v0;
    const float p1 = L.enc_id == 0 ? v.x1 : v.v1;
    const float p2 = L.enc_id == 0 ? v.x2 : v.v2;
    const int g_trig = e.num_freq >> 3;            // K blocks whose eight frequencies are all real
    const int fb = w.wave & 1;
    f32x4* fsave = nullptr;
    if (w.saved != nullptr && L.save_enc_slot >= 0 && block0 + fb < w.num_blocks)
        fsave = reinterpret_cast<f32x4*>(w.saved + ch.slot_offset[L.save_enc_slot] * w.num_blocks * 32) +
                (block0 + fb) * (int64_t)(ch.slot_channels[L.save_enc_slot] * 8);
    for (int G = rank; G < count && MV_VECTOR_WORK; G += stride) {
        const int k = k0 + G;
        float f[8];
        if (k < g_trig) features16_lockstep<true>(enc, k, w.h, p0, p1, p2, f);
        else features16_lockstep<false>(enc, k, w.h, p0, p1, p2, f);
        if (fsave != nullptr) {
            const int cq = 4 * k + 2 * w.h;
            f32x4 f0, f1;
#pragma unroll
            for (int p = 0; p < 4; ++p) { f0[p] = f[p]; f1[p] = f[4 + p]; }
            __builtin_nontemporal_store(f0, &fsave[saved_index16(cq, w.s)]);
            __builtin_nontemporal_store(f1, &fsave[saved_index16(cq + 1, w.s)]);
        }
        bf16x8 fp[3];
        split8x3(f, fp[0], fp[1], fp[2]);
        f32x4* dst = w.xbuf + (k & 15) * kMvKbVecs + fb * kMvBlkVecs + w.lane;
#pragma unroll
        for (int part = 0; part < 3; ++part) dst[64 * part] = __builtin_bit_cast(f32x4, fp[part]);
    }
}

// sign-mask byte of (slot, block, lane, tile o) for a step of eight tiles (mlp.hip's format)
__device__ __forceinline__ int64_t mv_mask_at(int slot, int64_t num_blocks, int64_t block, int lane, int o) {
    return (((int64_t)slot * num_blocks + block) * 64 + lane) * 16 + 4 * (o >> 1) + ((o & 1) ? 0 : 2);
}

// The epilogue of (tile o, this wave's block) in two parts.  What the matrix waves wait for -- ReLU, sign bits,
// fused head, the three-way split into res (the X stores follow behind a barrier) -- and what nobody waits
// for: the slab and mask stores to HBM (a nontemporal 1 KiB store costs the wave ~200 cycles beside the
// weight stream), which run BEHIND the barrier the matrix waves are released by.
struct MvDeferred {
    float y[16];                   // the tile's outputs for this wave's block (after the activation)
    unsigned sign_bits;
};

template <bool TRAIN>                                // (inference: no sign bits, nothing kept for the HBM stores)
__device__ __forceinline__ void mv_epilogue_t(const ffn_step& L, bool last_step, const MvCtx& w, MvVec& v, int o,
                                              const f32x16& acc, bf16x8 (&res)[2][3], MvDeferred& d) {
    if (!MV_VECTOR_WORK) return;
    const bool fused_head = L.head_off >= 0;
    const float* hw = w.bias_lds + (fused_head ? L.head_off : 0) + 4 + 16 * w.h;
    if (fused_head && o == 0 && w.h == 0) {
        const f32x4 hb = *reinterpret_cast<const f32x4*>(w.bias_lds + L.head_off);
#pragma unroll
        for (int c = 0; c < 4; ++c) v.logit[c] += hb[c];
    }
    const int relu_floor = L.relu ? 0 : (int)0x80000000;
    unsigned sign_bits = 0u;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = acc[8 * half + j];
            if (TRAIN) sign_bits = __builtin_amdgcn_alignbit(sign_bits, __builtin_bit_cast(unsigned, 0.0f - a), 31);
            y[j] = __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, a), relu_floor));
            if (TRAIN) d.y[8 * half + j] = y[j];
        }
        if (fused_head) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int group = 4 * o + 2 * half + (j >> 2);
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(hw + group * 32 + (j & 3) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) v.logit[c] = __builtin_fmaf(y[j], w4[c], v.logit[c]);
            }
        }
        if (!last_step) split8x3(y, res[half][0], res[half][1], res[half][2]);
    }
    d.sign_bits = sign_bits;
}
__device__ __forceinline__ void mv_epilogue(const ffn_step& L, bool last_step, const MvCtx& w, MvVec& v, int o,
                                            const f32x16& acc, bf16x8 (&res)[2][3], MvDeferred& d) {
    if (w.saved != nullptr || w.masks != nullptr) mv_epilogue_t<true>(L, last_step, w, v, o, acc, res, d);
    else mv_epilogue_t<false>(L, last_step, w, v, o, acc, res, d);
}

__device__ __forceinline__ void mv_epilogue_stores(const ffn_mlp_chain& ch, const ffn_step& L, const MvCtx& w, int o,
                                                   const MvDeferred& d) {
    if (!MV_VECTOR_WORK) return;
    int save_s = w.s, save_h = w.h, e_lane = w.lane;
    asm volatile("" : "+v"(save_s), "+v"(save_h), "+v"(e_lane));     // (see mlp_bf16.hip: no hoisted address tables)
    const int64_t block = w.block0 + w.sub;
    if (block >= w.num_blocks) return;
    if (w.saved != nullptr && L.out_slot >= 0) {
        f32x4* save_out = reinterpret_cast<f32x4*>(w.saved + ch.slot_offset[L.out_slot] * w.num_blocks * 32) +
                          block * (int64_t)(ch.slot_channels[L.out_slot] * 8);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 y0, y1;
#pragma unroll
            for (int p = 0; p < 4; ++p) { y0[p] = d.y[8 * half + p]; y1[p] = d.y[8 * half + 4 + p]; }
            const int cq = 2 * (4 * o + 2 * half) + save_h;
            __builtin_nontemporal_store(y0, &save_out[saved_index16(cq, save_s)]);
            __builtin_nontemporal_store(y1, &save_out[saved_index16(cq + 2, save_s)]);
        }
    }
    if (w.masks != nullptr && L.relu && L.mask_slot >= 0)
        *reinterpret_cast<uint16_t*>(w.masks + mv_mask_at(L.mask_slot, w.num_blocks, block, e_lane, o)) =
            (uint16_t)(d.sign_bits & 0xffffu);
}

// (tile o, this wave's block) of the output = its share of K blocks 2 o, 2 o + 1 of the next X image
__device__ __forceinline__ void mv_store_x(const MvCtx& w, int o, const bf16x8 (&res)[2][3]) {
    if (!MV_VECTOR_WORK) return;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4* dst = w.xbuf + (2 * o + half) * kMvKbVecs + w.sub * kMvBlkVecs + w.lane;
#pragma unroll
        for (int part = 0; part < 3; ++part) dst[64 * part] = __builtin_bit_cast(f32x4, res[half][part]);
    }
}

// a step seen from a vector wave, from its barrier S2 on (what comes before differs: the features'
// segments, or S1).  next != nullptr (the LAST step of a pass that has a successor): segment 0 of the
// NEXT pass's features is generated between S3b and S4 -- K blocks 0..7 of X are free from S3 on, the
// matrix waves start the next pass on them right behind S4.  The HBM stores of tile B are left to the
// caller (`late`): they go behind the next barrier.
__device__ __forceinline__ void mv_vector_tail(const ffn_mlp_chain& ch, const ffn_step& L, bool last_step,
                                               const MvCtx& w, MvVec& v, const MvVec* next, int64_t next_block0,
                                               MvDeferred& late) {
    f32x16 acc;
    bf16x8 res[2][3];
    MvDeferred early;
    MV_STAMP(w, 60);
    mv_barrier();                                  // S2: tile A handed over
    MV_STAMP(w, 61);
    mv_take_over(w, acc);
    mv_epilogue(L, last_step, w, v, w.m, acc, res, early);
    MV_STAMP(w, 62);
    mv_barrier();                                  // S3: K blocks 0..7 of X are consumed
    MV_STAMP(w, 63);
    if (!last_step) mv_store_x(w, w.m, res);
    MV_STAMP(w, 64);
    mv_barrier();                                  // S3b: K blocks 0..7 of the next image are in X
    MV_STAMP(w, 65);
    mv_epilogue_stores(ch, L, w, w.m, early);
    if (next != nullptr) {
        mv_generate(ch, ch.step[0], w, *next, next_block0, 0, 8, (w.wave - 4) >> 1, 2 * kMvVecPerSimd);
        MV_STAMP(w, 68);
    }
    mv_barrier();                                  // S4: tile B handed over, X consumed
    MV_STAMP(w, 66);
    mv_take_over(w, acc);
    mv_epilogue(L, last_step, w, v, w.m + 4, acc, res, late);
    if (!last_step) mv_store_x(w, w.m + 4, res);
    MV_STAMP(w, 67);
    // (the next barrier -- S1 of the next step, or F_0 of the next pass -- publishes K blocks 8..15)
}

// the SIMD pairs' partial logits of a pass meet through LDS: every vector wave leaves its sums there, and
// behind the next barrier (F_0 of the next pass) waves 4 and 5 add the four up, a block each
__device__ __forceinline__ void mv_leave_logits(const MvCtx& w, const MvVec& v) {
    f32x4 part;
#pragma unroll
    for (int c = 0; c < 4; ++c) part[c] = v.logit[c] + __shfl_xor(v.logit[c], 32);
    if (w.h == 0) w.logit_lds[(w.m * 2 + w.sub) * 32 + w.s] = part;
}
__device__ __forceinline__ void mv_collect_logits(const MvCtx& w, int64_t block0, int64_t n, float* logits) {
    if (w.m == 0 && w.h == 0) {                    // (waves 4 and 8: a block each)
        const int b = w.sub;
        const int64_t block = block0 + b;
        const int64_t sample = block * 32 + w.s;
        f32x4 out = w.logit_lds[b * 32 + w.s];
#pragma unroll
        for (int k = 1; k < 4; ++k) out += w.logit_lds[(k * 2 + b) * 32 + w.s];
        if (block < w.num_blocks && sample < n) reinterpret_cast<f32x4*>(logits)[sample] = out;
    }
}

// ---------------------------------------------------------------------------------- the kernel
// units of a pass in the matrix waves' order: entry = (flat K block of the chain) * 8 + (0: tile A, 4: tile B)
__device__ __forceinline__ int mv_build_units(const ffn_mlp_chain& ch, int* units) {
    int u = 0, flat = 0;
    for (int li = 0; li < ch.num_steps; ++li) {
        const int kb_act = ch.step[li].act_groups >> 1, kb_feat = ch.step[li].aux_groups >> 1;
        if (kb_act == 0) {
            for (int k0 = 0; k0 < kb_feat; k0 += 8)          // segment by segment, tile A then tile B
                for (int t = 0; t < 2; ++t)
                    for (int k = k0; k < k0 + 8; ++k) units[u++] = (flat + k) * 8 + 4 * t;
            flat += kb_feat;
        } else {
            for (int t = 0; t < 2; ++t)
                for (int k = 0; k < kb_act; ++k) units[u++] = (flat + k) * 8 + 4 * t;
            flat += kb_act;
        }
    }
    return u;
}

__global__ void __launch_bounds__(kMvThreads)
mlp_forward_bf16_mv_kernel(const ffn_mlp_chain ch, const uint16_t* __restrict__ packed,
                           const float* __restrict__ bias, const float* __restrict__ positions,
                           const float* __restrict__ views, int64_t n, float* __restrict__ logits,
                           float* __restrict__ saved, uint32_t* __restrict__ masks, int num_units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* enc_table = reinterpret_cast<float*>(smem + kMvXBytes + kMvHandBytes);
    float* bias_lds = reinterpret_cast<float*>(smem + kMvXBytes + kMvHandBytes + kEncTableBytes);
    int* tbl = reinterpret_cast<int*>(smem + kMvXBytes + kMvHandBytes + kEncTableBytes + kMvBiasFloats * 4);
    stage_encoding_tables(ch.enc, enc_table, threadIdx.x, kMvThreads);
    {
        const int staged = ch.bias_floats < kMvBiasFloats ? ch.bias_floats : kMvBiasFloats;
        for (int i = threadIdx.x; i < staged; i += kMvThreads) bias_lds[i] = bias[i];
    }
    if (threadIdx.x == 0) {
        // the refill table: tbl[i] = unit (i + 3) mod U, so that int4 entry t holds what trip t refills
        int* units = reinterpret_cast<int*>(smem);             // (X is not in use yet)
        const int u_total = mv_build_units(ch, units);
        for (int i = 0; i < u_total; ++i) tbl[i] = units[(i + 3) % u_total] * (kMvTileVecs * 16);       // (byte offsets)
    }
    MvCtx w;
    w.lane = threadIdx.x & 63;
    w.h = w.lane >> 5;
    w.s = w.lane & 31;
    w.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.m = w.wave & 3;
    w.sub = w.wave >= 4 ? (w.wave - 4) >> 2 : 0;
    w.xbuf = reinterpret_cast<f32x4*>(smem);
    w.hand = reinterpret_cast<f32x4*>(smem + kMvXBytes);
    w.logit_lds = reinterpret_cast<f32x4*>(smem + kMvXBytes + kMvHandBytes + kEncTableBytes + kMvBiasFloats * 4 + kMvMaxUnits * 4);
    w.enc_table = enc_table;
    w.bias_lds = bias_lds;
    w.bias_glb = bias;
    w.tbl4 = reinterpret_cast<const i32x4*>(tbl);
    w.trips_total = num_units >> 2;
    w.gw = reinterpret_cast<const f32x4*>(packed + ch.step[0].w_off);
    w.gwm = (mv_gptr)(w.gw + w.m * kMvTileVecs);
    w.lane16 = w.lane * 16;
    w.b_step1 = (int)ch.step[1].b_off;
    w.b_stride = ch.num_steps > 2 ? (int)(ch.step[2].b_off - ch.step[1].b_off) : 0;
    w.saved = saved;
    w.masks = reinterpret_cast<char*>(masks);
    w.num_blocks = (n + 31) / 32;
    const int64_t passes = (w.num_blocks + 1) / 2;
    const bool matrix = w.wave < 4;
    __syncthreads();                               // tables, biases and the unit table are staged

    const ffn_step& L0 = ch.step[0];
    const int segments = L0.aux_groups >> 4;       // (K blocks: a multiple of sixteen, mv_covers)
    const int num_steps = ch.num_steps, b_step0 = (int)L0.b_off;
    const int fb = w.wave & 1;
    auto inputs_of = [&](int64_t pass, MvVec& dst) {
        int64_t block = pass * 2 + fb;
        block = block < w.num_blocks ? block : w.num_blocks - 1;
        const int64_t sample = block * 32 + w.s;
        const int64_t src = sample < n ? sample : n - 1;
        dst.x0 = positions[src * 3 + 0]; dst.x1 = positions[src * 3 + 1]; dst.x2 = positions[src * 3 + 2];
        dst.v0 = dst.v1 = dst.v2 = 0.0f;
        if (views != nullptr) {
            dst.v0 = views[src * 3 + 0]; dst.v1 = views[src * 3 + 1]; dst.v2 = views[src * 3 + 2];
        }
    };
    // ---- the first pass's first segment of features: every wave generates (2 items each)
    MvVec v, vnext;
    inputs_of(blockIdx.x, v);
    mv_generate(ch, L0, w, v, (int64_t)blockIdx.x * 2, 0, 8, w.wave >> 1, kMvWaves / 2);
    mv_barrier();                                                                            // P1

    if (matrix) {
        // (the matrix pipe must never wait for an issue slot the partner's vector work took)
        __builtin_amdgcn_s_setprio(3);
        MvMat r;
        MvAcc c;
        // units 0, 1, 2 into ring slots 0, 1, 2 (table entry T - 1 = units U-1, 0, 1, 2); entry 0 next
        const i32x4 first = w.tbl4[w.trips_total - 1];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            mv_gptr base = reinterpret_cast<mv_gptr>(reinterpret_cast<mv_gbytes>(w.gwm) + (uint64_t)(unsigned)(first[j + 1] + w.lane16));
#pragma unroll
            for (int part = 0; part < 3; ++part) r.wr[j][part] = __builtin_bit_cast(bf16x8, base[part * 64]);
        }
        r.tq = 0;
        {
            const i32x4 entry = w.tbl4[0];
#pragma unroll
            for (int j = 0; j < 4; ++j) r.cur[j] = __builtin_amdgcn_readfirstlane(entry[j]);
        }
        for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
#ifdef MV_STAMPS
            w.stamp_on = blockIdx.x == 0 && pass == blockIdx.x + 2 * (int64_t)gridDim.x && w.wave == 0;
            w.stamp_n = 0;
#endif
            MV_STAMP(w, 1);
            mv_matrix_features(segments, b_step0, w, r, c);
            for (int li = 1; li < num_steps; ++li)
                mv_matrix_hidden(w.b_step1 + (li - 1) * w.b_stride, li + 1 == num_steps, w, r, c);
            mv_hand_over(w, c.b);                  // (the last tile of the pass: nothing follows to carry its barrier)
            mv_barrier();                                                                    // S4
            MV_STAMP(w, 24);
        }
        mv_barrier();                                                                        // R: the last pass's logits
        return;
    }

    // ---- a vector wave
    bool pending = false;                          // the partial logits of the pass before wait in LDS
    int64_t pending_block0 = 0;
    for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
#ifdef MV_STAMPS
        w.stamp_on = blockIdx.x == 0 && pass == blockIdx.x + 2 * (int64_t)gridDim.x && w.wave == 4;
        w.stamp_n = 0;
#endif
        MV_STAMP(w, 1);
        w.block0 = pass * 2;
        const bool has_next = pass + gridDim.x < passes;
        if (has_next) inputs_of(pass + gridDim.x, vnext);
#pragma unroll
        for (int c = 0; c < 4; ++c) v.logit[c] = 0.0f;
        for (int seg = 1; seg < segments; ++seg) {
            mv_generate(ch, L0, w, v, w.block0, 8 * seg, 8, (w.wave - 4) >> 1, 2 * kMvVecPerSimd);
            MV_STAMP(w, 50);
            mv_barrier();                                                                    // F_{seg-1}
            MV_STAMP(w, 51);
            if (seg == 1 && pending) {
                mv_collect_logits(w, pending_block0, n, logits);
                pending = false;
            }
        }
        MvDeferred late;
        mv_vector_tail(ch, L0, false, w, v, nullptr, 0, late);
        for (int li = 1; li < ch.num_steps; ++li) {
            const bool last = li + 1 == ch.num_steps;
            MV_STAMP(w, 70);
            mv_barrier();                                                                    // S1
            MV_STAMP(w, 71);
            mv_epilogue_stores(ch, ch.step[li - 1], w, w.m + 4, late);         // (tile B of the step before)
            mv_vector_tail(ch, ch.step[li], last, w, v, (last && has_next) ? &vnext : nullptr,
                           w.block0 + 2 * (int64_t)gridDim.x, late);
        }
        mv_epilogue_stores(ch, ch.step[ch.num_steps - 1], w, w.m + 4, late);
        mv_leave_logits(w, v);
        pending = true;
        pending_block0 = w.block0;
        if (has_next) {
            v.x0 = vnext.x0; v.x1 = vnext.x1; v.x2 = vnext.x2;
            v.v0 = vnext.v0; v.v1 = vnext.v1; v.v2 = vnext.v2;
        }
    }
    mv_barrier();                                                                            // R
    if (pending) mv_collect_logits(w, pending_block0, n, logits);
}

// Chains this organisation covers: a features-only first step of 16 j K blocks, then 256 -> 256 steps;
// eight output tiles everywhere; a bias buffer that fits its LDS copy.
bool mv_covers(const ffn_mlp_chain* chain, int* num_units) {
    if (chain->num_steps < 2 || chain->wide != 0 || chain->bias_floats > kMvBiasFloats) return false;
    int units = 0;
    for (int i = 0; i < chain->num_steps; ++i) {
        const ffn_step& L = chain->step[i];
        const int kb_act = L.act_groups >> 1, kb_feat = L.aux_groups >> 1;
        if (L.out_tiles != 8) return false;
        // (the matrix waves compute a step's bias offset: equally spaced from step 1 on)
        if (i >= 2 && L.b_off - chain->step[1].b_off != (i - 1) * (chain->step[2].b_off - chain->step[1].b_off)) return false;
        if (i == 0) {
            if (kb_act != 0 || kb_feat < 16 || (kb_feat & 15) != 0) return false;
            units += 2 * kb_feat;
        } else {
            if (kb_act != 16 || kb_feat != 0) return false;
            units += 32;
        }
    }
    if (units > kMvMaxUnits) return false;
    *num_units = units;
    return true;
}

int launch_forward_bf16x6_mv(const ffn_mlp_chain* chain, const uint16_t* packed_w, const float* bias,
                             const float* positions, const float* views, int64_t n, float* logits,
                             float* saved, uint32_t* masks, int num_units, void* stream) {
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int val = 0;
        if (hipDeviceGetAttribute(&val, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && val > 0) cus = val;
    }
    const int64_t passes = ((n + 31) / 32 + 1) / 2;
    const int64_t grid = passes < cus ? passes : cus;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_forward_bf16_mv_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMvLdsBytes);
    hipLaunchKernelGGL(mlp_forward_bf16_mv_kernel, dim3((unsigned)grid), dim3(kMvThreads), kMvLdsBytes,
                       (hipStream_t)stream, *chain, packed_w, bias, positions, views, n, logits, saved, masks, num_units);
    return 0;
}

// ================================================================================== backward data
// The same organisation for the backward-data chain of the same models: step 0 takes d(loss)/d(logits)
// through the fused head, every other step is 256 -> 256.  Two accumulators per (tile, block) -- one tile
// at a time in the matrix wave's registers: it adds them up and hands the tile over before the next one
// starts.  The vector waves' epilogue: the ReLU mask of the layer being differentiated, the dZ slab, the
// three-way split.
//
// STEP 0 IS THE VECTOR WAVES' (round 6): four real K rows per output -- 64 fused multiply-adds per lane
// and tile on head weights kept in LDS as f32 (hi + mid + lo of the pack, exactly).  As matrix work it was
// 48 matrix instructions (half of them on a zero K block) inside a chain of three barriers in which
// nothing overlapped: 6-9 k cycles of a 40 k pass.  Its output is the X image of step 1, written where a
// hidden step's epilogues write theirs: tiles 0..3 (K blocks 0..7) of the NEXT pass between S3 and S3b of
// the last step, tiles 4..7 (K blocks 8..15) behind the barrier that ends the pass, under the first half of
// the next pass's first K loop.  The matrix waves' stream is hidden steps only, pass after pass.
// (The head term in f32 arithmetic instead of six bf16 products: this chain's dZ differs from the
// two-waves-per-SIMD kernels' in the last bits -- 1e-7 relative -- where the forward's slabs are identical.)
#ifdef MV_KO_BWD_ONE_ACC            // (timing only: all six products on one accumulator)
constexpr bool kMvBwdTwo = false;
#else
constexpr bool kMvBwdTwo = true;
#endif
constexpr int kMvHeadBytes = 256 * 16;             // head weights W[channel][4] as f32
constexpr size_t kMvBwdLdsBytes = (size_t)kMvXBytes + kMvHandBytes + kMvMaxUnits * 4 + kMvHeadBytes;

struct MvAcc2 {
    f32x16 main[2], lo[2];
};

__device__ __forceinline__ void mv_meet_and_hand_over(const MvCtx& w, MvAcc2& c) {
    if constexpr (kMvBwdTwo) {
#pragma unroll
        for (int b = 0; b < 2; ++b) c.main[b] += c.lo[b];
    }
    mv_hand_over(w, c.main);
}

// a hidden step.  Barriers in the stream: S4 (tile B of the step before handed over; not in a pass's first
// step: nothing was), S1 (K blocks 8..15 are in X) in K(A); S2, S3, S3b in K(B) -- the LAST step's S3b in
// front of its last unit, whose operand reads are K block 0 of the next pass's image (the vector waves
// compute it from S3 on)
__device__ __forceinline__ void mv_matrix_bwd_hidden(bool first_step, bool last_step, const MvCtx& w, MvMat& r, MvAcc2& c) {
    MV_STAMP(w, 20);
    mv_k_loop2<2, kMvBwdTwo>(w, r, c.main, c.lo, 0, 4, 0, first_step ? 0x80u : 0x82u);           // (S4,) S1: units (1,) 7
    MV_STAMP(w, 21);
    mv_meet_and_hand_over(w, c);                   // tile A
    mv_k_loop2<2, kMvBwdTwo>(w, r, c.main, c.lo, 0, 4, 0, last_step ? 0x8082u : 0x882u);          // S2, S3, S3b: units 1, 7, 11 (15)
    MV_STAMP(w, 23);
    mv_meet_and_hand_over(w, c);                   // tile B
}

struct MvVecB {
    f32x4 dl;                      // d(loss)/d(logits) of this wave's block (sub), lane's sample: the pass being prepared
    const float* d_logits;
    float* dz;
    const char* masks;
    const f32x4* head_lds;         // LDS: W[channel][4]
    int64_t n;
};

__device__ __forceinline__ f32x4 mv_load_dlogits(const MvCtx& w, const MvVecB& v, int64_t block0) {
    int64_t block = block0 + w.sub;
    const bool real = block < w.num_blocks;
    block = real ? block : w.num_blocks - 1;
    const int64_t sample = block * 32 + w.s;
    // samples past n (the ragged tail of the last block) contribute zero
    return (real && sample < v.n) ? reinterpret_cast<const f32x4*>(v.d_logits)[sample] : (f32x4)(0.0f);
}

__device__ __forceinline__ unsigned mv_load_mask(const ffn_step& L, const MvCtx& w, const MvVecB& v, int64_t block0, int o) {
    if (L.mask_slot < 0) return 0xffffu;
    int64_t block = block0 + w.sub;
    block = block < w.num_blocks ? block : w.num_blocks - 1;
    return *reinterpret_cast<const uint16_t*>(v.masks + mv_mask_at(L.mask_slot, w.num_blocks, block, w.lane, o));
}

// the epilogue of (tile o, this wave's block): mask and the three-way split; the dZ slab goes to HBM behind
// the next barrier (mv_epilogue_bwd_stores: see MvDeferred)
__device__ __forceinline__ void mv_epilogue_bwd(bool last_step, unsigned word, const f32x16& acc, bf16x8 (&res)[2][3],
                                                MvDeferred& d) {
    if (!MV_VECTOR_WORK) return;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int bit = 15 - (8 * half + j);
            const int keep = ((int)(word << (31 - bit))) >> 31;
            // (through a scalar: __builtin_bit_cast applied to the vector ELEMENT acc[i] reads element 0
            // whatever i is -- hipcc 7.2; found by the parity check against the ws kernels)
            const float a = acc[8 * half + j];
            y[j] = __builtin_bit_cast(float, __builtin_bit_cast(int, a) & keep);
            d.y[8 * half + j] = y[j];
        }
        if (!last_step) split8x3(y, res[half][0], res[half][1], res[half][2]);
    }
}

__device__ __forceinline__ void mv_epilogue_bwd_stores(const ffn_mlp_chain& ch, const ffn_step& L, const MvCtx& w,
                                                       const MvVecB& v, int64_t block0, int o, const MvDeferred& d) {
    if (!MV_VECTOR_WORK) return;
    int save_s = w.s, save_h = w.h;
    asm volatile("" : "+v"(save_s), "+v"(save_h));
    const int64_t block = block0 + w.sub;
    if (L.out_slot < 0 || block >= w.num_blocks) return;
    f32x4* save_out = reinterpret_cast<f32x4*>(v.dz + ch.slot_offset[L.out_slot] * w.num_blocks * 32) +
                      block * (int64_t)(ch.slot_channels[L.out_slot] * 8);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 y0, y1;
#pragma unroll
        for (int p = 0; p < 4; ++p) { y0[p] = d.y[8 * half + p]; y1[p] = d.y[8 * half + 4 + p]; }
        const int cq = 2 * (4 * o + 2 * half) + save_h;
        __builtin_nontemporal_store(y0, &save_out[saved_index16(cq, save_s)]);
        __builtin_nontemporal_store(y1, &save_out[saved_index16(cq + 2, save_s)]);
    }
}

// Step 0 of (tile o, this wave's block) for the pass at block0: the head term in the accumulator layout
// (value 4 q + p of lane (h, s) = channel 32 o + 8 q + 4 h + p of sample s), masked, split, stored into X;
// its dZ slab is left to the caller (`d`: behind the next barrier).
__device__ __forceinline__ void mv_step0(const ffn_step& L0, const MvCtx& w, const MvVecB& v, const f32x4& dl, unsigned mask_word,
                                         int o, MvDeferred& d) {
    if (!MV_VECTOR_WORK) return;
    float dj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = L0.lg_col + j;
        const float x = c == 0 ? dl[0] : (c == 1 ? dl[1] : (c == 2 ? dl[2] : dl[3]));
        dj[j] = (j < L0.lg_n && c < 4) ? x : 0.0f;
    }
    const f32x4* hw = v.head_lds + 32 * o + 4 * w.h;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x4 w4 = hw[8 * q + p];
            acc[4 * q + p] = __builtin_fmaf(w4[3], dj[3], __builtin_fmaf(w4[2], dj[2], __builtin_fmaf(w4[1], dj[1], w4[0] * dj[0])));
        }
    bf16x8 res[2][3];
    mv_epilogue_bwd(false, mask_word, acc, res, d);
    mv_store_x(w, o, res);
}

// a hidden step seen from a vector wave, from its barrier S2 on.  next: the LAST step of a pass that has a
// successor computes tiles 0..3 of the next pass's step 0 (K blocks 0..7 of X: free from S3 on) between S3
// and S3b; their dZ slabs (`next_a`) and tile B's of this step (`late`) are left to the caller: behind the
// next barrier.  The last step's tile B is only TAKEN over here (`acc_b`, its mask word in `word_b_out`): what
// the matrix waves wait for next is the other half of the next pass's step 0, the caller computes that first.
__device__ __forceinline__ void mv_vector_tail_bwd(const ffn_mlp_chain& ch, const ffn_step& L, bool last_step, const MvCtx& w,
                                                   const MvVecB& v, bool next, const f32x4& dl_next, unsigned next_word_a,
                                                   MvDeferred& next_a, MvDeferred& late, f32x16& acc_b, unsigned& word_b_out) {
    f32x16 acc;
    bf16x8 res[2][3];
    MvDeferred early;
    const unsigned word_a = mv_load_mask(L, w, v, w.block0, w.m), word_b = mv_load_mask(L, w, v, w.block0, w.m + 4);
    mv_barrier();                                  // S2: tile A handed over
    mv_take_over(w, acc);
    mv_epilogue_bwd(last_step, word_a, acc, res, early);
    mv_barrier();                                  // S3: K blocks 0..7 of X are consumed
    if (!last_step) mv_store_x(w, w.m, res);
    else if (next) mv_step0(ch.step[0], w, v, dl_next, next_word_a, w.m, next_a);
    mv_barrier();                                  // S3b: K blocks 0..7 of the next image are in X
    mv_epilogue_bwd_stores(ch, L, w, v, w.block0, w.m, early);
    mv_barrier();                                  // S4: tile B handed over, X consumed
    if (last_step) {
        mv_take_over(w, acc_b);
        word_b_out = word_b;
        return;
    }
    mv_take_over(w, acc);
    mv_epilogue_bwd(false, word_b, acc, res, late);
    mv_store_x(w, w.m + 4, res);
}

// (step 0 has no units: its K blocks of the pack are skipped)
__device__ __forceinline__ int mv_build_units_bwd(const ffn_mlp_chain& ch, int* units) {
    int u = 0, flat = 0;
    for (int li = 0; li < ch.num_steps; ++li) {
        const int kb = (ch.step[li].act_groups >> 1) + (ch.step[li].aux_groups > 0 ? 2 : 0);
        if (li > 0)
            for (int t = 0; t < 2; ++t)
                for (int k = 0; k < kb; ++k) units[u++] = (flat + k) * 8 + 4 * t;
        flat += kb;
    }
    return u;
}

__global__ void __launch_bounds__(kMvThreads)
mlp_backward_bf16_mv_kernel(const ffn_mlp_chain ch, const uint16_t* __restrict__ packed,
                            const float* __restrict__ d_logits, int64_t n,
                            const uint32_t* __restrict__ masks, float* __restrict__ dz, int num_units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* tbl = reinterpret_cast<int*>(smem + kMvXBytes + kMvHandBytes);
    f32x4* head_lds = reinterpret_cast<f32x4*>(smem + kMvXBytes + kMvHandBytes + kMvMaxUnits * 4);
    if (threadIdx.x == 0) {
        int* units = reinterpret_cast<int*>(smem);             // (X is not in use yet)
        const int u_total = mv_build_units_bwd(ch, units);
        for (int i = 0; i < u_total; ++i) tbl[i] = units[(i + 3) % u_total] * (kMvTileVecs * 16);       // (byte offsets)
    }
    MvCtx w;
    w.lane = threadIdx.x & 63;
    w.h = w.lane >> 5;
    w.s = w.lane & 31;
    w.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    w.m = w.wave & 3;
    w.sub = w.wave >= 4 ? (w.wave - 4) >> 2 : 0;
    w.xbuf = reinterpret_cast<f32x4*>(smem);
    w.hand = reinterpret_cast<f32x4*>(smem + kMvXBytes);
    w.logit_lds = nullptr;
    w.enc_table = nullptr;
    w.bias_lds = nullptr;
    w.bias_glb = nullptr;
    w.tbl4 = reinterpret_cast<const i32x4*>(tbl);
    w.trips_total = num_units >> 2;
    w.gw = reinterpret_cast<const f32x4*>(packed + ch.step[0].w_off);
    w.gwm = (mv_gptr)(w.gw + w.m * kMvTileVecs);
    w.lane16 = w.lane * 16;
    w.b_step1 = w.b_stride = 0;
    w.saved = nullptr;
    w.masks = nullptr;
#ifdef MV_STAMPS
    w.stamp_on = 0;
    w.stamp_n = 0;
#endif
    w.num_blocks = (n + 31) / 32;
    const int64_t passes = (w.num_blocks + 1) / 2;
    const bool matrix = w.wave < 4;
    if (w.wave < 8 && w.h == 0) {
        // the head weights as f32: K rows 0..3 of step 0's K block 0, tile `wave` -- the A-operand layout keeps row
        // (= channel) r's K values 0..7 in lane r of the lower half; hi + mid + lo is the f32 value exactly
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 part[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) part[p] = __builtin_bit_cast(u32x4, w.gw[(w.wave * 3 + p) * 64 + w.lane]);
        f32x4 wt;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sum = 0.0f;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned word = part[p][j >> 1];
                sum += __builtin_bit_cast(float, (j & 1) ? (word & 0xffff0000u) : (word << 16));
            }
            wt[j] = sum;
        }
        head_lds[32 * w.wave + w.lane] = wt;
    }
    __syncthreads();                               // the unit table and the head weights are staged
    const int num_steps = ch.num_steps;

    if (matrix) {
        __builtin_amdgcn_s_setprio(3);
        MvMat r;
        MvAcc2 c;
        const i32x4 first = w.tbl4[w.trips_total - 1];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            mv_gptr base = reinterpret_cast<mv_gptr>(reinterpret_cast<mv_gbytes>(w.gwm) + (uint64_t)(unsigned)(first[j + 1] + w.lane16));
#pragma unroll
            for (int part = 0; part < 3; ++part) r.wr[j][part] = __builtin_bit_cast(bf16x8, base[part * 64]);
        }
        r.tq = 0;
        {
            const i32x4 entry = w.tbl4[0];
#pragma unroll
            for (int j = 0; j < 4; ++j) r.cur[j] = __builtin_amdgcn_readfirstlane(entry[j]);
        }
        mv_barrier();                                                                        // P1: the first pass's step-0 image
        asm volatile("" ::"v"(r.wr[2][2]));        // (see mv_matrix_features)
        mv_read_x0(w, r, 0);
        for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
#ifdef MV_STAMPS
            w.stamp_on = blockIdx.x == 0 && pass == blockIdx.x + 2 * (int64_t)gridDim.x && w.wave == 0;
            w.stamp_n = 0;
#endif
            MV_STAMP(w, 80);
            for (int li = 1; li < num_steps; ++li) mv_matrix_bwd_hidden(li == 1, li + 1 == num_steps, w, r, c);
            mv_barrier();                                                                    // S4 of the last step
            MV_STAMP(w, 24);
        }
        return;
    }

    // ---- a vector wave
    MvVecB v;
    v.d_logits = d_logits;
    v.dz = dz;
    v.masks = reinterpret_cast<const char*>(masks);
    v.head_lds = head_lds;
    v.n = n;
    const ffn_step& L0 = ch.step[0];
    w.block0 = (int64_t)blockIdx.x * 2;
    v.dl = mv_load_dlogits(w, v, w.block0);
    {   // the first pass's step 0, both halves (nothing streams yet: the slabs go out at once)
        MvDeferred da, db;
        mv_step0(L0, w, v, v.dl, mv_load_mask(L0, w, v, w.block0, w.m), w.m, da);
        mv_step0(L0, w, v, v.dl, mv_load_mask(L0, w, v, w.block0, w.m + 4), w.m + 4, db);
        mv_epilogue_bwd_stores(ch, L0, w, v, w.block0, w.m, da);
        mv_epilogue_bwd_stores(ch, L0, w, v, w.block0, w.m + 4, db);
    }
    mv_barrier();                                                                            // P1
    bool have_a = false;                           // tiles 0..3 of this pass's step 0 came out of the pass before: slabs pending
    bool have_last = false;                        // tile B of the pass before's last step: taken over, its epilogue pending
    int64_t last_block0 = 0;
    f32x16 last_acc;
    unsigned last_word = 0xffffu;
    MvDeferred step0_a, step0_b;
    unsigned word_b = 0xffffu;                     // step 0's mask word of tile m + 4, this pass (requested a pass ahead)
    const ffn_step& Llast = ch.step[num_steps - 1];
    for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
        const bool first_pass = pass == (int64_t)blockIdx.x;
        w.block0 = pass * 2;
        const bool has_next = pass + gridDim.x < passes;
        f32x4 dl_next = (f32x4)(0.0f);
        unsigned next_word_a = 0xffffu, next_word_b = 0xffffu;
        if (has_next) {
            const int64_t next_block0 = w.block0 + 2 * (int64_t)gridDim.x;
            dl_next = mv_load_dlogits(w, v, next_block0);
            next_word_a = mv_load_mask(L0, w, v, next_block0, w.m);
            next_word_b = mv_load_mask(L0, w, v, next_block0, w.m + 4);
        }
        // tiles 4..7 of this pass's step 0 (K blocks 8..15 of X: free since the barrier that ended the pass before),
        // under the first half of the matrix waves' first K loop -- the first thing behind that barrier
        if (!first_pass) mv_step0(L0, w, v, v.dl, word_b, w.m + 4, step0_b);
        MvDeferred late;
        for (int li = 1; li < num_steps; ++li) {
            const bool last = li + 1 == num_steps;
            mv_barrier();                                                                    // S1
            if (li == 1) {
                if (have_last) {                   // (tile B of the pass before's last step: no X image follows it)
                    bf16x8 none[2][3];
                    mv_epilogue_bwd(true, last_word, last_acc, none, late);
                    mv_epilogue_bwd_stores(ch, Llast, w, v, last_block0, w.m + 4, late);
                }
                if (!first_pass) {
                    if (have_a) mv_epilogue_bwd_stores(ch, L0, w, v, w.block0, w.m, step0_a);
                    mv_epilogue_bwd_stores(ch, L0, w, v, w.block0, w.m + 4, step0_b);
                }
            } else {
                mv_epilogue_bwd_stores(ch, ch.step[li - 1], w, v, w.block0, w.m + 4, late);      // (tile B of the step before)
            }
            mv_vector_tail_bwd(ch, ch.step[li], last, w, v, last && has_next, dl_next, next_word_a, step0_a, late, last_acc, last_word);
        }
        have_last = true;
        last_block0 = w.block0;
        have_a = has_next;
        v.dl = dl_next;
        word_b = next_word_b;
    }
    if (have_last) {
        MvDeferred late;
        bf16x8 none[2][3];
        mv_epilogue_bwd(true, last_word, last_acc, none, late);
        mv_epilogue_bwd_stores(ch, Llast, w, v, last_block0, w.m + 4, late);
    }
}

// backward chains this organisation covers: the d_logits term alone in step 0 (the vector waves'), then 256 -> 256 steps
bool mv_covers_bwd(const ffn_mlp_chain* chain, int* num_units) {
    if (chain->num_steps < 2 || chain->wide != 0) return false;
    int units = 0;
    for (int i = 0; i < chain->num_steps; ++i) {
        const ffn_step& L = chain->step[i];
        if (L.out_tiles != 8) return false;
        if (i == 0) {
            if (L.act_groups != 0 || L.aux_groups <= 0) return false;
            if (L.lg_col < 0 || L.lg_n < 1 || L.lg_col + L.lg_n > 4) return false;
        } else {
            if ((L.act_groups >> 1) != 16 || L.aux_groups != 0) return false;
            units += 32;
        }
    }
    if (units > kMvMaxUnits) return false;
    *num_units = units;
    return true;
}

int launch_backward_bf16x6_mv(const ffn_mlp_chain* chain, const uint16_t* packed_wt, const float* d_logits,
                              int64_t n, const uint32_t* masks, float* dz, int num_units, void* stream) {
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int val = 0;
        if (hipDeviceGetAttribute(&val, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && val > 0) cus = val;
    }
    const int64_t passes = ((n + 31) / 32 + 1) / 2;
    const int64_t grid = passes < cus ? passes : cus;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_backward_bf16_mv_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMvBwdLdsBytes);
    hipLaunchKernelGGL(mlp_backward_bf16_mv_kernel, dim3((unsigned)grid), dim3(kMvThreads), kMvBwdLdsBytes,
                       (hipStream_t)stream, *chain, packed_wt, d_logits, n, masks, dz, num_units);
    return 0;
}

}  // namespace ffn

#ifdef MV_STAMPS
extern "C" int ffn_debug_mv_stamps(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(ffn::mv_stamp_buf), sizeof(long long) * 2 * 1024);
}
#endif
