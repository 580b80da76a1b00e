// F32-ACCURATE split weight gradients (OPT-IN "bf16x6" training precision; wgrad.hip's exact-f32
// kernel stays the parity mode):  dW[j][k] = sum_s dZ[j][s] X[k][s]  with every f32 operand as THREE
// bf16 parts (hi, mid, lo: the f32 value exactly) and every f32 product as SIX
// v_mfma_f32_32x32x16_bf16 products, f32 accumulation -- 12 matrix cycles per K where
// v_mfma_f32_32x32x2_f32 takes 32.
//
// Units, segments, LDS-DMA staging (half-block stages, a ring of four) and the partial format are
// wgrad_bf16.hip's; what differs is the operand pipeline of a step.  A wave's 256 accumulator
// registers leave 256 for everything else, and two complete three-part operand sets (192) plus the
// raw values being converted (32) do not fit.  So only the A operand (the dZ window) is double
// buffered; the B operand (the input window) is converted JUST IN TIME, and PROGRESSIVELY: the
// three-way split produces hi first, then mid, then lo, and the step's products are ordered by the
// B part they need --
//      a_l b_h, a_m b_h, a_h b_h  |  a_m b_m, a_h b_m  |  a_h b_l
// -- so the matrix instructions start as soon as the hi parts exist (a packed convert per pair of
// values) and the mid / lo parts are computed under them.  One 32-register buffer carries the raw
// values: B of this step (until its lo parts are out), then A of the next step (converted into the
// other A set under the b_h / b_m products), then B of the next step (left raw for its own step).
#include <type_traits>

#include "wgrad_common.h"

namespace ffn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One stage of the three-way split of TWO components (2c, 2c + 1: one half of a float4) of eight
// float4 (eight samples): the bf16 part of what is left in x (round to nearest even; one packed
// convert per component and pair of samples -- its result is the operand layout), which is then
// subtracted -- x keeps the remainder for the next stage (the subtractions are exact in f32).  The
// two components of a sample sit in adjacent registers as they were read and every remainder is
// written back where its value was: no register moves (the per-component form, which paired up
// registers of different float4s, needed ~50 per step), 353 vector instructions per step where
// that form took 403.  LAST: no remainder is kept.
// (measured and NOT adopted, scripts/probes/wg6_variants/README.md: the subtraction as ONE
// v_dot2c_f32_bf16 per value -- x += (part_even, part_odd) . (-1, 0) -- is bit-identical, denormals
// included (scripts/probes/dot2_split_probe.hip), takes 255 instructions per step and is 4 % SLOWER:
// the dot instruction does not run beside the matrix pipe the way shift / mask / subtract do)
// (the raw values live as half float4s, pair-major: writing remainders back into the float4s they
// were read as makes hipcc keep several versions of the whole buffer alive)
struct Raw {
    f32x2v x[2][8];      // [component pair][sample]
};
template <bool LAST>
__device__ __forceinline__ void split_stage(f32x2v (&x)[8], bf16x8& part0, bf16x8& part1) {
    u32x4 w0, w1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x2v c0, c1;          // component 2c / 2c + 1 of samples 2t, 2t + 1
        c0[0] = x[2 * t][0];
        c0[1] = x[2 * t + 1][0];
        c1[0] = x[2 * t][1];
        c1[1] = x[2 * t + 1][1];
        const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(c0, bf16x2));
        const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(c1, bf16x2));
        if (!LAST) {
            f32x2v even, odd;   // the parts of sample 2t / 2t + 1 as f32
            even[0] = __builtin_bit_cast(float, h0 << 16);
            even[1] = __builtin_bit_cast(float, h1 << 16);
            odd[0] = __builtin_bit_cast(float, h0 & 0xffff0000u);
            odd[1] = __builtin_bit_cast(float, h1 & 0xffff0000u);
            x[2 * t] -= even;
            x[2 * t + 1] -= odd;
        }
        w0[t] = h0;
        w1[t] = h1;
    }
    part0 = __builtin_bit_cast(bf16x8, w0);
    part1 = __builtin_bit_cast(bf16x8, w1);
}
template <typename Parts>
__device__ __forceinline__ void split_all(f32x2v (&x)[8], Parts& o, int c) {
    split_stage<false>(x, o.h[2 * c], o.h[2 * c + 1]);
    split_stage<false>(x, o.m[2 * c], o.m[2 * c + 1]);
    split_stage<true>(x, o.l[2 * c], o.l[2 * c + 1]);
}

// LDS map: four stages [A half | B half] of 16 KiB each, then a 256-B row of zeros that idle lanes
// of a narrow window read instead of branching.
constexpr int kHalfBytes = 16 * 1024;
constexpr int kStageBytes = 2 * kHalfBytes;
constexpr int kStages = 4;
constexpr int kZeroRowAt = kStages * kStageBytes;
constexpr int kDmaLdsBytes = kZeroRowAt + 256;
static_assert(kDmaLdsBytes <= kUnitLdsBytes, "the head unit's images and the DMA ring share one allocation");

struct PartsA { bf16x8 h[4], m[4], l[4]; };      // [component = output row-set]
struct PartsB { bf16x8 h[4], m[4], l[4]; };      // [component = output column-set]

// s_waitcnt vmcnt(n) for a wave-uniform n in 0..16 (the count is an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define FFN_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        FFN_CASE(1) FFN_CASE(2) FFN_CASE(3) FFN_CASE(4) FFN_CASE(5) FFN_CASE(6) FFN_CASE(7) FFN_CASE(8)
        FFN_CASE(9) FFN_CASE(10) FFN_CASE(11) FFN_CASE(12) FFN_CASE(13) FFN_CASE(14) FFN_CASE(15)
        FFN_CASE(16)
#undef FFN_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// CA / CB: 4-KiB chunks of the full-block image (8 = a window wider than 128 channels)
template <int CA, int CB, bool BIAS>
__device__ __forceinline__ void unit_segment24(const ffn_mlp_chain& ch, const ffn_wgrad_unit& unit,
                                               const ffn_wgrad_segment& seg, char* smem,
                                               const float* __restrict__ saved,
                                               const float* __restrict__ dz, int64_t num_blocks,
                                               float* __restrict__ partials) {
    constexpr int NQ = (CA / 4) * (CB / 4);   // quadrants that exist: 4, 2 or 1
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int hh = lane >> 5;
    const int li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qd = wave & (NQ - 1);
    // waves that share a quadrant split the contraction steps (a 128x128 unit keeps two of its
    // four waves idle: they contribute zero partials)
    const int part = NQ == 4 ? 0 : (NQ == 2 ? wave >> 1 : wave);
    const int mp = CB == 8 ? qd >> 1 : qd, np = CB == 8 ? (qd & 1) : 0;
    const bool a_ok = li < unit.m_quads - 32 * mp;   // this lane's quad exists in the M window
    const bool b_ok = li < unit.n_quads - 32 * np;
    const int64_t a_stride = (int64_t)ch.slot_channels[unit.m_slot] * 128;   // bytes per block
    const int64_t b_stride = (int64_t)ch.slot_channels[unit.n_slot] * 128;
    const char* a_base = reinterpret_cast<const char*>(dz + ch.slot_offset[unit.m_slot] * num_blocks * 32) +
                         unit.m_cq0 * 512;
    const char* b_base = reinterpret_cast<const char*>(saved + ch.slot_offset[unit.n_slot] * num_blocks * 32) +
                         unit.n_cq0 * 512;
    const int64_t steps = 2 * (seg.blk_end - seg.blk_begin);      // contraction steps = stages

    f32x16 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.0f;
    f32x2v bsum[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};

    // ---- LDS-DMA (wgrad_bf16.hip): a stage = up to 32 pieces of 1 KiB; wave w issues pieces
    // 4w .. 4w + 3 of both halves
    const int a_pieces = unit.m_quads >> 2, b_pieces = unit.n_quads >> 2;    // <= 16 each
    auto clamp4 = [](int x) { return x < 0 ? 0 : (x > 4 ? 4 : x); };
    const int per_stage = clamp4(a_pieces - 4 * wave) + clamp4(b_pieces - 4 * wave);
    const int dma_lane = ((lane >> 4) * 512) + ((lane & 15) * 16);
    auto issue_stage = [&](int64_t st, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;        // both windows 256 channels: no tests
        const int64_t blk = seg.blk_begin + (st >> 1);
        const int half = (int)(st & 1) * 256;
        const char* ga = a_base + blk * a_stride + half + dma_lane;
        const char* gb = b_base + blk * b_stride + half + dma_lane;
        char* l = smem + (int)(st & (kStages - 1)) * kStageBytes;
        // (wave w requests the four CONSECUTIVE pieces 4w .. 4w + 3 of each half: their LDS
        // addresses differ by the instruction's immediate offset -- which the hardware adds to the
        // global address as well, hence the k * 1024 taken off it -- so that M0 is written once per
        // half, not once per request)
#define FFN_DMA(G, PIECES, LDS, K)                                                                 \
    if (FULL || 4 * wave + K < PIECES)                                                             \
        __builtin_amdgcn_global_load_lds(                                                          \
            (const __attribute__((address_space(1))) void*)(G + (4 * wave + K) * 2048 - K * 1024), \
            (__attribute__((address_space(3))) void*)(LDS + 4 * wave * 1024), 16, K * 1024, 2);
        FFN_DMA(ga, a_pieces, l, 0) FFN_DMA(ga, a_pieces, l, 1) FFN_DMA(ga, a_pieces, l, 2) FFN_DMA(ga, a_pieces, l, 3)
        FFN_DMA(gb, b_pieces, l + kHalfBytes, 0) FFN_DMA(gb, b_pieces, l + kHalfBytes, 1)
        FFN_DMA(gb, b_pieces, l + kHalfBytes, 2) FFN_DMA(gb, b_pieces, l + kHalfBytes, 3)
#undef FFN_DMA
    };

    // This lane's float4 of sample 8 hh + t of a stage sits at byte ((8 hh + t) ^ (li & 15)) * 16 of
    // its quad's 256-B row.  Idle lanes of a narrow window point into the zero row.
    const unsigned lane_x = (unsigned)(((8 * hh) ^ (li & 15)) << 4);
    const int a_row = a_ok ? (32 * mp + li) * 256 : -1;
    const int b_row = b_ok ? (32 * np + li) * 256 : -1;
    auto read_half = [&](int64_t st, int row, int half_off, Raw& raw) {
        const char* base = row >= 0 ? smem + (int)(st & (kStages - 1)) * kStageBytes + half_off + row
                                    : smem + kZeroRowAt;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(base + (lane_x ^ (unsigned)(t << 4)));
            raw.x[0][t][0] = q[0];
            raw.x[0][t][1] = q[1];
            raw.x[1][t][0] = q[2];
            raw.x[1][t][1] = q[3];
        }
    };
    auto add_bias = [&](const Raw& raw) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 8; ++t) bsum[c] += raw.x[c][t];
    };
    auto mine = [&](int64_t st) -> bool {       // (wave-uniform) does this wave contract step st?
        return NQ == 4 || (NQ == 2 ? (int)(st & 1) == part : (part < 2 && (int)(st & 1) == part));
    };
    auto split_a = [&](Raw& raw, PartsA& a) {
        if (BIAS) add_bias(raw);
#pragma unroll
        for (int c = 0; c < 2; ++c) split_all(raw.x[c], a, c);
    };
    auto split_b = [&](Raw& raw, PartsB& b) {
#pragma unroll
        for (int c = 0; c < 2; ++c) split_all(raw.x[c], b, c);
    };
    // the six partial products of a step, ordered by the B part they need (hi first)
#define FFN_PRODUCT(AP, BP)                                                                        \
    _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                              \
            acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.AP[p], b.BP[q], acc[p][q], 0, 0, 0);
    auto contract = [&](const PartsA& a, const PartsB& b) {
        FFN_PRODUCT(l, h) FFN_PRODUCT(m, h) FFN_PRODUCT(h, h)
        FFN_PRODUCT(m, m) FFN_PRODUCT(h, m)
        FFN_PRODUCT(h, l)
    };
    // every wave waits for ITS pieces of stage st (the stages issued after it stay in flight),
    // then the workgroup meets: stage st is complete in LDS and stage st - 1 has been read by all
    auto stage_ready = [&](int64_t st) {
        const int64_t younger = steps - 1 - st < kStages - 2 ? steps - 1 - st : kStages - 2;
        wait_vmcnt((int)younger * per_stage);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: stages 0..2 in flight; A of step 0 converted, B of step 0 raw in `v`
    for (int s = 0; s < kStages - 1 && s < steps; ++s) issue_stage(s, std::false_type{});
    PartsA a0, a1;
    PartsB b;
    Raw v;
    stage_ready(0);
    if (kStages - 1 < steps) issue_stage(kStages - 1, std::false_type{});

    // ---- full units: every wave contracts every step.  Step i (operands: A converted in `cur`,
    // B raw in `v`): stage i+1 becomes ready (every wave has also finished reading stage i) and
    // stage i+4 is requested; the hi parts of B are split off; then the 96 matrix instructions run
    // with, pinned behind them in program order: mid and lo parts of B, the raw A of step i+1 from
    // LDS and its conversion into `nxt`, the raw B of step i+1 from LDS (left in `v`).
    auto step_pipelined = [&](auto bulk_tag, int64_t i, PartsA& a, PartsA& nxt, bool last) {
        constexpr bool BULK = decltype(bulk_tag)::value;
        if (BULK) {
            asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if (!last) {
            stage_ready(i + 1);
        }
        // (the phases are fenced for the scheduler -- left alone it starts every LDS read and every
        // conversion as early as their inputs allow and keeps three raw buffers alive -- and inside
        // a phase the program order already alternates four matrix instructions (one output
        // row-set) with a slice of the vector work; the pin refines that to one matrix
        // instruction, then up to NV vector instructions)
#define FFN_FENCE() __builtin_amdgcn_sched_barrier(0)
#define FFN_ROWSET(AP, BP, P)                                                                      \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
        acc[P][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.AP[P], b.BP[q], acc[P][q], 0, 0, 0);
 // one LDS read behind each of eight matrix instructions (a burst of eight 1-KiB reads per wave,
        // four waves at once, holds the LDS for 256+ cycles)
#define FFN_PIN_READS()                                                                            \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }
#define FFN_PIN_READS2()      /* two behind each of four: the values are needed eight instructions on */  \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                         \
    }
#define FFN_PIN_DMA()         /* one request behind each of the step's first eight */              \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                         \
    }                                                                                              \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                         \
    }
#define FFN_PIN16(NV)                                                                              \
    _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                        \
    }
#define FFN_PIN12(NV)                                                                              \
    _Pragma("unroll") for (int k = 0; k < 12; ++k) {                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                        \
    }
#define FFN_PIN8(NV)                                                                              \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                        \
    }
        // (the hi parts of this step's B were split off under the last products of the step before:
        // the matrix instructions start right behind the barrier, the request for stage i + 4 --
        // its slot is the one every wave has finished reading -- goes out under the first four)
        FFN_FENCE();
        // a_l b_h, with the requests and the mid parts of B behind it
        FFN_ROWSET(l, h, 0)
        if (BULK) {
            issue_stage(i + kStages, std::true_type{});
        } else if (!last) {
            if (i + kStages < steps) issue_stage(i + kStages, std::false_type{});
        }
        FFN_ROWSET(l, h, 1) split_stage<false>(v.x[0], b.m[0], b.m[1]);
        FFN_ROWSET(l, h, 2)
        FFN_ROWSET(l, h, 3) split_stage<false>(v.x[1], b.m[2], b.m[3]);
        FFN_PIN_DMA()
        FFN_FENCE();
        // a_m b_h, with the lo parts of B behind it
        FFN_ROWSET(m, h, 0) split_stage<true>(v.x[0], b.l[0], b.l[1]);
        FFN_ROWSET(m, h, 1)
        FFN_ROWSET(m, h, 2) split_stage<true>(v.x[1], b.l[2], b.l[3]);
        FFN_ROWSET(m, h, 3)
        FFN_PIN16(2)
        FFN_FENCE();
        // a_h b_h, over the LDS round trip of the next step's raw A (v is free)
        if (!last) read_half(i + 1, a_row, 0, v);
        FFN_ROWSET(h, h, 0) FFN_ROWSET(h, h, 1) FFN_ROWSET(h, h, 2) FFN_ROWSET(h, h, 3)
        FFN_PIN_READS()
        FFN_FENCE();
        // a_m b_m and a_h b_m, with the conversion of the next step's A behind them
        FFN_ROWSET(m, m, 0)
        if (!last && BIAS) add_bias(v);       // (before the splits turn the raw values into remainders)
        FFN_ROWSET(m, m, 1)
        if (!last) split_all(v.x[0], nxt, 0);
        FFN_ROWSET(m, m, 2)
        FFN_ROWSET(m, m, 3)
        FFN_PIN16(6)
        FFN_FENCE();
        FFN_ROWSET(h, m, 0)
        if (!last) split_all(v.x[1], nxt, 1);
        FFN_ROWSET(h, m, 1)
        FFN_ROWSET(h, m, 2)
        FFN_ROWSET(h, m, 3)
        FFN_PIN16(5)
        FFN_FENCE();
        // a_h b_l, over the LDS round trip of the next step's raw B (stays in v)
        // ... and, once they have landed, the hi parts of that B (b.h is free since a_h b_h): the
        // next step starts with matrix instructions
        if (!last) read_half(i + 1, b_row, kHalfBytes, v);
        FFN_ROWSET(h, l, 0) FFN_ROWSET(h, l, 1)
        FFN_PIN_READS2()
        FFN_FENCE();
        FFN_ROWSET(h, l, 2)
        if (!last) split_stage<false>(v.x[0], b.h[0], b.h[1]);
        FFN_ROWSET(h, l, 3)
        if (!last) split_stage<false>(v.x[1], b.h[2], b.h[3]);
        FFN_PIN8(8)
        FFN_FENCE();
#undef FFN_FENCE
#undef FFN_PIN_READS
#undef FFN_PIN_READS2
#undef FFN_PIN_DMA
#undef FFN_PIN12
#undef FFN_PIN8
#undef FFN_ROWSET
#undef FFN_PIN16
    };
    // narrow units (fewer than four quadrants: the waves of a quadrant take turns; conversions not
    // overlapped): stage i is read, converted and multiplied out FIRST -- the barrier of
    // stage_ready(i + 1) is what tells the other waves that its slot may be overwritten by the DMA
    // of stage i + 4, which is requested behind it
    auto step = [&](int64_t i, PartsA& a) {
        if (mine(i)) {
            read_half(i, a_row, 0, v);
            split_a(v, a);
            read_half(i, b_row, kHalfBytes, v);
            split_b(v, b);
            contract(a, b);
        }
        if (i + 1 < steps) {
            stage_ready(i + 1);
            if (i + kStages < steps) issue_stage(i + kStages, std::false_type{});
        }
    };
    if (NQ == 4) {
        read_half(0, a_row, 0, v);
        split_a(v, a0);
        read_half(0, b_row, kHalfBytes, v);
        split_stage<false>(v.x[0], b.h[0], b.h[1]);
        split_stage<false>(v.x[1], b.h[2], b.h[3]);
        int64_t i = 0;
        if (a_pieces == 16 && b_pieces == 16) {
            for (; i + kStages + 1 < steps; i += 2) {    // stages i+4 and i+5 exist
                step_pipelined(std::true_type{}, i, a0, a1, false);
                step_pipelined(std::true_type{}, i + 1, a1, a0, false);
            }
        }
        for (; i + 2 < steps; i += 2) {
            step_pipelined(std::false_type{}, i, a0, a1, false);
            step_pipelined(std::false_type{}, i + 1, a1, a0, false);
        }
        step_pipelined(std::false_type{}, steps - 2, a0, a1, false);
        step_pipelined(std::false_type{}, steps - 1, a1, a0, true);       // nothing to prefetch
    } else {
        for (int64_t i = 0; i < steps; ++i) step(i, a0);
    }
#undef FFN_PRODUCT
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // the ring is idle: the next segment may reuse it

    {
        float* out = partials + (int64_t)(seg.slot + wave) * kPartialFloats;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[((p * 4 + q) * 16 + r) * 64 + lane] = acc[p][q][r];
        reinterpret_cast<f32x4*>(out + 16 * 16 * 64)[lane] = f32x4{bsum[0][0], bsum[0][1], bsum[1][0], bsum[1][1]};
    }
}

__global__ void __launch_bounds__(256, 1)
wgrad_unit_bf16x6_kernel(const ffn_mlp_chain ch, const ffn_wgrad_unit* __restrict__ units,
                         const ffn_wgrad_segment* __restrict__ segments,
                         const int32_t* __restrict__ seg_start, const float* __restrict__ saved,
                         const float* __restrict__ dz, const float* __restrict__ d_logits, int64_t n,
                         float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t num_blocks = (n + 31) / 32;
    const int seg_lo = seg_start[blockIdx.x], seg_hi = seg_start[blockIdx.x + 1];
    for (int si = seg_lo; si < seg_hi; ++si) {
        ffn_wgrad_segment seg = segments[si];
        if (seg.blk_end > num_blocks) seg.blk_end = num_blocks;
        if (seg.blk_end <= seg.blk_begin) {
            float* out = partials + (int64_t)(seg.slot + (threadIdx.x >> 6)) * kPartialFloats;
            for (int e = threadIdx.x & 63; e < kPartialFloats; e += 64) out[e] = 0.0f;
            continue;
        }
        const ffn_wgrad_unit unit = units[seg.job];
        // the zero rows: the one behind each image (head unit) / the DMA ring's (units)
        if (unit.kind == 1) {
            for (int k = threadIdx.x; k < 4 * 128; k += 256)
                reinterpret_cast<float*>(smem + (k >> 7) * kImageStride + kImageBytes)[k & 127] = 0.0f;
        } else if (threadIdx.x < 64) {
            reinterpret_cast<float*>(smem + kZeroRowAt)[threadIdx.x] = 0.0f;
        }
        __syncthreads();
        if (unit.kind == 1) {
            // (the logits-head unit: exact f32, register-staged, wgrad_common.h)
            head_segment(ch, unit, seg, smem, saved, d_logits, n, num_blocks, partials);
        } else {
            const bool m_wide = unit.m_quads > 32, n_wide = unit.n_quads > 32;
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            const bool bias = unit.want_bias != 0 && (!n_wide || (wave & 1) == 0);
#define FFN_UNIT(CA, CB)                                                                         \
    do {                                                                                         \
        if (bias) unit_segment24<CA, CB, true>(ch, unit, seg, smem, saved, dz, num_blocks, partials);   \
        else unit_segment24<CA, CB, false>(ch, unit, seg, smem, saved, dz, num_blocks, partials);       \
    } while (0)
            if (m_wide && n_wide) FFN_UNIT(8, 8);
            else if (m_wide) FFN_UNIT(8, 4);
            else if (n_wide) FFN_UNIT(4, 8);
            else FFN_UNIT(4, 4);
#undef FFN_UNIT
        }
        __syncthreads();
    }
}

}  // namespace ffn

using namespace ffn;

extern "C" int ffn_mlp_wgrad_units_bf16x6(const ffn_mlp_chain* chain, const ffn_wgrad_unit* units,
                                          const ffn_wgrad_segment* segments, const int32_t* seg_start,
                                          int num_groups, const float* saved, const float* dz,
                                          const float* d_logits, int64_t n, float* partials,
                                          void* stream) {
    if (n <= 0 || num_groups <= 0) return fail_arg("ffn_mlp_wgrad_units_bf16x6: shape");
    const size_t lds = kUnitLdsBytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_unit_bf16x6_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wgrad_unit_bf16x6_kernel, dim3(num_groups), dim3(256), lds, (hipStream_t)stream,
                       *chain, units, segments, seg_start, saved, dz, d_logits, n, partials);
    return check_launch("ffn_mlp_wgrad_units_bf16x6");
}
