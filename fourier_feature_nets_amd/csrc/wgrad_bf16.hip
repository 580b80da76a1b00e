// Split-bf16 weight gradients (OPT-IN "bf16x3" training precision; wgrad.hip's exact-f32 kernel
// stays the parity mode):  dW[j][k] = sum_s dZ[j][s] X[k][s]  with every f32 product as three
// v_mfma_f32_32x32x16_bf16 products (hi*lo + lo*hi + hi*hi, f32 accumulation).
//
// Same units, segments and partial format as the f32 kernel (the reduce kernel is shared); a wave
// owns a 128x128 quadrant of dW in 256 accumulator registers.  The bf16 instruction contracts
// SIXTEEN samples, eight consecutive ones per lane: lane (g, i) reads the float4 of channel quad i
// at samples 16 ks + 8 g + 0..7, splits its 32 values into (hi, lo) bf16 -- component p of the
// eight float4 is the operand of output tile row-set p -- and issues 48 matrix instructions per
// sixteen samples (3.1k matrix cycles per 32-sample block).
//
// Round 3: the kernel used to be bound by instruction issue, not by HBM (its staging
// pattern alone streams at 7.2 TB/s; one in-order wave per SIMD issued ~450 conversion
// instructions, 96 matrix instructions, 32 operand reads, 16 staging loads and 16 deposits per
// block one after the other).  Now
//   * slabs go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, non-temporal): no staging
//     registers, no ds_write pass.  A stage is HALF a block (the sixteen samples of one
//     contraction step: 16 KiB of the dZ window + 16 KiB of the input window, rows of 256 B
//     gathered from the slab's 512-B rows); four stages ring through LDS, three in flight
//     (96 KiB per CU) while one is consumed; one workgroup barrier per stage;
//   * the 64 registers that staged the slabs hold a SECOND operand set: while the matrix
//     instructions of step i run on one set, the operands of step i+1 are read from LDS and
//     converted into the other.
#include <type_traits>

#include "wgrad_common.h"

namespace ffn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// component P of eight float4 -> (hi, lo) bf16 operand, three vector instructions per value: per
// pair one v_cvt_pk_bf16_f32 (both hi parts, round to nearest), a shift / a mask back to f32, two
// subtractions, one v_cvt_pk_bf16_f32 (both lo parts) -- written on the packed words so that
// hipcc does not assemble the eight-wide operands element by element (73 moves per two steps).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int P>
__device__ __forceinline__ void split_component(const f32x4 (&v)[8], bf16x8& hi, bf16x8& lo) {
    u32x4 hw, lw;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x2v x;
        x[0] = v[2 * t][P];
        x[1] = v[2 * t + 1][P];
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        f32x2v r;
        r[0] = x[0] - __builtin_bit_cast(float, h << 16);
        r[1] = x[1] - __builtin_bit_cast(float, h & 0xffff0000u);
        hw[t] = h;
        lw[t] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    }
    hi = __builtin_bit_cast(bf16x8, hw);
    lo = __builtin_bit_cast(bf16x8, lw);
}

// LDS map: four stages [A half | B half] of 16 KiB each, then a 256-B row of zeros that idle lanes
// of a narrow window read instead of branching.
constexpr int kHalfBytes = 16 * 1024;
constexpr int kStageBytes = 2 * kHalfBytes;
constexpr int kStages = 4;
constexpr int kZeroRowAt = kStages * kStageBytes;
constexpr int kDmaLdsBytes = kZeroRowAt + 256;
static_assert(kDmaLdsBytes <= kUnitLdsBytes, "the head unit's images and the DMA ring share one allocation");

struct Operands {
    bf16x8 ah[4], al[4], bh[4], bl[4];
};

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
__device__ __forceinline__ void unit_segment16(const ffn_mlp_chain& ch, const ffn_wgrad_unit& unit,
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
    f32x4 bsum = zero4();

    // ---- LDS-DMA: a stage = up to 32 pieces of 1 KiB (16 for the A half, 16 for the B half),
    // piece p of a half = rows 4p..4p+3 (quads) x 256 B; lane l of the issuing wave fetches the
    // 16 B at row 4p + (l >> 4), column l & 15.  Wave w issues the four CONSECUTIVE pieces 4w .. 4w + 3
    // of both halves -- their LDS addresses differ by the request's immediate offset, which the
    // hardware adds to the global address as well (hence the K * 1024 taken off it): M0 is written
    // once per half, not once per request; pieces past the end of a window are not issued
    // (`per_stage` = what this wave issues).
    const int a_pieces = unit.m_quads >> 2, b_pieces = unit.n_quads >> 2;    // <= 16 each
    auto clamp4 = [](int x) { return x < 0 ? 0 : (x > 4 ? 4 : x); };
    const int per_stage = clamp4(a_pieces - 4 * wave) + clamp4(b_pieces - 4 * wave);
    const int dma_lane = ((lane >> 4) * 512) + ((lane & 15) * 16);
    auto issue_stage = [&](int64_t st, auto full_tag, int which = 2) {      // which: 0 = the A half, 1 = the B half, 2 = both
        constexpr bool FULL = decltype(full_tag)::value;        // both windows 256 channels: no tests
        const int64_t blk = seg.blk_begin + (st >> 1);
        const int half = (int)(st & 1) * 256;
        const char* ga = a_base + blk * a_stride + half + dma_lane;
        const char* gb = b_base + blk * b_stride + half + dma_lane;
        char* l = smem + (int)(st & (kStages - 1)) * kStageBytes;
#define FFN_DMA(G, PIECES, LDS, K)                                                                 \
    if (FULL || 4 * wave + K < PIECES)                                                             \
        __builtin_amdgcn_global_load_lds(                                                          \
            (const __attribute__((address_space(1))) void*)(G + (4 * wave + K) * 2048 - K * 1024), \
            (__attribute__((address_space(3))) void*)(LDS + 4 * wave * 1024), 16, K * 1024, 2);
        if (which != 1) {
            FFN_DMA(ga, a_pieces, l, 0) FFN_DMA(ga, a_pieces, l, 1) FFN_DMA(ga, a_pieces, l, 2) FFN_DMA(ga, a_pieces, l, 3)
        }
        if (which != 0) {
            FFN_DMA(gb, b_pieces, l + kHalfBytes, 0) FFN_DMA(gb, b_pieces, l + kHalfBytes, 1)
            FFN_DMA(gb, b_pieces, l + kHalfBytes, 2) FFN_DMA(gb, b_pieces, l + kHalfBytes, 3)
        }
#undef FFN_DMA
    };

    // This lane's float4 of sample 8 hh + t of a stage sits at byte ((8 hh + t) ^ (li & 15)) * 16 of
    // its quad's 256-B row.  Idle lanes of a narrow window point into the zero row.
    const unsigned lane_x = (unsigned)(((8 * hh) ^ (li & 15)) << 4);
    const int a_row = a_ok ? (32 * mp + li) * 256 : -1;
    const int b_row = b_ok ? (32 * np + li) * 256 : -1;
    auto read_half = [&](int64_t st, int row, int half_off, f32x4 (&v)[8], int t0 = 0, int t1 = 8) {
        const char* base = row >= 0 ? smem + (int)(st & (kStages - 1)) * kStageBytes + half_off + row
                                    : smem + kZeroRowAt;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t >= t0 && t < t1) v[t] = *reinterpret_cast<const f32x4*>(base + (lane_x ^ (unsigned)(t << 4)));
    };
    auto mine = [&](int64_t st) -> bool {       // (wave-uniform) does this wave contract step st?
        return NQ == 4 || (NQ == 2 ? (int)(st & 1) == part : (part < 2 && (int)(st & 1) == part));
    };
    auto load_operands = [&](int64_t st, Operands& o) {
        f32x4 v[8];
        read_half(st, a_row, 0, v);
        if (BIAS) {
#pragma unroll
            for (int t = 0; t < 8; ++t) bsum += v[t];
        }
        split_component<0>(v, o.ah[0], o.al[0]); split_component<1>(v, o.ah[1], o.al[1]);
        split_component<2>(v, o.ah[2], o.al[2]); split_component<3>(v, o.ah[3], o.al[3]);
        read_half(st, b_row, kHalfBytes, v);
        split_component<0>(v, o.bh[0], o.bl[0]); split_component<1>(v, o.bh[1], o.bl[1]);
        split_component<2>(v, o.bh[2], o.bl[2]); split_component<3>(v, o.bh[3], o.bl[3]);
    };
    auto contract = [&](const Operands& o) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.ah[p], o.bl[q], acc[p][q], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.al[p], o.bh[q], acc[p][q], 0, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.ah[p], o.bh[q], acc[p][q], 0, 0, 0);
    };
    // every wave waits for ITS pieces of stage st (the stages issued after it stay in flight),
    // then the workgroup meets: stage st is complete in LDS and stage st - 1 has been read by all
    auto stage_ready = [&](int64_t st) {
        const int64_t younger = steps - 1 - st < kStages - 2 ? steps - 1 - st : kStages - 2;
        wait_vmcnt((int)younger * per_stage);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: stages 0..2 in flight, operands of step 0 in set 0
    for (int s = 0; s < kStages - 1 && s < steps; ++s) issue_stage(s, std::false_type{});
    Operands set0, set1;
    stage_ready(0);
    if (kStages - 1 < steps) issue_stage(kStages - 1, std::false_type{});
    if (mine(0)) load_operands(0, set0);

    // ---- steady state, two steps per trip (the operand sets alternate statically).  Step i:
    // stage i+1 becomes ready (wait + barrier: every wave has also finished READING stage i, whose
    // slot the next DMA may overwrite), stage i+4 is requested, then the matrix instructions of
    // step i run while the operands of step i+1 are read and converted.
    //
    // Full units (all four quadrants exist: every wave contracts every step) get the interleaving
    // spelled out and pinned: an in-order wave overlaps nothing by itself, and left to hipcc the
    // conversions sit in front of the matrix instructions.  Per pass of 16 matrix instructions
    // (4 per output row-set p), the ~30 conversion instructions of component p of the next
    // step's operand follow the 4 instructions of row-set p: one matrix instruction, then seven
    // vector instructions, sixteen times (sched_group_barrier masks: 0x008 MFMA, 0x002 VALU,
    // 0x100 DS read).
    auto step_pipelined = [&](auto bulk_tag, int64_t i, Operands& cur, Operands& nxt) {
        // BULK: both windows are 256 channels wide (8 pieces per wave and stage) and stage i+4
        // exists -- no branch, no jump table in the steady state
        constexpr bool BULK = decltype(bulk_tag)::value;
        if (BULK) {
            asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            stage_ready(i + 1);
        }
        // (a step STARTS with matrix instructions behind its barrier -- both operand sets of this
        // step are ready --: the requests for stage i + 4 and the LDS reads of the next step's raw A
        // go out under the first four (four reads behind each of the first two, four requests behind each of
        // the others); they used to sit in
        // front of the step: ~35 + 8 instructions with the matrix pipe idle)
#define FFN_FENCE() __builtin_amdgcn_sched_barrier(0)
#define FFN_ROWSET(X, Y, P)                                                                    \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                              \
        acc[P][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.X[P], cur.Y[q], acc[P][q], 0, 0, 0);
#define FFN_SPLIT_A(P) split_component<P>(v, nxt.ah[P], nxt.al[P])
#define FFN_SPLIT_B(P) split_component<P>(v, nxt.bh[P], nxt.bl[P])
#define FFN_PIN(N, NV)                                                                         \
    _Pragma("unroll") for (int k = 0; k < N; ++k) {                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                    \
    }
        f32x4 v[8];
        auto request = [&](int which) {
            if (BULK) {
                issue_stage(i + kStages, std::true_type{}, which);
            } else {
                if (i + kStages < steps) issue_stage(i + kStages, std::false_type{}, which);
            }
        };
#define FFN_ONE(X, Y, P, Q) acc[P][Q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.X[P], cur.Y[Q], acc[P][Q], 0, 0, 0);
        // (every group of this row-set fenced: left to the pins hipcc bursts the eight reads)
        FFN_FENCE();
        FFN_ONE(ah, bl, 0, 0) read_half(i + 1, a_row, 0, v, 0, 4);
        FFN_FENCE();
        FFN_ONE(ah, bl, 0, 1) read_half(i + 1, a_row, 0, v, 4, 8);
        FFN_FENCE();
        FFN_ONE(ah, bl, 0, 2) request(0);
        FFN_FENCE();
        FFN_ONE(ah, bl, 0, 3) request(1);
        FFN_FENCE();
        if (BIAS) {
#pragma unroll
            for (int t = 0; t < 8; ++t) bsum += v[t];
        }
        FFN_ROWSET(ah, bl, 1) FFN_SPLIT_A(0); FFN_SPLIT_A(1);
        FFN_ROWSET(ah, bl, 2) FFN_SPLIT_A(2);
        FFN_ROWSET(ah, bl, 3) FFN_SPLIT_A(3);
        FFN_PIN(12, BIAS ? 9 : 8)
        FFN_FENCE();
        FFN_ONE(al, bh, 0, 0) read_half(i + 1, b_row, kHalfBytes, v, 0, 4);
        FFN_FENCE();
        FFN_ONE(al, bh, 0, 1) read_half(i + 1, b_row, kHalfBytes, v, 4, 8);
        FFN_FENCE();
        FFN_ONE(al, bh, 0, 2) FFN_ONE(al, bh, 0, 3)
        FFN_FENCE();
        FFN_ROWSET(al, bh, 1) FFN_SPLIT_B(0); FFN_SPLIT_B(1);
        FFN_ROWSET(al, bh, 2) FFN_SPLIT_B(2);
        FFN_ROWSET(al, bh, 3) FFN_SPLIT_B(3);
        FFN_PIN(12, 8)
        FFN_FENCE();
        FFN_ROWSET(ah, bh, 0) FFN_ROWSET(ah, bh, 1) FFN_ROWSET(ah, bh, 2) FFN_ROWSET(ah, bh, 3)
        FFN_FENCE();
#undef FFN_FENCE
#undef FFN_ROWSET
#undef FFN_SPLIT_A
#undef FFN_SPLIT_B
#undef FFN_PIN
#undef FFN_ONE
    };
    auto step = [&](int64_t i, Operands& cur, Operands& nxt) {
        if (i + 1 < steps) {
            stage_ready(i + 1);
            if (i + kStages < steps) issue_stage(i + kStages, std::false_type{});
            if (mine(i + 1)) load_operands(i + 1, nxt);
        }
        if (mine(i)) contract(cur);
    };
    if (NQ == 4) {
        int64_t i = 0;
        if (a_pieces == 16 && b_pieces == 16) {
            for (; i + kStages + 1 < steps; i += 2) {    // stages i+4 and i+5 exist
                step_pipelined(std::true_type{}, i, set0, set1);
                step_pipelined(std::true_type{}, i + 1, set1, set0);
            }
        }
        for (; i + 2 < steps; i += 2) {
            step_pipelined(std::false_type{}, i, set0, set1);
            step_pipelined(std::false_type{}, i + 1, set1, set0);
        }
        step_pipelined(std::false_type{}, steps - 2, set0, set1);
        contract(set1);                                  // the last step has nothing to prefetch
    } else {
        for (int64_t i = 0; i < steps; i += 2) {
            step(i, set0, set1);
            if (i + 1 < steps) step(i + 1, set1, set0);
        }
    }
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
        reinterpret_cast<f32x4*>(out + 16 * 16 * 64)[lane] = bsum;
    }
}

__global__ void __launch_bounds__(256, 1)
wgrad_unit_bf16_kernel(const ffn_mlp_chain ch, const ffn_wgrad_unit* __restrict__ units,
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
            // (the logits-head unit: f32, register-staged, wgrad_common.h)
            head_segment(ch, unit, seg, smem, saved, d_logits, n, num_blocks, partials);
        } else {
            const bool m_wide = unit.m_quads > 32, n_wide = unit.n_quads > 32;
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            const bool bias = unit.want_bias != 0 && (!n_wide || (wave & 1) == 0);
#define FFN_UNIT(CA, CB)                                                                         \
    do {                                                                                         \
        if (bias) unit_segment16<CA, CB, true>(ch, unit, seg, smem, saved, dz, num_blocks, partials);   \
        else unit_segment16<CA, CB, false>(ch, unit, seg, smem, saved, dz, num_blocks, partials);       \
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

extern "C" int ffn_mlp_wgrad_units_bf16x3(const ffn_mlp_chain* chain, const ffn_wgrad_unit* units,
                                          const ffn_wgrad_segment* segments, const int32_t* seg_start,
                                          int num_groups, const float* saved, const float* dz,
                                          const float* d_logits, int64_t n, float* partials,
                                          void* stream) {
    if (n <= 0 || num_groups <= 0) return fail_arg("ffn_mlp_wgrad_units_bf16x3: shape");
    const size_t lds = kUnitLdsBytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_unit_bf16_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wgrad_unit_bf16_kernel, dim3(num_groups), dim3(256), lds, (hipStream_t)stream,
                       *chain, units, segments, seg_start, saved, dz, d_logits, n, partials);
    return check_launch("ffn_mlp_wgrad_units_bf16x3");
}
