// Split-bf16 weight gradients (OPT-IN "bf16x3" training precision; wgrad.hip's exact-f32 kernel
// stays the parity mode):  dW[j][k] = sum_s dZ[j][s] X[k][s]  with every f32 product as three
// v_mfma_f32_32x32x16_bf16 products (hi*lo + lo*hi + hi*hi, f32 accumulation).
//
// Same units, segments, LDS images, staging and partial format as the f32 kernel -- the f32
// slabs are copied into LDS exactly as they sit in HBM, a wave owns a 128x128 quadrant of dW in
// 256 accumulator registers, and the reduce kernel is shared.  What changes is the contraction
// step: the bf16 instruction contracts SIXTEEN samples, eight consecutive ones per lane, so lane
// (g, i) reads the float4 of channel quad i at samples 16*ks + 8*g + 0..7 (eight conflict-free
// ds_read_b128, the same LDS traffic per sample as the f32 kernel), splits its 32 values into
// (hi, lo) bf16 -- component p of the eight float4 is the operand of output tile row-set p --
// and issues 48 matrix instructions per sixteen samples where the f32 kernel issues 128 at
// four times the cost each.  At that rate the kernel is bound by the slab traffic (2 x 32 KiB per
// block and unit), not by the matrix pipe.
//
// FEAT (regenerate_features, OFF by default): units whose input window is a slab of encoding
// features do not read that slab -- each wave regenerates its share of the window (the feature
// code of the forward kernels, fourier_features.h: same instructions, same bits) from the block's
// 32 sample positions straight into the LDS image, and the forward pass need not save features
// (2 of the 5 KiB per sample it writes, 2 of the 9 KiB this kernel reads, tiny model).  Built,
// bit-identical, and measured SLOWER: the block's vector work (450 conversion + 280 feature
// instructions, which this in-order wave does not overlap with its 96 matrix instructions)
// then exceeds the 9.4k cycles the 64 KiB of slab traffic cost -- weight gradients 8.9 -> 11.9 ms
// for 0.7 ms saved in the forward pass (knock-outs: no matrix instructions -3.2 ms, no feature
// generation -2.6, no conversions -1.2, no contraction at all -3.1).
#include <type_traits>

#include "fourier_features.h"
#include "wgrad_common.h"

namespace ffn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// component P of eight float4 -> (hi, lo) bf16 operand
template <int P>
__device__ __forceinline__ void split_component(const f32x4 (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float x = v[t][P];
        const __bf16 h = (__bf16)x;
        hi[t] = h;
        lo[t] = (__bf16)(x - (float)h);
    }
}

// the encoding whose features fill a slab slot, -1 if the slot holds activations
__device__ __forceinline__ int encoding_of_slot(const ffn_mlp_chain& ch, int slot) {
    for (int i = 0; i < ch.num_steps; ++i)
        if (ch.step[i].save_enc_slot == slot) return ch.step[i].enc_id;
    return -1;
}

template <int CA, int CB, bool BIAS, bool FEAT>
__device__ __forceinline__ void unit_segment16(const ffn_mlp_chain& ch, const ffn_wgrad_unit& unit,
                                               const ffn_wgrad_segment& seg, char* smem,
                                               const float* __restrict__ saved,
                                               const float* __restrict__ dz, int64_t num_blocks,
                                               float* __restrict__ partials,
                                               const float* __restrict__ points, int64_t n,
                                               int enc_id) {
    constexpr int NQ = (CA / 4) * (CB / 4);   // quadrants that exist: 4, 2 or 1
    constexpr int NCH = CA + CB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int hh = lane >> 5;
    const int li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qd = wave & (NQ - 1);
    // waves that share a quadrant split the block's two 16-sample steps (a 128x128 unit keeps
    // two of its four waves idle: they contribute zero partials)
    const int part = NQ == 4 ? 0 : (NQ == 2 ? wave >> 1 : wave);
    const int ks_begin = NQ == 4 ? 0 : part, ks_end = NQ == 4 ? 2 : (part < 2 ? part + 1 : part);
    const int mp = CB == 8 ? qd >> 1 : qd, np = CB == 8 ? (qd & 1) : 0;
    const bool a_ok = li < unit.m_quads - 32 * mp;   // this lane's quad exists in the M window
    const bool b_ok = li < unit.n_quads - 32 * np;
    const int64_t a_stride = (int64_t)ch.slot_channels[unit.m_slot] * 128;   // bytes per block
    const int64_t b_stride = (int64_t)ch.slot_channels[unit.n_slot] * 128;
    const char* a_s = reinterpret_cast<const char*>(dz + ch.slot_offset[unit.m_slot] * num_blocks * 32) +
                      unit.m_cq0 * 512 + seg.blk_begin * a_stride;
    const char* b_s = reinterpret_cast<const char*>(saved + ch.slot_offset[unit.n_slot] * num_blocks * 32) +
                      unit.n_cq0 * 512 + seg.blk_begin * b_stride;
    a_s = uniform_ptr(a_s);
    b_s = uniform_ptr(b_s);
    const int ca_last = (unit.m_quads >> 3) - 1, cb_last = (unit.n_quads >> 3) - 1;
    const int t16 = tid * 16;

    f32x16 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.0f;
    f32x4 bsum = zero4();

    // staging registers: R[j] = chunk j of the A image (j < CA) / chunk j-CA of the B image
    // (wgrad.hip: chunks past the end of a window re-read its last chunk)
    f32x4 R[NCH];
    typedef const f32x4 __attribute__((address_space(1)))* gptr;
    constexpr int NST = FEAT ? CA : NCH;      // chunks that are staged (FEAT: the A image only)

    // FEAT: this lane's sample of a block, and the window's K groups generated into LDS image
    // `buf`: wave w takes groups w, w+4, ... (group g = channel quads 2g and 2g+1 = lane halves)
    const EncRegs enc = load_enc(ch.enc[FEAT ? enc_id : 0],
                                 reinterpret_cast<const float*>(smem + kUnitLdsBytes) + (FEAT ? enc_id : 0) * kEncTablePitch);
    const int groups = unit.n_quads >> 1, g_first = unit.n_cq0 >> 1;
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    auto load_point = [&](int64_t blk) {
        if (!FEAT) return;
        blk = blk < num_blocks ? blk : num_blocks - 1;
        int64_t sample = blk * 32 + li;
        sample = sample < n ? sample : n - 1;             // the forward pass clamps the same way
        px = points[sample * 3 + 0]; py = points[sample * 3 + 1]; pz = points[sample * 3 + 2];
    };
    // Trips of four K groups (two feature_oct calls = four independent packed sincos chains).
    // Wave w owns the contiguous groups [w*per, (w+1)*per).
    auto generate = [&](int buf) {
        if (!FEAT) return;
        const f32x2 s0 = (f32x2)(enc.scale * px), s1 = (f32x2)(enc.scale * py), s2 = (f32x2)(enc.scale * pz);
        const f32x4 q0 = (f32x4)(enc.scale * px), q1 = (f32x4)(enc.scale * py), q2 = (f32x4)(enc.scale * pz);
        const int per = ((groups + 15) >> 4) << 2;          // groups per wave, a multiple of 4
        const int gl_end = (wave + 1) * per < groups ? (wave + 1) * per : groups;
        for (int gl = wave * per; gl < gl_end; gl += 4) {
            const int g = g_first + gl;
            f32x4 v[4];
            if (gl + 3 < gl_end && 4 * (g + 3) + 3 < enc.F) {
                feature_oct(enc, g, hh, q0, q1, q2, v[0], v[1]);
                feature_oct(enc, g + 2, hh, q0, q1, q2, v[2], v[3]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = feature_quad<false>(enc, g + u, hh, px, py, pz, s0, s1, s2);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cq = 2 * (gl + u) + hh;
                if (gl + u < gl_end)
                    *reinterpret_cast<f32x4*>(smem + image_b(buf) + (cq * 32 + (li ^ (cq & 15))) * 16) = v[u];
            }
        }
    };
#define FFN_REQUEST(j)                                                                         \
    do {                                                                                       \
        gptr chunk = (j) < CA ? (gptr)(a_s + ((j) < ca_last ? (j) : ca_last) * 4096)           \
                              : (gptr)(b_s + ((j) - CA < cb_last ? (j) - CA : cb_last) * 4096);  \
        asm volatile("" : "+s"(chunk));                                                        \
        R[j] = __builtin_nontemporal_load(&chunk[tid]);                                                                     \
    } while (0)
#define FFN_DEPOSIT(cur, j)                                                                    \
    *reinterpret_cast<f32x4*>(smem + ((j) < CA ? image_a(cur) + (j) * 4096                     \
                                               : image_b(cur) + ((j) - CA) * 4096) + t16) = R[j]

    // ---- prologue: first block -> LDS buffer 0, second block -> registers
    load_point(seg.blk_begin);
#pragma unroll
    for (int j = 0; j < NST; ++j) FFN_REQUEST(j);
#pragma unroll
    for (int j = 0; j < NST; ++j) FFN_DEPOSIT(0, j);
    generate(0);
    load_point(seg.blk_begin + 1);
    a_s += a_stride;
    b_s += b_stride;
    if (seg.blk_begin + 1 < seg.blk_end) {
#pragma unroll
        for (int j = 0; j < NST; ++j) FFN_REQUEST(j);
    }
    a_s += a_stride;       // from here on: the block after next
    b_s += b_stride;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // This lane's float4 of sample u = 16 ks + 8 hh + t sits at byte (u ^ (li & 15)) * 16 of its
    // quad row = lane_x ^ ((16 ks + t) << 4) with lane_x = ((8 hh) ^ (li & 15)) << 4 (disjoint
    // bits).  Idle lanes of a narrow window point into the image's zero row.
    const unsigned lane_x = (unsigned)(((8 * hh) ^ (li & 15)) << 4);
    const char* a_row = smem + image_a(0) + (a_ok ? (32 * mp + li) * 512 : kImageBytes);
    const char* b_row = smem + image_b(0) + (b_ok ? (32 * np + li) * 512 : kImageBytes);

    auto block_body = [&](auto cur_tag, int64_t blk_of_body, bool has1, bool has2) {
        constexpr int CUR = decltype(cur_tag)::value;
        constexpr int kToggle = CUR * kImageStride;
        // The copy of the next block (registers -> free LDS buffer) comes first and the requests
        // for the block after it follow at once: a block is only ~6k cycles of work here, and the
        // requests must be in flight for all of it to cover the HBM round trip (the f32 kernel,
        // 16k cycles per block, requests in the middle of the block).
        if (has1) {
#pragma unroll
            for (int j = 0; j < NST; ++j) FFN_DEPOSIT(1 - CUR, j);
            generate(1 - CUR);               // (the point of block b+1 was requested a block ago)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has2) {
#pragma unroll
            for (int j = 0; j < NST; ++j) FFN_REQUEST(j);
            load_point(blk_of_body + 2);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bool mine = ks >= ks_begin && ks < ks_end;        // (wave-uniform)
            if (mine) {
                f32x4 av[8], bv[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned off = lane_x ^ (unsigned)((16 * ks + t) << 4);
                    av[t] = *reinterpret_cast<const f32x4*>(a_row + kToggle + off);
                    bv[t] = *reinterpret_cast<const f32x4*>(b_row + kToggle + off);
                }
                if (BIAS) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) bsum += av[t];
                }
                bf16x8 ah[4], al[4], bh[4], bl[4];
                split_component<0>(av, ah[0], al[0]); split_component<1>(av, ah[1], al[1]);
                split_component<2>(av, ah[2], al[2]); split_component<3>(av, ah[3], al[3]);
                split_component<0>(bv, bh[0], bl[0]); split_component<1>(bv, bh[1], bl[1]);
                split_component<2>(bv, bh[2], bl[2]); split_component<3>(bv, bh[3], bl[3]);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[p], bl[q], acc[p][q], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[p], bh[q], acc[p][q], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[p], bh[q], acc[p][q], 0, 0, 0);
            }
        }
        a_s += a_stride;
        b_s += b_stride;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    for (int64_t blk = seg.blk_begin; blk < seg.blk_end; blk += 2) {
        block_body(std::integral_constant<int, 0>{}, blk, blk + 1 < seg.blk_end, blk + 2 < seg.blk_end);
        if (blk + 1 < seg.blk_end)
            block_body(std::integral_constant<int, 1>{}, blk + 1, blk + 2 < seg.blk_end, blk + 3 < seg.blk_end);
    }
#undef FFN_REQUEST
#undef FFN_DEPOSIT

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
                       float* __restrict__ partials, const float* __restrict__ positions,
                       const float* __restrict__ views, int regenerate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int k = threadIdx.x; k < 4 * 128; k += 256)      // the zero row behind each image
        reinterpret_cast<float*>(smem + (k >> 7) * kImageStride + kImageBytes)[k & 127] = 0.0f;
    if (regenerate) stage_encoding_tables(ch.enc, reinterpret_cast<float*>(smem + kUnitLdsBytes), threadIdx.x, 256);
    __syncthreads();
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
        if (unit.kind == 1) {
            // (the logits-head unit streams X at the HBM rate in f32 already)
            head_segment(ch, unit, seg, smem, saved, d_logits, n, num_blocks, partials);
        } else {
            const bool m_wide = unit.m_quads > 32, n_wide = unit.n_quads > 32;
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            const bool bias = unit.want_bias != 0 && (!n_wide || (wave & 1) == 0);
            const int enc_id = regenerate ? encoding_of_slot(ch, unit.n_slot) : -1;
            const float* points = enc_id == 1 ? views : positions;
#define FFN_UNIT(CA, CB)                                                                         \
    do {                                                                                         \
        if (enc_id >= 0) {                                                                       \
            if (bias) unit_segment16<CA, CB, true, true>(ch, unit, seg, smem, saved, dz, num_blocks, partials, points, n, enc_id);   \
            else unit_segment16<CA, CB, false, true>(ch, unit, seg, smem, saved, dz, num_blocks, partials, points, n, enc_id);      \
        } else {                                                                                 \
            if (bias) unit_segment16<CA, CB, true, false>(ch, unit, seg, smem, saved, dz, num_blocks, partials, points, n, 0);     \
            else unit_segment16<CA, CB, false, false>(ch, unit, seg, smem, saved, dz, num_blocks, partials, points, n, 0);        \
        }                                                                                        \
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
                                          const float* positions, const float* views,
                                          int regenerate_features, void* stream) {
    if (n <= 0 || num_groups <= 0) return fail_arg("ffn_mlp_wgrad_units_bf16x3: shape");
    if (regenerate_features && positions == nullptr)
        return fail_arg("ffn_mlp_wgrad_units_bf16x3: regenerating features needs the sample positions");
    const size_t lds = kUnitLdsBytes + kEncTableBytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_unit_bf16_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wgrad_unit_bf16_kernel, dim3(num_groups), dim3(256), lds, (hipStream_t)stream,
                       *chain, units, segments, seg_start, saved, dz, d_logits, n, partials, positions, views,
                       regenerate_features);
    return check_launch("ffn_mlp_wgrad_units_bf16x3");
}
