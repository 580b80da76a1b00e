// Weight gradients of the fused MLP: dW_l[j][k] = sum_s dZ_l[j][s] * X_l[k][s], exact-f32 MFMA.
//
// The contraction runs over SAMPLES, so both operands are read "transposed" relative to how
// the forward/dgrad kernels produced them: lane (hh, i) takes the float4 holding channel quad
// i (4 consecutive channels) of sample 2u+hh from the block-layout slabs.  The four
// components of that float4 feed four different 32-row output tiles (rows 4i+p, p=0..3), so
// one 16-byte read per operand drives 16 MFMAs (a 128x128 patch of dW) per 2 samples.
// The XOR in the slab layout makes a lane group's 16-byte LDS reads conflict-free.
// Every input of a layer is a slab: hidden activations saved by the forward kernel, and the
// encoding features it saved as well (regenerating them here costs VALU time that does not
// hide under f32 MFMA).  Persistent workgroups walk a host-built, cost-balanced list of
// (unit, sample-block range) segments, keep a 128x128 patch per wave in 256 accumulator
// registers, and dump one partial per wave and segment; a second kernel sums the partials in
// a fixed order (deterministic) and scatters into the natural nn.Linear gradient layout.
#include <type_traits>

#include "wgrad_common.h"

namespace ffn {

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float comp_of(const f32x4& v, int q) { return v[q]; }
__device__ __forceinline__ float comp_of(const f32x2& v, int q) { return v[q]; }
__device__ __forceinline__ float comp_of(float v, int) { return v; }
template <int NF> struct BOperand { typedef f32x4 type; };
template <> struct BOperand<2> { typedef f32x2 type; };
template <> struct BOperand<4> { typedef float type; };

// CA / CB = 4 KiB chunks staged per block for the A / B image: 8 for a window wider than 128
// channels (two quadrants along that side), 4 otherwise; BIAS = this wave also sums dZ.
//
// NF = fold of a NARROW input window (the encoding features: 63 or 27 channels).  A lane's
// float4 is a channel quad and its four components feed four different column tiles, so a window
// of <= 16 quads would leave half of every tile's 32 columns on the zero row.  Folded, lane group
// g = li / (32 / NF) reads quad li % (32 / NF) and takes components g * (4 / NF) ... of it (an
// 8- or 4-byte LDS read at +g * 16 / NF): 4 / NF full column tiles per step instead of four
// half-empty ones -- half / a quarter of the matrix instructions, and only the window's 2 / 1
// chunks staged.  Column j of tile q' is then channel 4 * (j % (32/NF)) + (j / (32/NF)) * (4/NF)
// + q' (ffn_reduce_job.n_fold tells the reducer).
template <int CA, int CB, bool BIAS, int NF = 1>
__device__ __forceinline__ void unit_segment(const ffn_mlp_chain& ch, const ffn_wgrad_unit& unit,
                                             const ffn_wgrad_segment& seg, char* smem,
                                             const float* __restrict__ saved,
                                             const float* __restrict__ dz, int64_t num_blocks,
                                             float* __restrict__ partials) {
    static_assert(NF == 1 || CB == 4, "only a window of one quadrant folds");
    constexpr int NQ = (CA / 4) * (CB / 4);   // quadrants that exist: 4, 2 or 1
    constexpr int CBS = CB / NF;              // chunks of the B image that are staged
    constexpr int NCH = CA + CBS;
    constexpr int S = 4 * NQ;                 // sample pairs of a block this wave multiplies
    constexpr int NQM = 4 / NF;               // column tiles per step
    constexpr int kGroupLanes = 32 / NF;
    // chunks [k * NCH / (S/4), (k+1) * NCH / (S/4)) are deposited in step k of [0, S/4);
    // chunks [k * NCH / (S/2), ...) requested in step S/4 + k of [S/4, 3S/4)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int hh = lane >> 5;
    const int li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qd = wave & (NQ - 1);
    const int part = NQ == 4 ? 0 : (NQ == 2 ? wave >> 1 : wave);
    const int mp = CB == 8 ? qd >> 1 : qd, np = CB == 8 ? (qd & 1) : 0;
    const int qi = li & (kGroupLanes - 1);           // B: this lane's quad ...
    const int fg = li / kGroupLanes;                 // ... and which of its components (folded windows)
    const bool a_ok = li < unit.m_quads - 32 * mp;   // this lane's quad exists in the M window
    const bool b_ok = qi < unit.n_quads - 32 * np;
    const int64_t a_stride = (int64_t)ch.slot_channels[unit.m_slot] * 128;   // bytes per block
    const int64_t b_stride = (int64_t)ch.slot_channels[unit.n_slot] * 128;
    // wave-uniform base of the block being requested; lanes add tid*16
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

    // staging registers: R[j] = chunk j of the A image (j < CA) / chunk j-CA of the B image.
    // Chunks past the end of a window re-read its last chunk (never consumed: the lanes that
    // would are pointed at the zero row), which keeps the loop free of branches.
    f32x4 R[NCH];
    // uniform chunk pointer indexed by the thread: scalar base + lane offset addressing, the
    // chunk select and the block walk stay on the SALU
    typedef const f32x4 __attribute__((address_space(1)))* gptr;
#define FFN_REQUEST(j)                                                                         \
    do {                                                                                       \
        gptr chunk = (j) < CA ? (gptr)(a_s + ((j) < ca_last ? (j) : ca_last) * 4096)           \
                              : (gptr)(b_s + ((j) - CA < cb_last ? (j) - CA : cb_last) * 4096);  \
        asm volatile("" : "+s"(chunk));    /* its own scalar base: no per-lane 64-bit adds */  \
        R[j] = __builtin_nontemporal_load(&chunk[tid]);                                                                     \
    } while (0)
#define FFN_DEPOSIT(cur, j)                                                                    \
    *reinterpret_cast<f32x4*>(smem + ((j) < CA ? image_a(cur) + (j) * 4096                     \
                                               : image_b(cur) + ((j) - CA) * 4096) + t16) = R[j]

    // ---- prologue: first block -> LDS buffer 0, second block -> registers
#pragma unroll
    for (int j = 0; j < NCH; ++j) FFN_REQUEST(j);
#pragma unroll
    for (int j = 0; j < NCH; ++j) FFN_DEPOSIT(0, j);
    a_s += a_stride;
    b_s += b_stride;
    if (seg.blk_begin + 1 < seg.blk_end) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) FFN_REQUEST(j);
    }
    a_s += a_stride;       // from here on: the block after next
    b_s += b_stride;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // LDS addresses of this lane's operand float4 for each of its S steps, in the EVEN
    // buffers (the odd ones are +kImageStride, an immediate).  Sample pair u of a quad row
    // sits at byte ((2u + hh) ^ (li & 15)) * 16 = x0 ^ (i << 5) for u = part*S + i.
    // Idle lanes of a narrow window point into the image's zero row.
    const unsigned x0 = (unsigned)(((hh ^ (li & 15)) << 4) ^ ((part * S) << 5));
    const unsigned x0b = (unsigned)(((hh ^ (qi & 15)) << 4) ^ ((part * S) << 5));
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned a_row = lds0 + image_a(0) + (a_ok ? (32 * mp + li) * 512 : kImageBytes);
    const unsigned b_row = lds0 + image_b(0) + (b_ok ? (32 * np + qi) * 512 : kImageBytes) +
                           (NF > 1 ? fg * (16 / NF) : 0);
    unsigned a_at[S], b_at[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        a_at[i] = a_row + (x0 ^ (unsigned)(i << 5));
        b_at[i] = b_row + (x0b ^ (unsigned)(i << 5));
    }
    typedef typename BOperand<NF>::type bvec;
#define FFN_READ_OPERANDS(dst_a, dst_b, idx, tail)                                             \
    do {                                                                                       \
        if constexpr (NF == 1)                                                                 \
            asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4" tail \
                         : "=&v"(dst_a), "=&v"(dst_b) : "v"(a_at[idx]), "v"(b_at[idx]), "i"(kToggle) : "memory"); \
        else if constexpr (NF == 2)                                                            \
            asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4" tail  \
                         : "=&v"(dst_a), "=&v"(dst_b) : "v"(a_at[idx]), "v"(b_at[idx]), "i"(kToggle) : "memory"); \
        else                                                                                   \
            asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b32 %1, %3 offset:%4" tail  \
                         : "=&v"(dst_a), "=&v"(dst_b) : "v"(a_at[idx]), "v"(b_at[idx]), "i"(kToggle) : "memory"); \
    } while (0)

    // one block out of the buffers of parity CUR (compile-time: the toggle is an immediate).
    // Operand reads are hand-issued (inline asm, so hipcc's waitcnt pass does not see them)
    // right behind the step's MFMAs have started, and waited for by hand at the end of the
    // step: a full ~1000 cycles of matrix work covers the LDS latency.
    auto block_body = [&](auto cur_tag, bool has1, bool has2) {
        constexpr int CUR = decltype(cur_tag)::value;
        constexpr int kToggle = CUR * kImageStride;
        f32x4 a, a_n;
        bvec b, b_n;
        FFN_READ_OPERANDS(a, b, 0, "\n\ts_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int i = 0; i < S; ++i) {
            constexpr int kLast = S - 1;
            const int in = i + 1 < S ? i + 1 : kLast;
            FFN_READ_OPERANDS(a_n, b_n, in, "");
            __builtin_amdgcn_sched_barrier(0);
            if (BIAS) bsum += a;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int q = 0; q < NQM; ++q)
                    acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], comp_of(b, q), acc[p][q], 0, 0, 0);
                if (p == 0) {
                    if (i < S / 4) {
                        if (has1) {
#pragma unroll
                            for (int j = (i * NCH) / (S / 4); j < ((i + 1) * NCH) / (S / 4); ++j)
                                FFN_DEPOSIT(1 - CUR, j);
                        }
                    } else if (i < 3 * S / 4) {
                        if (has2) {
#pragma unroll
                            for (int j = ((i - S / 4) * NCH) / (S / 2); j < ((i - S / 4 + 1) * NCH) / (S / 2); ++j)
                                FFN_REQUEST(j);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            a = a_n;
            b = b_n;
        }
        a_s += a_stride;
        b_s += b_stride;
        __builtin_amdgcn_s_barrier();     // LDS traffic of this block is complete (lgkmcnt(0) above)
    };

    for (int64_t blk = seg.blk_begin; blk < seg.blk_end; blk += 2) {
        block_body(std::integral_constant<int, 0>{}, blk + 1 < seg.blk_end, blk + 2 < seg.blk_end);
        if (blk + 1 < seg.blk_end)
            block_body(std::integral_constant<int, 1>{}, blk + 2 < seg.blk_end, blk + 3 < seg.blk_end);
    }
#undef FFN_REQUEST
#undef FFN_DEPOSIT
#undef FFN_READ_OPERANDS

    {
        float* out = partials + (int64_t)(seg.slot + wave) * kPartialFloats;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < NQM; ++q)      // (the reducer does not read the tiles a fold leaves out)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[((p * 4 + q) * 16 + r) * 64 + lane] = acc[p][q][r];
        reinterpret_cast<f32x4*>(out + 16 * 16 * 64)[lane] = bsum;
    }
}

__global__ void __launch_bounds__(256, 1)
wgrad_unit_kernel(const ffn_mlp_chain ch, const ffn_wgrad_unit* __restrict__ units,
                  const ffn_wgrad_segment* __restrict__ segments,
                  const int32_t* __restrict__ seg_start, const float* __restrict__ saved,
                  const float* __restrict__ dz, const float* __restrict__ d_logits, int64_t n,
                  float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int k = threadIdx.x; k < 4 * 128; k += 256)      // the zero row behind each image
        reinterpret_cast<float*>(smem + (k >> 7) * kImageStride + kImageBytes)[k & 127] = 0.0f;
    __syncthreads();
    const int64_t num_blocks = (n + 31) / 32;
    const int seg_lo = seg_start[blockIdx.x], seg_hi = seg_start[blockIdx.x + 1];
    for (int si = seg_lo; si < seg_hi; ++si) {
        ffn_wgrad_segment seg = segments[si];
        // the host may plan for a rounded-up block count (one plan serves a range of batch
        // sizes): clamp, and give the reducer zeros for a segment that fell off the end
        if (seg.blk_end > num_blocks) seg.blk_end = num_blocks;
        if (seg.blk_end <= seg.blk_begin) {
            float* out = partials + (int64_t)(seg.slot + (threadIdx.x >> 6)) * kPartialFloats;
            for (int e = threadIdx.x & 63; e < kPartialFloats; e += 64) out[e] = 0.0f;
            continue;
        }
        const ffn_wgrad_unit unit = units[seg.job];
        if (unit.kind == 1) {
            head_segment(ch, unit, seg, smem, saved, d_logits, n, num_blocks, partials);
        } else {
            const bool m_wide = unit.m_quads > 32, n_wide = unit.n_quads > 32;
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            // the bias gradient rides on the waves of the layer's first window that own
            // an n-half 0 quadrant (all variants execute the same barriers)
            const bool bias = unit.want_bias != 0 && (!n_wide || (wave & 1) == 0);
#define FFN_UNIT(CA, CB, NF)                                                                   \
    do {                                                                                       \
        if (bias) unit_segment<CA, CB, true, NF>(ch, unit, seg, smem, saved, dz, num_blocks, partials);  \
        else unit_segment<CA, CB, false, NF>(ch, unit, seg, smem, saved, dz, num_blocks, partials);      \
    } while (0)
            const int fold = ffn_wgrad_fold(unit.n_quads);
            if (m_wide && n_wide) FFN_UNIT(8, 8, 1);
            else if (n_wide) FFN_UNIT(4, 8, 1);
            else if (m_wide) {
                if (fold == 4) FFN_UNIT(8, 4, 4);
                else if (fold == 2) FFN_UNIT(8, 4, 2);
                else FFN_UNIT(8, 4, 1);
            } else {
                if (fold == 4) FFN_UNIT(4, 4, 4);
                else if (fold == 2) FFN_UNIT(4, 4, 2);
                else FFN_UNIT(4, 4, 1);
            }
#undef FFN_UNIT
        }
        __syncthreads();
    }
}

// Sums a job's partials in slot order and scatters into the flat natural-layout gradient.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const ffn_reduce_job* __restrict__ rjobs, const float* __restrict__ partials,
                    float* __restrict__ grads) {
    const ffn_reduce_job job = rjobs[blockIdx.y];
    const int elems = job.kind == 0 ? 16 * 16 * 64 : 4 * 16 * 64;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < elems + 256; e += gridDim.x * blockDim.x) {
        float sum = 0.0f;
        if (e < elems) {
            const int lane = e & 63;
            const int r = (e >> 6) & 15;
            const int tile = e >> 10;
            const int hh = lane >> 5, jj = lane & 31;
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
            int row, kint;
            if (job.kind == 0) {
                // folded window (n_fold 2 / 4): 4 / n_fold column tiles exist, column jj is quad
                // jj % (32 / n_fold), component (jj / (32 / n_fold)) * (4 / n_fold) + q
                const int fold = job.n_fold > 1 ? job.n_fold : 1;
                const int group_lanes = 32 / fold, tiles = 4 / fold;
                const int p = tile >> 2, q = tile & 3;
                if (q >= tiles) continue;
                const int quad = jj % group_lanes;
                row = job.m_ch0 + 4 * i + p;
                kint = job.k_base + 4 * (job.n_quad0 + quad) + (jj / group_lanes) * tiles + q;
                if (quad >= job.n_quads) continue;
            } else {
                // head unit (wgrad_common.h): float (q * 4 + ch) * 64 + lane, lane = (sample
                // block, logits column); the thread of sample block 0 sums the sixteen
                const int q = e >> 8, chn = (e >> 6) & 3, col_j = lane & 3;
                row = col_j;
                kint = job.k_base + 4 * (job.n_quad0 + q) + chn;
                if ((lane >> 2) != 0 || col_j >= job.lg_n || q >= job.n_quads) continue;
            }
            if (row >= job.rows) continue;
            const int col = job.col_map[kint];
            if (col < 0) continue;
            for (int s = job.slot_begin; s < job.slot_end; s += job.slot_stride) {
                const float* part = partials + (int64_t)s * kPartialFloats + e;
                if (job.kind == 0) {
                    sum += part[0];
                } else {
#pragma unroll
                    for (int b = 0; b < 16; ++b) sum += part[4 * b];
                }
            }
            grads[job.w_grad_off + (int64_t)row * job.ld + col] = sum;
        } else if (job.has_bias) {
            // bias strip: full job -> float4 per lane (channels 4i+p); head -> one float per lane
            const int b = e - elems;
            if (job.kind == 0) {
                const int i = b >> 2, p = b & 3;  // i in 0..63 covers both halves; use i < 32
                if (i >= 32) continue;
                for (int s = job.slot_begin; s < job.slot_end; s += job.slot_stride) {
                    const float* strip = partials + (int64_t)s * kPartialFloats + 16 * 16 * 64;
                    sum += strip[i * 4 + p] + strip[(32 + i) * 4 + p];
                }
                const int row = job.m_ch0 + 4 * i + p;
                if (row < job.rows) grads[job.b_grad_off + row] = sum;
            } else {
                if (b >= job.lg_n) continue;
                for (int s = job.slot_begin; s < job.slot_end; s += job.slot_stride) {
                    const float* strip = partials + (int64_t)s * kPartialFloats + 16 * 16 * 64;
#pragma unroll
                    for (int sbk = 0; sbk < 16; ++sbk) sum += strip[4 * sbk + b];
                }
                grads[job.b_grad_off + b] = sum;
            }
        }
    }
}

}  // namespace ffn

using namespace ffn;

extern "C" int ffn_mlp_wgrad_units(const ffn_mlp_chain* chain, const ffn_wgrad_unit* units,
                                   const ffn_wgrad_segment* segments, const int32_t* seg_start,
                                   int num_groups, const float* saved, const float* dz,
                                   const float* d_logits, int64_t n, float* partials,
                                   void* stream) {
    if (n <= 0 || num_groups <= 0) return fail_arg("ffn_mlp_wgrad_units: shape");
    const size_t lds = kUnitLdsBytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_unit_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wgrad_unit_kernel, dim3(num_groups), dim3(256), lds, (hipStream_t)stream,
                       *chain, units, segments, seg_start, saved, dz, d_logits, n, partials);
    return check_launch("ffn_mlp_wgrad_units");
}

extern "C" int ffn_mlp_wgrad_reduce(const ffn_reduce_job* jobs, int num_jobs,
                                    const float* partials, float* grads, void* stream) {
    if (num_jobs <= 0) return fail_arg("ffn_mlp_wgrad_reduce: no jobs");
    // one element per thread: with ~20 jobs that is ~1300 workgroups, enough loads in flight to
    // pull the partials (64 MiB for 256 segments) at the HBM rate -- 17 blocks per job left the
    // kernel latency-bound (65 us where the traffic is worth ~20)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(66, num_jobs), dim3(256), 0, (hipStream_t)stream,
                       jobs, partials, grads);
    return check_launch("ffn_mlp_wgrad_reduce");
}

extern "C" int64_t ffn_mlp_wgrad_partial_floats(void) { return kPartialFloats; }
