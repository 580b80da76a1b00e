// Weight gradients of the fused MLP: dW_l[j][k] = sum_s dZ_l[j][s] * X_l[k][s], exact-f32 MFMA.
//
// The contraction runs over SAMPLES, so both operands are read "transposed" relative to how
// the forward/dgrad kernels produced them: lane (hh, i) takes the float4 holding channel quad
// i (4 consecutive channels) of sample 2u+hh from the block-layout slabs.  The four
// components of that float4 feed four different 32-row output tiles (rows 4i+p, p=0..3), so
// one 16-byte load per operand drives 16 MFMAs (a 128x128 patch of dW) per 2 samples.
// The XOR in the slab layout makes a lane group's 16-byte reads land on distinct 64-byte
// sectors.  Persistent waves walk a host-built, cost-balanced list of (job, sample-block
// range) segments, keep the 128x128 patch in 256 accumulator registers, and dump one
// partial per segment; a second kernel sums the partials in a fixed order (deterministic)
// and scatters into the natural nn.Linear gradient layout.
#include "common.h"

namespace ffn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPartialFloats = 16 * 16 * 64 + 256;  // 16 tiles + bias strip

struct EncRegsW {
    const float* b; const float* a; int F; int Fi; int raw; float scale;
};

__device__ __forceinline__ EncRegsW load_enc_w(const ffn_encoding& e, const float* table = nullptr) {
    EncRegsW r;
    r.F = e.num_freq; r.Fi = e.num_freq > 0 ? e.num_freq : 1;
    r.b = table != nullptr ? table : e.b;              // LDS copy when the kernel staged one
    r.a = table != nullptr ? table + 3 * r.Fi : e.a;
    r.raw = (e.include_input != 0 || e.num_freq == 0) ? 1 : 0;
    r.scale = e.scale;
    return r;
}

// same internal feature order as mlp.hip: quad cq holds frequencies 2cq and 2cq+1
__device__ __forceinline__ f32x4 feature_quad(const EncRegsW& enc, int cq, float x0, float x1,
                                              float x2) {
    f32x4 v;
    const float s0 = enc.scale * x0, s1 = enc.scale * x1, s2 = enc.scale * x2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = 2 * cq + j;
        const int kk = k < enc.Fi ? k : enc.Fi - 1;
        float ang = s0 * enc.b[kk];
        ang = __builtin_fmaf(s1, enc.b[enc.Fi + kk], ang);
        ang = __builtin_fmaf(s2, enc.b[2 * enc.Fi + kk], ang);
        float sn, cs;
        fast_sincos(ang, sn, cs);
        const float amp = enc.a[kk];
        const int c = 2 * (k - enc.F);
        const float raw_even = (enc.raw && c == 0) ? x0 : ((enc.raw && c == 2) ? x2 : 0.0f);
        const float raw_odd = (enc.raw && c == 0) ? x1 : 0.0f;
        const bool trig = k < enc.F;
        v[2 * j] = trig ? amp * cs : raw_even;
        v[2 * j + 1] = trig ? amp * sn : raw_odd;
    }
    return v;
}

struct PanelSrc {
    const f32x4* base;   // slab start (float4 units) or nullptr for an encoding
    int64_t block_stride;// float4 per 32-sample block
    int cq;              // this lane's channel quad inside the slab / encoding
    bool valid;
};

__device__ __forceinline__ f32x4 zero4() { f32x4 z; z[0] = z[1] = z[2] = z[3] = 0.0f; return z; }

// operand of lane (hh, i) for sample sp (= 2u+hh) of block blk
template <bool ENC>
__device__ __forceinline__ f32x4 panel_load(const PanelSrc& src, const EncRegsW& enc,
                                            const float* __restrict__ xyz, int64_t n, int64_t blk,
                                            int sp) {
    f32x4 v;
    if (ENC) {
        int64_t sample = blk * 32 + sp;
        sample = sample < n ? sample : n - 1;
        const float x0 = xyz[sample * 3 + 0], x1 = xyz[sample * 3 + 1], x2 = xyz[sample * 3 + 2];
        v = feature_quad(enc, src.cq, x0, x1, x2);
    } else {
        v = src.base[blk * src.block_stride + src.cq * 32 + (sp ^ (src.cq & 15))];
    }
    return src.valid ? v : zero4();
}

// dW patch [128 dZ channels] x [128 X channels] of one segment
template <bool ENC>
__device__ __forceinline__ void full_job(const PanelSrc& zs, const PanelSrc& xs,
                                         const EncRegsW& enc, const float* __restrict__ xyz,
                                         int64_t n, const ffn_wgrad_segment& seg, int hh, int lane,
                                         float* __restrict__ out) {
    f32x16 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.0f;
    f32x4 bsum = zero4();
    const int64_t steps = (seg.blk_end - seg.blk_begin) * 16;
    f32x4 a = panel_load<false>(zs, enc, xyz, n, seg.blk_begin, hh);
    f32x4 b = panel_load<ENC>(xs, enc, xyz, n, seg.blk_begin, hh);
    for (int64_t t = 0; t < steps; ++t) {
        const int64_t tn = t + 1 < steps ? t + 1 : t;
        const int64_t blk_n = seg.blk_begin + (tn >> 4);
        const int sp_n = 2 * (int)(tn & 15) + hh;
        const f32x4 a_n = panel_load<false>(zs, enc, xyz, n, blk_n, sp_n);
        const f32x4 b_n = panel_load<ENC>(xs, enc, xyz, n, blk_n, sp_n);
        bsum += a;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], b[q], acc[p][q], 0, 0, 0);
        if (ENC) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
        }
        a = a_n;
        b = b_n;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[((p * 4 + q) * 16 + r) * 64 + lane] = acc[p][q][r];
    reinterpret_cast<f32x4*>(out + 16 * 16 * 64)[lane] = bsum;
}

// head rows: dW^T patch [128 X channels] x [<=4 d_logits columns]
template <bool ENC>
__device__ __forceinline__ void head_job(const PanelSrc& xs, const EncRegsW& enc,
                                         const float* __restrict__ xyz,
                                         const float* __restrict__ d_logits, int lg_col, int lg_n,
                                         int64_t n, const ffn_wgrad_segment& seg, int hh, int li,
                                         int lane, float* __restrict__ out) {
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    float bsum = 0.0f;
    const bool col_ok = li < lg_n;
    const int col = col_ok ? lg_col + li : 0;
    const int64_t steps = (seg.blk_end - seg.blk_begin) * 16;
    f32x4 a = panel_load<ENC>(xs, enc, xyz, n, seg.blk_begin, hh);
    int64_t s0 = seg.blk_begin * 32 + hh;
    float b = (col_ok && s0 < n) ? d_logits[s0 * 4 + col] : 0.0f;
    for (int64_t t = 0; t < steps; ++t) {
        const int64_t tn = t + 1 < steps ? t + 1 : t;
        const int64_t blk_n = seg.blk_begin + (tn >> 4);
        const int sp_n = 2 * (int)(tn & 15) + hh;
        const f32x4 a_n = panel_load<ENC>(xs, enc, xyz, n, blk_n, sp_n);
        const int64_t sn = blk_n * 32 + sp_n;
        const int64_t sc = sn < n ? sn : n - 1;
        float b_n = d_logits[sc * 4 + col];
        b_n = (col_ok && sn < n) ? b_n : 0.0f;
        bsum += b;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], b, acc[p], 0, 0, 0);
        a = a_n;
        b = b_n;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(p * 16 + r) * 64 + lane] = acc[p][r];
    out[16 * 16 * 64 + lane] = bsum;
}

__global__ void __launch_bounds__(256, 1)
wgrad_kernel(const ffn_mlp_chain ch, const ffn_wgrad_job* __restrict__ jobs,
             const ffn_wgrad_segment* __restrict__ segments, const int32_t* __restrict__ seg_start,
             const float* __restrict__ saved, const float* __restrict__ dz,
             const float* __restrict__ d_logits, const float* __restrict__ positions,
             const float* __restrict__ views, int64_t n, float* __restrict__ partials) {
    const int lane = threadIdx.x & 63;
    const int hh = lane >> 5;
    const int li = lane & 31;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t num_blocks = (n + 31) / 32;
    const int seg_lo = seg_start[wave], seg_hi = seg_start[wave + 1];
    for (int si = seg_lo; si < seg_hi; ++si) {
        const ffn_wgrad_segment seg = segments[si];
        const ffn_wgrad_job job = jobs[seg.job];
        float* out = partials + (int64_t)seg.slot * kPartialFloats;
        if (seg.blk_end <= seg.blk_begin) continue;

        // the X panel (N side for a full job, M side for a head job)
        PanelSrc xs;
        const bool x_enc = job.n_kind == 1;
        const EncRegsW enc = load_enc_w(ch.enc[x_enc ? job.n_slot : 0]);
        const float* xyz = (x_enc && job.n_slot == 1) ? views : positions;
        xs.valid = li < job.n_quads;
        xs.cq = job.n_cq0 + (xs.valid ? li : 0);   // idle lanes re-read quad 0, then get zeroed
        xs.base = nullptr;
        xs.block_stride = 0;
        if (!x_enc) {
            xs.base = reinterpret_cast<const f32x4*>(saved + ch.slot_offset[job.n_slot] * num_blocks * 32);
            xs.block_stride = ch.slot_channels[job.n_slot] * 8;
        }
        if (job.kind == 0) {
            PanelSrc zs;
            zs.base = reinterpret_cast<const f32x4*>(dz + ch.slot_offset[job.m_slot] * num_blocks * 32);
            zs.block_stride = ch.slot_channels[job.m_slot] * 8;
            zs.valid = li < job.m_quads;
            zs.cq = job.m_cq0 + (zs.valid ? li : 0);
            if (x_enc) full_job<true>(zs, xs, enc, xyz, n, seg, hh, lane, out);
            else full_job<false>(zs, xs, enc, xyz, n, seg, hh, lane, out);
        } else {
            if (x_enc) head_job<true>(xs, enc, xyz, d_logits, job.lg_col, job.lg_n, n, seg, hh, li, lane, out);
            else head_job<false>(xs, enc, xyz, d_logits, job.lg_col, job.lg_n, n, seg, hh, li, lane, out);
        }
    }
}

// ---------------------------------------------------------------------------------- units
// LDS-staged variant for full 256x256 products.  A 256-thread workgroup owns one unit
// (dZ slab window of <=256 channels  x  input window of <=256 channels) over a range of
// sample blocks.  Per block the two 32 KiB operand images are brought into LDS exactly as
// they sit in HBM (global_load_lds, 1 KiB per wave-instruction, double buffered), or -- for
// an encoding input -- generated into LDS by the workgroup, one frequency pair per thread
// per MFMA step.  Each wave then computes its 128x128 quadrant from LDS with two
// conflict-free ds_read_b128 per 16 MFMAs.  HBM/L2 traffic = the unique operand bytes.
constexpr int kUnitBufBytes = 64 * 1024;   // A image 32 KiB + B image 32 KiB

__device__ __forceinline__ void stage_slab(const f32x4* __restrict__ block_base, int quads,
                                           char* lds_part, int tid, int wave) {
    // chunk k = 8 quads = 4 KiB = one 16-byte piece per thread
    for (int k = 0; k < (quads >> 3); ++k) {
        const char* g = reinterpret_cast<const char*>(block_base) + k * 4096 + tid * 16;
        char* l = lds_part + k * 4096 + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
}

// LDS map of the unit kernel: [2 x (A image 32 KiB | B image 32 KiB)] [encoding tables]
// [512 B of zeros: the row idle lanes of a narrow window read instead of branching/selecting]
constexpr int kUnitZeroOffset = 2 * kUnitBufBytes + kEncTableBytes;
constexpr int kUnitLdsBytes = kUnitZeroOffset + 512;

template <bool ENC>
__device__ __forceinline__ void unit_segment(const ffn_mlp_chain& ch, const ffn_wgrad_unit& unit,
                                             const ffn_wgrad_segment& seg, char* smem,
                                             const float* __restrict__ saved,
                                             const float* __restrict__ dz,
                                             const float* __restrict__ xyz, int64_t n,
                                             int64_t num_blocks, float* __restrict__ partials) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int hh = lane >> 5;
    const int li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mp = wave >> 1, np = wave & 1;
    const bool a_ok = li < unit.m_quads - 32 * mp;   // this lane's quad exists in the M window
    const bool b_ok = li < unit.n_quads - 32 * np;
    const f32x4* a_slab = reinterpret_cast<const f32x4*>(dz + ch.slot_offset[unit.m_slot] * num_blocks * 32) + unit.m_cq0 * 32;
    const int64_t a_stride = ch.slot_channels[unit.m_slot] * 8;
    const f32x4* b_slab = nullptr;
    int64_t b_stride = 0;
    const int enc_id = ENC ? unit.n_slot : 0;
    EncRegsW enc = load_enc_w(ch.enc[enc_id], reinterpret_cast<const float*>(smem + 2 * kUnitBufBytes) +
                                                  enc_id * kEncTablePitch);
    if (!ENC) {
        b_slab = reinterpret_cast<const f32x4*>(saved + ch.slot_offset[unit.n_slot] * num_blocks * 32) + unit.n_cq0 * 32;
        b_stride = ch.slot_channels[unit.n_slot] * 8;
    }
    // feature generation: thread -> sample tid&31, quads (tid>>5) + 8j, j = 0..7, one
    // frequency (cos, sin) per MFMA step: step u makes frequency 2*quad + (u&1) of j = u>>1
    const int f_s = tid & 31;
    const int f_q = tid >> 5;
    struct Tab { float b0, b1, b2, amp; };
    auto table_of = [&](int u) {
        Tab t;
        const int cq = f_q + 8 * (u >> 1);
        const int k = 2 * (unit.n_cq0 + cq) + (u & 1);
        const int kk = k < enc.Fi ? k : enc.Fi - 1;
        t.b0 = enc.b[kk]; t.b1 = enc.b[enc.Fi + kk]; t.b2 = enc.b[2 * enc.Fi + kk]; t.amp = enc.a[kk];
        return t;
    };
    auto feature_store = [&](char* buf, int u, const Tab& t, float x0, float x1, float x2) {
        const int cq = f_q + 8 * (u >> 1);
        const int k = 2 * (unit.n_cq0 + cq) + (u & 1);
        const float s0 = enc.scale * x0, s1 = enc.scale * x1, s2 = enc.scale * x2;
        float ang = s0 * t.b0;
        ang = __builtin_fmaf(s1, t.b1, ang);
        ang = __builtin_fmaf(s2, t.b2, ang);
        float sn, cs;
        fast_sincos(ang, sn, cs);
        float2 v;
        // wave-uniform: the highest frequency any lane of this wave touches in this step
        const int k_top = 2 * (unit.n_cq0 + 2 * wave + 1 + 8 * (u >> 1)) + 1;
        if (k_top < enc.F && 2 * wave + 1 + 8 * (u >> 1) < unit.n_quads) {
            v.x = t.amp * cs;                      // all-trig fast path: no selects
            v.y = t.amp * sn;
            *reinterpret_cast<float2*>(buf + 32 * 1024 + (cq * 32 + (f_s ^ (cq & 15))) * 16 + (u & 1) * 8) = v;
            return;
        }
        const int c = 2 * (k - enc.F);
        const float raw_even = (enc.raw && c == 0) ? x0 : ((enc.raw && c == 2) ? x2 : 0.0f);
        const float raw_odd = (enc.raw && c == 0) ? x1 : 0.0f;
        const bool trig = k < enc.F;
        v.x = trig ? t.amp * cs : raw_even;
        v.y = trig ? t.amp * sn : raw_odd;
        if (cq < unit.n_quads)
            *reinterpret_cast<float2*>(buf + 32 * 1024 + (cq * 32 + (f_s ^ (cq & 15))) * 16 + (u & 1) * 8) = v;
    };
    auto load_xyz = [&](int64_t blk, float& x0, float& x1, float& x2) {
        int64_t sample = blk * 32 + f_s;
        sample = sample < n ? sample : n - 1;
        x0 = xyz[sample * 3 + 0]; x1 = xyz[sample * 3 + 1]; x2 = xyz[sample * 3 + 2];
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.0f;
    f32x4 bsum = zero4();

    // ---- prologue: stage the first block into buffer 0
    {
        char* buf = smem;
        stage_slab(a_slab + seg.blk_begin * a_stride, unit.m_quads, buf, tid, wave);
        if (ENC) {
            float x0, x1, x2;
            load_xyz(seg.blk_begin, x0, x1, x2);
            for (int u = 0; u < 16; ++u) feature_store(buf, u, table_of(u), x0, x1, x2);
        } else {
            stage_slab(b_slab + seg.blk_begin * b_stride, unit.n_quads, buf + 32 * 1024, tid, wave);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int sw = li & 15;
    const int zero_row = kUnitZeroOffset / 16;     // float4 index of the zero row
    for (int64_t blk = seg.blk_begin; blk < seg.blk_end; ++blk) {
        const int cur = (int)((blk - seg.blk_begin) & 1);
        char* buf = smem + cur * kUnitBufBytes;
        char* nxt = smem + (cur ^ 1) * kUnitBufBytes;
        const bool more = blk + 1 < seg.blk_end;
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (more) {
            stage_slab(a_slab + (blk + 1) * a_stride, unit.m_quads, nxt, tid, wave);
            if (ENC) load_xyz(blk + 1, x0, x1, x2);
            else stage_slab(b_slab + (blk + 1) * b_stride, unit.n_quads, nxt + 32 * 1024, tid, wave);
        }
        // idle lanes of a narrow window read the zero row: no select in the MFMA stream.
        // Operand reads are hand-issued (inline asm, so hipcc's waitcnt pass does not see
        // them) right behind the step's MFMAs have started, and waited for by hand at the end
        // of the step: a full ~1000 cycles of matrix work covers the LDS latency.
        const unsigned lds0 = (unsigned)(size_t)smem;
        const unsigned a_base = a_ok ? lds0 + cur * kUnitBufBytes + (32 * mp + li) * 512 : lds0 + kUnitZeroOffset;
        const unsigned b_base = b_ok ? lds0 + cur * kUnitBufBytes + 32 * 1024 + (32 * np + li) * 512 : lds0 + kUnitZeroOffset;
        f32x4 a, b, a_n, b_n;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(a_base + ((hh ^ sw) << 4)), "v"(b_base + ((hh ^ sw) << 4)) : "memory");
        Tab tn = table_of(0);
#pragma unroll 2
        for (int u = 0; u < 16; ++u) {
            const int un = u + 1 < 16 ? u + 1 : u;
            const unsigned off_n = (unsigned)(((2 * un + hh) ^ sw) << 4);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3"
                         : "=&v"(a_n), "=&v"(b_n) : "v"(a_base + off_n), "v"(b_base + off_n) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (ENC) {
                const Tab t = tn;
                tn = table_of(un);
                if (more) feature_store(nxt, u, t, x0, x1, x2);
            }
            bsum += a;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], b[q], acc[p][q], 0, 0, 0);
            if (ENC) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            a = a_n;
            b = b_n;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    {   // (idle quadrants store zeros into their own, never-read slot)
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

// Logits-head rows inside the LDS-staged kernel: dW_head[j][k] = sum_s dl[s][j] * X[k][s] for
// the <= 4 head outputs j.  X (<= 256 channels) is staged like any slab; wave w owns channel
// half (w & 1) and sample half (w >> 1) of every block: 8 steps x 4 MFMAs -- the kernel is
// HBM-bound (32 KiB per ~2k cycles per CU), which is the point: no scattered global reads.
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
    const f32x4* x_slab = reinterpret_cast<const f32x4*>(saved + ch.slot_offset[unit.n_slot] * num_blocks * 32) + unit.n_cq0 * 32;
    const int64_t x_stride = ch.slot_channels[unit.n_slot] * 8;
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    float bsum = 0.0f;
    // this wave's 8 sample pairs of a block; d_logits straight from HBM (512 B / block),
    // fetched one block ahead so that their latency hides behind the previous block
    auto load_dl = [&](int64_t blk, float (&dst)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t sample = blk * 32 + 2 * (8 * sh + k) + hh;
            const int64_t sc = sample < n ? sample : n - 1;
            const float v = d_logits[sc * 4 + col];
            dst[k] = (col_ok && sample < n) ? v : 0.0f;
        }
    };
    float dl[8], dl_next[8];
    load_dl(seg.blk_begin, dl);
    stage_slab(x_slab + seg.blk_begin * x_stride, unit.n_quads, smem, tid, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int sw = li & 15;
    const f32x4* all = reinterpret_cast<const f32x4*>(smem);
    for (int64_t blk = seg.blk_begin; blk < seg.blk_end; ++blk) {
        const int cur = (int)((blk - seg.blk_begin) & 1);
        char* buf = smem + cur * kUnitBufBytes;
        const int64_t nb = blk + 1 < seg.blk_end ? blk + 1 : blk;
        if (blk + 1 < seg.blk_end)
            stage_slab(x_slab + (blk + 1) * x_stride, unit.n_quads, smem + (cur ^ 1) * kUnitBufBytes, tid, wave);
        load_dl(nb, dl_next);
        const f32x4* lx = x_ok ? reinterpret_cast<const f32x4*>(buf) + (32 * half + li) * 32 : all + kUnitZeroOffset / 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 a = lx[(2 * (8 * sh + k) + hh) ^ sw];
            bsum += dl[k];
#pragma unroll
            for (int p = 0; p < 4; ++p)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], dl[k], acc[p], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) dl[k] = dl_next[k];
    }
    float* out = partials + (int64_t)(seg.slot + wave) * kPartialFloats;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(p * 16 + r) * 64 + lane] = acc[p][r];
    out[16 * 16 * 64 + lane] = bsum;
}

__global__ void __launch_bounds__(256, 1)
wgrad_unit_kernel(const ffn_mlp_chain ch, const ffn_wgrad_unit* __restrict__ units,
                  const ffn_wgrad_segment* __restrict__ segments,
                  const int32_t* __restrict__ seg_start, const float* __restrict__ saved,
                  const float* __restrict__ dz, const float* __restrict__ d_logits,
                  const float* __restrict__ positions, const float* __restrict__ views, int64_t n,
                  float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stage_encoding_tables(ch.enc, reinterpret_cast<float*>(smem + 2 * kUnitBufBytes), threadIdx.x, 256);
    if (threadIdx.x < 128) reinterpret_cast<float*>(smem + kUnitZeroOffset)[threadIdx.x] = 0.0f;
    __syncthreads();
    const int64_t num_blocks = (n + 31) / 32;
    const int seg_lo = seg_start[blockIdx.x], seg_hi = seg_start[blockIdx.x + 1];
    for (int si = seg_lo; si < seg_hi; ++si) {
        const ffn_wgrad_segment seg = segments[si];
        if (seg.blk_end <= seg.blk_begin) continue;
        const ffn_wgrad_unit unit = units[seg.job];
        if (unit.kind == 1) {
            head_segment(ch, unit, seg, smem, saved, d_logits, n, num_blocks, partials);
        } else if (unit.n_kind == 1) {
            const float* xyz = unit.n_slot == 1 ? views : positions;
            unit_segment<true>(ch, unit, seg, smem, saved, dz, xyz, n, num_blocks, partials);
        } else {
            unit_segment<false>(ch, unit, seg, smem, saved, dz, positions, n, num_blocks, partials);
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
            for (int s = job.slot_begin; s < job.slot_end; s += job.slot_stride)
                sum += partials[(int64_t)s * kPartialFloats + e];
            const int lane = e & 63;
            const int r = (e >> 6) & 15;
            const int tile = e >> 10;
            const int hh = lane >> 5, jj = lane & 31;
            const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
            int row, kint;
            if (job.kind == 0) {
                const int p = tile >> 2, q = tile & 3;
                row = job.m_ch0 + 4 * i + p;
                kint = job.k_base + 4 * (job.n_quad0 + jj) + q;
                if (jj >= job.n_quads) continue;
            } else {
                row = jj;
                kint = job.k_base + 4 * (job.n_quad0 + i) + tile;
                if (jj >= job.lg_n || i >= job.n_quads) continue;
            }
            if (row >= job.rows) continue;
            const int col = job.col_map[kint];
            if (col < 0) continue;
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
                    sum += strip[b] + strip[32 + b];
                }
                grads[job.b_grad_off + b] = sum;
            }
        }
    }
}

}  // namespace ffn

using namespace ffn;

extern "C" int ffn_mlp_wgrad(const ffn_mlp_chain* chain, const ffn_wgrad_job* jobs,
                             const ffn_wgrad_segment* segments, const int32_t* seg_start,
                             int num_waves, const float* saved, const float* dz,
                             const float* d_logits, const float* positions, const float* views,
                             int64_t n, float* partials, void* stream) {
    if (n <= 0 || num_waves <= 0 || (num_waves & 3)) return fail_arg("ffn_mlp_wgrad: shape");
    hipLaunchKernelGGL(wgrad_kernel, dim3(num_waves / 4), dim3(256), 0, (hipStream_t)stream, *chain,
                       jobs, segments, seg_start, saved, dz, d_logits, positions, views, n, partials);
    return check_launch("ffn_mlp_wgrad");
}

extern "C" int ffn_mlp_wgrad_units(const ffn_mlp_chain* chain, const ffn_wgrad_unit* units,
                                   const ffn_wgrad_segment* segments, const int32_t* seg_start,
                                   int num_groups, const float* saved, const float* dz,
                                   const float* d_logits, const float* positions,
                                   const float* views, int64_t n, float* partials, void* stream) {
    if (n <= 0 || num_groups <= 0) return fail_arg("ffn_mlp_wgrad_units: shape");
    const size_t lds = kUnitLdsBytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_unit_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wgrad_unit_kernel, dim3(num_groups), dim3(256), lds, (hipStream_t)stream,
                       *chain, units, segments, seg_start, saved, dz, d_logits, positions, views, n,
                       partials);
    return check_launch("ffn_mlp_wgrad_units");
}

extern "C" int ffn_mlp_wgrad_reduce(const ffn_reduce_job* jobs, int num_jobs,
                                    const float* partials, float* grads, void* stream) {
    if (num_jobs <= 0) return fail_arg("ffn_mlp_wgrad_reduce: no jobs");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(17, num_jobs), dim3(256), 0, (hipStream_t)stream,
                       jobs, partials, grads);
    return check_launch("ffn_mlp_wgrad_reduce");
}

extern "C" int64_t ffn_mlp_wgrad_partial_floats(void) { return kPartialFloats; }
