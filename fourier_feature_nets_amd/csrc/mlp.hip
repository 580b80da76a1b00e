// Fused Fourier-feature MLP on the gfx950 matrix cores, exact-f32 mode
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain).
//
// Work decomposition: ONE WAVEFRONT OWNS 32 CONSECUTIVE SAMPLES for the whole network.
// The layer product is computed transposed, D[out][sample] = W[out][in] * X[in][sample]:
// weights are the MFMA A operand, activations the B operand.  With that orientation the
// accumulator layout of layer l (lane = (h, sample), registers = output channels
// 8q+4h+p) IS the B-operand layout of layer l+1 for the K ordering c' = 8g+4h+p, so a
// layer's output goes through bias+ReLU and one lane-linear ds_write_b128 into the wave's
// private 32 KiB LDS slab and comes back as one lane-linear ds_read_b128 per 32 MFMAs --
// no transposes, no inter-wave barriers, no HBM round trips between layers.
// Weights are streamed from L2 in a pre-packed operand order (ffn_mlp_pack), 1 KiB
// coalesced per wave-load, double buffered in registers against the 64-cycle MFMAs.
// f32 VALU work does not overlap f32 MFMA on gfx950, so nothing but loads rides in the K
// loops: Fourier features are generated in packed-f32 bursts into the slab between K loops
// (and saved for the weight gradients when training), biases are the accumulators' initial
// value, logits heads are folded into the producing layer's epilogue.  The same interpreter
// runs the backward-data chain (dZ_{l-1} = relu'(H_{l-1}) * W_l^T dZ_l) with transposed
// weight packs and 1-bit ReLU masks.  A "wide" variant (two waves per 32-sample block, 64 KiB
// slab) covers 512-channel layers.
#include "fourier_features.h"
#include "composite_terms.h"
#include "focus_terms.h"
#include "occupancy_map.h"

namespace ffn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the activation slabs live in LDS and are addressed through LDS-typed pointers (built from the
// 32-bit LDS address): no generic-pointer aperture checks, ds_read / ds_write by construction
typedef f32x4 __attribute__((address_space(3))) lds_f32x4;
__device__ __forceinline__ lds_f32x4* lds_slab(const char* generic_lds_address) {
    return (lds_f32x4*)(uint32_t)(uintptr_t)generic_lds_address;
}
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSamplesPerWave = 32;
constexpr int kWavesPerBlock = 4;
constexpr int kActBytesPerWave = 32 * 1024;  // 256 channels x 32 samples x 4 B
constexpr int kBiasLdsFloats = 4096;         // LDS copy of the bias buffer's first floats: every
                                             // fused-head block, then as many step biases as fit

enum Mode { kInfer = 0, kTrainFwd = 1, kBackward = 2 };

// ---------------------------------------------------------------------------------- pack
__global__ void __launch_bounds__(256)
pack_kernel(const float* __restrict__ src, int rows, int cols, int ld, int transpose,
            const int32_t* __restrict__ row_map, const int32_t* __restrict__ col_map, int groups,
            int tiles, float* __restrict__ dst) {
    const int64_t total = (int64_t)groups * tiles * 256;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        const int64_t go = e >> 8;
        const int o = (int)(go % tiles);
        const int g = (int)(go / tiles);
        int r = 32 * o + (lane & 31);             // operand row  (M index)
        int c = 8 * g + 4 * (lane >> 5) + p;      // operand col  (K index)
        if (row_map != nullptr) r = row_map[r];
        if (col_map != nullptr) c = col_map[c];
        float v = 0.0f;
        if (r >= 0 && c >= 0) {
            const int sr = transpose ? c : r;
            const int sc = transpose ? r : c;
            if (sr < rows && sc < cols) v = src[(int64_t)sr * ld + sc];
        }
        dst[e] = v;
    }
}

// Every operand pack, bias block and fused-head block of a model in ONE launch (blockIdx.y = job):
// what MlpProgram.pack() used to issue as one pack launch + one device copy per layer and head
// -- a dozen ~5 us launches in front of every optimisation step, 6 % of the step at the
// reference's default batch of 1024 rays.
__global__ void __launch_bounds__(256)
pack_jobs_kernel(const ffn_pack_job* __restrict__ jobs) {
    const ffn_pack_job job = jobs[blockIdx.y];
    if (job.kind == 1) {          // strided copy: dst[c*dst_cs + r*dst_rs] = src[r*ld + c]
        const int64_t total = (int64_t)job.rows * job.cols;
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
             e += (int64_t)gridDim.x * blockDim.x) {
            const int r = (int)(e / job.cols), c = (int)(e % job.cols);
            job.dst[(int64_t)c * job.dst_cs + (int64_t)r * job.dst_rs] = job.src[(int64_t)r * job.ld + c];
        }
        return;
    }
    const int64_t total = (int64_t)job.groups * job.tiles * 256;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(e & 3);
        const int lane = (int)((e >> 2) & 63);
        const int64_t go = e >> 8;
        const int o = (int)(go % job.tiles);
        const int g = (int)(go / job.tiles);
        const int r = 32 * o + (lane & 31);
        int c = 8 * g + 4 * (lane >> 5) + p;
        if (job.col_map != nullptr) c = job.col_map[c];
        float v = 0.0f;
        if (c >= 0) {
            const int sr = job.transpose ? c : r;
            const int sc = job.transpose ? r : c;
            if (sr < job.rows && sc < job.cols) v = job.src[(int64_t)sr * job.ld + sc];
        }
        job.dst[e] = v;
    }
}

// K groups [c0, c0+count) of an encoding into the wave's slab (and, when training, into the
// saved-feature slab `fsave`).  Four independent groups per trip give the in-order wave the
// instruction-level parallelism that hides the packed-FMA latency.
template <bool SAVE>
__device__ __forceinline__ void generate_features(const EncRegs& enc, int c0, int count, int h,
                                                  int s, int lane, float x0, float x1, float x2,
                                                  lds_f32x4* act, f32x4* fsave, int g_begin, int g_step) {
    // trips of 4 K groups: g_begin, g_begin + g_step, ... (two waves sharing a slab split them)
    const f32x2 s0 = (f32x2)(enc.scale * x0), s1 = (f32x2)(enc.scale * x1), s2 = (f32x2)(enc.scale * x2);
    const f32x4 q0 = (f32x4)(enc.scale * x0), q1 = (f32x4)(enc.scale * x1), q2 = (f32x4)(enc.scale * x2);
    int g_trig = (enc.F >> 2) - c0;               // groups with 4g+3 < F
    g_trig = g_trig < 0 ? 0 : (g_trig > count ? count : g_trig);
    g_trig &= ~3;
    for (int g = g_begin; g < count; g += g_step) {
        f32x4 v[4];
        if (g < g_trig) {
            feature_oct(enc, c0 + g, h, q0, q1, q2, v[0], v[1]);
            feature_oct(enc, c0 + g + 2, h, q0, q1, q2, v[2], v[3]);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = feature_quad<false>(enc, c0 + g + u, h, x0, x1, x2, s0, s1, s2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            act[(g + u) * 64 + lane] = v[u];
            if (SAVE) __builtin_nontemporal_store(v[u], &fsave[(2 * (c0 + g + u) + h) * 32 + (s ^ ((2 * (c0 + g + u) + h) & 15))]);
        }
    }
}

// ---------------------------------------------------------------------------------- MFMA block
template <int OT>
__device__ __forceinline__ void mma_group(f32x16 (&acc)[OT], const f32x4 (&a)[OT], f32x4 x) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int o = 0; o < OT; ++o)
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[o][p], x[p], acc[o], 0, 0, 0);
}

// Scheduling pipelines (sched_group_barrier masks: 0x002 VALU, 0x008 MFMA, 0x020 VMEM read,
// 0x100 DS read).  A single in-order wave per SIMD has nothing to hide an s_waitcnt behind, so
// every load (weights from L2, operands / tables from LDS) must ISSUE at the head of a
// half-trip and be consumed a full MFMA block later; left alone, the scheduler sinks the
// ds_reads next to their users and each lgkmcnt wait stalls the matrix pipe ~100 cycles.
template <int OT>
__device__ __forceinline__ void pipeline_plain() {
    __builtin_amdgcn_sched_group_barrier(0x020, 2 * OT, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * OT, 0);
}
// Inference: the weight loads of a half-trip (2 * OT global_load_dwordx4, consumed one half-trip
// = 8 * OT MFMAs later) are spread ONE PER FOUR MFMAs.  A lone wave spends ~10 issue cycles per L2
// load, and sixteen of them in a row in front of the MFMA block leave the matrix pipe idle for
// ~150 cycles per half-trip once the previous block's last MFMA has drained (knock-out without
// the loads: -3.3 % tiny, -4 % full NeRF; scripts/probes/variants_kloop.py); behind a running
// MFMA their issue is free: -1.7 % on the inference forward, same box
// (scripts/gpu/r3_ab_kloop.sh).  The pattern orders ONE basic block, and the training / backward
// trips are several (their save stores are conditional).  Measured there and not adopted: the
// condition hoisted into two copies of the loop (150-190 spilled registers, training forward
// +14 %); the loads moved next to their half-trip's MFMAs (training forward +-0, full-NeRF
// backward data +5 %) -- with stores in the trip the in-order vmcnt, not the issue slots, decides.
template <int OT, bool SAVES = false>
__device__ __forceinline__ void pipeline_spread() {
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    // training: the half-trip's two save stores go FIRST -- vmcnt retires in order, so the loads
    // behind them wait for their acknowledgement, but those loads are consumed a half-trip later
    if (SAVES) __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
#pragma unroll
    for (int i = 0; i < 2 * OT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
}
// `wg` is the WAVE-UNIFORM address of tile 0 of the wanted K group; tiles are 64 float4
// apart, lanes 16 B.  Keeping the running pointer uniform lets the loads use the scalar-base
// addressing mode: the group walk and its end-of-panel clamp are SALU work, not VALU work
// in the MFMA stream.
template <int OT>
__device__ __forceinline__ void load_group(f32x4 (&a)[OT], const f32x4* __restrict__ wg, int lane) {
    // tiles 4..7 lie past the 12-bit immediate offset: give them their own scalar base (the
    // empty asm keeps the compiler from folding it back into per-lane 64-bit address math)
    typedef const f32x4 __attribute__((address_space(1)))* gptr;
    const gptr lo = (gptr)wg;
    gptr hi = (gptr)(wg + 4 * 64);
    if (OT > 4) asm volatile("" : "+s"(hi));
#pragma unroll
    for (int o = 0; o < OT; ++o) a[o] = (o < 4 ? lo : hi)[(o & 3) * 64 + lane];
}

struct WaveCtx {
    int lane, h, s;          // lane = 32*h + s
    float x0, x1, x2;        // position of this lane's sample
    float v0, v1, v2;        // view direction of this lane's sample
    f32x4 dl;                // backward: d(loss)/d(logits) of this lane's sample
    lds_f32x4* act;          // this wave's LDS slab, indexed [group*64 + lane]
    const float* enc_table;  // LDS copies of the encoding tables
    const float* bias_lds;   // LDS copy of the head of the bias buffer (kBiasLdsFloats floats)
    const float* bias_glb;   // the whole bias buffer (steps whose bias lies past the LDS copy)
    uint4* masks;            // ReLU sign masks: [slot][block][half][lane] x 128 bit
    int64_t block;           // 32-sample block id inside this launch
    int64_t num_blocks;
    int64_t slab_block0;     // slabs are indexed by block id of the WHOLE batch: this launch's first
    int64_t slab_blocks;     // block and the batch's block count (a launch may cover a sub-range)
    float logit[4];
    int half;                // wide mode: which half of a step's output tiles this wave owns
    bool active;             // wide mode: false = lockstep dummy pass (block clamped, no stores)
};

// float4 index of (channel quad cq, sample s) inside a saved-activation block
__device__ __forceinline__ int saved_index(int cq, int s) { return cq * 32 + (s ^ (cq & 15)); }

__device__ __forceinline__ f32x4* slab_block(const ffn_mlp_chain& ch, float* base, int slot,
                                             const WaveCtx& w) {
    return reinterpret_cast<f32x4*>(base + ch.slot_offset[slot] * w.slab_blocks * 32) +
           (w.slab_block0 + w.block) * (int64_t)(ch.slot_channels[slot] * 8);
}

// Wide mode (a layer wider than 256 channels): two waves share one 64 KiB slab and one block
// of 32 samples; each owns half of every step's output tiles (OT = tiles / 2) and reads all
// of its K groups.  The in-place slab hand-off then needs workgroup barriers: one before the
// epilogue overwrites the slab (every reader of this step is done) and one after it.
__device__ __forceinline__ void team_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// FLAV specialises the epilogue on the step's (wave-uniform) flavour, so that its flags are not
// tested inside the unrolled tile loop (each runtime flag there is a scalar branch per channel
// quad: 3-4 fetch bubbles x 32 quads per step, in a loop with ~25 instructions per quad):
//   0 generic (flags read from the step at run time -- the narrow tile counts)
//   1 slab destination, no fused head, no output save   (hidden layers; dgrad steps)
//   2 slab destination, output saved                    (forward: fused head + save; dgrad: last step)
//   3 slab destination, fused head, no save             (forward inference)
//
// BIG (chains with a layer wider than 512 channels, up to 1024; ffn_mlp_chain.wide == 3): the team
// of FOUR waves with up to eight output tiles per wave and the whole 128 KiB slab area as its one
// slab (128 K groups: 1024 channels x 32 samples).  The fused-head weights of such a chain (4 per
// channel: 4100 floats for ONE 1024-channel head) outgrow the LDS copy of the bias buffer, so BIG
// steps read head blocks from the bias buffer itself (L2) -- like biases past the LDS copy.
template <int OT, int MODE, int TWP, int FLAV, bool BIG = false>
__device__ __forceinline__ void run_step(const ffn_mlp_chain& ch, const ffn_step& L,
                                         const ffn_step* next, WaveCtx& w,
                                         const float* __restrict__ packed_w,
                                         f32x4 (&pre)[16],               // groups 0,1 weights, prefetched
                                         float* __restrict__ slab_out) { // fwd: saved; bwd: dZ
    constexpr bool WIDE = TWP > 1;
    constexpr int TW = TWP;                       // waves sharing a step's output tiles (1, 2 or 4)
    constexpr int kChunk = BIG ? 128 : (WIDE ? 64 : 32);        // K groups the slab holds
    const int half = WIDE ? w.half : 0;
    // accumulators start at the bias (forward) or zero (backward): the 32 LDS reads go out
    // back to back here instead of sitting, each with its own wait, in the epilogue
    f32x16 acc[OT];
    if (MODE == kBackward) {
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][r] = 0.0f;
    } else if (L.b_off + 32 * OT * TW <= kBiasLdsFloats) {
        const float* bv = w.bias_lds + L.b_off + 32 * OT * half + 4 * w.h;
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + 32 * o + 8 * q);
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[o][4 * q + p] = b4[p];
            }
    } else {
        // a deep / 512-wide chain whose biases outgrow the LDS copy (a 512-wide full NeRF: 7.9k
        // floats): this step's bias comes from L2 -- two distinct 16-byte addresses per load
        const float* bv = w.bias_glb + L.b_off + 32 * OT * half + 4 * w.h;
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + 32 * o + 8 * q);
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[o][4 * q + p] = b4[p];
            }
    }

    const f32x4* wp = reinterpret_cast<const f32x4*>(packed_w + L.w_off) + half * OT * 64;   // uniform
    const int GA = L.act_groups;   // multiple of 4
    const int GX = L.aux_groups;   // multiple of 4; encoding features (fwd) or d_logits (bwd)
    const int G = GA + GX;
    constexpr int kGroupStride = TW * OT * 64;   // float4 between consecutive K groups
    // weights are double buffered two K groups deep: while pair A feeds the MFMAs, pair B
    // (two groups = 4096 MFMA cycles ahead) is in flight from L2.  `wnext` walks the packed
    // panel group by group: one 64-bit add per pair of groups, constant offsets otherwise.
    f32x4 wa0[OT], wa1[OT], wb0[OT], wb1[OT];
    f32x4 x0, x1, x2, x3;
#pragma unroll
    for (int o = 0; o < OT; ++o) { wa0[o] = pre[o]; wa1[o] = pre[8 + o]; }
    // (the group walk is kept as a uniform integer so that advancing and clamping it is
    // SALU work; the loads address  scalar base + lane*16 + immediate)
    int gnext = 2;                                       // next K-group pair to fetch
    const int glast = G - 2;
    const f32x4* wnext = wp + 2 * kGroupStride;
    // backward: the ReLU sign mask of the layer being differentiated, fetched a layer ahead
    const int64_t mask_at = (((int64_t)(L.mask_slot < 0 ? 0 : L.mask_slot) * w.num_blocks + w.block) * TW + half) * 64 + w.lane;
    uint4 mbits = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    if (MODE == kBackward && L.mask_slot >= 0) mbits = w.masks[mask_at];

    // ---- K segments.  Every operand comes out of the slab: first the activations the
    // previous step left there (GA groups), then the encoding features, generated into the
    // slab in bursts of up to kChunk K groups between two K loops.
    // Narrow training forward: EVERY K-loop trip saves the four operand groups it reads (hidden
    // activations on consume, encoding features likewise) -- the host plans it so
    // (validate_chain) -- which makes the trip one straight line whose issue order can be pinned.
    constexpr bool kTripSaves = MODE == kTrainFwd && !WIDE;
    const int feat_chunks = MODE == kBackward ? 0 : (GX + kChunk - 1) / kChunk;
    const int segs = (GA > 0 ? 1 : 0) + feat_chunks;
    for (int sg = 0; sg < segs; ++sg) {
        const bool feat = !(GA > 0 && sg == 0);
        int count = GA;
        f32x4* save = nullptr;
        if (!feat) {
            if (kTripSaves || (MODE != kInfer && L.save_in_slot >= 0 && w.active))
                save = slab_block(ch, slab_out, L.save_in_slot, w);
        } else {
            const int c0 = (sg - (GA > 0 ? 1 : 0)) * kChunk;
            count = GX - c0 < kChunk ? GX - c0 : kChunk;
            const EncRegs enc = load_enc(ch.enc[L.enc_id], w.enc_table + L.enc_id * kEncTablePitch);
            const float p0 = L.enc_id == 0 ? w.x0 : w.v0;
            const float p1 = L.enc_id == 0 ? w.x1 : w.v1;
            const float p2 = L.enc_id == 0 ? w.x2 : w.v2;
            if (WIDE && sg > 0) team_barrier();   // the partner may still read what we overwrite
            if (kTripSaves) {
                // the features are saved like activations: "on consume", by the K-loop trips
                generate_features<false>(enc, c0, count, w.h, w.s, w.lane, p0, p1, p2, w.act, nullptr,
                                         4 * half, 4 * TW);
                save = slab_block(ch, slab_out, L.save_enc_slot, w) + c0 * 64;   // 1 KiB per K group
            } else if (MODE == kTrainFwd && L.save_enc_slot >= 0 && w.active) {
                generate_features<true>(enc, c0, count, w.h, w.s, w.lane, p0, p1, p2, w.act,
                                        slab_block(ch, slab_out, L.save_enc_slot, w), 4 * half, 4 * TW);
            } else {
                generate_features<false>(enc, c0, count, w.h, w.s, w.lane, p0, p1, p2, w.act, nullptr,
                                         4 * half, 4 * TW);
            }
            if (WIDE) team_barrier();
        }
        // team mode: each wave of the team saves its share of what all consume (of the four K groups
        // of a trip: a pair's waves two each, a quad's waves one each)
        const bool saving = MODE != kInfer && save != nullptr;
        const bool save_0 = saving && (TW == 1 || half == 0);
        const bool save_1 = saving && (TW == 1 || half == (TW == 2 ? 0 : 1));
        const bool save_2 = saving && (TW == 1 || half == (TW == 2 ? 1 : 2));
        const bool save_3 = saving && (TW == 1 || half == (TW == 2 ? 1 : 3));
        // byte offset of this lane's float4 of K group g inside a saved block:
        //   16 * ((2g + h) * 32 + (s ^ ((2g + h) & 15)))  =  g * 1024  +  (lane_part ^ (((2g) & 15) << 4))
        // with lane_part = h * 512 + ((s ^ h) << 4): one v_xor with a scalar per store, the rest
        // rides in the scalar base
        const unsigned save_lane = (unsigned)(w.h * 512 + ((w.s ^ w.h) << 4));
        char* save_s = reinterpret_cast<char*>(save);
#define FFN_SAVE(g, value)                                                                     \
    __builtin_nontemporal_store((value), reinterpret_cast<f32x4*>(save_s + (int64_t)(g) * 1024 +   \
                              (save_lane ^ (unsigned)((((g) * 2) & 15) << 4))))
        const lds_f32x4* xa = w.act + w.lane;
        x0 = xa[0];
        x1 = xa[64];
        if (MODE == kInfer || kTripSaves) {
            // one basic block per trip (pipeline_spread): the next trip's first operands are read
            // unconditionally -- the last trip re-reads its own groups, never consumed
            for (int g = 0; g < count; g += 4) {
                load_group<OT>(wb0, wnext, w.lane);
                load_group<OT>(wb1, wnext + kGroupStride, w.lane);
                x2 = xa[128];
                x3 = xa[192];
                if (kTripSaves) {
                    FFN_SAVE(g, x0);
                    FFN_SAVE(g + 1, x1);
                }
                mma_group<OT>(acc, wa0, x0);
                mma_group<OT>(acc, wa1, x1);
                gnext = gnext + 2 < glast ? gnext + 2 : glast;      // clamp at the end of the panel
                wnext = wp + (int64_t)gnext * kGroupStride;
                load_group<OT>(wa0, wnext, w.lane);
                load_group<OT>(wa1, wnext + kGroupStride, w.lane);
                xa += g + 4 < count ? 256 : 0;
                if (kTripSaves) {
                    FFN_SAVE(g + 2, x2);
                    FFN_SAVE(g + 3, x3);
                }
                x0 = xa[0];
                x1 = xa[64];
                mma_group<OT>(acc, wb0, x2);
                mma_group<OT>(acc, wb1, x3);
                gnext = gnext + 2 < glast ? gnext + 2 : glast;
                wnext = wp + (int64_t)gnext * kGroupStride;
                pipeline_spread<OT, kTripSaves>();
                pipeline_spread<OT, kTripSaves>();
            }
        } else {
            for (int g = 0; g < count; g += 4) {
                load_group<OT>(wb0, wnext, w.lane);
                load_group<OT>(wb1, wnext + kGroupStride, w.lane);
                x2 = xa[128];
                x3 = xa[192];
                if (save_0) FFN_SAVE(g, x0);
                if (save_1) FFN_SAVE(g + 1, x1);
                mma_group<OT>(acc, wa0, x0);
                mma_group<OT>(acc, wa1, x1);
                gnext = gnext + 2 < glast ? gnext + 2 : glast;      // clamp at the end of the panel
                wnext = wp + (int64_t)gnext * kGroupStride;
                load_group<OT>(wa0, wnext, w.lane);
                load_group<OT>(wa1, wnext + kGroupStride, w.lane);
                xa += 256;
                if (g + 4 < count) {
                    x0 = xa[0];
                    x1 = xa[64];
                }
                if (save_2) FFN_SAVE(g + 2, x2);
                if (save_3) FFN_SAVE(g + 3, x3);
                mma_group<OT>(acc, wb0, x2);
                mma_group<OT>(acc, wb1, x3);
                gnext = gnext + 2 < glast ? gnext + 2 : glast;
                wnext = wp + (int64_t)gnext * kGroupStride;
                pipeline_plain<OT>();
                pipeline_plain<OT>();
            }
        }
#undef FFN_SAVE
    }

    // ---- backward: the d_logits columns are one more (register) K group --------------
    if (MODE == kBackward && GX > 0) {
        // d_logits columns [lg_col, lg_col+lg_n) are K channels 0..lg_n-1: group 0, h == 0
        f32x4 d = w.dl;
        f32x4 sel;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) v = (c == L.lg_col + p && p < L.lg_n) ? d[c] : v;
            sel[p] = w.h == 0 ? v : 0.0f;
        }
        mma_group<OT>(acc, wa0, sel);   // group GA; the padding groups are all zero
    }

    // ---- next step's first K pair is fetched from inside the epilogue loop (two loads per
    // output tile), so that their issue slots sit between the epilogue's VALU work instead of
    // in front of it (a lone wave pays ~64 cycles per L2 load it cannot overlap)
    const int ot_next = next != nullptr ? next->out_tiles / TW : 0;
    const f32x4* wn = next != nullptr
        ? reinterpret_cast<const f32x4*>(packed_w + next->w_off) + half * ot_next * 64 + w.lane
        : nullptr;

    // ---- epilogue: bias / activation / mask, hand-off -------------------------------
    f32x4* save_out = nullptr;
    if (MODE != kInfer && L.save_out_slot >= 0 && w.active) save_out = slab_block(ch, slab_out, L.save_out_slot, w);
    // fused logits head: this lane holds channels 8*group + 4h + p of its sample; their
    // products with the head's rows accumulate per lane, the partial sums meet at the end
    const bool fused_head = FLAV == 0 ? (MODE != kBackward && L.head_off >= 0)
                                      : (MODE != kBackward && FLAV >= 2);
    const bool to_slab = FLAV == 0 ? L.dst == 0 : true;
    const bool save_y = FLAV == 0 ? save_out != nullptr : FLAV == 2;
    const float* head_base = BIG ? w.bias_glb : w.bias_lds;
    const float* hw = head_base + (fused_head ? L.head_off : 0) + 4 + 16 * w.h;
    if (fused_head && w.h == 0 && half == 0) {
        const f32x4 hb = *reinterpret_cast<const f32x4*>(head_base + L.head_off);
#pragma unroll
        for (int c = 0; c < 4; ++c) w.logit[c] += hb[c];
    }
    if (WIDE) team_barrier();        // every K loop of this step has finished reading the slab
    // Wide mode: the epilogue's slab / save addresses are lane constants, and with the wide
    // kernels' register budget hipcc computes them once per kernel, SPILLS them (106 dwords in the
    // training forward) and reloads them here -- scratch reloads that wait, through the in-order
    // vmcnt, for every global store issued before them.  Opaque lane terms keep them
    // two-instruction recomputations.
    int e_lane = w.lane, e_h = w.h, e_s = w.s;
    if (WIDE) asm volatile("" : "+v"(e_lane), "+v"(e_h), "+v"(e_s));
    unsigned sign_bits[4] = {0u, 0u, 0u, 0u};
    // ReLU as a signed-integer max on the bit pattern: floats below +0 (and -0) are negative
    // integers, so max(bits, 0) is relu(t) in ONE instruction -- fmaxf costs three here (IEEE
    // canonicalisation of the input, the max, a select on the runtime relu flag); a step
    // without ReLU uses the floor INT_MIN (identity).
    const int relu_floor = L.relu ? 0 : (int)0x80000000;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        if (o < ot_next) {               // (uniform) the next step may have more tiles than this one
            pre[o] = wn[o * 64];
            pre[8 + o] = wn[(int64_t)(TW * ot_next + o) * 64];
        }
        if (o >= OT) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int group = 4 * (OT * half + o) + q;   // K group of the next step
            f32x4 y;
            if (MODE == kBackward) {
                const unsigned word = o < 2 ? mbits.x : (o < 4 ? mbits.y : (o < 6 ? mbits.z : mbits.w));
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    // (a word that received a single tile -- OT == 1 -- was shifted 16 times only)
                    const int filled = ((OT & 1) && o == OT - 1) ? 16 : 32;
                    const int bit = filled - 1 - (16 * (o & 1) + 4 * q + p);
                    // all-ones / zero from the mask bit (one v_bfe_i32), then one v_and
                    const int keep = ((int)(word << (31 - bit))) >> 31;
                    const float a = acc[o][4 * q + p];     // (a scalar copy: bit_cast of a vector
                    y[p] = __builtin_bit_cast(float, __builtin_bit_cast(int, a) & keep);   // element reads lane 0)
                }
                w.act[group * 64 + e_lane] = y;
                if (save_y) __builtin_nontemporal_store(y, &save_out[saved_index(2 * group + e_h, e_s)]);
            } else {
                // fused head: this quad's four weight rows are requested BEFORE its ReLU / sign-bit
                // work and its slab write (which the compiler must keep them ordered against), so
                // that work covers the LDS latency instead of a wait in front of the FMAs
                f32x4 hw4[4];
                if (fused_head) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) hw4[p] = *reinterpret_cast<const f32x4*>(hw + group * 32 + p * 4);
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float t = acc[o][4 * q + p];
                    // sign bit of (0 - t) is set exactly when t > 0 (0 - (+-0) = +0); shifting it in
                    // from the right is one v_alignbit: value (o&1, q, p) ends at bit 31 - (16(o&1)+4q+p)
                    if (MODE == kTrainFwd)
                        sign_bits[o >> 1] = __builtin_amdgcn_alignbit(sign_bits[o >> 1],
                                                                      __builtin_bit_cast(unsigned, 0.0f - t), 31);
                    y[p] = __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, t), relu_floor));
                }
                if (to_slab) {
                    w.act[group * 64 + e_lane] = y;
                    if (fused_head) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) w.logit[c] = __builtin_fmaf(y[p], hw4[p][c], w.logit[c]);
                        }
                        if (MODE == kTrainFwd && save_y) __builtin_nontemporal_store(y, &save_out[saved_index(2 * group + e_h, e_s)]);
                    }
                } else if (!WIDE && o == 0 && q == 0) {
                    // real outputs = rows 0..out_n-1 of tile 0 = registers 0..3 of h == 0
#pragma unroll
                    for (int c = 0; c < 4; ++c) {   // static register index, runtime select
                        const int p = c - L.out_col;
                        float val = w.logit[c];
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) val = (p == pp && pp < L.out_n) ? y[pp] : val;
                        w.logit[c] = val;
                    }
                }
            }
        }
    }
    if (MODE == kTrainFwd && L.relu && L.mask_slot >= 0 && w.active)
        w.masks[mask_at] = make_uint4(sign_bits[0], sign_bits[1], sign_bits[2], sign_bits[3]);
    if (WIDE) team_barrier();        // the step's output is in the slab
}

template <int MODE, int TWP, bool BIG = false>
__device__ __forceinline__ void run_chain(const ffn_mlp_chain& ch, WaveCtx& w,
                                          const float* __restrict__ packed_w,
                                          float* __restrict__ slab_out) {
    constexpr bool WIDE = TWP > 1;
    constexpr int TW = TWP;
    f32x4 pre[16];
    {
        const ffn_step& first = ch.step[0];
        const int ot = first.out_tiles / TW;
        const f32x4* wp = reinterpret_cast<const f32x4*>(packed_w + first.w_off) + (WIDE ? w.half : 0) * ot * 64 + w.lane;
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (o < ot) {
                pre[o] = wp[o * 64];
                pre[8 + o] = wp[(int64_t)(TW * ot + o) * 64];
            }
    }
    for (int li = 0; li < ch.num_steps; ++li) {
        const ffn_step& L = ch.step[li];
        const ffn_step* next = li + 1 < ch.num_steps ? &ch.step[li + 1] : nullptr;
        const int ot = L.out_tiles / TW;
        if ((TWP < 4 || BIG) && ot == 8) {
            // the 256-channel steps (all the time of every supported model) get a specialised
            // epilogue; the flavour is a property of the step, known before its K loops start.
            // (Only as many specialisations as the register allocator digests: every inlined
            // copy of run_step shares the kernel's 512 registers.)
            const bool saves = MODE != kInfer && L.save_out_slot >= 0 && w.active;
            const bool head = MODE != kBackward && L.head_off >= 0;
            if (MODE == kBackward) {
                if (saves) run_step<8, MODE, TWP, 2, BIG>(ch, L, next, w, packed_w, pre, slab_out);
                else run_step<8, MODE, TWP, 1, BIG>(ch, L, next, w, packed_w, pre, slab_out);
            } else if (!head && L.dst == 0) run_step<8, MODE, TWP, 1, BIG>(ch, L, next, w, packed_w, pre, slab_out);
            else run_step<8, MODE, TWP, 0, BIG>(ch, L, next, w, packed_w, pre, slab_out);
        } else if ((TWP < 4 || BIG) && ot == 4) run_step<4, MODE, TWP, 0, BIG>(ch, L, next, w, packed_w, pre, slab_out);
        else if (ot == 2) run_step<2, MODE, TWP, 0, BIG>(ch, L, next, w, packed_w, pre, slab_out);
        else run_step<1, MODE, TWP, 0, BIG>(ch, L, next, w, packed_w, pre, slab_out);
    }
}

// Persistent launch: one workgroup per CU; its four waves walk the 32-sample blocks
// independently (block = first, first + stride, ...) -- no workgroup turnover, no tail of
// SIMDs idling until the slowest sibling wave retires.  Wide mode: two pairs of waves per
// workgroup, every pair runs the same number of passes (the barriers are workgroup-wide);
// a pair past the end re-runs the last block with its stores switched off.
constexpr int kTeamScratchBytes = 3072;    // team modes: partial logits of a team's waves 1.. (pairs: 2 x 1 KiB; a quad: 3 KiB)

template <int TWP>
__device__ __forceinline__ void wave_setup(WaveCtx& w, char* smem, int64_t n, int64_t& stride) {
    constexpr bool WIDE = TWP > 1;
    w.lane = threadIdx.x & 63;
    w.h = w.lane >> 5;
    w.s = w.lane & 31;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int teams = kWavesPerBlock / TWP;
    const int team = wave_in_block / TWP;
    w.half = WIDE ? (wave_in_block % TWP) : 0;
    stride = (int64_t)gridDim.x * teams;
    w.act = lds_slab(smem + team * (kActBytesPerWave * kWavesPerBlock / teams));
    w.enc_table = reinterpret_cast<const float*>(smem + kWavesPerBlock * kActBytesPerWave);
    w.bias_lds = reinterpret_cast<const float*>(smem + kWavesPerBlock * kActBytesPerWave + kEncTableBytes);
    w.num_blocks = (n + kSamplesPerWave - 1) / kSamplesPerWave;
    w.block = (int64_t)blockIdx.x * teams + team;
    w.slab_block0 = 0;
    w.slab_blocks = w.num_blocks;
    w.active = true;
}

template <int MODE, int TWP, bool BIG = false>
__global__ void __launch_bounds__(256, 1)
mlp_forward_kernel(const ffn_mlp_chain ch, const float* __restrict__ packed_w,
                   const float* __restrict__ bias, const float* __restrict__ positions,
                   const float* __restrict__ views, int64_t n, float* __restrict__ logits,
                   float* __restrict__ saved, uint32_t* __restrict__ masks, int64_t slab_block0,
                   int64_t slab_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stage_encoding_tables(ch.enc, reinterpret_cast<float*>(smem + kWavesPerBlock * kActBytesPerWave),
                          threadIdx.x, 256);
    {
        float* bl = reinterpret_cast<float*>(smem + kWavesPerBlock * kActBytesPerWave + kEncTableBytes);
        const int staged = ch.bias_floats < kBiasLdsFloats ? ch.bias_floats : kBiasLdsFloats;
        for (int i = threadIdx.x; i < staged; i += 256) bl[i] = bias[i];
    }
    __syncthreads();                  // narrow mode: the only barrier of the kernel
    constexpr bool WIDE = TWP > 1;
    WaveCtx w;
    int64_t stride;
    wave_setup<TWP>(w, smem, n, stride);
    w.bias_glb = bias;
    if (slab_blocks > 0) { w.slab_block0 = slab_block0; w.slab_blocks = slab_blocks; }
    w.masks = reinterpret_cast<uint4*>(masks);
    // (team scratch: the partial logits of a team's waves 1 .. TWP-1, 1 KiB each)
    f32x4* scratch = reinterpret_cast<f32x4*>(smem + kWavesPerBlock * kActBytesPerWave + kEncTableBytes +
                                              kBiasLdsFloats * 4) + ((threadIdx.x >> 6) / TWP) * (TWP - 1) * 64;
    const int64_t passes = WIDE ? (w.num_blocks + stride - 1) / stride : 0;
    const int64_t first = w.block;
    // the inputs of a block are requested one block ahead: a lone wave has nothing else to
    // cover a ~2k-cycle HBM latency in front of the first feature burst
    float in_next[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto request_inputs = [&](int64_t block) {
        block = block < w.num_blocks ? block : w.num_blocks - 1;
        const int64_t sample = block * kSamplesPerWave + w.s;
        const int64_t src = sample < n ? sample : n - 1;  // tail lanes recompute the last sample
        in_next[0] = positions[src * 3 + 0]; in_next[1] = positions[src * 3 + 1]; in_next[2] = positions[src * 3 + 2];
        if (views != nullptr) {
            in_next[3] = views[src * 3 + 0]; in_next[4] = views[src * 3 + 1]; in_next[5] = views[src * 3 + 2];
        }
    };
    request_inputs(first);
    for (int64_t pass = 0; WIDE ? pass < passes : w.block < w.num_blocks; ++pass) {
        if (WIDE) {
            w.block = first + pass * stride;
            w.active = w.block < w.num_blocks;
            if (!w.active) w.block = w.num_blocks - 1;
        }
        const int64_t sample = w.block * kSamplesPerWave + w.s;
        w.x0 = in_next[0]; w.x1 = in_next[1]; w.x2 = in_next[2];
        w.v0 = in_next[3]; w.v1 = in_next[4]; w.v2 = in_next[5];
        request_inputs(first + (pass + 1) * stride);
        w.logit[0] = w.logit[1] = w.logit[2] = w.logit[3] = 0.0f;
        run_chain<MODE, TWP, BIG>(ch, w, packed_w, saved);
        f32x4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c)   // MFMA heads leave their rows on h == 0, fused heads on both halves
            out[c] = w.logit[c] + __shfl_xor(w.logit[c], 32);
        if (WIDE) {                   // the waves of a team hold partial sums: added in wave order
            if (w.half > 0) scratch[(w.half - 1) * 64 + w.lane] = out;
            team_barrier();
            if (w.half == 0) {
#pragma unroll
                for (int k = 1; k < TWP; ++k) out += scratch[(k - 1) * 64 + w.lane];
            }
        }
        if (w.h == 0 && w.half == 0 && w.active && sample < n) {
            reinterpret_cast<f32x4*>(logits)[sample] = out;
        }
        if (!WIDE) w.block += stride;
    }
}

// ---------------------------------------------------------------------------------- fused render
// Inference in ONE launch: t-sampling -> sample positions -> Fourier-feature MLP -> sigmoid /
// softplus -> front-to-back compositing (K2 + K3 + K4 + K5; ray_caster.py:103-159 renders a
// frame as sample / model / composite passes over HBM-resident (R,S,3) and (R,S,4) arrays).
// A wavefront owns whole rays.  A ray's samples go through the chain interpreter in blocks of
// 32 (the forward kernel's unit); the logits of two consecutive blocks land on the two lane
// halves, which is exactly the "sample s on lane s & 63" row layout of the composite scan, so
// the scan consumes them straight from registers.  HBM traffic per ray: 8 B ray id + 32 B ray
// state in, 20 B (or 3 B of u8 pixel) out -- samples, features and logits never leave the CU.
//
// Optional empty-space skipping (occupancy grid, K9 semantics: a sample in an empty cell has
// sigma = 0, so its weight is 0 and its transmittance factor min(1, 1 + 1e-10) = 1): the wave
// first compacts the ray's occupied samples (ballot + popcount ranks, a 512-B slot list in
// LDS), runs only ceil(m / 32) blocks, and composites over the kept samples with each one's
// OWN delta = t[j+1] - t[j] -- the same colour / alpha as evaluating every sample.
struct RenderParams {
    const float* starts; const float* dirs; const float* near_far; int64_t total_rays;
    const int64_t* ray_index; int64_t ray_base; const uint8_t* valid; int num_rays; int S;
    const float* unit; const float* t_values;
    const uint32_t* occ_bits; GridMap map;
    float* color; float* alpha; float* depth; int32_t* nan_flag;
    uint8_t* image; int64_t pixel_offset;
};

// 512-wide chains (WIDE): a PAIR of waves owns a ray (each computes half of every step's output
// channels, like the wide forward kernel).  The chain's barriers are workgroup-wide, so the two
// pairs of a workgroup run in step: they take NEIGHBOURING rays (validity and occupancy of
// adjacent pixels nearly always agree, so little is wasted), agree on max(blocks) per round and a
// pair with fewer blocks -- or no ray -- runs the remaining passes on a dummy sample with its
// results ignored.  The even wave of a pair adds the odd wave's partial logits and composites.
constexpr int kRenderScratchBytes = 4096;   // [0,2K) partial logits (wide) | slot lists; [2K,3K) wide slot lists; [3K,..) block counts

template <bool WIDE>
__global__ void __launch_bounds__(256, 1)
render_fused_kernel(const ffn_mlp_chain ch, const float* __restrict__ packed_w,
                    const float* __restrict__ bias, const RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stage_encoding_tables(ch.enc, reinterpret_cast<float*>(smem + kWavesPerBlock * kActBytesPerWave),
                          threadIdx.x, 256);
    {
        float* bl = reinterpret_cast<float*>(smem + kWavesPerBlock * kActBytesPerWave + kEncTableBytes);
        const int staged = ch.bias_floats < kBiasLdsFloats ? ch.bias_floats : kBiasLdsFloats;
        for (int i = threadIdx.x; i < staged; i += 256) bl[i] = bias[i];
    }
    __syncthreads();
    WaveCtx w;
    int64_t stride;
    wave_setup<(WIDE ? 2 : 1)>(w, smem, kSamplesPerWave, stride);
    w.bias_glb = bias;
    w.block = 0;                       // nothing is saved in inference: slab addressing is unused
    w.masks = nullptr;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int kTeams = WIDE ? 2 : kWavesPerBlock;
    const int team = WIDE ? wave_in_block >> 1 : wave_in_block;
    char* scratch = smem + kWavesPerBlock * kActBytesPerWave + kEncTableBytes + kBiasLdsFloats * 4;
    f32x4* partial = reinterpret_cast<f32x4*>(scratch) + team * 64;
    uint16_t* slots = reinterpret_cast<uint16_t*>(scratch + (WIDE ? 2048 : 0)) + team * 256;
    volatile int* team_blocks = reinterpret_cast<volatile int*>(scratch + 3072);
    const int S = p.S;
    const int rays_per_round = gridDim.x * kTeams;
    const int rounds = (p.num_rays + rays_per_round - 1) / rays_per_round;
    for (int round = 0; round < rounds; ++round) {
        const int r = WIDE ? (round * (int)gridDim.x + (int)blockIdx.x) * 2 + team
                           : (int)blockIdx.x * kWavesPerBlock + team + round * rays_per_round;
        bool have = r < p.num_rays;
        const int rr = have ? r : 0;       // a pair without a ray walks ray 0's addresses
        const int64_t ray = p.ray_index != nullptr ? p.ray_index[rr] : p.ray_base + rr;
        if (have && p.valid != nullptr && p.valid[ray] == 0) {      // misses the volume: black, no pixel
            if (w.lane == 0 && w.half == 0) {
                if (p.color != nullptr) {
                    p.color[(int64_t)r * 3 + 0] = 0.f; p.color[(int64_t)r * 3 + 1] = 0.f;
                    p.color[(int64_t)r * 3 + 2] = 0.f;
                }
                if (p.alpha != nullptr) p.alpha[r] = 0.f;
                if (p.depth != nullptr) p.depth[r] = 0.f;
            }
            have = false;
        }
        if (!WIDE && !have) continue;
        const float sx = p.starts[ray * 3 + 0], sy = p.starts[ray * 3 + 1], sz = p.starts[ray * 3 + 2];
        const float dx = p.dirs[ray * 3 + 0], dy = p.dirs[ray * 3 + 1], dz = p.dirs[ray * 3 + 2];
        const float near = p.near_far[ray], far = p.near_far[p.total_rays + ray];
        const float span = sub_rn(far, near);
        // t of sample j: given, or near + linspace(0,1,S)[j] * (far - near) with separately
        // rounded multiply and add like the sampling kernel (bit-identical t and positions)
        const float* trow = p.t_values != nullptr ? p.t_values + (int64_t)rr * S : nullptr;
        auto t_of = [&](int j) -> float {
            return trow != nullptr ? trow[j] : mul_add_rn(p.unit[j], span, near);
        };
        int m = have ? S : 0;
        if (have && p.occ_bits != nullptr) {
            m = 0;
            for (int base = 0; base < S; base += 64) {
                const int j = base + w.lane;
                const bool in_ray = j < S;
                const float t = t_of(in_ray ? j : S - 1);
                const float px = mul_add_rn(t, dx, sx), py = mul_add_rn(t, dy, sy), pz = mul_add_rn(t, dz, sz);
                const bool keep = in_ray && occupied_at(p.map, p.occ_bits, px, py, pz);
                const uint64_t mask = __ballot(keep);
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (keep) slots[m + rank] = (uint16_t)j;   // wide: both waves of the pair write the same list
                m += __popcll(mask);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        RayAccum acc;
        acc.reset();
        const int nblk = (m + 31) >> 5;
        int passes = nblk;
        if (WIDE) {                        // both pairs run max(blocks) passes
            if (w.lane == 0 && w.half == 0) team_blocks[team] = nblk;
            team_barrier();
            const int other = team_blocks[team ^ 1];
            team_barrier();                // the next round's count may not overtake this read
            passes = __builtin_amdgcn_readfirstlane(other > nblk ? other : nblk);
        }
        for (int row = 0; 2 * row < passes; ++row) {
            float4 lg = make_float4(0.f, 0.f, 0.f, 0.f);
            float tj = 0.0f;
            int jj = 0;
            bool act = false;
            for (int hb = 0; hb < 2; ++hb) {
                const int k = 2 * row + hb;
                if (k >= passes) break;
                const bool mine = k < nblk;                   // else: a pass for the neighbour's sake
                const int slot = 32 * k + w.s;
                const bool valid = mine && slot < m;
                const int sl = valid ? slot : (m > 0 ? m - 1 : 0);   // tail lanes recompute the last sample
                const int j = !mine ? 0 : (p.occ_bits != nullptr ? (int)slots[sl] : sl);
                const float t = t_of(j);
                w.x0 = mul_add_rn(t, dx, sx);
                w.x1 = mul_add_rn(t, dy, sy);
                w.x2 = mul_add_rn(t, dz, sz);
                w.v0 = dx; w.v1 = dy; w.v2 = dz;
                w.logit[0] = w.logit[1] = w.logit[2] = w.logit[3] = 0.0f;
                run_chain<kInfer, (WIDE ? 2 : 1)>(ch, w, packed_w, nullptr);
                f32x4 out;
#pragma unroll
                for (int c = 0; c < 4; ++c) out[c] = w.logit[c] + __shfl_xor(w.logit[c], 32);
                if (WIDE) {               // the two waves of a pair hold partial sums
                    if (w.half == 1) partial[w.lane] = out;
                    team_barrier();
                    if (w.half == 0) out += partial[w.lane];
                }
                if (w.h == hb && mine) {
                    lg = make_float4(out[0], out[1], out[2], out[3]);
                    tj = t; jj = j; act = valid;
                }
            }
            if (2 * row >= nblk || w.half != 0) continue;
            const bool last = jj == S - 1;
            const float tnext = (act && !last) ? t_of(jj + 1) : 0.0f;
            const SampleTerms q = make_terms(lg, tj, tnext, last, act, p.nan_flag);
            acc.row(q, w.lane, jj, act && !last);
        }
        if (!have || w.half != 0) continue;
        acc.finish();
        if (w.lane == 0) {
            if (p.color != nullptr) {
                p.color[(int64_t)r * 3 + 0] = acc.cr; p.color[(int64_t)r * 3 + 1] = acc.cg;
                p.color[(int64_t)r * 3 + 2] = acc.cb;
            }
            if (p.alpha != nullptr) p.alpha[r] = acc.asum;
            if (p.depth != nullptr) p.depth[r] = t_of(acc.depth_pick(S));
            if (p.image != nullptr) {      // (x * 255) truncated to u8 (ray_sampler.py:193-196)
                uint8_t* px = p.image + (ray - p.pixel_offset) * 3;
                px[0] = (uint8_t)(int)(acc.cr * 255.0f);
                px[1] = (uint8_t)(int)(acc.cg * 255.0f);
                px[2] = (uint8_t)(int)(acc.cb * 255.0f);
            }
        }
    }
}

// ---------------------------------------------------------------------------------- fused focus
// Opacity-guided sampling with a LIVE coarse model in one launch (ray_sampler.py:234-357 and
// :388-392, which the reference runs as a start-up pass over every ray into a (rays, S_f - 1)
// table): per ray, the wave evaluates the coarse model on the S_f <= 64 probe points
// t = linspace(near, far, S_f) through the chain interpreter (two 32-sample blocks whose sigma
// logits land on the two lane halves = one row of the scan), builds the blend-weight CDF in
// registers and its own -- now idle -- LDS slab, draws the S_f inverse-transform samples, merges
// them with the uniform half already in t_io and writes the sorted row.  Probe positions, logits,
// opacities and CDF rows never exist in HBM.  Arithmetic = K2a/K2c/K2d's (focus_terms.h), so
// the t-values are the table path's bit for bit.
struct FocusParams {
    const float* starts; const float* dirs; const float* near_far; int64_t total_rays;
    const int64_t* ray_index; int num_rays; int S; int n_focus;
    const float* unit_focus; const float* u; float* t_io;
};

// 512-wide opacity models (WIDE): a PAIR of waves owns a ray, like the fused render -- each wave
// computes half of every step's output channels, the even wave adds the odd wave's partial logits
// and does the CDF / sampling / merge in the pair's slab.  The chain's barriers are workgroup-wide:
// the two pairs run in step (every ray takes the same number of passes here), a pair without a
// ray walks ray 0's probe points with its stores switched off, and one more barrier per ray keeps
// the odd wave's next feature burst out of the slab the even wave is still merging in.
template <bool WIDE>
__global__ void __launch_bounds__(256, 1)
focus_fused_kernel(const ffn_mlp_chain ch, const float* __restrict__ packed_w,
                   const float* __restrict__ bias, const FocusParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stage_encoding_tables(ch.enc, reinterpret_cast<float*>(smem + kWavesPerBlock * kActBytesPerWave),
                          threadIdx.x, 256);
    {
        float* bl = reinterpret_cast<float*>(smem + kWavesPerBlock * kActBytesPerWave + kEncTableBytes);
        const int staged = ch.bias_floats < kBiasLdsFloats ? ch.bias_floats : kBiasLdsFloats;
        for (int i = threadIdx.x; i < staged; i += 256) bl[i] = bias[i];
    }
    __syncthreads();
    WaveCtx w;
    int64_t stride;
    wave_setup<(WIDE ? 2 : 1)>(w, smem, kSamplesPerWave, stride);
    w.bias_glb = bias;
    w.block = 0;
    w.masks = nullptr;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int kTeams = WIDE ? 2 : kWavesPerBlock;
    const int team = WIDE ? wave_in_block >> 1 : wave_in_block;
    f32x4* partial = reinterpret_cast<f32x4*>(smem + kWavesPerBlock * kActBytesPerWave + kEncTableBytes +
                                              kBiasLdsFloats * 4) + team * 64;
    const int teams = gridDim.x * kTeams;
    const int n = p.n_focus;
    const int nblk = (n + 31) >> 5;
    const int rounds = (p.num_rays + teams - 1) / teams;
    for (int round = 0; round < rounds; ++round) {
        const int r0 = round * teams + (int)blockIdx.x * kTeams + team;
        const bool have = r0 < p.num_rays;
        if (!WIDE && !have) continue;
        const int r = have ? r0 : 0;          // (a pair without a ray keeps the barriers company)
        const int64_t ray = p.ray_index[r];
        const float sx = p.starts[ray * 3 + 0], sy = p.starts[ray * 3 + 1], sz = p.starts[ray * 3 + 2];
        const float dx = p.dirs[ray * 3 + 0], dy = p.dirs[ray * 3 + 1], dz = p.dirs[ray * 3 + 2];
        const float near = p.near_far[ray], far = p.near_far[p.total_rays + ray];
        const float span = sub_rn(far, near);
        auto t_of = [&](int j) -> float { return mul_add_rn(p.unit_focus[j], span, near); };
        float sigma_logit = 0.0f;
        for (int hb = 0; hb < 2; ++hb) {
            if (hb >= nblk) break;
            const int slot = 32 * hb + w.s;
            const int j = slot < n ? slot : n - 1;
            const float t = t_of(j);
            w.x0 = mul_add_rn(t, dx, sx);
            w.x1 = mul_add_rn(t, dy, sy);
            w.x2 = mul_add_rn(t, dz, sz);
            w.v0 = dx; w.v1 = dy; w.v2 = dz;
            w.logit[0] = w.logit[1] = w.logit[2] = w.logit[3] = 0.0f;
            run_chain<kInfer, (WIDE ? 2 : 1)>(ch, w, packed_w, nullptr);
            float out = w.logit[3] + __shfl_xor(w.logit[3], 32);
            if (WIDE) {                   // the two waves of a pair hold partial sums
                f32x4 mine;
                mine[0] = mine[1] = mine[2] = 0.0f;
                mine[3] = out;
                if (w.half == 1) partial[w.lane] = mine;
                team_barrier();
                if (w.half == 0) out += partial[w.lane][3];
            }
            if (w.h == hb) sigma_logit = out;
        }
        if (WIDE && (w.half != 0 || !have)) {
            team_barrier();               // (matches the barrier behind the even wave's merge)
            continue;
        }
        // probe sample s sits on lane s
        float sigma[1], delta[1];
        sigma[0] = w.lane < n ? softplus_probe(sigma_logit) : 0.0f;
        delta[0] = w.lane < n - 1 ? sub_rn(t_of(w.lane + 1), t_of(w.lane)) : 0.0f;
        float* c = (float*)w.act;                         // the wave's slab is idle between rays
        float* tv = c + 256;
        cdf_of_probe<1>(sigma, delta, n, w.lane, c);
        float* row = p.t_io + (int64_t)r * p.S;
        for (int i = w.lane; i < p.S - n; i += 64) tv[i] = row[i];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        focus_merge_ray(near, span, c, tv, p.u + (int64_t)r * n, p.unit_focus, p.S, n, w.lane, row);
        if (WIDE) team_barrier();         // the slab is free for the next ray's features
    }
}

// Backward-data chain: consumes d_logits (N,4) and the saved forward activations, writes
// dZ of every hidden layer (block layout) for the weight-gradient kernel.
template <int TWP, bool BIG = false>
__global__ void __launch_bounds__(256, 1)
mlp_backward_data_kernel(const ffn_mlp_chain ch, const float* __restrict__ packed_wt,
                         const float* __restrict__ d_logits, int64_t n,
                         uint32_t* __restrict__ masks, float* __restrict__ dz, int64_t slab_block0,
                         int64_t slab_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool WIDE = TWP > 1;
    WaveCtx w;
    int64_t stride;
    wave_setup<TWP>(w, smem, n, stride);
    w.bias_glb = nullptr;              // (a backward chain has no biases)
    if (slab_blocks > 0) { w.slab_block0 = slab_block0; w.slab_blocks = slab_blocks; }
    w.masks = reinterpret_cast<uint4*>(masks);
    w.x0 = w.x1 = w.x2 = w.v0 = w.v1 = w.v2 = 0.0f;
    const int64_t passes = WIDE ? (w.num_blocks + stride - 1) / stride : 0;
    const int64_t first = w.block;
    // d_logits of a block are requested one block ahead (see the forward kernel)
    f32x4 dl_next;
    auto request_dl = [&](int64_t block) {
        block = block < w.num_blocks ? block : w.num_blocks - 1;
        const int64_t sample = block * kSamplesPerWave + w.s;
        const f32x4 v = reinterpret_cast<const f32x4*>(d_logits)[sample < n ? sample : n - 1];
        f32x4 zero; zero[0] = zero[1] = zero[2] = zero[3] = 0.0f;
        dl_next = sample < n ? v : zero;
    };
    request_dl(first);
    for (int64_t pass = 0; WIDE ? pass < passes : w.block < w.num_blocks; ++pass) {
        if (WIDE) {
            w.block = first + pass * stride;
            w.active = w.block < w.num_blocks;
            if (!w.active) w.block = w.num_blocks - 1;
        }
        w.dl = dl_next;
        request_dl(first + (pass + 1) * stride);
        run_chain<kBackward, TWP, BIG>(ch, w, packed_wt, dz);
        if (!WIDE) w.block += stride;
    }
}

}  // namespace ffn

using namespace ffn;

extern "C" int ffn_mlp_pack(const float* src, int rows, int cols, int ld, int transpose,
                            const int32_t* row_map, const int32_t* col_map, int groups, int tiles,
                            float* dst, void* stream) {
    if (groups <= 0 || tiles <= 0) return fail_arg("ffn_mlp_pack: shape");
    const int64_t total = (int64_t)groups * tiles * 256;
    int64_t grid = (total + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, src, rows,
                       cols, ld, transpose, row_map, col_map, groups, tiles, dst);
    return check_launch("ffn_mlp_pack");
}

extern "C" int ffn_mlp_pack_jobs(const ffn_pack_job* jobs, int num_jobs, void* stream) {
    if (num_jobs == 0) return 0;
    if (num_jobs < 0 || jobs == nullptr) return fail_arg("ffn_mlp_pack_jobs: no jobs");
    hipLaunchKernelGGL(pack_jobs_kernel, dim3(64, num_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return check_launch("ffn_mlp_pack_jobs");
}

static int validate_chain(const ffn_mlp_chain* ch, bool backward, bool train = false) {
    if (ch == nullptr || ch->num_steps < 1 || ch->num_steps > FFN_MAX_STEPS) return 1;
    if (ch->bias_floats < 0) return 1;
    const bool wide = ch->wide != 0;
    const bool quad = ch->wide == 2;       // four waves per block: narrow chains of >= 128 channels
    const bool big = ch->wide == 3;        // four waves per block, up to 1024 channels per layer
    if (ch->wide < 0 || ch->wide > 3) return 1;
    for (int i = 0; i < ch->num_steps; ++i) {
        const ffn_step& L = ch->step[i];
        const int ot = L.out_tiles;
        if (quad && (!(ot == 4 || ot == 8) || L.act_groups > 32 || L.dst != 0)) return 1;
        if (big ? !(ot == 4 || ot == 8 || ot == 16 || ot == 32)
                : (wide ? !(ot == 2 || ot == 4 || ot == 8 || ot == 16) : !(ot == 1 || ot == 2 || ot == 4 || ot == 8))) return 1;
        if (L.act_groups < 0 || L.aux_groups < 0 || L.act_groups > (big ? 128 : (wide ? 64 : 32))) return 1;
        if ((L.act_groups & 3) || (L.aux_groups & 3) || L.act_groups + L.aux_groups == 0) return 1;
        if (!backward && L.aux_groups > 0 && (L.enc_id < 0 || L.enc_id > 1)) return 1;
        // fused-head blocks (4 bias floats + 4 per channel) are read from the LDS copy only
        if (!backward && L.head_off >= 0 &&
            L.head_off + 4 + 128 * ot > (big ? ch->bias_floats : kBiasLdsFloats)) return 1;    // (big: from L2)
        if (!backward && (L.b_off < 0 || L.b_off + 32 * ot > ch->bias_floats)) return 1;
        if (!backward && (ch->enc[0].num_freq > 256 || ch->enc[1].num_freq > 256)) return 1;
        if (backward && L.aux_groups != 0 && L.aux_groups != 4) return 1;
        if (backward && L.aux_groups && (L.lg_n < 1 || L.lg_col < 0 || L.lg_col + L.lg_n > 4)) return 1;
        if (!backward && L.dst == 1 &&
            (wide || L.out_n < 1 || L.out_n > 4 || L.out_col < 0 || L.out_col + L.out_n > 4))
            return 1;
        // the narrow training forward saves from every K-loop trip
        if (train && !wide && ((L.act_groups > 0 && L.save_in_slot < 0) || (L.aux_groups > 0 && L.save_enc_slot < 0)))
            return 1;
    }
    return 0;
}

static const size_t kLdsBytes = (size_t)kWavesPerBlock * kActBytesPerWave + kEncTableBytes +
                                kBiasLdsFloats * 4 + kTeamScratchBytes;

// one resident workgroup per CU (its ~150 KiB of LDS admit no second one)
static int64_t persistent_grid(int64_t blocks32, int teams) {
    static int cu_count[64] = {0};          // per device ordinal (an immutable attribute cache)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = cu_count[dev];
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        cu_count[dev] = cus;
    }
    const int64_t wgs = (blocks32 + teams - 1) / teams;
    return wgs < cus ? wgs : cus;
}

static const size_t kRenderLdsBytes = kLdsBytes - kTeamScratchBytes + kRenderScratchBytes;

template <typename K>
static void allow_big_lds(K kernel, size_t bytes = kLdsBytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int MODE, int TWP, bool BIG = false>
static void launch_forward(const ffn_mlp_chain* chain, const float* packed_w, const float* bias,
                           const float* positions, const float* views, int64_t n, float* logits,
                           float* saved, uint32_t* masks, int64_t slab_block0, int64_t slab_blocks,
                           void* stream) {
    const int64_t blocks32 = (n + kSamplesPerWave - 1) / kSamplesPerWave;
    const int64_t grid = persistent_grid(blocks32, kWavesPerBlock / TWP);
    allow_big_lds(&mlp_forward_kernel<MODE, TWP, BIG>);
    hipLaunchKernelGGL((mlp_forward_kernel<MODE, TWP, BIG>), dim3((unsigned)grid), dim3(256), kLdsBytes,
                       (hipStream_t)stream, *chain, packed_w, bias, positions, views, n, logits, saved,
                       masks, slab_block0, slab_blocks);
}

extern "C" int ffn_mlp_forward(const ffn_mlp_chain* chain, const float* packed_w,
                               const float* bias, const float* positions, const float* views,
                               int64_t n, float* logits, float* saved, uint32_t* masks,
                               int64_t slab_block0, int64_t slab_blocks, void* stream) {
    if (n == 0) return 0;
    if (n < 0 || validate_chain(chain, false, saved != nullptr))
        return fail_arg("ffn_mlp_forward: bad chain or size (training chains save every K segment: "
                        "save_in_slot / save_enc_slot set wherever a step has activation / encoding inputs)");
    if ((saved == nullptr) != (masks == nullptr)) return fail_arg("ffn_mlp_forward: saved and masks go together");
    if (slab_blocks != 0 && (slab_block0 < 0 || slab_block0 + (n + 31) / 32 > slab_blocks))
        return fail_arg("ffn_mlp_forward: the launch's blocks must lie inside [0, slab_blocks)");
    const bool train = saved != nullptr;
    if (chain->wide == 3) {            // layers of up to 1024 channels: a team of four waves per block
        if (train) launch_forward<kTrainFwd, 4, true>(chain, packed_w, bias, positions, views, n, logits, saved, masks, slab_block0, slab_blocks, stream);
        else launch_forward<kInfer, 4, true>(chain, packed_w, bias, positions, views, n, logits, saved, masks, slab_block0, slab_blocks, stream);
    } else if (chain->wide == 2) {     // four waves per block: training launches (a batch's short last round)
        if (!train) return fail_arg("ffn_mlp_forward: four-waves-per-block chains (wide == 2) are training-only");
        launch_forward<kTrainFwd, 4>(chain, packed_w, bias, positions, views, n, logits, saved, masks, slab_block0, slab_blocks, stream);
    } else if (chain->wide) {
        if (train) launch_forward<kTrainFwd, 2>(chain, packed_w, bias, positions, views, n, logits, saved, masks, slab_block0, slab_blocks, stream);
        else launch_forward<kInfer, 2>(chain, packed_w, bias, positions, views, n, logits, saved, masks, slab_block0, slab_blocks, stream);
    } else {
        if (train) launch_forward<kTrainFwd, 1>(chain, packed_w, bias, positions, views, n, logits, saved, masks, slab_block0, slab_blocks, stream);
        else launch_forward<kInfer, 1>(chain, packed_w, bias, positions, views, n, logits, saved, masks, slab_block0, slab_blocks, stream);
    }
    return check_launch("ffn_mlp_forward");
}

extern "C" int ffn_render_fused_fwd(const ffn_mlp_chain* chain, const float* packed_w,
                                    const float* bias, const ffn_render_rays* rays,
                                    const ffn_occupancy* occupancy, const ffn_render_out* out,
                                    void* stream) {
    if (rays == nullptr || out == nullptr) return fail_arg("ffn_render_fused_fwd: null argument");
    if (rays->num_rays == 0) return 0;
    if (rays->num_rays < 0 || rays->num_samples < 1 || rays->num_samples > 256)
        return fail_arg("ffn_render_fused_fwd: need 1 <= num_samples <= 256");
    if (validate_chain(chain, false) || chain->wide > 1)
        return fail_arg("ffn_render_fused_fwd: bad chain (narrow and 512-wide forward chains only)");
    if (rays->t_values == nullptr && rays->unit == nullptr)
        return fail_arg("ffn_render_fused_fwd: unit or t_values is required");
    RenderParams p;
    p.starts = rays->starts; p.dirs = rays->directions; p.near_far = rays->near_far;
    p.total_rays = rays->num_rays_total; p.ray_index = rays->ray_index;
    p.ray_base = rays->ray_base; p.valid = rays->valid;
    p.num_rays = rays->num_rays; p.S = rays->num_samples;
    p.unit = rays->unit; p.t_values = rays->t_values;
    p.occ_bits = nullptr;
    const float zero3[3] = {0.f, 0.f, 0.f}, one3[3] = {1.f, 1.f, 1.f};
    p.map = make_map(zero3, one3, 1);      // unused without a grid
    if (occupancy != nullptr && occupancy->bits != nullptr) {
        if (occupancy->resolution < 1 || occupancy->resolution > 1024)
            return fail_arg("ffn_render_fused_fwd: occupancy resolution");
        p.occ_bits = occupancy->bits;
        p.map = make_map(occupancy->box_min, occupancy->box_size, occupancy->resolution);
    }
    p.color = out->color; p.alpha = out->alpha; p.depth = out->depth; p.nan_flag = out->nan_flag;
    p.image = out->image; p.pixel_offset = out->pixel_offset;
    if (chain->wide) {                 // a pair of waves per ray, two pairs per workgroup
        const int64_t grid = persistent_grid(rays->num_rays, 2);
        allow_big_lds(&render_fused_kernel<true>, kRenderLdsBytes);
        hipLaunchKernelGGL(render_fused_kernel<true>, dim3((unsigned)grid), dim3(256), kRenderLdsBytes,
                           (hipStream_t)stream, *chain, packed_w, bias, p);
    } else {
        const int64_t grid = persistent_grid(rays->num_rays, kWavesPerBlock);
        allow_big_lds(&render_fused_kernel<false>, kRenderLdsBytes);
        hipLaunchKernelGGL(render_fused_kernel<false>, dim3((unsigned)grid), dim3(256), kRenderLdsBytes,
                           (hipStream_t)stream, *chain, packed_w, bias, p);
    }
    return check_launch("ffn_render_fused_fwd");
}

extern "C" int ffn_focus_fused(const ffn_mlp_chain* chain, const float* packed_w, const float* bias,
                               const float* starts, const float* directions, const float* near_far,
                               int64_t num_rays_total, const int64_t* ray_index, int num_rays,
                               int num_samples, int n_focus, const float* unit_focus, const float* u,
                               float* t_io, void* stream) {
    if (num_rays == 0) return 0;
    if (num_rays < 0 || num_samples > 256 || n_focus < 3 || n_focus > 64 || n_focus > num_samples)
        return fail_arg("ffn_focus_fused: need 3 <= n_focus <= 64, n_focus <= S <= 256");
    if (validate_chain(chain, false) || chain->wide > 1)
        return fail_arg("ffn_focus_fused: bad chain (narrow and 512-wide forward chains only)");
    FocusParams p;
    p.starts = starts; p.dirs = directions; p.near_far = near_far; p.total_rays = num_rays_total;
    p.ray_index = ray_index; p.num_rays = num_rays; p.S = num_samples; p.n_focus = n_focus;
    p.unit_focus = unit_focus; p.u = u; p.t_io = t_io;
    if (chain->wide) {                 // a pair of waves per ray, two pairs per workgroup
        const int64_t grid = persistent_grid(num_rays, 2);
        allow_big_lds(&focus_fused_kernel<true>);
        hipLaunchKernelGGL(focus_fused_kernel<true>, dim3((unsigned)grid), dim3(256), kLdsBytes,
                           (hipStream_t)stream, *chain, packed_w, bias, p);
        return check_launch("ffn_focus_fused");
    }
    const int64_t wgs = ((int64_t)num_rays + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t grid = persistent_grid(wgs * kWavesPerBlock, kWavesPerBlock);
    allow_big_lds(&focus_fused_kernel<false>);
    hipLaunchKernelGGL(focus_fused_kernel<false>, dim3((unsigned)grid), dim3(256), kLdsBytes,
                       (hipStream_t)stream, *chain, packed_w, bias, p);
    return check_launch("ffn_focus_fused");
}

template <int TWP, bool BIG = false>
static void launch_backward(const ffn_mlp_chain* chain, const float* packed_wt, const float* d_logits,
                            int64_t n, uint32_t* masks, float* dz, int64_t slab_block0,
                            int64_t slab_blocks, void* stream) {
    const int64_t blocks32 = (n + kSamplesPerWave - 1) / kSamplesPerWave;
    const int64_t grid = persistent_grid(blocks32, kWavesPerBlock / TWP);
    allow_big_lds(&mlp_backward_data_kernel<TWP, BIG>);
    hipLaunchKernelGGL((mlp_backward_data_kernel<TWP, BIG>), dim3((unsigned)grid), dim3(256), kLdsBytes,
                       (hipStream_t)stream, *chain, packed_wt, d_logits, n, masks, dz, slab_block0, slab_blocks);
}

extern "C" int ffn_mlp_backward_data(const ffn_mlp_chain* chain, const float* packed_wt,
                                     const float* d_logits, int64_t n, uint32_t* masks,
                                     float* dz, int64_t slab_block0, int64_t slab_blocks,
                                     void* stream) {
    if (n == 0) return 0;
    if (n < 0 || validate_chain(chain, true)) return fail_arg("ffn_mlp_backward_data: bad chain or size");
    if (slab_blocks != 0 && (slab_block0 < 0 || slab_block0 + (n + 31) / 32 > slab_blocks))
        return fail_arg("ffn_mlp_backward_data: the launch's blocks must lie inside [0, slab_blocks)");
    if (chain->wide == 3) launch_backward<4, true>(chain, packed_wt, d_logits, n, masks, dz, slab_block0, slab_blocks, stream);
    else if (chain->wide == 2) launch_backward<4>(chain, packed_wt, d_logits, n, masks, dz, slab_block0, slab_blocks, stream);
    else if (chain->wide) launch_backward<2>(chain, packed_wt, d_logits, n, masks, dz, slab_block0, slab_blocks, stream);
    else launch_backward<1>(chain, packed_wt, d_logits, n, masks, dz, slab_block0, slab_blocks, stream);
    return check_launch("ffn_mlp_backward_data");
}
