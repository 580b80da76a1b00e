// Split-bf16 inference mode of the fused Fourier-feature MLP (OPT-IN, separately labelled; the
// exact-f32 kernels of mlp.hip stay the parity mode and the headline).
//
// v_mfma_f32_32x32x2_f32 runs on the FP32 vector lanes (157 TFLOP/s, and no VALU work overlaps
// it); v_mfma_f32_32x32x16_bf16 runs on the matrix units at 16x that rate.  Every f32 operand
// is split into two bf16 parts, x = x_hi + x_lo (16 mantissa bits together), and every product
// into three matrix instructions  w_hi x_lo + w_lo x_hi + w_hi x_hi  with f32 accumulation: a
// relative error of ~2^-16 per product instead of f32's 2^-24, at 3/16 of the matrix time.
//
// At that speed the f32 kernel's organisation no longer works: streaming every weight from L2
// per 32-sample block would need ~48 TB/s.  Here the four waves of a workgroup advance in
// LOCKSTEP through the chain, each on its own block of 32 samples, and share every weight
// K block through LDS (a ring of eight 16 KiB K blocks fed five blocks ahead of their use by
// all 256 threads, operands read one block ahead, one workgroup barrier per four K blocks); the
// activations never touch LDS at all: with the bf16 operand layout the accumulators of layer l
// (lane = (h, sample), registers r <-> channel 8q + 4h + p) become the B operand of layer l+1
// in place -- K block G of the next layer is the eight accumulator registers 8(G&1) .. 8(G&1)+7
// of tile G/2, converted to (hi, lo) bf16 pairs.  The weight packs are permuted to that K order
// on the host (ffn_mlp_pack_bf16).  Encoding features are generated per K block in registers
// (four angles per lane), overlapping the previous block's matrix instructions.
#include "bf16_ring.h"

namespace ffn {

constexpr int kBiasFloats16 = 4096;
constexpr size_t kLdsBytes16 = (size_t)kRingBlocks16 * kBlockVecs16 * 16 + kEncTableBytes + kBiasFloats16 * 4;

// ---------------------------------------------------------------------------------- pack
// dst[(((G*tiles + o)*parts + part)*64 + lane)*8 + j] = part(src[32*o + (lane & 31)][col_map[16*G + 8*(lane >> 5) + j]])
// (rows past the matrix are zero: the forward kernel always runs tiles = 8)
// part 0 = bf16(v) (round to nearest even), part p = bf16(v - float(part 0) - .. - float(part p-1));
// parts = 2: the (hi, lo) operands of the bf16x3 kernels, parts = 3: (hi, mid, lo), exactly v (bf16x6).
// transpose: the operand is src^T -- tile rows walk src's columns, col_map maps K to src's rows.
__global__ void __launch_bounds__(256)
pack_bf16_kernel(const float* __restrict__ src, int rows, int cols, int ld,
                 const int32_t* __restrict__ col_map, int kblocks, int tiles, int transpose, int parts,
                 uint16_t* __restrict__ dst) {
    const int64_t total = (int64_t)kblocks * tiles * 64 * 8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7);
        const int lane = (int)((e >> 3) & 63);
        const int64_t go = e >> 9;
        const int o = (int)(go % tiles);
        const int G = (int)(go / tiles);
        const int r = 32 * o + (lane & 31);
        const int c = col_map[16 * G + 8 * (lane >> 5) + j];
        float v = 0.0f;
        if (!transpose) {
            if (c >= 0 && r < rows && c < cols) v = src[(int64_t)r * ld + c];
        } else if (c >= 0 && c < rows && r < cols) {
            v = src[(int64_t)c * ld + r];
        }
        const int64_t base = ((go * parts) * 64 + lane) * 8 + j;
        for (int p = 0; p < parts; ++p) {
            const __bf16 part = (__bf16)v;
            dst[base + (int64_t)p * 64 * 8] = __builtin_bit_cast(uint16_t, part);
            v -= (float)part;
        }
    }
}

struct Ctx16 : Ring16 {
    int h, s;
    float x0, x1, x2, v0, v1, v2;
    float logit[4];
    const float* enc_table;   // LDS
    const float* bias_lds;    // LDS
    // training only: where this wave's block saves its activations / ReLU sign masks
    int64_t block, num_blocks;
    bool active;
    float* saved;
    uint4* masks;
};

// the block of a slab slot: the layout of mlp.hip (the backward kernels read what this kernel saves)
__device__ __forceinline__ f32x4* slab_block16(const ffn_mlp_chain& ch, int slot, const Ctx16& w) {
    return reinterpret_cast<f32x4*>(w.saved + ch.slot_offset[slot] * w.num_blocks * 32) +
           w.block * (int64_t)(ch.slot_channels[slot] * 8);
}

// One dense step.  cur_hi / cur_lo: the step's activation
// input as bf16 pairs, K block G = channels 16G..16G+15 in the hand-off order; overwritten with
// the step's output.  Every step runs all eight output tiles (narrower layers are zero-padded
// by the pack: one instantiation -- a second one costs hipcc 1.5 KB of scratch per lane -- and
// 1/24 more matrix work on the full NeRF); the tiles past the layer's width get no bias, no
// head terms, and are never consumed.
//
// TRAIN: the step also leaves what the backward pass needs, in the f32 kernels' formats -- the
// encoding features it generated (step.save_enc_slot), its output (step.out_slot) and the ReLU sign mask (step.mask_slot).
constexpr int OT = OT16;
template <bool TRAIN>
__device__ __forceinline__ void step16(const ffn_mlp_chain& ch, const ffn_step& L, Ctx16& w,
                                       bf16x8 (&cur_hi)[16], bf16x8 (&cur_lo)[16], f32x4 (&stage)[2][4],
                                       bf16x8 (&wh)[2][8], bf16x8 (&wl)[2][8]) {
    const int ot = L.out_tiles;                    // real tiles of this layer
    const int kb_act = L.act_groups >> 1, kb_feat = L.aux_groups >> 1;

    f32x16 acc[OT];
    {
        const float* bv = w.bias_lds + L.b_off + 4 * w.h;
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + (o < ot ? 32 * o + 8 * q : 0));
                if (o >= ot) b4 = (f32x4)(0.0f);
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[o][4 * q + p] = b4[p];
            }
    }

#pragma unroll
    for (int G = 0; G < 16; G += 2)
        if (G < kb_act) {
            ring_kblock<0>(w, acc, cur_hi[G], cur_lo[G], stage, wh, wl);
            ring_kblock<1>(w, acc, cur_hi[G + 1], cur_lo[G + 1], stage, wh, wl);
        }
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
        // K blocks whose eight frequencies (both lane halves) are all real ones take the
        // select-free feature code; the tail (raw inputs, padding) the generic one
        int g_trig = e.num_freq >> 3;
        g_trig = g_trig < kb_feat ? g_trig : kb_feat;
        // (computing block G+1's features under block G's matrix instructions was measured: the
        // interleaved issue made the block 4 % SLOWER on this compiler, so the features run between blocks)
        for (int G = 0; G < kb_feat; G += 2) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                float f[8];
                if (G + sub < g_trig) features16<true>(enc, G + sub, w.h, p0, p1, p2, f);
                else features16<false>(enc, G + sub, w.h, p0, p1, p2, f);
                if (TRAIN && L.save_enc_slot >= 0 && w.active) {
                    f32x4* fsave = slab_block16(ch, L.save_enc_slot, w);
                    const int cq = 4 * (G + sub) + 2 * w.h;
                    f32x4 f0, f1;
#pragma unroll
                    for (int p = 0; p < 4; ++p) { f0[p] = f[p]; f1[p] = f[4 + p]; }
                    __builtin_nontemporal_store(f0, &fsave[saved_index16(cq, w.s)]);
                    __builtin_nontemporal_store(f1, &fsave[saved_index16(cq + 1, w.s)]);
                }
                bf16x8 fh, fl;
                split8(f, fh, fl);
                if (sub == 0) ring_kblock<0>(w, acc, fh, fl, stage, wh, wl);
                else ring_kblock<1>(w, acc, fh, fl, stage, wh, wl);
            }
        }
    }
    // ---- epilogue: ReLU, fused head, hand-off as bf16 pairs
    const bool fused_head = L.head_off >= 0;
    const float* hw = w.bias_lds + (fused_head ? L.head_off : 0) + 4 + 16 * w.h;
    if (fused_head && w.h == 0) {
        const f32x4 hb = *reinterpret_cast<const f32x4*>(w.bias_lds + L.head_off);
#pragma unroll
        for (int c = 0; c < 4; ++c) w.logit[c] += hb[c];
    }
    const int relu_floor = L.relu ? 0 : (int)0x80000000;
    f32x4* save_out = nullptr;
    if (TRAIN && L.out_slot >= 0 && w.active) save_out = slab_block16(ch, L.out_slot, w);
    // The 32 store addresses of this epilogue are lane constants (cq*32 + (s ^ (cq & 15))): left
    // alone, hipcc computes them all once per kernel, SPILLS them, and reloads them here -- and a
    // scratch reload waits (in-order vmcnt) for every global store issued before it, which cost
    // the training variant ~3 ms.  Making the lane term opaque per step keeps them two-instruction
    // recomputations.
    int save_s = w.s, save_h = w.h;
    if (TRAIN) asm volatile("" : "+v"(save_s), "+v"(save_h));
    unsigned sign_bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = acc[o][8 * half + j];
                // (mlp.hip's mask format: value (o&1, q, p) of word o/2 ends at bit 31 - (16(o&1)+4q+p))
                if (TRAIN)
                    sign_bits[o >> 1] = __builtin_amdgcn_alignbit(sign_bits[o >> 1],
                                                                  __builtin_bit_cast(unsigned, 0.0f - t), 31);
                y[j] = __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, t), relu_floor));
            }
            if (TRAIN && save_out != nullptr && o < ot) {
                f32x4 y0, y1;
#pragma unroll
                for (int p = 0; p < 4; ++p) { y0[p] = y[p]; y1[p] = y[4 + p]; }
                const int cq = 2 * (4 * o + 2 * half) + save_h;    // K group 4o + q, q = 2 half (+1)
                __builtin_nontemporal_store(y0, &save_out[saved_index16(cq, save_s)]);
                __builtin_nontemporal_store(y1, &save_out[saved_index16(cq + 2, save_s)]);
            }
            if (fused_head && o < ot) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // register 8*half + j = channel 32o + 8q + 4h + p with q = 2*half + j/4, p = j%4
                    const int group = 4 * o + 2 * half + (j >> 2);
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(hw + group * 32 + (j & 3) * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) w.logit[c] = __builtin_fmaf(y[j], w4[c], w.logit[c]);
                }
            }
            split8(y, cur_hi[2 * o + half], cur_lo[2 * o + half]);
        }
    }
    if (TRAIN && L.relu && L.mask_slot >= 0 && w.active) {
        // (a one-tile layer's word holds 16 bits in the f32 kernels: right-aligned)
        if (ot == 1) sign_bits[0] >>= 16;
        w.masks[((int64_t)L.mask_slot * w.num_blocks + w.block) * 64 + w.lane] =
            make_uint4(sign_bits[0], sign_bits[1], sign_bits[2], sign_bits[3]);
    }
}

template <bool TRAIN>
__global__ void __launch_bounds__(256, 1)
mlp_forward_bf16_kernel(const ffn_mlp_chain ch, const uint16_t* __restrict__ packed,
                        const float* __restrict__ bias, const float* __restrict__ positions,
                        const float* __restrict__ views, int64_t n, float* __restrict__ logits,
                        float* __restrict__ saved, uint32_t* __restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* enc_table = reinterpret_cast<float*>(smem + (size_t)kRingBlocks16 * kBlockVecs16 * 16);
    float* bias_lds = reinterpret_cast<float*>(smem + (size_t)kRingBlocks16 * kBlockVecs16 * 16 + kEncTableBytes);
    stage_encoding_tables(ch.enc, enc_table, threadIdx.x, 256);
    for (int i = threadIdx.x; i < ch.bias_floats; i += 256) bias_lds[i] = bias[i];
    Ctx16 w;
    w.tid = threadIdx.x;
    w.lane = threadIdx.x & 63;
    w.h = w.lane >> 5;
    w.s = w.lane & 31;
    w.wbuf = reinterpret_cast<f32x4*>(smem);
    w.enc_table = enc_table;
    w.bias_lds = bias_lds;
    w.gweights = reinterpret_cast<const f32x4*>(packed + ch.step[0].w_off);
    w.total_kb = 0;
    for (int li = 0; li < ch.num_steps; ++li) w.total_kb += (ch.step[li].act_groups + ch.step[li].aux_groups) >> 1;
    w.saved = saved;
    w.masks = reinterpret_cast<uint4*>(masks);
    w.num_blocks = (n + 31) / 32;
    f32x4 stage[2][4];
    bf16x8 wh[2][8], wl[2][8];          // weight operands: set (g & 1) belongs to ring position g
    ring_prime(w, stage, wh, wl);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t num_blocks = (n + 31) / 32;
    const int64_t groups = (num_blocks + 3) / 4;            // 4 blocks (one per wave) per pass
    // the inputs of a pass are requested one pass ahead (an HBM round trip is ~5 % of a pass)
    float in_next[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto request_inputs = [&](int64_t pass) {
        int64_t block = pass * 4 + wave;
        block = block < num_blocks ? block : num_blocks - 1;
        const int64_t sample = block * 32 + w.s;
        const int64_t src = sample < n ? sample : n - 1;
        in_next[0] = positions[src * 3 + 0]; in_next[1] = positions[src * 3 + 1]; in_next[2] = positions[src * 3 + 2];
        if (views != nullptr) {
            in_next[3] = views[src * 3 + 0]; in_next[4] = views[src * 3 + 1]; in_next[5] = views[src * 3 + 2];
        }
    };
    request_inputs(blockIdx.x);
    for (int64_t pass = blockIdx.x; pass < groups; pass += gridDim.x) {
        const int64_t block = pass * 4 + wave;
        const bool active = block < num_blocks;
        const int64_t sample = (active ? block : num_blocks - 1) * 32 + w.s;
        w.block = active ? block : num_blocks - 1;
        w.active = active;
        w.x0 = in_next[0]; w.x1 = in_next[1]; w.x2 = in_next[2];
        w.v0 = in_next[3]; w.v1 = in_next[4]; w.v2 = in_next[5];
        request_inputs(pass + gridDim.x < groups ? pass + gridDim.x : pass);
        w.logit[0] = w.logit[1] = w.logit[2] = w.logit[3] = 0.0f;
        bf16x8 cur_hi[16], cur_lo[16];
#pragma unroll
        for (int G = 0; G < 16; ++G) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { cur_hi[G][j] = (__bf16)0.0f; cur_lo[G][j] = (__bf16)0.0f; }
        }
        for (int li = 0; li < ch.num_steps; ++li) step16<TRAIN>(ch, ch.step[li], w, cur_hi, cur_lo, stage, wh, wl);
        f32x4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = w.logit[c] + __shfl_xor(w.logit[c], 32);
        if (w.h == 0 && active && sample < n) reinterpret_cast<f32x4*>(logits)[sample] = out;
    }
}

}  // namespace ffn

using namespace ffn;

static int pack_bf16(const char* what, const float* src, int rows, int cols, int ld, const int32_t* col_map,
                     int kblocks, int tiles, int transpose, int parts, uint16_t* dst, void* stream) {
    if (kblocks <= 0 || tiles <= 0 || col_map == nullptr || parts < 2 || parts > 3) return fail_arg(what);
    const int64_t total = (int64_t)kblocks * tiles * 512;
    int64_t grid = (total + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, src, rows,
                       cols, ld, col_map, kblocks, tiles, transpose, parts, dst);
    return check_launch(what);
}

extern "C" int ffn_mlp_pack_bf16(const float* src, int rows, int cols, int ld, const int32_t* col_map,
                                 int kblocks, int tiles, int transpose, uint16_t* dst, void* stream) {
    return pack_bf16("ffn_mlp_pack_bf16: shape", src, rows, cols, ld, col_map, kblocks, tiles, transpose, 2, dst, stream);
}

extern "C" int ffn_mlp_pack_bf16_parts(const float* src, int rows, int cols, int ld, const int32_t* col_map,
                                       int kblocks, int tiles, int transpose, int parts, uint16_t* dst,
                                       void* stream) {
    return pack_bf16("ffn_mlp_pack_bf16_parts: shape or parts (2 or 3)", src, rows, cols, ld, col_map, kblocks,
                     tiles, transpose, parts, dst, stream);
}

static int launch_forward16(const char* what, const ffn_mlp_chain* chain, const uint16_t* packed_w,
                            const float* bias, const float* positions, const float* views,
                            int64_t n, float* logits, float* saved, uint32_t* masks, void* stream) {
    if (n == 0) return 0;
    if (n < 0 || chain == nullptr || chain->num_steps < 1 || chain->num_steps > FFN_MAX_STEPS)
        return fail_arg(what);
    const bool wide = chain->wide != 0;
    if (chain->bias_floats < 0) return fail_arg(what);
    int kb_feat = 0, kb_all = 0;
    for (int i = 0; i < chain->num_steps; ++i) {
        const ffn_step& L = chain->step[i];
        const int ot = L.out_tiles;
        // (slab-destination steps with fused heads only)
        const bool tiles_ok = wide ? (ot == 2 || ot == 4 || ot == 8 || ot == 16) : (ot == 1 || ot == 2 || ot == 4 || ot == 8);
        if (!tiles_ok || L.dst != 0 || (L.act_groups & 3) ||
            (L.aux_groups & 3) || L.act_groups < 0 || L.act_groups > (wide ? 64 : 32) || L.aux_groups < 0 ||
            L.act_groups + L.aux_groups == 0 || (L.aux_groups > 0 && (L.enc_id < 0 || L.enc_id > 1)) ||
            (L.head_off >= 0 && L.head_off + 4 + 128 * ot > 4096))
            return fail_arg(what);
        kb_feat += L.aux_groups >> 1;
        kb_all += (L.act_groups + L.aux_groups) >> 1;
    }
    // Two organisations of the same arithmetic (bf16_ring.h): 512-wide chains exist only in the
    // two-waves-per-SIMD one.  Of the narrow chains it is the faster one for the training forward
    // (tiny NeRF -7 %, full NeRF -3 %) and for inference where the encoding is a large part of the
    // work (tiny NeRF -4 %); the ring kernels keep the inference of encoding-light chains (full
    // NeRF: +6 % otherwise).  Measured on one box, interleaved (scripts/microbench_bf16_chain.py).
    // FFN_BF16_KERNELS=ring|ws overrides (the ring kernels need all biases in their LDS copy).
    const bool ring_ok = !wide && chain->bias_floats <= kBiasFloats16;
    if (!ring_ok || prefer_ws_kernels(saved != nullptr || 4 * kb_feat >= kb_all)) {
        launch_forward16_ws(chain, packed_w, bias, positions, views, n, logits, saved, masks, stream);
        return check_launch(what);
    }
    const int64_t groups = ((n + 31) / 32 + 3) / 4;
    int cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int64_t grid = groups < cus ? groups : cus;
    if (saved != nullptr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_forward_bf16_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes16);
        hipLaunchKernelGGL(mlp_forward_bf16_kernel<true>, dim3((unsigned)grid), dim3(256), kLdsBytes16,
                           (hipStream_t)stream, *chain, packed_w, bias, positions, views, n, logits, saved, masks);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_forward_bf16_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes16);
        hipLaunchKernelGGL(mlp_forward_bf16_kernel<false>, dim3((unsigned)grid), dim3(256), kLdsBytes16,
                           (hipStream_t)stream, *chain, packed_w, bias, positions, views, n, logits, nullptr, nullptr);
    }
    return check_launch(what);
}

extern "C" int ffn_mlp_forward_bf16x3(const ffn_mlp_chain* chain, const uint16_t* packed_w,
                                      const float* bias, const float* positions, const float* views,
                                      int64_t n, float* logits, void* stream) {
    return launch_forward16("ffn_mlp_forward_bf16x3: unsupported chain or size", chain, packed_w, bias,
                            positions, views, n, logits, nullptr, nullptr, stream);
}

extern "C" int ffn_mlp_forward_bf16x3_train(const ffn_mlp_chain* chain, const uint16_t* packed_w,
                                            const float* bias, const float* positions, const float* views,
                                            int64_t n, float* logits, float* saved, uint32_t* masks,
                                            void* stream) {
    if (n == 0) return 0;          // (an empty batch has empty slabs: nothing to write, no pointers to check)
    if (saved == nullptr || masks == nullptr) return fail_arg("ffn_mlp_forward_bf16x3_train: saved and masks are required");
    return launch_forward16("ffn_mlp_forward_bf16x3_train: unsupported chain or size", chain, packed_w, bias,
                            positions, views, n, logits, saved, masks, stream);
}
