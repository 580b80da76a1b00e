// Split-bf16 backward-data pass (OPT-IN "bf16x3" training precision; mlp.hip's exact-f32
// backward stays the parity mode).  Same organisation as the forward kernel of mlp_bf16.hip --
// four lockstep waves per workgroup, each on its own block of 32 samples, weights shared through
// the LDS ring, activations handed from step to step in registers -- walking the network from
// the logits to the first layer:
//
//   dZ_j = mask_j * ( W_c^T dZ_c  [+ W_head^T d_logits] )
//
// per producer layer j with hidden consumer c (and, possibly, a logits head).  The operand
// packs are the TRANSPOSED weights in the forward kernel's tile format (ffn_mlp_pack_bf16 with
// transpose = 1); the d_logits term is one K block of four real rows plus one zero K block (every
// step keeps an even number of K blocks).  mask_j is the ReLU sign mask the forward pass saved;
// every dZ_j is written to its slab of `dz` in the f32 kernels' format, for the weight-gradient
// kernel.
#include "bf16_ring.h"

namespace ffn {

struct CtxB16 : Ring16 {
    int h, s;
    f32x4 dl;                 // d(loss)/d(logits) of this lane's sample
    int64_t block, num_blocks;
    bool active;
    float* dz;
    const uint4* masks;
};

__device__ __forceinline__ void step16_bwd(const ffn_mlp_chain& ch, const ffn_step& L, CtxB16& w,
                                           bf16x8 (&cur_hi)[16], bf16x8 (&cur_lo)[16],
                                           f32x4 (&stage)[2][4], bf16x8 (&wh)[2][8], bf16x8 (&wl)[2][8]) {
    const int ot = L.out_tiles;
    const int kb_act = L.act_groups >> 1;
    // the sign mask of the layer being differentiated: requested now, used after the K loop
    uint4 mbits = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    if (L.mask_slot >= 0) mbits = w.masks[((int64_t)L.mask_slot * w.num_blocks + w.block) * 64 + w.lane];

    f32x16 acc[OT16];
#pragma unroll
    for (int o = 0; o < OT16; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[o][r] = 0.0f;

#pragma unroll
    for (int G = 0; G < 16; G += 2)
        if (G < kb_act) {
            ring_kblock<0>(w, acc, cur_hi[G], cur_lo[G], stage, wh, wl);
            ring_kblock<1>(w, acc, cur_hi[G + 1], cur_lo[G + 1], stage, wh, wl);
        }
    if (L.aux_groups > 0) {
        // K rows 0..lg_n-1 of the head block = logits lg_col .. lg_col+lg_n-1 (lane half 0)
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = L.lg_col + j;
            const float d = c == 0 ? w.dl[0] : (c == 1 ? w.dl[1] : (c == 2 ? w.dl[2] : w.dl[3]));
            v[j] = (w.h == 0 && j < L.lg_n && c < 4) ? d : 0.0f;
            v[4 + j] = 0.0f;
        }
        bf16x8 dh, dlo;
        split8(v, dh, dlo);
        ring_kblock<0>(w, acc, dh, dlo, stage, wh, wl);
#pragma unroll
        for (int j = 0; j < 8; ++j) { dh[j] = (__bf16)0.0f; dlo[j] = (__bf16)0.0f; }
        ring_kblock<1>(w, acc, dh, dlo, stage, wh, wl);
    }
    // ---- epilogue: mask, save dZ, hand-off as bf16 pairs
    f32x4* save_out = nullptr;
    if (L.out_slot >= 0 && w.active)
        save_out = reinterpret_cast<f32x4*>(w.dz + ch.slot_offset[L.out_slot] * w.num_blocks * 32) +
                   w.block * (int64_t)(ch.slot_channels[L.out_slot] * 8);
    const int top = ot == 1 ? 15 : 31;      // (a one-tile layer's mask word holds 16 bits)
#pragma unroll
    for (int o = 0; o < OT16; ++o) {
        const unsigned word = o < 2 ? mbits.x : (o < 4 ? mbits.y : (o < 6 ? mbits.z : mbits.w));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int bit = top - (16 * (o & 1) + 8 * half + j);
                const int keep = ((int)(word << (31 - bit))) >> 31;
                const float a = acc[o][8 * half + j];      // (scalar copy: bit_cast of a vector element reads lane 0)
                y[j] = __builtin_bit_cast(float, __builtin_bit_cast(int, a) & keep);
            }
            if (save_out != nullptr && o < ot) {
                f32x4 y0, y1;
#pragma unroll
                for (int p = 0; p < 4; ++p) { y0[p] = y[p]; y1[p] = y[4 + p]; }
                const int cq = 2 * (4 * o + 2 * half) + w.h;
                __builtin_nontemporal_store(y0, &save_out[saved_index16(cq, w.s)]);
                __builtin_nontemporal_store(y1, &save_out[saved_index16(cq + 2, w.s)]);
            }
            split8(y, cur_hi[2 * o + half], cur_lo[2 * o + half]);
        }
    }
}

__global__ void __launch_bounds__(256, 1)
mlp_backward_bf16_kernel(const ffn_mlp_chain ch, const uint16_t* __restrict__ packed,
                         const float* __restrict__ d_logits, int64_t n,
                         const uint32_t* __restrict__ masks, float* __restrict__ dz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CtxB16 w;
    w.tid = threadIdx.x;
    w.lane = threadIdx.x & 63;
    w.h = w.lane >> 5;
    w.s = w.lane & 31;
    w.wbuf = reinterpret_cast<f32x4*>(smem);
    w.gweights = reinterpret_cast<const f32x4*>(packed + ch.step[0].w_off);
    w.total_kb = 0;
    for (int li = 0; li < ch.num_steps; ++li)
        w.total_kb += (ch.step[li].act_groups >> 1) + (ch.step[li].aux_groups > 0 ? 2 : 0);
    w.dz = dz;
    w.masks = reinterpret_cast<const uint4*>(masks);
    w.num_blocks = (n + 31) / 32;
    f32x4 stage[2][4];
    bf16x8 wh[2][8], wl[2][8];
    ring_prime(w, stage, wh, wl);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t num_blocks = w.num_blocks;
    const int64_t groups = (num_blocks + 3) / 4;            // 4 blocks (one per wave) per pass
    f32x4 dl_next = (f32x4)(0.0f);
    auto request_inputs = [&](int64_t pass) {
        int64_t block = pass * 4 + wave;
        block = block < num_blocks ? block : num_blocks - 1;
        const int64_t sample = block * 32 + w.s;
        // samples past n (the ragged tail of the last block) contribute zero
        dl_next = sample < n ? reinterpret_cast<const f32x4*>(d_logits)[sample] : (f32x4)(0.0f);
    };
    request_inputs(blockIdx.x);
    for (int64_t pass = blockIdx.x; pass < groups; pass += gridDim.x) {
        const int64_t block = pass * 4 + wave;
        w.active = block < num_blocks;
        w.block = w.active ? block : num_blocks - 1;
        w.dl = dl_next;
        request_inputs(pass + gridDim.x < groups ? pass + gridDim.x : pass);
        bf16x8 cur_hi[16], cur_lo[16];
#pragma unroll
        for (int G = 0; G < 16; ++G) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { cur_hi[G][j] = (__bf16)0.0f; cur_lo[G][j] = (__bf16)0.0f; }
        }
        for (int li = 0; li < ch.num_steps; ++li) step16_bwd(ch, ch.step[li], w, cur_hi, cur_lo, stage, wh, wl);
    }
}

}  // namespace ffn

using namespace ffn;

extern "C" int ffn_mlp_backward_data_bf16x3(const ffn_mlp_chain* chain, const uint16_t* packed_wt,
                                            const float* d_logits, int64_t n, const uint32_t* masks,
                                            float* dz, void* stream) {
    if (n == 0) return 0;
    const char* what = "ffn_mlp_backward_data_bf16x3: unsupported chain or size";
    if (n < 0 || chain == nullptr || chain->num_steps < 1 || chain->num_steps > FFN_MAX_STEPS)
        return fail_arg(what);
    const bool wide = chain->wide != 0;
    for (int i = 0; i < chain->num_steps; ++i) {
        const ffn_step& L = chain->step[i];
        const int ot = L.out_tiles;
        const bool tiles_ok = wide ? (ot == 2 || ot == 4 || ot == 8 || ot == 16) : (ot == 1 || ot == 2 || ot == 4 || ot == 8);
        if (!tiles_ok || (L.act_groups & 3) || L.act_groups < 0 ||
            L.act_groups > (wide ? 64 : 32) || (L.act_groups == 0 && L.aux_groups == 0) ||
            (L.aux_groups > 0 && (L.lg_col < 0 || L.lg_n < 1 || L.lg_col + L.lg_n > 4)))
            return fail_arg(what);
    }
    // (512-wide chains exist only in the two-waves-per-SIMD organisation; for narrow chains it is
    // level with the ring kernels on the tiny NeRF and 4 % faster on the full one -- both are
    // bound by the dZ stores: 2.9 of 3.8 ms without them)
    if (wide || prefer_ws_kernels(true)) {
        launch_backward16_ws(chain, packed_wt, d_logits, n, masks, dz, stream);
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
    const size_t lds = (size_t)kRingBlocks16 * kBlockVecs16 * 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_backward_bf16_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mlp_backward_bf16_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream,
                       *chain, packed_wt, d_logits, n, masks, dz);
    return check_launch("ffn_mlp_backward_data_bf16x3");
}
