// What a LONE wave per SIMD can sustain in the K loop of a three-part (bf16x6) chain: the question
// behind a one-wave-per-SIMD organisation of the split kernels (512 registers per wave, the epilogue
// of one tile software-pipelined under the K loop of the next).  A workgroup is four
// waves, one per SIMD; a wave owns NT output tiles of two 32-sample blocks; per K block it issues
// 6 NT x 2 matrix instructions, streams its weights (NT x 3 KiB) L2 -> registers DEPTH K blocks ahead
// out of a 1.5 MB operand pack that every workgroup walks (the tiny model's), reads the X operands
// (6 KiB: two blocks x three parts) from LDS one K block ahead, and carries FILL independent vector
// instructions behind every matrix instruction.  Every group (one matrix instruction, at most one
// memory instruction, the fillers) sits between two scheduling fences, like the fenced trips of
// scripts/probes/r6_variants/fenced_trips.patch.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/chain_stream_probe.hip -o scripts/probes/chain_stream_probe
#include <hip/hip_runtime.h>

#include <cstring>
#include <random>
#include <vector>
#include <cstdio>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKBlocks = 64;                 // K blocks of the chain (tiny NeRF: 32 + 16 + 16)
constexpr int kTiles = 8;
constexpr int kKbVecs = kTiles * 3 * 64;     // float4 per K block of the pack: 8 tiles x 3 parts x 64 lanes

// PATTERN (round 6, the backward kernels' two accumulators): 0 = every product on acc[t][b]; 1 = the first five
// products of a K block on a second accumulator set lo[t][b], the sixth on acc (mlp_bf16_ws.hip's backward);
// 2 = the same with the sixth product issued FIRST; 3 = the second set takes products 0..2, acc 3..5
template <int NT, int DEPTH, int FILL, bool LOADW, bool READX, int PARTNER = 0, int PATTERN = 0>
__global__ void __launch_bounds__(PARTNER ? 512 : 256, 1)
probe(const f32x4* __restrict__ pack, float* out, int passes, long long* cycles, const f32x4* __restrict__ xinit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* xbuf = reinterpret_cast<f32x4*>(smem);        // X: [K block 0..15][block][part][lane]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16 * 2 * 3 * 64; i += (PARTNER ? 512 : 256)) {
        f32x4 v; v[0] = i * 1e-3f; v[1] = 1.0f; v[2] = -1.0f; v[3] = 0.5f;
        xbuf[i] = xinit != nullptr ? xinit[i] : v;
    }
    if (threadIdx.x == 0) *reinterpret_cast<volatile int*>(smem + 16 * 2 * 3 * 64 * 16) = 0;
    __syncthreads();
    if (PARTNER && wave >= 4) {
        // the co-resident wave of every SIMD: an "epilogue" -- dependent vector arithmetic with LDS stores
        // and reads in between -- for as long as the K-loop waves run (they raise a flag at their end)
        volatile int* flag = reinterpret_cast<volatile int*>(smem + 16 * 2 * 3 * 64 * 16);
        f32x4* mine = xbuf + 16 * 2 * 3 * 64 + (wave - 4) * 64 * 8 + lane;
        float a = lane, b = lane + 1, c = 0.5f;
        int spins = 0;
        while (*flag == 0 && spins < (1 << 22)) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                a = __builtin_fmaf(a, 1.0001f, b);
                b = __builtin_fmaf(b, 0.9999f, c);
                c = __builtin_fmaf(c, 1.0002f, a);
            }
            f32x4 v; v[0] = a; v[1] = b; v[2] = c; v[3] = 1.0f;
            mine[(spins & 7) * 64] = v;
            const f32x4 r = mine[((spins + 3) & 7) * 64];
            a += r[0] * 1e-9f;
            ++spins;
        }
        out[blockIdx.x * 512 + threadIdx.x] = a + b + c;
        return;
    }
    if (PARTNER == 2) __builtin_amdgcn_s_setprio(1);
    typedef const f32x4 __attribute__((address_space(1)))* gptr;
    f32x16 acc[NT][2], lo[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[t][b][r] = 0.0f; lo[t][b][r] = 0.0f; }
    bf16x8 wr[DEPTH][NT][3];
    bf16x8 x[2][2][3];
    float fill[8];
    float fc0 = 1.0001f, fc1 = 0.5f;
    asm volatile("" : "+v"(fc0), "+v"(fc1));
#pragma unroll
    for (int i = 0; i < 8; ++i) fill[i] = lane + i;
    // prologue: DEPTH K blocks of weights in flight, X of K block 0
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                wr[d][t][p] = __builtin_bit_cast(bf16x8, pack[d * kKbVecs + ((wave + 4 * t) * 3 + p) * 64 + lane]);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int p = 0; p < 3; ++p) x[0][b][p] = __builtin_bit_cast(bf16x8, xbuf[(b * 3 + p) * 64 + lane]);
    constexpr int PW[6] = {0, 2, 1, 0, 1, 0}, PX[6] = {2, 0, 1, 1, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    int kb = 0;                                  // K block of the chain (0 .. kKBlocks - 1), uniform
    for (int pass = 0; pass < passes; ++pass) {
        for (int k0 = 0; k0 < kKBlocks; k0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int hb = d & 1;
                // the registers of the K block that multiplied BEFORE this one (ring slot d - 1) are
                // requested again, DEPTH K blocks ahead of their next use, one request per group
                const int dp = (d + DEPTH - 1) % DEPTH;
                int ahead = kb + DEPTH - 1;
                ahead -= ahead >= kKBlocks ? kKBlocks : 0;
                gptr base = (gptr)(pack + (long long)ahead * kKbVecs + wave * 3 * 64);
                asm volatile("" : "+s"(base));
                const f32x4* xp = xbuf + ((kb + 1) & 15) * (2 * 3 * 64) + lane;
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const int pq = PATTERN == 2 ? (q == 0 ? 5 : q - 1) : q;       // product issued in slot q
                            const bool on_lo = PATTERN == 0 ? false : (PATTERN == 3 ? pq < 3 : pq < 5);
                            if (on_lo) lo[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[d][t][PW[pq]], x[hb][b][PX[pq]], lo[t][b], 0, 0, 0);
                            else acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[d][t][PW[pq]], x[hb][b][PX[pq]], acc[t][b], 0, 0, 0);
                            const int g = (q * NT + t) * 2 + b;      // group index 0 .. 12 NT - 1
                            if (READX && g < 6) {                    // the six X reads of the next K block
                                const int rb = g / 3, rp = g % 3;
                                x[hb ^ 1][rb][rp] = __builtin_bit_cast(bf16x8, xp[(rb * 3 + rp) * 64]);
                            }
                            if (LOADW && g >= 6 && g < 6 + 3 * NT) { // then the 3 NT weight requests
                                const int rt = (g - 6) / 3, rp = (g - 6) % 3;
                                wr[dp][rt][rp] = __builtin_bit_cast(bf16x8, base[(4 * rt * 3 + rp) * 64 + lane]);
                            }
#pragma unroll
                            for (int j = 0; j < FILL; ++j) {
                                // (inline assembly: left to itself hipcc packs pairs of these into
                                // v_pk_fma_f32, which is slow beside matrix instructions -- the first
                                // version of this probe measured THAT: 3.4 cycles per filler)
                                const int idx = (g * FILL + j) & 7;
                                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fill[idx]) : "v"(fc0), "v"(fc1));
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                kb = kb + 1 < kKBlocks ? kb + 1 : 0;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[t][b][r] + lo[t][b][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += fill[i];
    if (PARTNER && wave == 0 && lane == 0) *reinterpret_cast<volatile int*>(smem + 16 * 2 * 3 * 64 * 16) = 1;
    out[blockIdx.x * (PARTNER ? 512 : 256) + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

static int g_passes = 64;                    // (`sustained`: 2400 -- ~30 ms per launch, the clock settles)
static const f32x4* g_xinit = nullptr;       // (`sustained`: three-part splits of ReLU'd normal activations)

template <int NT, int DEPTH, int FILL, bool LOADW, bool READX, int PARTNER = 0, int PATTERN = 0>
void run(const f32x4* pack, float* out, long long* cyc) {
    const int passes = g_passes, grid = 256;
    const size_t lds = 16 * 2 * 3 * 64 * 16 + 64 + 4 * 64 * 8 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NT, DEPTH, FILL, LOADW, READX, PARTNER, PATTERN>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NT, DEPTH, FILL, LOADW, READX, PARTNER, PATTERN><<<grid, PARTNER ? 512 : 256, lds>>>(pack, out, g_passes > 64 ? g_passes : 2, cyc, g_xinit);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NT, DEPTH, FILL, LOADW, READX, PARTNER, PATTERN><<<grid, PARTNER ? 512 : 256, lds>>>(pack, out, passes, cyc, g_xinit);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_wave = (double)passes * kKBlocks * 12 * NT;
    const double mfma = (double)grid * 4 * per_wave;
    printf("pattern %d  partner %d  tiles/wave %d  depth %d  fill %d  weights %s  X %s: %8.3f ms  %7.1f TFLOP/s issued (2500 peak)  %5.1f ticks/MFMA  "
           "weights %.2f TB/s\n", PATTERN, PARTNER, NT, DEPTH, FILL, LOADW ? "L2  " : "none", READX ? "LDS " : "none", ms,
           mfma * 32768 / ms / 1e9, (double)h / per_wave,
           LOADW ? (double)grid * 4 * passes * kKBlocks * NT * 3072.0 / ms / 1e9 : 0.0);
}

static unsigned short bf16_rne(float v) {
    unsigned u;
    memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// count x [3 parts][64 lanes] x 8 values: the three-part split (hi, mid, lo) of draws of `draw`
template <class F>
static std::vector<unsigned short> split_parts(size_t count, F draw) {
    std::vector<unsigned short> out(count * 3 * 64 * 8);
    for (size_t s = 0; s < count; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                float v = draw();
                for (int p = 0; p < 3; ++p) {
                    const unsigned short h = bf16_rne(v);
                    out[((s * 3 + p) * 64 + lane) * 8 + j] = h;
                    const unsigned u = (unsigned)h << 16;
                    float back;
                    memcpy(&back, &u, 4);
                    v -= back;
                }
            }
    return out;
}

int main(int argc, char** argv) {
    f32x4* pack; float* out; long long* cyc;
    const size_t pack_bytes = (size_t)kKBlocks * kKbVecs * 16;
    hipMalloc(&pack, pack_bytes); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    hipMemset(pack, 0x3c, pack_bytes);
    if (argc > 1 && strcmp(argv[1], "sustained") == 0) {
        // What the stream SUSTAINS: launches of ~30 ms (the chip clocks to its power budget within a few ms)
        // on operands with the chain kernels' statistics -- three-part splits of normal weights and of
        // ReLU'd normal activations (constant data toggles nothing and clocks 20-25 % higher).
        std::mt19937 rng(7);
        std::normal_distribution<float> nw(0.f, 0.0625f), nx(0.f, 1.f);
        const std::vector<unsigned short> wp = split_parts((size_t)kKBlocks * kTiles, [&] { return nw(rng); });
        const std::vector<unsigned short> xp = split_parts(16 * 2, [&] { const float v = nx(rng); return v > 0.f ? v : 0.f; });
        hipMemcpy(pack, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
        f32x4* xinit;
        hipMalloc(&xinit, xp.size() * 2);
        hipMemcpy(xinit, xp.data(), xp.size() * 2, hipMemcpyHostToDevice);
        g_xinit = xinit;
        g_passes = 2400;
        printf("SUSTAINED: 2400 passes of 64 K blocks per launch, random three-part operands\n");
        run<1, 4, 0, false, false>(pack, out, cyc);      // the matrix pipe alone
        run<1, 4, 0, false, true>(pack, out, cyc);       // + LDS operand reads
        run<1, 4, 0, true, true>(pack, out, cyc);        // + the weight stream
        run<1, 8, 2, true, true>(pack, out, cyc);        // + vector work in the shadows
        run<1, 8, 4, true, true>(pack, out, cyc);
        run<1, 8, 6, true, true>(pack, out, cyc);
        run<1, 4, 0, true, true, 1>(pack, out, cyc);     // + a vector-work partner wave on every SIMD
        run<1, 4, 0, true, true>(pack, out, cyc);
        return 0;
    }
    printf("operand pack %.2f MB, 256 workgroups x 4 waves (one per SIMD), 64 passes of 64 K blocks\n", pack_bytes / 1e6);
    run<1, 4, 0, false, false>(pack, out, cyc);      // the matrix pipe alone
    run<1, 4, 0, false, true>(pack, out, cyc);       // + LDS operand reads
    run<1, 2, 0, true, true>(pack, out, cyc);        // + the weight stream, 2 / 4 / 8 K blocks ahead
    run<1, 4, 0, true, true>(pack, out, cyc);
    run<1, 8, 0, true, true>(pack, out, cyc);
    run<1, 8, 2, true, true>(pack, out, cyc);        // + vector work in the shadows
    run<1, 8, 4, true, true>(pack, out, cyc);
    run<1, 8, 5, true, true>(pack, out, cyc);
    run<1, 8, 6, true, true>(pack, out, cyc);
    run<1, 8, 8, true, true>(pack, out, cyc);
    run<1, 4, 4, false, false>(pack, out, cyc);      // (fillers beside the bare matrix stream)
    run<1, 4, 6, false, false>(pack, out, cyc);
    run<2, 2, 0, true, true>(pack, out, cyc);        // two tiles per wave and K block: half the LDS reads per instruction
    run<2, 4, 0, true, true>(pack, out, cyc);
    run<2, 4, 2, true, true>(pack, out, cyc);
    run<2, 4, 4, true, true>(pack, out, cyc);
    run<2, 4, 6, true, true>(pack, out, cyc);
    // with a second wave per SIMD in vector + LDS work (an epilogue beside the K loop), at equal
    // priority and with the K-loop waves at s_setprio 1
    run<1, 2, 0, true, true, 1>(pack, out, cyc);
    run<1, 2, 0, true, true, 2>(pack, out, cyc);
    run<1, 4, 0, true, true, 1>(pack, out, cyc);
    run<1, 4, 0, true, true, 2>(pack, out, cyc);
    // two accumulator sets (the backward kernels)
    run<1, 4, 0, true, true, 0, 0>(pack, out, cyc);
    run<1, 4, 0, true, true, 0, 1>(pack, out, cyc);
    run<1, 4, 0, true, true, 0, 2>(pack, out, cyc);
    run<1, 4, 0, true, true, 0, 3>(pack, out, cyc);
    run<1, 4, 0, false, false, 0, 0>(pack, out, cyc);
    run<1, 4, 0, false, false, 0, 1>(pack, out, cyc);
    run<1, 4, 0, false, false, 0, 2>(pack, out, cyc);
    run<1, 4, 0, false, false, 0, 3>(pack, out, cyc);
    return 0;
}
