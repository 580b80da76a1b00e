# cycle stamps (s_memtime) at the phase boundaries of the two-waves-per-SIMD forward kernels
# (bf16x3 and bf16x6 instantiations alike): workgroup 0, every wave, its LAST pass; read back
# through ffn_debug_read_stamps (variant-only export).  Round-5 text of mlp_bf16_ws.hip.
SUBS = [
("""__device__ __forceinline__ void ws_barrier() {""",
 """__device__ unsigned long long g_stamps[16 * 64];
__device__ __forceinline__ void ws_stamp(int wave, int& idx) {
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && idx < 64) g_stamps[wave * 64 + idx] = __builtin_amdgcn_s_memtime();
    idx++;
}
#define STAMP() ws_stamp(w.wave, w.sidx)
__device__ __forceinline__ void ws_barrier() {"""),
("""    bool stale;                // the weight registers do not hold chunks cpos, cpos + 1""",
 """    bool stale;                // the weight registers do not hold chunks cpos, cpos + 1
    int sidx;"""),
("""    if (kb_act > 0) swap_due = run(kb_act, 0);     // X holds the previous step's output (filled)""",
 """    STAMP();
    if (kb_act > 0) swap_due = run(kb_act, 0);     // X holds the previous step's output (filled)
    STAMP();"""),
("""                if (k_first) {
                    swap_due = run(count, side * HALF);
                    if (next > 0) generate(c1, next, (side ^ 1) * HALF);
                } else {
                    if (next > 0) generate(c1, next, (side ^ 1) * HALF);
                    swap_due = run(count, side * HALF);
                }
                if (next > 0) ws_barrier();""", """                STAMP();
                if (k_first) {
                    swap_due = run(count, side * HALF);
                    STAMP();
                    if (next > 0) generate(c1, next, (side ^ 1) * HALF);
                } else {
                    if (next > 0) generate(c1, next, (side ^ 1) * HALF);
                    STAMP();
                    swap_due = run(count, side * HALF);
                }
                STAMP();
                if (next > 0) ws_barrier();"""),
("""    if (swap_due) ws_swap<S>(wreg);
    ws_barrier();                                  // every K loop of this step has read X
""", """    if (swap_due) ws_swap<S>(wreg);
    STAMP();
    ws_barrier();                                  // every K loop of this step has read X
    STAMP();
"""),
("""        ws_barrier();                              // the step's output is in X
    }
}""", """        STAMP();
        ws_barrier();                              // the step's output is in X
        STAMP();
    }
}"""),
("""        w.block0 = pass * NB;
        w.x0 = in_next[0];""", """        w.block0 = pass * NB;
        w.sidx = 0;
        w.x0 = in_next[0];"""),
("""int launch_backward16_ws(const ffn_mlp_chain* chain, const uint16_t* packed_wt, const float* d_logits,
                         int64_t n, const uint32_t* masks, float* dz, void* stream) {""", """}
extern "C" int ffn_debug_read_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ffn::g_stamps), sizeof(unsigned long long) * 16 * 64);
}
namespace ffn {
int launch_backward16_ws(const ffn_mlp_chain* chain, const uint16_t* packed_wt, const float* d_logits,
                         int64_t n, const uint32_t* masks, float* dz, void* stream) {"""),
]
