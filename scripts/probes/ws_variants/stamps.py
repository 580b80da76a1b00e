# cycle stamps (s_memtime) at the phase boundaries of the WS forward kernel: workgroup 0, every
# wave, last pass; read back through ffn_debug_read_stamps (variant-only export)
SUBS = [
("""__device__ __forceinline__ void ws_barrier() {""",
 """__device__ unsigned long long g_stamps[8 * 64];
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
("""    bf16x8 res[TPW][NB][2][2];
    int save_s = w.s, save_h = w.h, e_lane = w.lane;
    asm volatile("" : "+v"(save_s), "+v"(save_h), "+v"(e_lane));     // (see mlp_bf16.hip: no hoisted address tables)""",
 """    bf16x8 res[TPW][NB][2][2];
    STAMP();
    int save_s = w.s, save_h = w.h, e_lane = w.lane;
    asm volatile("" : "+v"(save_s), "+v"(save_h), "+v"(e_lane));     // (see mlp_bf16.hip: no hoisted address tables)"""),
("""    ws_barrier();                                  // every K loop of this step has read X
""", """    STAMP();
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
("""int launch_backward16_ws(""", """}
extern "C" int ffn_debug_read_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ffn::g_stamps), sizeof(unsigned long long) * 8 * 64);
}
namespace ffn {
int launch_backward16_ws("""),
]
