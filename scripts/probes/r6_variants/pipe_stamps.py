# cycle stamps (s_memtime) of the half-step pipeline (csrc/mlp_bf16_ws.hip, forward kernel): workgroup 0,
# every wave, its LAST pass; read back through ffn_debug_read_stamps (variant-only export).
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
# forward step: K phase
("""    if (S::PIPE && w.owes_barrier && !pipelined) { ws_barrier(); w.owes_barrier = false; }
    if (kb_act > 0) {                               // X holds the previous step's output""",
 """    STAMP();
    if (S::PIPE && w.owes_barrier && !pipelined) { ws_barrier(); w.owes_barrier = false; }
    if (kb_act > 0) {                               // X holds the previous step's output"""),
("""            swap_due = run(kb_act - g0 < n_old ? kb_act - g0 : n_old, g0);
            if (S::PIPE && w.owes_barrier) { ws_barrier(); w.owes_barrier = false; }   // "young half written"
        }
    }
    if (kb_feat > 0) {""",
 """            swap_due = run(kb_act - g0 < n_old ? kb_act - g0 : n_old, g0);
            STAMP();
            if (S::PIPE && w.owes_barrier) { ws_barrier(); w.owes_barrier = false; }   // "young half written"
            STAMP();
        }
    }
    if (kb_feat > 0) {"""),
("""    if (pipelined && !w.older) ws_barrier();       // meet 1 (younger waves): every K loop of this step is done
""", """    STAMP();
    if (pipelined && !w.older) ws_barrier();       // meet 1 (younger waves): every K loop of this step is done
    STAMP();
"""),
("""    if (pipelined) {
        write_next();                              // older: the other old-half buffer; younger: the young half, behind meet 1
        ws_barrier();                              // older: meet 1 ("old half written, my K loops done"); younger: meet 2
        if (w.older) w.owes_barrier = true;        // (meet 2 comes in the middle of the next K loop)
    } else {""", """    STAMP();
    if (pipelined) {
        write_next();                              // older: the other old-half buffer; younger: the young half, behind meet 1
        STAMP();
        ws_barrier();                              // older: meet 1 ("old half written, my K loops done"); younger: meet 2
        STAMP();
        if (w.older) w.owes_barrier = true;        // (meet 2 comes in the middle of the next K loop)
    } else {"""),
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

SUBS_NOLOAD = SUBS + [("""    ws_kblock<S, NT, 0>(w, acc, wreg[P][0], ops, g + 1);
    ws_load_kblock<S>(w, wreg[P][0], c2, 0);""", """    ws_kblock<S, NT, 0>(w, acc, wreg[P][0], ops, g + 1);"""),
("""    ws_kblock<S, NT, 1>(w, acc, wreg[P][1], ops, nxt);
    ws_load_kblock<S>(w, wreg[P][1], c2, 1);""", """    ws_kblock<S, NT, 1>(w, acc, wreg[P][1], ops, nxt);""")]
SUBS_NOX = SUBS + [("""            x[HB ^ 1][b][2 - q] = __builtin_bit_cast(bf16x8, p[b * S::kBlkVecs + (2 - q) * 64]);""", """            if (false) x[HB ^ 1][b][2 - q] = __builtin_bit_cast(bf16x8, p[b * S::kBlkVecs + (2 - q) * 64]);""")]
