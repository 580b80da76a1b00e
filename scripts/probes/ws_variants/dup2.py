# every weight request issued TWICE, the duplicate into a shadow register set that is "used" where
# the real one is consumed (no extra waits): what would streaming a step's weights twice per pass
# (two block pairs ping-ponging) cost on the L1 / L2 path?  Timing only.
SUBS = [
("""template <class S, int NT, int P>
__device__ __forceinline__ void ws_chunk(""", """__device__ bf16x8 g_dup_sink;
template <class S, int NT, int P>
__device__ __forceinline__ void ws_chunk("""),
("""    ws_kblock<S, NT, 0>(w, acc, wreg[P][0], x, xh, g + 1);
    ws_load_kblock<S>(w, wreg[P][0], c2, 0);""", """    asm volatile("" ::"v"(w.dupreg[P][0][0][0]), "v"(w.dupreg[P][0][0][1]));
    ws_kblock<S, NT, 0>(w, acc, wreg[P][0], x, xh, g + 1);
    ws_load_kblock<S>(w, wreg[P][0], c2, 0);
    ws_load_kblock<S>(w, reinterpret_cast<bf16x8 (&)[S::TPW][2]>(w.dupreg[P][0]), c2 == 0 ? 1 : c2 - 1, 0);"""),
("""    ws_kblock<S, NT, 1>(w, acc, wreg[P][1], x, xh, nxt);
    ws_load_kblock<S>(w, wreg[P][1], c2, 1);""", """    asm volatile("" ::"v"(w.dupreg[P][1][0][0]), "v"(w.dupreg[P][1][0][1]));
    ws_kblock<S, NT, 1>(w, acc, wreg[P][1], x, xh, nxt);
    ws_load_kblock<S>(w, wreg[P][1], c2, 1);
    ws_load_kblock<S>(w, reinterpret_cast<bf16x8 (&)[S::TPW][2]>(w.dupreg[P][1]), c2 == 0 ? 1 : c2 - 1, 1);"""),
("""    bool stale;                // the weight registers do not hold chunks cpos, cpos + 1""",
 """    bool stale;                // the weight registers do not hold chunks cpos, cpos + 1
    bf16x8 dupreg[2][2][2][2];"""),
]
