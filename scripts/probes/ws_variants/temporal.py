# slab stores of the two-waves-per-SIMD kernels as ordinary (L2 write-back) stores instead of
# non-temporal ones: do the epilogue bursts get absorbed by the L2?
SUBS = [("__builtin_nontemporal_store(f0, &fsave[saved_index16(cq, w.s)]);", "fsave[saved_index16(cq, w.s)] = f0;"),
        ("__builtin_nontemporal_store(f1, &fsave[saved_index16(cq + 1, w.s)]);", "fsave[saved_index16(cq + 1, w.s)] = f1;"),
        ("__builtin_nontemporal_store(y0, &save_out[saved_index16(cq, save_s)]);", "save_out[saved_index16(cq, save_s)] = y0;"),
        ("__builtin_nontemporal_store(y1, &save_out[saved_index16(cq + 2, save_s)]);", "save_out[saved_index16(cq + 2, save_s)] = y1;")]
