# non-temporal slab traffic: loads of the weight-gradient kernels, stores of the forward / dgrad saves
NT_LOAD = {
    "wgrad.hip": [("R[j] = chunk[tid];", "R[j] = __builtin_nontemporal_load(&chunk[tid]);")],
    "wgrad_bf16.hip": [("R[j] = chunk[tid];", "R[j] = __builtin_nontemporal_load(&chunk[tid]);")],
}
NT_STORE = {
    "mlp.hip": [
        ("    *reinterpret_cast<f32x4*>(save_s + (int64_t)(g) * 1024 +                                   \\\n                              (save_lane ^ (unsigned)((((g) * 2) & 15) << 4))) = (value)",
         "    __builtin_nontemporal_store((value), reinterpret_cast<f32x4*>(save_s + (int64_t)(g) * 1024 +   \\\n                              (save_lane ^ (unsigned)((((g) * 2) & 15) << 4))))"),
        ("if (SAVE) fsave[(2 * (c0 + g + u) + h) * 32 + (s ^ ((2 * (c0 + g + u) + h) & 15))] = v[u];",
         "if (SAVE) __builtin_nontemporal_store(v[u], &fsave[(2 * (c0 + g + u) + h) * 32 + (s ^ ((2 * (c0 + g + u) + h) & 15))]);"),
        ("if (save_y) save_out[saved_index(2 * group + e_h, e_s)] = y;\n            } else {",
         "if (save_y) __builtin_nontemporal_store(y, &save_out[saved_index(2 * group + e_h, e_s)]);\n            } else {"),
        ("if (MODE == kTrainFwd && save_y) save_out[saved_index(2 * group + e_h, e_s)] = y;",
         "if (MODE == kTrainFwd && save_y) __builtin_nontemporal_store(y, &save_out[saved_index(2 * group + e_h, e_s)]);"),
    ],
    "mlp_bf16.hip": [
        ("fsave[saved_index16(cq, w.s)] = f0;", "__builtin_nontemporal_store(f0, &fsave[saved_index16(cq, w.s)]);"),
        ("fsave[saved_index16(cq + 1, w.s)] = f1;", "__builtin_nontemporal_store(f1, &fsave[saved_index16(cq + 1, w.s)]);"),
        ("save_out[saved_index16(cq, save_s)] = y0;", "__builtin_nontemporal_store(y0, &save_out[saved_index16(cq, save_s)]);"),
        ("save_out[saved_index16(cq + 2, save_s)] = y1;", "__builtin_nontemporal_store(y1, &save_out[saved_index16(cq + 2, save_s)]);"),
    ],
    "mlp_bf16_bwd.hip": [
        ("save_out[saved_index16(cq, w.s)] = y0;", "__builtin_nontemporal_store(y0, &save_out[saved_index16(cq, w.s)]);"),
        ("save_out[saved_index16(cq + 2, w.s)] = y1;", "__builtin_nontemporal_store(y1, &save_out[saved_index16(cq + 2, w.s)]);"),
    ],
}
VARIANTS = {
    "ntload": NT_LOAD,
    "ntstore": NT_STORE,
    "ntboth": {**NT_LOAD, **NT_STORE},
}
