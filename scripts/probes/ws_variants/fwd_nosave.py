# training forward without its activation / feature stores (masks stay): what the stores cost
SUBS = [("__builtin_nontemporal_store(f0, &fsave[saved_index16(cq, w.s)]);", "if (w.num_blocks < 0) fsave[saved_index16(cq, w.s)] = f0;"),
        ("__builtin_nontemporal_store(f1, &fsave[saved_index16(cq + 1, w.s)]);", "if (w.num_blocks < 0) fsave[saved_index16(cq + 1, w.s)] = f1;"),
        ("""            if (TRAIN && L.out_slot >= 0 && live)
                save_out = reinterpret_cast<f32x4*>(w.saved +""", """            if (TRAIN && L.out_slot >= 0 && live && w.num_blocks < 0)
                save_out = reinterpret_cast<f32x4*>(w.saved +""")]
