# training forward without its ACTIVATION stores (feature stores and masks stay)
SUBS = [("""            if (TRAIN && L.out_slot >= 0 && live)
                save_out = reinterpret_cast<f32x4*>(w.saved +""", """            if (TRAIN && L.out_slot >= 0 && live && w.num_blocks < 0)
                save_out = reinterpret_cast<f32x4*>(w.saved +""")]
