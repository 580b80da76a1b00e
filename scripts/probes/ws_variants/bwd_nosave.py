# backward-data kernels without their dZ stores: the compute part alone (8 vs 16 waves)
SUBS = [("""            if (L.out_slot >= 0 && live)
                save_out = reinterpret_cast<f32x4*>(w.dz + ch.slot_offset[L.out_slot] * w.num_blocks * 32) +""",
         """            if (L.out_slot >= 0 && live && w.num_blocks < 0)
                save_out = reinterpret_cast<f32x4*>(w.dz + ch.slot_offset[L.out_slot] * w.num_blocks * 32) +""")]
