# training forward whose activation / feature stores all land in block 0's rows (they hit the L2
# and never reach HBM): is the cost of the stores their issue or the HBM write stream?
SUBS = [("""                save_out = reinterpret_cast<f32x4*>(w.saved + ch.slot_offset[L.out_slot] * w.num_blocks * 32) +
                           (blk0 + b) * (int64_t)(ch.slot_channels[L.out_slot] * 8);
            unsigned sign_bits = 0u;""", """                save_out = reinterpret_cast<f32x4*>(w.saved + ch.slot_offset[L.out_slot] * w.num_blocks * 32) +
                           ((blk0 + b) & 255) * (int64_t)(ch.slot_channels[L.out_slot] * 8);
            unsigned sign_bits = 0u;"""),
        ("(w.block0 + fb) * (int64_t)(ch.slot_channels[L.save_enc_slot] * 8);", "((w.block0 + fb) & 255) * (int64_t)(ch.slot_channels[L.save_enc_slot] * 8);")]
