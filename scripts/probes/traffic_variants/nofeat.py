# TIMING / TRAFFIC ONLY (wrong gradients): a training step that neither SAVES the encoding
# features in the forward pass nor STREAMS them in the weight-gradient units (the units of an
# encoding window re-read their first block: L2 hits) -- the HBM traffic and the kernel times a
# step with regenerated features could at best have (the generation itself is not in it).
SUBS = {
    "mlp.hip": [
        ("""                save = slab_block(ch, slab_out, L.save_enc_slot, w) + c0 * 64;   // 1 KiB per K group""",
         """                save = reinterpret_cast<f32x4*>(slab_out + ch.slot_offset[L.save_enc_slot] * w.slab_blocks * 32) +
                       (w.block & 255) * (int64_t)(ch.slot_channels[L.save_enc_slot] * 8) + c0 * 64;   // (L2-resident rows)"""),
        ("""            } else if (MODE == kTrainFwd && L.save_enc_slot >= 0 && w.active) {
                generate_features<true>(""", """            } else if (MODE == kTrainFwd && L.save_enc_slot >= 0 && w.active && w.slab_blocks < 0) {
                generate_features<true>("""),
    ],
    "wgrad.hip": [
        ("""    const int64_t b_stride = (int64_t)ch.slot_channels[unit.n_slot] * 128;
    // wave-uniform base""", """    const int64_t b_stride = unit.n_slot >= ch.num_slots ? 0 : (int64_t)ch.slot_channels[unit.n_slot] * 128;
    // wave-uniform base"""),
    ],
    "mlp_bf16_ws.hip": [
        ("""        if (TRAIN && L.save_enc_slot >= 0 && w.block0 + fb < w.num_blocks)""",
         """        if (TRAIN && L.save_enc_slot >= 0 && w.block0 + fb < w.num_blocks && w.num_blocks < 0)"""),
    ],
    "wgrad_bf16.hip": [
        ("""    const int64_t b_stride = (int64_t)ch.slot_channels[unit.n_slot] * 128;
    const char* a_base""", """    const int64_t b_stride = unit.n_slot >= ch.num_slots ? 0 : (int64_t)ch.slot_channels[unit.n_slot] * 128;
    const char* a_base"""),
    ],
}
