SUBS = [("""                if (!last_step) {
                    bf16x8 yh, yl;
                    split8(y, yh, yl);
                    f32x4* dst = w.xbuf + ws_x_index<TPW>(2 * o + half, b, 0) + e_lane;
                    dst[0] = __builtin_bit_cast(f32x4, yh);
                    dst[64] = __builtin_bit_cast(f32x4, yl);
                }
            }
            if (TRAIN && L.relu""", """                if (!last_step && y[0] == 123.456f) {
                    bf16x8 yh, yl;
                    split8(y, yh, yl);
                    f32x4* dst = w.xbuf + ws_x_index<TPW>(2 * o + half, b, 0) + e_lane;
                    dst[0] = __builtin_bit_cast(f32x4, yh);
                    dst[64] = __builtin_bit_cast(f32x4, yl);
                }
            }
            if (TRAIN && L.relu""")]
