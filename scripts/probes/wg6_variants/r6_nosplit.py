# timing-only knock-out (WRONG results): wgrad_bf16x6.hip with the three-way split reduced to one packed convert per stage
SUBS = {"wgrad_bf16x6.hip": [("""        if (!LAST) {
            x[2 * t] = pair[0] - __builtin_bit_cast(float, h << 16);
            x[2 * t + 1] = pair[1] - __builtin_bit_cast(float, h & 0xffff0000u);
        }""", "")]}
