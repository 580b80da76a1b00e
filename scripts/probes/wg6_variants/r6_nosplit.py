# timing-only knock-out (WRONG results): wgrad_bf16x6.hip with the three-way split reduced to one packed convert per stage
SUBS = {"wgrad_bf16x6.hip": [("""        if (!LAST) {
            f32x2v even, odd;   // the parts of sample 2t / 2t + 1 as f32
            even[0] = __builtin_bit_cast(float, h0 << 16);
            even[1] = __builtin_bit_cast(float, h1 << 16);
            odd[0] = __builtin_bit_cast(float, h0 & 0xffff0000u);
            odd[1] = __builtin_bit_cast(float, h1 & 0xffff0000u);
            x[2 * t] -= even;
            x[2 * t + 1] -= odd;
        }""", "")]}
