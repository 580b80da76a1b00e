# timing-only: fewer vector instructions per matrix instruction in the pins (4 / 2 / 5 / 4)
SUBS = {"wgrad_bf16x6.hip": [("FFN_PIN16(6)\n        FFN_FENCE();\n        // a_m b_h", "FFN_PIN16(5)\n        FFN_FENCE();\n        // a_m b_h"),
                             ("FFN_PIN16(8)", "FFN_PIN16(7)"),
                             ("FFN_ROWSET(h, m, 3)\n        FFN_PIN16(6)", "FFN_ROWSET(h, m, 3)\n        FFN_PIN16(5)")]}
