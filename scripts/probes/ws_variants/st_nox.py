exec(open('scripts/probes/ws_variants/stamps.py').read())
SUBS = SUBS + [("""    for (int b = 0; b < NB; ++b) x[b][1] = __builtin_bit_cast(bf16x8, p[b * 128 + 64]);""", """    (void)p;"""),
               ("""        if (HB) x[b][0] = __builtin_bit_cast(bf16x8, p[b * 128]);
        else xh[b] = __builtin_bit_cast(bf16x8, p[b * 128]);""", """        (void)p;""")]
