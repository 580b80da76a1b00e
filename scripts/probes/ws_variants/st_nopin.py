exec(open('scripts/probes/ws_variants/stamps.py').read())
SUBS = SUBS + [("""        ws_pin_kblock<TPW, NT, 0>();
        ws_pin_kblock<TPW, NT, 2 * TPW>();""", "")]
