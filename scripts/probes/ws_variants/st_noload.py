exec(open('scripts/probes/ws_variants/stamps.py').read())
SUBS = SUBS + [("    ws_load_kblock<TPW>(w, wreg[P][0], c2, 0);\n", ""), ("    ws_load_kblock<TPW>(w, wreg[P][1], c2, 1);\n", "")]
