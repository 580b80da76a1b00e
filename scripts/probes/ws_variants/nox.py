SUBS = [("    ws_read_x<TPW>(w, xb[1], g + 1);\n", ""), ("    ws_read_x<TPW>(w, xb[0], nxt);\n", "")]
