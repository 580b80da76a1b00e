SUBS = [("    ws_load_chunk<TPW>(w, wreg[P], c2);\n    w.cpos", "    w.cpos")]
