"""Knock-outs of the exact-f32 K loop (timing only, results are wrong): what do the weight loads
from L2 and the operand reads from LDS cost on top of the 64-cycle matrix instructions?
   python scripts/probes/make_variants.py scripts/probes/variants_kloop.py"""
NOLOAD = ("    for (int o = 0; o < OT; ++o) a[o] = (o < 4 ? lo : hi)[(o & 3) * 64 + lane];\n}",
          "    for (int o = 0; o < OT; ++o) if (lane < 0) a[o] = (o < 4 ? lo : hi)[(o & 3) * 64 + lane];\n}")
NOLDS = [("            x2 = xa[128];\n            x3 = xa[192];\n", "            x2 = x0;\n            x3 = x1;\n"),
         ("            if (g + 4 < count) {\n                x0 = xa[0];\n                x1 = xa[64];\n            }\n",
          "            asm volatile(\"\" : \"+v\"(x0), \"+v\"(x1));\n")]
VARIANTS = {
    "kloop_noload": {"mlp.hip": [NOLOAD]},
    "kloop_nolds": {"mlp.hip": NOLDS},
    "kloop_neither": {"mlp.hip": [NOLOAD] + NOLDS},
}
