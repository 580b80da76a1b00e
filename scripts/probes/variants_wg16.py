# knock-outs of the split-bf16 weight-gradient kernel: what bounds it?  (timing only)
NOCONTRACT = ("            const bool mine = ks >= ks_begin && ks < ks_end;        // (wave-uniform)",
              "            const bool mine = false && ks >= ks_begin && ks < ks_end;")
NODEPOSIT = ("            for (int j = 0; j < NST; ++j) FFN_DEPOSIT(1 - CUR, j);\n            generate(1 - CUR);",
             "            for (int j = 0; j < NST; ++j) asm volatile(\"\" :: \"v\"(R[j]));\n            generate(1 - CUR);")
NOREQUEST = ("            for (int j = 0; j < NST; ++j) FFN_REQUEST(j);\n            load_point(blk_of_body + 2);",
             "            for (int j = 0; j < NST; ++j) asm volatile(\"\" : \"+v\"(R[j]));\n            load_point(blk_of_body + 2);")
NOBARRIER = ("        a_s += a_stride;\n        b_s += b_stride;\n        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n        __builtin_amdgcn_s_barrier();\n    };",
             "        a_s += a_stride;\n        b_s += b_stride;\n        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n    };")
VARIANTS = {
    "wg16_nc_nodep": {"wgrad_bf16.hip": [NOCONTRACT, NODEPOSIT]},
    "wg16_nc_noreq": {"wgrad_bf16.hip": [NOCONTRACT, NOREQUEST]},
    "wg16_nc_nodep_noreq": {"wgrad_bf16.hip": [NOCONTRACT, NODEPOSIT, NOREQUEST]},
}
