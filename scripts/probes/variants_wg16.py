# knock-outs of the LDS-DMA / pipelined split-bf16 weight-gradient kernel (timing only)
NOSPLIT = [("#define FFN_SPLIT_A(P) split_component<P>(v, nxt.ah[P], nxt.al[P])", "#define FFN_SPLIT_A(P) nxt.ah[P][0] = (__bf16)v[P][0]"),
           ("#define FFN_SPLIT_B(P) split_component<P>(v, nxt.bh[P], nxt.bl[P])", "#define FFN_SPLIT_B(P) nxt.bh[P][0] = (__bf16)v[P][0]")]
NODMA = [("            issue_stage(i + kStages, std::true_type{});\n", "")]
NOMFMA = [("        acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.X[0], cur.Y[q], acc[0][q], 0, 0, 0);  \\", "        acc[0][q][0] += (float)cur.X[0][0] * (float)cur.Y[q][0];  \\"),
          ("        acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.X[1], cur.Y[q], acc[1][q], 0, 0, 0);  \\", "        acc[1][q][0] += (float)cur.X[1][0] * (float)cur.Y[q][0];  \\"),
          ("        acc[2][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.X[2], cur.Y[q], acc[2][q], 0, 0, 0);  \\", "        acc[2][q][0] += (float)cur.X[2][0] * (float)cur.Y[q][0];  \\"),
          ("        acc[3][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.X[3], cur.Y[q], acc[3][q], 0, 0, 0);  \\", "        acc[3][q][0] += (float)cur.X[3][0] * (float)cur.Y[q][0];  \\")]
VARIANTS = {
    "wg16_nosplit": {"wgrad_bf16.hip": NOSPLIT},
    "wg16_nodma": {"wgrad_bf16.hip": NODMA},
    "wg16_nosplit_nodma": {"wgrad_bf16.hip": NOSPLIT + NODMA},
    "wg16_nomfma": {"wgrad_bf16.hip": NOMFMA},
}
