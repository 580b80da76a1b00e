SUBS = [("acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[t][0], x[b][1], acc[t][b], 0, 0, 0);", "acc[t][b][0] += (float)wk[t][0][0] * (float)x[b][1][0];"),
        ("acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[t][1], x[b][0], acc[t][b], 0, 0, 0);", "acc[t][b][1] += (float)wk[t][1][0] * (float)x[b][0][0];"),
        ("acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[t][0], x[b][0], acc[t][b], 0, 0, 0);", "acc[t][b][2] += (float)wk[t][0][1] * (float)x[b][0][1];")]
