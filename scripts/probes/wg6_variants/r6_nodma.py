# timing-only knock-out (WRONG results): the pipelined step of wgrad_bf16x6.hip without the LDS-DMA requests of its bulk loop
SUBS = {"wgrad_bf16x6.hip": [("""        if (BULK) {
            issue_stage(i + kStages, std::true_type{});
        } else if (!last) {""", """        if (BULK) {
        } else if (!last) {""")]}
