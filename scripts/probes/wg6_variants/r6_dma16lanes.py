# timing-only knock-out (WRONG results): the bulk loop's LDS-DMA requests of wgrad_bf16x6.hip with 16 of 64 lanes
# active (a quarter of the bytes, the same number of requests)
SUBS = {"wgrad_bf16x6.hip": [("""        if (BULK) {
            issue_stage(i + kStages, std::true_type{});
        } else if (!last) {""", """        if (BULK) {
            if (lane < 16) issue_stage(i + kStages, std::true_type{});
        } else if (!last) {""")]}
