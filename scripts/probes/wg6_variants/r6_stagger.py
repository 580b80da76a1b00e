# experiment: the four waves of a workgroup request their pieces of stage i + 4 at FOUR DIFFERENT points
# of the step (wave w at the boundary behind product group 1 / 2 / 4 / 5) instead of all at once
SUBS = {"wgrad_bf16x6.hip": [("""        FFN_ROWSET(l, h, 0)
        if (BULK) {
            issue_stage(i + kStages, std::true_type{});
        } else if (!last) {
            if (i + kStages < steps) issue_stage(i + kStages, std::false_type{});
        }
""", """        auto dma_turn = [&](int turn) {
            if (wave == turn) {
                if (BULK) {
                    issue_stage(i + kStages, std::true_type{});
                } else if (!last) {
                    if (i + kStages < steps) issue_stage(i + kStages, std::false_type{});
                }
            }
        };
        FFN_ROWSET(l, h, 0)
"""), ("""        FFN_PIN16(6)
        FFN_FENCE();
        // a_m b_h, with the lo parts of B behind it""", """        FFN_PIN16(6)
        FFN_FENCE();
        dma_turn(0);
        // a_m b_h, with the lo parts of B behind it"""), ("""        FFN_PIN16(2)
        FFN_FENCE();""", """        FFN_PIN16(2)
        FFN_FENCE();
        dma_turn(1);"""), ("""        FFN_ROWSET(m, m, 3)
        FFN_PIN16(6)
        FFN_FENCE();""", """        FFN_ROWSET(m, m, 3)
        FFN_PIN16(6)
        FFN_FENCE();
        dma_turn(2);"""), ("""        FFN_ROWSET(h, m, 3)
        FFN_PIN16(5)
        FFN_FENCE();""", """        FFN_ROWSET(h, m, 3)
        FFN_PIN16(5)
        FFN_FENCE();
        dma_turn(3);""")]}
