# lever (a), variant 1: the training forward's two save stores of a half-trip issued BEHIND its
# matrix instructions (pinned last) instead of in front of them
SUBS = [("""    if (SAVES) __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
#pragma unroll
    for (int i = 0; i < 2 * OT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }""", """#pragma unroll
    for (int i = 0; i < 2 * OT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    if (SAVES) __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);""")]
