# lever (a), variant 2: the two save stores of a half-trip spread into the MIDDLE of its matrix
# instructions (one after a quarter, one after half), loads around them
SUBS = [("""    if (SAVES) __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);
#pragma unroll
    for (int i = 0; i < 2 * OT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }""", """#pragma unroll
    for (int i = 0; i < 2 * OT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (SAVES && (i == OT / 2 || i == OT)) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
    }""")]
