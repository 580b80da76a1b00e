# timing-only: no issue-order pins in the pipelined step of wgrad_bf16x6.hip
SUBS = {"wgrad_bf16x6.hip": [("""        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \\
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                                        \\""", """        \\""")]}
