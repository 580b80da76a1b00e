# timing-only knock-out (WRONG results): the pipelined step of wgrad_bf16x6.hip without its workgroup barrier
SUBS = {"wgrad_bf16x6.hip": [("""            asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if (!last) {""", """            asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        } else if (!last) {""")]}
