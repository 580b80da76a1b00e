SUBS = [("""    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// the wave's slice""", """    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// the wave's slice""")]
