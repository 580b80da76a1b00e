# timing-only: no scheduler fences between the phases of the pipelined step
SUBS = {"wgrad_bf16x6.hip": [("#define FFN_FENCE() __builtin_amdgcn_sched_barrier(0)", "#define FFN_FENCE()")]}
