# PMC passes over the hidden-layer micro-benchmark (no kernel-trace domains besides kernel-trace)
mkdir -p gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc2 -o a --output-format csv -- python scripts/microbench_mlp.py --iters 1 --model raw --hidden 9 > gpurun_out/pmc2/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR -d gpurun_out/pmc2 -o b --output-format csv -- python scripts/microbench_mlp.py --iters 1 --model raw --hidden 9 > gpurun_out/pmc2/b.log 2>&1
ls gpurun_out/pmc2
