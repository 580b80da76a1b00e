# stall / LDS counters of the training kernels (kernel-trace only next to --pmc)
mkdir -p gpurun_out/pmc3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-render"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc3 -o a --output-format csv -- $B > gpurun_out/pmc3/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/pmc3 -o b --output-format csv -- $B > gpurun_out/pmc3/b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_IFETCH SQ_ACTIVE_INST_FLAT -d gpurun_out/pmc3 -o c --output-format csv -- $B > gpurun_out/pmc3/c.log 2>&1
ls gpurun_out/pmc3
