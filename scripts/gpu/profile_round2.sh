# Round-2 measurement artifacts (one GPU): kernel stats, HBM traffic (FETCH / WRITE in separate
# passes), SQ counters, HBM micro-benchmarks.  --pmc passes carry --kernel-trace only.
OUT=gpurun_out/prof2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-shape --no-config3 --no-config5 --no-skip-leg --no-bf16-leg"
rocprofv3 --kernel-trace --stats -d $OUT -o stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- $B > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT -o sq --output-format csv -- $B > $OUT/sq.log 2>&1
python scripts/microbench_hbm.py > $OUT/hbm_microbench.json 2> $OUT/hbm_microbench.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT -o sqhbm --output-format csv -- python scripts/microbench_hbm.py > $OUT/sqhbm.log 2>&1
ls -la $OUT | head -40
tail -2 $OUT/sq.log | cut -c1-300
