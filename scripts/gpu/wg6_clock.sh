# SQ counters (busy cycles, matrix pipe busy, clock) of the bf16x6 weight-gradient kernel under the in-tree build
# and knock-out variants: does a knock-out save CYCLES or buy CLOCK?   LIBS="new wg6_nodma ..." bash scripts/gpu/wg6_clock.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wg6clock; mkdir -p $OUT
for lib in ${LIBS:-new wg6_nodma wg6_dma16lanes wg6_nosplit}; do
  if [ $lib = new ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_$lib.so; fi
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT -o $lib --output-format csv -- python scripts/microbench_train_kernels.py --iters 3 --modes bf16x6 > $OUT/$lib.log 2>&1
  python scripts/pmc_counter_summary.py $OUT/${lib}_counter_collection.csv $OUT/$lib.json "variant $lib" > /dev/null
  python - <<PY
import json
d = json.load(open("$OUT/$lib.json"))["kernels"]
for k, v in d.items():
    if "wgrad_unit_bf16x6" in k:
        print("$lib", {a: v[a] for a in ("avg_us_under_pmc", "mfma_busy_frac", "effective_clock_ghz", "valu_insts_per_mfma_inst", "issue_stall_frac_of_wave_cycles")}, "busy cycles per SE", round(v["counters"]["SQ_BUSY_CYCLES"] / 32))
PY
done
rm -f $OUT/*_kernel_trace.csv $OUT/*counter_collection.csv $OUT/*agent_info.csv
