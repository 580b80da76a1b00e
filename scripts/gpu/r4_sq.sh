# SQ counters of the split-bf16 chain kernels (tiny model, 2^22 samples): ring vs two-waves-per-SIMD
OUT=gpurun_out/r4sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HEAD=$(cat .git_head 2>/dev/null || echo unknown)
for which in ring ws; do
  FFN_BF16_KERNELS=$which rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT -o sq_$which --output-format csv -- python scripts/microbench_bf16_chain.py --models tiny > $OUT/sq_$which.log 2>&1
  python scripts/pmc_counter_summary.py $OUT/sq_${which}_counter_collection.csv $OUT/sq_$which.json "rocprofv3 --kernel-trace --pmc (8 SQ counters) on: FFN_BF16_KERNELS=$which python scripts/microbench_bf16_chain.py --models tiny" $HEAD
  FFN_BF16_KERNELS=$which rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT -o sq2_$which --output-format csv -- python scripts/microbench_bf16_chain.py --models tiny > $OUT/sq2_$which.log 2>&1
  python scripts/pmc_counter_summary.py $OUT/sq2_${which}_counter_collection.csv $OUT/sq2_$which.json "second pass: wait / LDS counters" $HEAD
done
rm -f $OUT/*_kernel_trace.csv $OUT/*counter_collection.csv
python - <<'PY'
import json
for which in ("ring", "ws"):
    for f in ("sq", "sq2"):
        try:
            d = json.load(open("gpurun_out/r4sq/%s_%s.json" % (f, which)))
        except Exception as e:
            print(f, which, "missing", e); continue
        for k, v in d["kernels"].items():
            if "bf16" in k:
                print(which, f, k[:60], {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "counters"})
                print("    ", {a: round(b / 1e6, 2) for a, b in v.get("counters", {}).items()})
PY
