# Round-3 measurement artifacts (one GPU), all at the SAME commit: kernel stats (bench workload, the
# north-star full-NeRF launch, the config-5 step, the exact / split-bf16 training kernels side by
# side), HBM traffic (FETCH / WRITE in separate passes; f32 bench and the split-bf16 kernels),
# SQ counters, HBM micro-benchmarks.  --pmc passes carry --kernel-trace only.
OUT=gpurun_out/prof3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-shape --no-config3 --no-config5 --no-skip-leg --no-bf16-leg"
M="python scripts/microbench_train_kernels.py --iters 3"
N="python scripts/microbench_train_kernels.py --model nerf --rays 65536 --samples 128 --iters 2 --modes f32"
C="python bench.py --model gaussian512 --rays 32768 --samples 128 --size 800 --cameras 25 --steps 3 --warmup 1 --no-cpu-baseline --no-render"
rocprofv3 --kernel-trace --stats -d $OUT -o stats --output-format csv -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o train --output-format csv -- $M > $OUT/train.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o northstar --output-format csv -- $N > $OUT/northstar.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o config5 --output-format csv -- $C > $OUT/config5.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- $B > $OUT/write.log 2>&1
mkdir -p $OUT/bf16
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/bf16 -o fetch --output-format csv -- $M > $OUT/bf16/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/bf16 -o write --output-format csv -- $M > $OUT/bf16/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT -o sq --output-format csv -- $B > $OUT/sq.log 2>&1
python scripts/microbench_hbm.py > $OUT/hbm_microbench.json 2> $OUT/hbm_microbench.err
HEAD=$(cat .git_head 2>/dev/null || echo unknown)
for n in stats train northstar config5; do python scripts/kernel_stats_csv.py $OUT/${n}_kernel_stats.csv $OUT/r03_kernel_stats_${n}.csv; done
python scripts/pmc_traffic_summary.py $OUT $OUT/r03_hbm_traffic.json $HEAD
python scripts/pmc_traffic_summary.py $OUT/bf16 $OUT/r03_hbm_traffic_train_kernels_f32_and_bf16x3.json $HEAD "$M"
python scripts/pmc_counter_summary.py $OUT/sq_counter_collection.csv $OUT/r03_sq_counters.json "rocprofv3 --kernel-trace --pmc (8 SQ counters, one pass) on: $B" $HEAD
rm -f $OUT/*_kernel_trace.csv $OUT/*counter_collection.csv $OUT/bf16/*counter_collection.csv $OUT/bf16/*_kernel_trace.csv
ls $OUT | head -40
head -8 $OUT/r03_kernel_stats_stats.csv; head -6 $OUT/r03_kernel_stats_northstar.csv; head -6 $OUT/r03_kernel_stats_config5.csv; head -9 $OUT/r03_kernel_stats_train.csv
python - <<'PY'
import json
for f in ("gpurun_out/prof3/r03_hbm_traffic.json", "gpurun_out/prof3/r03_hbm_traffic_train_kernels_f32_and_bf16x3.json"):
    d = json.load(open(f)); print(f, d["training_step_mlp_kernels_hbm_bytes"] / 1e9, {k: round(v["hbm_bytes"] / 1e9, 2) for k, v in d["kernels"].items() if v["hbm_bytes"] > 1e8})
PY
