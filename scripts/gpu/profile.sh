# A round's measurement artifacts (one GPU), all at ONE commit (run scripts/gpu/stamp.sh first):
# every -m gpu test in exact f32 and again with --precision bf16x6, kernel stats of the bench workload
# and of the training kernels in the three arithmetic modes side by side (exact f32, bf16x3, bf16x6),
# FETCH / WRITE traffic of those kernels (separate passes), SQ counters, the HBM-bound kernels against
# fill / copy, then the default bench line.  --pmc passes carry --kernel-trace only.
#   ROUND=r06 bash scripts/gpu/profile.sh      -> gpurun_out/prof/${ROUND}_*  (copy what is judged into profiles/)
ROUND=${ROUND:-r06}
OUT=gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HEAD=$(cat .git_head 2>/dev/null || echo unknown)
S=$(date +%s); timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/tests.log | tail -12
S=$(date +%s); timeout 1500 python -m pytest tests -m gpu -q --precision bf16x6 -rs > $OUT/tests_bf16x6.log 2>&1; echo "tests --precision bf16x6 rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR\|^SKIPPED" $OUT/tests_bf16x6.log | cut -c1-200 | tail -16
S=$(date +%s); timeout 1500 python -m pytest tests -m gpu -q --precision bf16x3 -rs > $OUT/tests_bf16x3.log 2>&1; echo "tests --precision bf16x3 (NOT an f32-accurate mode: tolerance failures expected) rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed" $OUT/tests_bf16x3.log | tail -2
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-shape --no-config3 --no-config5 --no-skip-leg --no-bf16-leg --no-render"
M="python scripts/microbench_train_kernels.py --iters 3 --modes f32,bf16x3,bf16x6"
rocprofv3 --kernel-trace --stats -d $OUT -o stats --output-format csv -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o train --output-format csv -- $M > $OUT/train.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- $M > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- $M > $OUT/write.log 2>&1
mkdir -p $OUT/bench_traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/bench_traffic -o fetch --output-format csv -- $B > $OUT/bfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/bench_traffic -o write --output-format csv -- $B > $OUT/bwrite.log 2>&1
python scripts/pmc_traffic_summary.py $OUT/bench_traffic $OUT/${ROUND}_hbm_traffic.json $HEAD "$B"
rm -rf $OUT/bench_traffic
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT -o sq --output-format csv -- $M > $OUT/sq.log 2>&1
for n in stats train; do python scripts/kernel_stats_csv.py $OUT/${n}_kernel_stats.csv $OUT/${ROUND}_kernel_stats_${n}.csv; done
python scripts/pmc_traffic_summary.py $OUT $OUT/${ROUND}_hbm_traffic_train_kernels.json $HEAD "$M"
python scripts/pmc_counter_summary.py $OUT/sq_counter_collection.csv $OUT/${ROUND}_sq_counters_train_kernels.json "rocprofv3 --kernel-trace --pmc (8 SQ counters, one pass) on: $M" $HEAD
python scripts/microbench_hbm.py > $OUT/${ROUND}_hbm_microbench.json 2> $OUT/hbm.err
S=$(date +%s); python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s"
rm -f $OUT/*_kernel_trace.csv $OUT/*counter_collection.csv $OUT/*agent_info.csv
export ROUND
python - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/prof/bench.json") if l.startswith("{")][-1])
print("value", b["value"], "ms/step", b["ms_per_step"], "roofline", b["roofline"]["frac"], b["roofline"]["kernel"])
for k in ("f32_accurate_split", "split_bf16_training"):
    v = b.get(k) or {}
    print(k, v.get("train_step_ms_interleaved_runs"), v.get("speedup_vs_exact_f32_step"), {a: c["avg_ms"] for a, c in (v.get("kernels") or {}).items()})
print("default_batch", {k: (v["ms_per_step"], v["per_ray_rate_vs_large_batch"], v["host_enqueue_ms_per_step"]) for k, v in b["default_batch_step"].items() if isinstance(v, dict)})
print("cpu_baseline", {k: b["cpu_baseline"][k] for k in ("value", "cores", "thread_sweep_rays_per_s", "at_8192_rays_per_step_rays_per_s")})
import os
t = json.load(open("gpurun_out/prof/%s_hbm_traffic_train_kernels.json" % os.environ.get("ROUND", "r06")))
print({k: round(v["hbm_bytes"] / 1e9, 2) for k, v in t["kernels"].items()})
PY
