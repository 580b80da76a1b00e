# Round-4 measurement artifacts (one GPU), all at the SAME commit: the default bench line, kernel
# stats (bench workload, north-star full-NeRF launch, config-5 step, exact / split-bf16 training
# kernels side by side), HBM traffic (FETCH / WRITE in separate passes), SQ counters of the bench
# and of the split-bf16 chain kernels (ring vs two-waves-per-SIMD), the default-batch timeline.
# --pmc passes carry --kernel-trace only.
OUT=gpurun_out/prof4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HEAD=$(cat .git_head 2>/dev/null || echo unknown)
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
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT -o sq --output-format csv -- $B > $OUT/sq.log 2>&1
for n in stats train northstar config5; do python scripts/kernel_stats_csv.py $OUT/${n}_kernel_stats.csv $OUT/r04_kernel_stats_${n}.csv; done
python scripts/pmc_traffic_summary.py $OUT $OUT/r04_hbm_traffic.json $HEAD
# the default bench line AFTER the traffic passes, so that its roofline.traffic comes from this call
cp $OUT/r04_hbm_traffic.json profiles/r04_hbm_traffic.json
S=$(date +%s); python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s"
python scripts/pmc_counter_summary.py $OUT/sq_counter_collection.csv $OUT/r04_sq_counters.json "rocprofv3 --kernel-trace --pmc (8 SQ counters, one pass) on: $B" $HEAD
# the split-bf16 chain kernels: both organisations, tiny and full NeRF, two counter passes each
for which in ring ws; do
  K="python scripts/microbench_bf16_chain.py"
  FFN_BF16_KERNELS=$which rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT -o sq16_$which --output-format csv -- $K > $OUT/sq16_$which.log 2>&1
  python scripts/pmc_counter_summary.py $OUT/sq16_${which}_counter_collection.csv $OUT/r04_sq_counters_bf16_$which.json "FFN_BF16_KERNELS=$which: rocprofv3 --kernel-trace --pmc (8 SQ counters) on: $K (tiny NeRF then full NeRF, 2^22 samples: the per-kernel means mix both models; per-model numbers in r04_bf16_chain_kernels.json)" $HEAD
  FFN_BF16_KERNELS=$which rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT -o sq16b_$which --output-format csv -- $K --models tiny > $OUT/sq16b_$which.log 2>&1
  python scripts/pmc_counter_summary.py $OUT/sq16b_${which}_counter_collection.csv $OUT/r04_sq_counters_bf16_${which}_waits.json "FFN_BF16_KERNELS=$which: wait / LDS counters, tiny NeRF" $HEAD
done
python - <<'PY' > gpurun_out/prof4/r04_bf16_chain_kernels.json
import json, os, subprocess, sys
out = {"what": "split-bf16 chain kernels alone, 2^22 samples, interleaved ring / ws (8 waves) / ws (16 waves); ms",
       "commit": open(".git_head").read().split()[0] if os.path.exists(".git_head") else None, "runs": []}
for rep in range(2):
    for which, waves in (("ring", None), ("ws", "8"), ("ws", "16")):
        env = dict(os.environ, FFN_BF16_KERNELS=which)
        if waves:
            env["FFN_BF16_WAVES"] = waves
        r = subprocess.run([sys.executable, "scripts/microbench_bf16_chain.py"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            row = json.loads(line[-1]); row["waves"] = waves; row["rep"] = rep
            out["runs"].append(row)
print(json.dumps(out, indent=1))
PY
python scripts/microbench_hbm.py > $OUT/hbm_microbench.json 2> $OUT/hbm_microbench.err
# kernel timeline of one optimisation step at the reference's default batch (1024 x 128, tiny NeRF)
rocprofv3 --kernel-trace --stats -d $OUT/small -o small --output-format csv -- python scripts/probes/default_batch.py tiny 200 > $OUT/small.log 2>&1
python - <<'PY' > gpurun_out/prof4/r04_default_batch_timeline.txt
import csv, glob
f = [p for p in glob.glob("gpurun_out/prof4/small/**/*kernel_trace.csv", recursive=True)][0]
rows = list(csv.DictReader(open(f)))
names = [r["Kernel_Name"].split("(")[0] for r in rows]
idx = [i for i, n in enumerate(names) if "norm_adam" in n]
i0, i1 = idx[100], idx[101]
t0 = int(rows[i0]["End_Timestamp"])
print("# one optimisation step at the reference's default batch (1024 rays x 128 samples, tiny NeRF): rocprofv3 --kernel-trace timeline")
print("# (scripts/gpu/profile_round4.sh; ideal at the large-batch rate: forward 428, backward data 206, weight gradients 384 us)")
print("--- one default-batch step (us since previous step's Adam end): start, duration, kernel")
for r in rows[i0 + 1:i1 + 1]:
    print("%9.1f %8.2f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][:80]))
PY
rm -rf $OUT/small
cat $OUT/r04_default_batch_timeline.txt | tail -22
rm -f $OUT/*_kernel_trace.csv $OUT/*counter_collection.csv
ls $OUT | head -50
head -8 $OUT/r04_kernel_stats_stats.csv; head -6 $OUT/r04_kernel_stats_northstar.csv; head -6 $OUT/r04_kernel_stats_config5.csv; head -9 $OUT/r04_kernel_stats_train.csv
python - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/prof4/bench.json").read().strip().split("\n") if l.startswith("{")][-1])
print("ms/step", b["ms_per_step"], "value", b["value"], "roofline", b["roofline"]["frac"])
for k in ("render", "north_star_shape", "config3_step", "default_batch_step", "split_bf16_inference", "split_bf16_training", "cpu_baseline"):
    v = b.get(k)
    if isinstance(v, dict):
        v = {a: (c if not isinstance(c, dict) else "{...}") for a, c in v.items() if a not in ("workload", "label", "path", "includes", "sample", "metric")}
    print(k, v)
c5 = b["config5_step"]
print({k: (v if not isinstance(v, dict) else {a: c for a, c in v.items() if a != "kernels"}) for k, v in c5.items() if k != "workload"})
d = json.load(open("gpurun_out/prof4/r04_hbm_traffic.json")); print("traffic", d["training_step_mlp_kernels_hbm_bytes"] / 1e9)
PY
