# Per-kernel durations of the training microbenchmark (exact-f32 and split-bf16 kernels).
OUT=gpurun_out/wg16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python scripts/microbench_bf16_train.py "$@" > $OUT/log.txt 2>&1
tail -8 $OUT/log.txt
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/wg16/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
