mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q > gpurun_out/r4c/round4.log 2>&1; echo "rc=$?" >> gpurun_out/r4c/round4.log
grep -v "^  \|^$" gpurun_out/r4c/round4.log | tail -40
OUT=gpurun_out/r4c/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python scripts/microbench_bf16_chain.py --models tiny > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4c/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
