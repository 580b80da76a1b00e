# kernel timeline of one optimisation step at the reference's default batch (1024 x 128): ${1:-tiny}
OUT=gpurun_out/r4t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
M=${1:-tiny}
python scripts/probes/default_batch.py $M 200 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d $OUT/small -o small --output-format csv -- python scripts/probes/default_batch.py $M 200 > $OUT/small.log 2>&1
python - <<PY > $OUT/default_batch_timeline_$M.txt
import csv, glob
f = [p for p in glob.glob("gpurun_out/r4t/small/**/*kernel_trace.csv", recursive=True)][0]
rows = list(csv.DictReader(open(f)))
names = [r["Kernel_Name"].split("(")[0] for r in rows]
idx = [i for i, n in enumerate(names) if "norm_adam" in n]
i0, i1 = idx[100], idx[101]
t0 = int(rows[i0]["End_Timestamp"])
print("# one optimisation step at the reference's default batch (1024 rays x 128 samples, $M model): rocprofv3 --kernel-trace timeline")
print("--- one default-batch step (us since previous step's Adam end): start, duration, kernel")
for r in rows[i0 + 1:i1 + 1]:
    print("%9.1f %8.2f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][:80]))
PY
rm -rf $OUT/small
cat $OUT/default_batch_timeline_$M.txt
