mkdir -p gpurun_out/r3s
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r3s/prof -o small --output-format csv -- python scripts/probes/default_batch.py tiny 200 > gpurun_out/r3s/prof.log 2>&1
tail -1 gpurun_out/r3s/prof.log | cut -c1-300
find gpurun_out/r3s/prof -type f | head
python - <<'PY'
import csv, glob, collections
f = [p for p in glob.glob("gpurun_out/r3s/prof/**/*kernel_trace.csv", recursive=True)][0]
rows = list(csv.DictReader(open(f)))
# take the 200 timed small steps: find the window by kernel order -- aggregate everything, then report per-launch averages
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-72s calls %6d  total %9.0f us  avg %8.2f us  %5.1f%%" % (k, a[0], a[1], a[1] / a[0], 100 * a[1] / tot))
# timeline of one small step in the middle: the kernels between two consecutive clip_value launches
names = [r["Kernel_Name"].split("(")[0] for r in rows]
idx = [i for i, n in enumerate(names) if "norm_adam" in n]
i0, i1 = idx[100], idx[101]
t0 = int(rows[i0]["End_Timestamp"])
print("--- one default-batch step (us since previous step's Adam end): start, duration, kernel")
for r in rows[i0 + 1:i1 + 1]:
    print("%9.1f %8.2f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][:80]))
PY
