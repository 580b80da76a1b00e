mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum"
rocprofv3 --kernel-trace --pmc $P -d gpurun_out/pmc -o probe --output-format csv -- env PROBE_RANDOM=1 ./scripts/probes/stream_read_probe > gpurun_out/pmc/probe.log 2>&1
rocprofv3 --kernel-trace --pmc $P -d gpurun_out/pmc -o wg --output-format csv -- python scripts/probes/wgrad_alone.py > gpurun_out/pmc/wg.log 2>&1
tail -2 gpurun_out/pmc/wg.log | cut -c1-200
python - <<'PY'
import csv, collections
for f in ("gpurun_out/pmc/probe_counter_collection.csv", "gpurun_out/pmc/wg_counter_collection.csv"):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        a = acc.setdefault(k, {"n": set(), "ns": 0})
        if r["Dispatch_Id"] not in a["n"]:
            a["n"].add(r["Dispatch_Id"]); a["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, a in acc.items():
        n = len(a["n"])
        if a.get("TCP_TCC_READ_REQ_sum", 0) < 1e6: continue
        print("%-60s n=%d %.2f ms  L1->L2 read latency %.0f cyc  EA read latency(level/req) %.0f  L2 hit %.3f  EA rdreq/launch %.3g"
              % (k, n, a["ns"] / n / 1e6, a["TCP_TCC_READ_REQ_LATENCY_sum"] / a["TCP_TCC_READ_REQ_sum"],
                 a["TCC_EA0_RDREQ_LEVEL_sum"] / max(a["TCC_EA0_RDREQ_sum"], 1), a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a["TCC_MISS_sum"], 1),
                 a["TCC_EA0_RDREQ_sum"] / n))
PY
