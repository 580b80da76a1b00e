# What would a training step move and cost that neither saves nor re-reads the encoding features?
# Timing-only variant (scripts/probes/traffic_variants/nofeat.py; wrong gradients): kernel times
# interleaved with the product on one box, then FETCH / WRITE passes (separate) of both.
OUT=gpurun_out/r4nf
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
M="timeout 150 python scripts/microbench_train_kernels.py --iters 3"
V=$PWD/scripts/probes/variants/libffn_nofeat.so
echo "== nofeat first"; FFN_HIP_LIBRARY=$V $M 2>&1 | tail -1 | grep -q sum_ms || { echo "variant failed"; exit 1; }
for rep in 0 1; do
  echo "== product rep $rep"; $M 2>&1 | tail -1
  echo "== nofeat rep $rep"; FFN_HIP_LIBRARY=$V $M 2>&1 | tail -1
done
for which in nofeat; do
  if [ $which = product ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$V; fi
  mkdir -p $OUT/$which
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/$which -o fetch --output-format csv -- $M > $OUT/$which/fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/$which -o write --output-format csv -- $M > $OUT/$which/write.log 2>&1
  python - <<PY
import sys
sys.path.insert(0, "scripts")
import pmc_traffic_summary as p
p.main("$OUT/$which", "$OUT/traffic_$which.json", open(".git_head").read().split()[0], command="python scripts/microbench_train_kernels.py --iters 3 ($which library)")
import json
d = json.load(open("$OUT/traffic_$which.json"))
for k, v in d["kernels"].items():
    if "mlp_" in k or "wgrad_unit" in k:
        print("%-70s %8.2f GB" % (k[:70], v["hbm_bytes"] / 1e9))
PY
  rm -f $OUT/$which/*_counter_collection.csv $OUT/$which/*kernel_trace.csv
done
