# Scaling curve of the headline step in ONE call: N = 1, 2, 4, 8 ranks back to back (as many as
# the node has GPUs; FFN_BENCH_SHARE_GPU=1 runs every N on cuda:0 over gloo -- a functional check of
# the sharding / collective / timing fields on a one-GPU box, labelled so, never a measurement).
# Writes gpurun_out/scale/bench_N.json per N and gpurun_out/scale/scale_curve.json
# (copy to profiles/rNN_scale_curve.json): value, ms/step, efficiency vs N = 1, the collective
# block (avg/max us, share of the step, per-rank max/min step time) per N.
# Both scaling modes (SCALINGS="weak strong"): weak = RAYS per GPU per step (the driver's
# contract), strong = ONE global batch of RAYS rays per step sharded over the ranks.
#   RAYS=65536 STEPS=10 WARMUP=2 NS="1 2 4 8" SCALINGS="weak strong" bash scripts/gpu/scale.sh
mkdir -p gpurun_out/scale
RAYS=${RAYS:-65536}; STEPS=${STEPS:-10}; WARMUP=${WARMUP:-2}; NS=${NS:-"1 2 4 8"}; SCALINGS=${SCALINGS:-"weak strong"}
GPUS=$(python -c "import torch; print(torch.cuda.device_count())")
for MODE in $SCALINGS; do
for N in $NS; do
  if [ "$N" -gt "$GPUS" ] && [ "${FFN_BENCH_SHARE_GPU:-0}" != "1" ]; then echo "N=$N: only $GPUS GPU(s), skipped"; continue; fi
  EXTRA=""
  if [ "$N" -gt 1 ]; then EXTRA="--no-render"; fi
  S=$(date +%s)
  if [ "$N" = 1 ] && [ "$MODE" = strong ] && [ -f gpurun_out/scale/bench_weak_1.json ]; then
    cp gpurun_out/scale/bench_weak_1.json gpurun_out/scale/bench_strong_1.json; continue   # (N = 1: the same run)
  fi
  python bench.py --gpus $N --steps $STEPS --warmup $WARMUP --rays $RAYS --scaling $MODE --no-target-shape --no-config3 \
      --no-config5 --no-bf16-leg --no-skip-leg $EXTRA > gpurun_out/scale/bench_${MODE}_$N.json 2> gpurun_out/scale/bench_${MODE}_$N.err
  echo "$MODE N=$N rc=$? wall=$(( $(date +%s) - S ))s"
done
done
python - <<'PY'
import glob, json, os
curves = {}
for mode in ("weak", "strong"):
  rows = {}
  for path in sorted(glob.glob("gpurun_out/scale/bench_%s_*.json" % mode)):
    lines = [l for l in open(path).read().strip().split("\n") if l.startswith("{")]
    if not lines:
        continue
    b = json.loads(lines[-1])
    rows[b["n_gpus"]] = b
  base = rows.get(1)
  curve = curves.setdefault(mode, [])
  for n in sorted(rows):
    b = rows[n]
    col = b.get("collective") or {}
    curve.append({
        "n_gpus": n, "rays_per_s": b["value"], "ms_per_step": b["ms_per_step"],
        "global_batch_rays": b["config"].get("global_batch_rays"), "rays_per_gpu": b["config"].get("rays_per_gpu"),
        "speedup_vs_1": None if base is None else b["value"] / base["value"],
        "efficiency_vs_1": None if base is None else b["value"] / base["value"] / n,
        "scaling": b["scaling"], "backend": col.get("backend"), "shared_gpu": col.get("shared_gpu"),
        "all_reduce_span_incl_overlapped_sampling_avg_us": col.get("span_issue_to_wait_avg_us"),
        "sampling_under_the_collective_avg_us": col.get("sampling_under_the_collective_avg_us"),
        "launch_stream_waited_for_all_reduce_avg_us": col.get("launch_stream_waited_avg_us"),
        "launch_stream_waited_for_all_reduce_max_us": col.get("launch_stream_waited_max_us"),
        "waited_frac_of_step": col.get("waited_frac_of_step"), "rank_step_ms": col.get("rank_step_ms"),
        "placement": col.get("placement"),
        "cpu_baseline_source": (b.get("cpu_baseline") or {}).get("source"),
        "commit": b["config"].get("commit")})
out = {"what": "scaling of the headline training step: weak = RAYS (65 536) rays per GPU per step, strong = one "
               "global batch of RAYS rays per step sharded over the ranks",
       "functional_only": any(c.get("shared_gpu") for cs in curves.values() for c in cs),
       "curve": curves.get("weak", []), "curve_strong": curves.get("strong", [])}
json.dump(out, open("gpurun_out/scale/scale_curve.json", "w"), indent=1)
for mode, curve in curves.items():
  for c in curve:
    print(mode, c["n_gpus"], "GPUs: %.0f rays/s  %.2f ms/step  efficiency %s  waited for the all-reduce %s us"
          % (c["rays_per_s"], c["ms_per_step"], c["efficiency_vs_1"], c["launch_stream_waited_for_all_reduce_avg_us"]))
PY
