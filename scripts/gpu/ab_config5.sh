mkdir -p gpurun_out/ab
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = new ]; then export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_new.so; else unset FFN_HIP_LIBRARY; fi
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-config3 --no-skip-leg --no-bf16-leg --no-target-shape --no-render > gpurun_out/ab/${lib}_${rep}.json 2> gpurun_out/ab/${lib}_${rep}.err
    python - <<PY
import json
b = json.loads(open("gpurun_out/ab/${lib}_${rep}.json").read().strip().split("\n")[-1])
c5 = b["config5_step"]
print("${lib} ${rep}", round(b["ms_per_step"], 3), {k.split("_kernel")[0]: v["avg_ms"] for k, v in b["kernels"].items()}, "config5", c5["full"]["step_ms"], c5["with_occupancy_grid"]["step_ms"], {k.split("_kernel")[0]: v["avg_ms"] for k, v in c5["full"]["kernels"].items()})
PY
  done
done
