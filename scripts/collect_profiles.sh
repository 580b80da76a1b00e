# copies what scripts/gpu/r3_final.sh left under gpurun_out/ (scratch) into profiles/ (tracked)
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/prof3; Z=gpurun_out/r3z
python - <<'PY'
import json
line = [l for l in open("gpurun_out/r3z/bench.json").read().strip().split("\n") if l.startswith("{")][-1]
json.dump(json.loads(line), open("profiles/r03_bench_1gpu.json", "w"), indent=1)
line = [l for l in open("gpurun_out/r3z/bench_8ranks_shared_gpu.json").read().strip().split("\n") if l.startswith("{")][-1]
json.dump(json.loads(line), open("profiles/r03_bench_8ranks_shared_gpu_functional.json", "w"), indent=1)
PY
cp $Z/psnr_parity_400.json profiles/r03_psnr_parity_400.json
cp $Z/psnr_5000.json profiles/r03_psnr_5000_steps.json
cp $P/r03_kernel_stats_stats.csv profiles/r03_kernel_stats.csv
cp $P/r03_kernel_stats_northstar.csv profiles/r03_kernel_stats_north_star.csv
cp $P/r03_kernel_stats_config5.csv profiles/r03_kernel_stats_config5.csv
cp $P/r03_kernel_stats_train.csv profiles/r03_kernel_stats_train_f32_bf16x3.csv
cp $P/r03_hbm_traffic.json $P/r03_hbm_traffic_train_kernels_f32_and_bf16x3.json $P/r03_sq_counters.json profiles/
cp $P/hbm_microbench.json profiles/r03_hbm_microbench.json
grep -h '"commit"' profiles/r03_hbm_traffic.json profiles/r03_sq_counters.json | head -3
