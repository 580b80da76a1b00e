# copies what scripts/gpu/profile_round4.sh, r4_quality.sh and r4_tests_and_scale.sh left under gpurun_out/ (scratch)
# into profiles/ (tracked)
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/prof4; Q=gpurun_out/r4q
python - <<'PY'
import json
line = [l for l in open("gpurun_out/prof4/bench.json").read().strip().split("\n") if l.startswith("{")][-1]
json.dump(json.loads(line), open("profiles/r04_bench_1gpu.json", "w"), indent=1)
PY
cp $P/r04_kernel_stats_stats.csv profiles/r04_kernel_stats.csv
cp $P/r04_kernel_stats_northstar.csv profiles/r04_kernel_stats_north_star.csv
cp $P/r04_kernel_stats_config5.csv profiles/r04_kernel_stats_config5.csv
cp $P/r04_kernel_stats_train.csv profiles/r04_kernel_stats_train_f32_bf16x3.csv
cp $P/r04_hbm_traffic.json $P/r04_sq_counters.json $P/r04_bf16_chain_kernels.json profiles/
cp $P/r04_sq_counters_bf16_ring.json $P/r04_sq_counters_bf16_ws.json profiles/
cp $P/r04_sq_counters_bf16_ring_waits.json $P/r04_sq_counters_bf16_ws_waits.json profiles/
cp $P/hbm_microbench.json profiles/r04_hbm_microbench.json
cp $P/r04_default_batch_timeline.txt profiles/
[ -f $Q/psnr_parity_bf16x3.json ] && cp $Q/psnr_parity_bf16x3.json profiles/r04_psnr_parity_bf16x3.json
[ -f $Q/psnr_parity_config3.json ] && cp $Q/psnr_parity_config3.json profiles/r04_psnr_parity_config3.json
[ -f gpurun_out/r4j/scale_curve_shared_gpu_functional.json ] && cp gpurun_out/r4j/scale_curve_shared_gpu_functional.json profiles/r04_scale_curve_shared_gpu_functional.json
grep -h '"commit"' profiles/r04_hbm_traffic.json profiles/r04_sq_counters.json profiles/r04_bf16_chain_kernels.json | head -4
