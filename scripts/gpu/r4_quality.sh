# quality evidence of round 4: the HIP half of the PSNR ensemble (5 seeds x 1000 steps through
# Raycaster.fit), and the re-synchronised-segment PSNR parity of the split-bf16 training mode
mkdir -p gpurun_out/r4q
S=$(date +%s)
python -m tests.psnr_ensemble hip --seeds 5 --out gpurun_out/r4q/psnr_ensemble_hip.json > gpurun_out/r4q/ensemble.log 2>&1; echo "ensemble rc=$? $(( $(date +%s) - S ))s"
grep "^seed" gpurun_out/r4q/ensemble.log
S=$(date +%s)
python -m tests.psnr_parity hip --oracle profiles/r03_psnr_parity_oracle.json --ckpt-dir tests/golden/_psnr_oracle --precision bf16x3 --out gpurun_out/r4q/psnr_parity_bf16x3.json > gpurun_out/r4q/bf16.log 2>&1; echo "bf16 parity rc=$? $(( $(date +%s) - S ))s"
tail -1 gpurun_out/r4q/bf16.log | cut -c1-600
# BASELINE config 3 (full NeRF + live focus sampling): same-weights PSNR and three re-synchronised
# 100-step segments against the oracle's checkpoints (tests/psnr_parity_config3.py)
S=$(date +%s)
python -m tests.psnr_parity_config3 hip --oracle profiles/r04_psnr_parity_config3_oracle.json --ckpt-dir tests/golden/_psnr_oracle_config3 --out gpurun_out/r4q/psnr_parity_config3.json > gpurun_out/r4q/config3.log 2>&1; echo "config3 parity rc=$? $(( $(date +%s) - S ))s"
tail -1 gpurun_out/r4q/config3.log | cut -c1-700
python -m tests.psnr_parity_config3 hip --precision bf16x3 --oracle profiles/r04_psnr_parity_config3_oracle.json --ckpt-dir tests/golden/_psnr_oracle_config3 --out gpurun_out/r4q/psnr_parity_config3_bf16x3.json > gpurun_out/r4q/config3_bf16.log 2>&1; echo "config3 bf16x3 rc=$?"
tail -1 gpurun_out/r4q/config3_bf16.log | cut -c1-700
