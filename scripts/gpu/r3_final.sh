# final artifacts of the round at ONE commit: profiles first (the bench line quotes their traffic and
# commit), then tests, smoke, the bench line, the PSNR runs and the 8-rank functional run
mkdir -p gpurun_out/r3z
bash scripts/gpu/profile_round3.sh > gpurun_out/r3z/profile.log 2>&1; tail -3 gpurun_out/r3z/profile.log
cp gpurun_out/prof3/r03_hbm_traffic.json profiles/r03_hbm_traffic.json
( time python -m pytest tests -m gpu -q ) > gpurun_out/r3z/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r3z/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py ) > gpurun_out/r3z/bench.json 2> gpurun_out/r3z/bench.err; echo "bench rc=$?"
python -m tests.psnr_parity hip --oracle profiles/r03_psnr_parity_oracle.json --ckpt-dir tests/golden/_psnr_oracle --out gpurun_out/r3z/psnr_parity_400.json 2>&1 | tail -1 | cut -c1-300
python -m tests.psnr_parity hip --steps 5000 --every 250 --out gpurun_out/r3z/psnr_5000.json 2>&1 | tail -1 | cut -c1-300
FFN_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --rays 8192 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r3z/bench_8ranks_shared_gpu.json 2> gpurun_out/r3z/bench8.err; echo "bench8 rc=$?"
