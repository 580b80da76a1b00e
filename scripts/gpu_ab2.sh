mkdir -p gpurun_out/ab
python -m pytest tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -k "mlp or nerf or fit or step or wide" 2>&1 | tail -2
bash scripts/gpu_ab.sh
