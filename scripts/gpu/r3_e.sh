FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_hwsin.so python -m pytest tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -m gpu -q 2>&1 | tail -6
VARIANTS="hwsin" MB_ARGS="--modes f32" bash scripts/gpu/r3_ab_nt.sh
