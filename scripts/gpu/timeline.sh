mkdir -p gpurun_out/tl
FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_dbg_epi.so python scripts/probes/epilogue_timeline.py > gpurun_out/tl/epi.log 2>&1
FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_dbg_tiles.so python scripts/probes/tiles_timeline.py > gpurun_out/tl/tiles.log 2>&1
cat gpurun_out/tl/epi.log | tail -14; cat gpurun_out/tl/tiles.log | tail -7
