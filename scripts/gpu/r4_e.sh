export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_ws_stamps.so
python scripts/probes/ws_stamps.py nerf 2>&1 | tail -4
