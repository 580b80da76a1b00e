# BASELINE config 3 end to end through the drivers, on a procedural 400x400 scene of 120 cameras:
# tiny NeRF (coarse model) for 3000 steps, then the full NeRF with it as the opacity model (live
# focus sampling) for 3000 steps of 4096 rays x 128 samples; logs land in gpurun_out/nerf_run/
set -e
OUT=gpurun_out/nerf_run
mkdir -p $OUT
python scripts/make_synthetic_npz.py /tmp/scene400.npz --size 400 --cameras 120 > /dev/null
( time python scripts/train_tiny_nerf.py /tmp/scene400.npz positional /tmp/tiny400 --num-steps 3000 --report-interval 500 \
    --image-interval 100000 --batch-size 4096 --num-samples 64 ) > $OUT/tiny.log 2>&1
cp /tmp/tiny400/log.txt $OUT/tiny_log.txt
( time python scripts/train_nerf.py /tmp/scene400.npz /tmp/nerf400 --opacity-model /tmp/tiny400/tiny_nerf.pt \
    --num-steps 3000 --report-interval 500 --image-interval 100000 --batch-size 4096 --num-samples 128 ) > $OUT/nerf.log 2>&1
cp /tmp/nerf400/log.txt $OUT/nerf_log.txt
tail -3 $OUT/tiny_log.txt; tail -4 $OUT/nerf_log.txt; grep real $OUT/tiny.log $OUT/nerf.log
