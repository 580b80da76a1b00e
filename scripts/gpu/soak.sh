# Integration soak of the driver scripts with the opt-in modes on a synthetic scene:
# tiny NeRF (exact / bf16x3 / bf16x3 + empty-space skipping), then a full NeRF with the tiny model
# as its opacity model (live focus sampling), then an orbit render.  PSNRs land in gpurun_out/soak/.
set -e
OUT=gpurun_out/soak
mkdir -p $OUT
python scripts/make_synthetic_npz.py $OUT/scene.npz > /dev/null
for mode in f32 bf16x3; do
  python scripts/train_tiny_nerf.py $OUT/scene.npz positional $OUT/tiny_$mode --num-steps 600 --report-interval 200 \
      --image-interval 100000 --batch-size 4096 --num-samples 64 --crop-steps 100 --precision $mode > $OUT/tiny_$mode.log 2>&1
  tail -1 $OUT/tiny_$mode/log.txt
done
python scripts/train_tiny_nerf.py $OUT/scene.npz positional $OUT/tiny_skip --num-steps 600 --report-interval 200 \
    --image-interval 100000 --batch-size 4096 --num-samples 64 --crop-steps 100 --precision bf16x3 \
    --skip-empty-space --skip-warmup 200 --skip-refresh 200 > $OUT/tiny_skip.log 2>&1
tail -1 $OUT/tiny_skip/log.txt
FFN_FOCUS_MODE=live python scripts/train_nerf.py $OUT/scene.npz $OUT/nerf --opacity-model $OUT/tiny_f32/tiny_nerf.pt \
    --num-steps 300 --report-interval 100 --image-interval 100000 --batch-size 4096 --num-samples 64 --crop-steps 50 \
    --precision bf16x3 > $OUT/nerf.log 2>&1
tail -1 $OUT/nerf/log.txt
python scripts/orbit_video.py $OUT/nerf/nerf.pt 200 $OUT/orbit --num-frames 4 --num-samples 64 --precision bf16x3 > $OUT/orbit.log 2>&1
ls $OUT/orbit | wc -l
# round 3: the YCrCb colour space end to end, the table / live focus modes of the drivers, two ranks
python scripts/train_tiny_nerf.py $OUT/scene.npz positional $OUT/tiny_ycc --num-steps 300 --report-interval 100 \
    --image-interval 150 --batch-size 4096 --num-samples 64 --crop-steps 50 --color-space YCrCb > $OUT/tiny_ycc.log 2>&1
tail -1 $OUT/tiny_ycc/log.txt
for fm in table live; do
  python scripts/train_nerf.py $OUT/scene.npz $OUT/nerf_$fm --opacity-model $OUT/tiny_f32/tiny_nerf.pt --focus-mode $fm \
      --num-steps 100 --report-interval 50 --image-interval 100000 --batch-size 1024 --num-samples 64 --crop-steps 20 > $OUT/nerf_$fm.log 2>&1
  tail -1 $OUT/nerf_$fm/log.txt
done
FFN_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    scripts/train_tiny_nerf.py $OUT/scene.npz positional $OUT/tiny_dp2 --num-steps 200 --report-interval 100 \
    --image-interval 100000 --batch-size 4096 --num-samples 64 --crop-steps 50 --device cuda:0 > $OUT/tiny_dp2.log 2>&1
tail -1 $OUT/tiny_dp2/log.txt
