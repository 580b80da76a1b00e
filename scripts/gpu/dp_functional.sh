# Functional checks of the data-parallel path on a ONE-GPU box (never measurements):
#  (a) one rank under torch.distributed.run with the RCCL backend -- the asynchronous all-reduce and the
#      look-ahead sampling under it, on the real communicator;
#  (b) 1 / 2 / 4 ranks sharing cuda:0 over gloo, weak and strong: every field of the line.
OUT=gpurun_out/dp
mkdir -p $OUT
B="--steps 4 --warmup 1 --no-target-shape --no-config3 --no-config5 --no-bf16-leg --no-skip-leg --no-render --no-cpu-baseline"
for MODE in weak strong; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
      bench.py --gpus 1 --scaling $MODE $B > $OUT/rccl_1rank_$MODE.json 2> $OUT/rccl_1rank_$MODE.err
  echo "rccl one rank $MODE rc=$?"
done
python - <<'PY'
import json
for mode in ("weak", "strong"):
    lines = [l for l in open("gpurun_out/dp/rccl_1rank_%s.json" % mode) if l.startswith("{")]
    b = json.loads(lines[-1])
    print(mode, b["value"], b["ms_per_step"], b["scaling"], json.dumps(b.get("collective"))[:400])
PY
FFN_BENCH_SHARE_GPU=1 NS="1 2 4" STEPS=4 WARMUP=1 bash scripts/gpu/scale.sh
cp gpurun_out/scale/scale_curve.json $OUT/scale_curve_shared_gpu_functional.json
tail -3 gpurun_out/scale/*.err | tail -30
