OUT=gpurun_out/r5d
mkdir -p $OUT
for v in default train default train; do
  if [ $v = train ]; then export FFN_BF16X6_INFER=train; else unset FFN_BF16X6_INFER; fi
  echo "== inference instantiation: $v"
  timeout 300 python - <<'PY'
import torch, time
import fourier_feature_nets_amd as ffn
dev = torch.device("cuda:0")
for name in ("mlp8", "tiny", "nerf"):
    torch.manual_seed(2)
    model = {"mlp8": lambda: ffn.MLP(3, 4, num_layers=8, num_channels=256), "tiny": lambda: ffn.PositionalFourierMLP(3, 4, 5.5),
             "nerf": lambda: ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True)}[name]().to(dev)
    prog = model.program()
    n = 1 << 22 if name != "nerf" else 1 << 21
    x = torch.rand(n, 3, device=dev) * 2 - 1
    v = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1) if model.use_view else None
    prog.forward(x, v, None, precision="bf16x6"); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        y = prog.forward(x, v, None, precision="bf16x6")
    e1.record(); torch.cuda.synchronize()
    print(name, "bf16x6 inference %.3f ms" % (e0.elapsed_time(e1) / 4), float(y.abs().max()))
PY
done 2>&1 | tee $OUT/ab.txt
