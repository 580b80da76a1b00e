import sys, os, io, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import fourier_feature_nets_amd as ffn
from tests.test_pipeline_gpu import _small_model
from tests.conftest import GOLDEN
dev = torch.device("cuda:0")
g = np.load(os.path.join(GOLDEN, "training.npz"))
data = np.load(os.path.join(GOLDEN, "scene16.npz"))
coarse = _small_model(g)
n_train = int(data["split_counts"][0])
cams = [ffn.CameraInfo.create("c%d" % i, ffn.Resolution(16, 16), data["intrinsics"][i], data["extrinsics"][i]) for i in range(n_train)]
for S, strat in ((16, False), (128, True), (37, True), (64, False), (7, True)):
    with contextlib.redirect_stdout(io.StringIO()):
        smp = ffn.RaySampler(data["bounds"], cams, S, strat, coarse, 64, device=dev, focus_mode="live")
    idx = smp.valid_index(torch.arange(1, smp.num_rays, 4, device=dev))
    torch.manual_seed(S); a = smp.sample_t(idx, None)
    smp.fused_focus = False
    torch.manual_seed(S); b = smp.sample_t(idx, None)
    d = (a - b).abs()
    print("S", S, "strat", strat, "mismatch", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()), "rows", int((d > 0).any(1).sum()))
    if float(d.max()) > 0:
        r = int((d > 0).any(1).nonzero()[0])
        print("  row", r, a[r][:8].tolist(), b[r][:8].tolist())
        # compare cdf rows: live rows vs ... 

# is the fused render bit-identical to the three-pass render (same MLP logits)?
with contextlib.redirect_stdout(io.StringIO()):
    plain = ffn.RaySampler(data["bounds"], cams, 64, False, device=dev)
caster = ffn.Raycaster(coarse)
ids = plain.valid_index(torch.arange(0, plain.num_rays, 3, device=dev))
with torch.no_grad():
    a = caster.render_rays(plain, ids, include_depth=True)
    b = caster.render(plain.sample(ids, None), True)
print("render color equal", torch.equal(a.color, b.color), float((a.color - b.color).abs().max()),
      "alpha equal", torch.equal(a.alpha, b.alpha))
# logits of the probe through model() vs positions recomputed
smp.fused_focus = False
n_focus = 4

# constant-logit opacity model: the MLP arithmetic drops out; any mismatch is in the focus pieces
const = _small_model(g)
with torch.no_grad():
    for layer in const.layers:
        layer.weight.zero_(); layer.bias.zero_()
    const.layers[-1].bias[3] = 0.3
const.invalidate_packed()
for S, strat in ((16, False), (128, True)):
    with contextlib.redirect_stdout(io.StringIO()):
        smp = ffn.RaySampler(data["bounds"], cams, S, strat, const, 64, device=dev, focus_mode="live")
    idx = smp.valid_index(torch.arange(1, smp.num_rays, 4, device=dev))
    torch.manual_seed(S); a = smp.sample_t(idx, None)
    smp.fused_focus = False
    torch.manual_seed(S); b = smp.sample_t(idx, None)
    d = (a - b).abs()
    print("const model S", S, "mismatch", int((d > 0).sum()), "max", float(d.max()))
