"""Parity of every HIP kernel against the CPU oracle / the golden fixtures.  Needs an MI355X.

All calls go through the C ABI (fourier_feature_nets_amd.ops -> ctypes -> libffn_hip.so).
Tolerances (fp32): bit-exact for indices, t-values and positions given identical inputs;
1e-5 absolute on rendered colour/alpha (transcendentals + scan order); 2e-5 relative on MLP
outputs (summation order of a K<=512 dot product).
"""

import math

import numpy as np
import pytest
import torch

from oracle import ffn_oracle as orc
from tests.helpers import formula_fill, look_at_camera

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def ops():
    from fourier_feature_nets_amd import ops as _ops
    return _ops


# ----------------------------------------------------------------------------------- K1
def _rig(n_cam, width, height, seed=0):
    rng = np.random.RandomState(seed)
    intr, ext = [], []
    for c in range(n_cam):
        ang = 2 * np.pi * c / n_cam + 0.1
        eye = [4 * np.cos(ang), 0.6 + rng.rand(), 4 * np.sin(ang)]
        k, e = look_at_camera(eye, width, height)
        intr.append(k)
        ext.append(e)
    return np.stack(intr), np.stack(ext)


def _device_rays(ops, intr, ext, width, height, bounds):
    unproj = np.stack([orc.unprojection(k, e) for k, e in zip(intr, ext)]).astype(np.float32)
    cam = np.stack([e[:3, 3] for e in ext]).astype(np.float32)
    lo, hi = orc.aabb_from_bounds(bounds)
    return ops.raygen_nearfar(_t(unproj).to(dev()), _t(cam).to(dev()), width, height,
                              lo[0], hi[0])


@pytest.mark.parametrize("tag", ["eye2", "scale2"])
def test_raygen_against_golden(ops, golden, tag):
    g = golden("raygen")
    W, H = int(g["width"]), int(g["height"])
    starts, dirs, nf, valid = _device_rays(ops, g["intrinsics"], g["extrinsics"], W, H,
                                           g["bounds_" + tag])
    np.testing.assert_allclose(starts.cpu().numpy(), g["starts_" + tag], rtol=0, atol=0)
    np.testing.assert_allclose(dirs.cpu().numpy(), g["directions_" + tag], rtol=0, atol=3e-7)
    invalid = np.nonzero(valid.cpu().numpy() == 0)[0]
    assert np.array_equal(invalid, g["invalid_" + tag])
    ok = valid.cpu().numpy() == 1
    np.testing.assert_allclose(nf.cpu().numpy()[:, ok], g["near_far_" + tag][:, ok],
                               rtol=2e-6, atol=2e-6)


def test_raygen_full_size_properties(ops):
    """400x400 x 8 cameras: unit directions, ids in x-fastest order, valid <=> near < far."""
    intr, ext = _rig(8, 400, 400)
    bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
    starts, dirs, nf, valid = _device_rays(ops, intr, ext, 400, 400, bounds)
    assert starts.shape == (8 * 160000, 3)
    norms = dirs.norm(dim=-1)
    assert float((norms - 1).abs().max()) < 1e-6
    st = orc.sampler_state(bounds, intr[:1], ext[:1], 400, 400)
    np.testing.assert_allclose(dirs[:160000].cpu().numpy(), st["directions"].numpy(), atol=3e-7)
    v = valid.bool()
    assert bool(((nf[0] < nf[1]) == v).all())
    assert 0.3 < float(v.float().mean()) < 0.95
    assert float(nf[0][v].min()) >= 0.1
    bad_cpu = set(st["invalid"].tolist())
    bad_gpu = set(np.nonzero(valid[:160000].cpu().numpy() == 0)[0].tolist())
    assert len(bad_cpu ^ bad_gpu) <= 4      # grazing rays may flip on a 1-ulp direction change


# ----------------------------------------------------------------------------------- K2
@pytest.mark.parametrize("step", [None, 0, 500, 5000])
@pytest.mark.parametrize("stratified", [False, True])
def test_sampling_bit_exact(ops, golden, step, stratified):
    g = golden("sampling")
    r = golden("raygen")
    key = "none" if step is None else str(step)
    S = 16
    near_far = _t(r["near_far_eye2"]).to(dev())
    starts = _t(r["starts_eye2"]).to(dev())
    dirs = _t(r["directions_eye2"]).to(dev())
    idx = _t(g["idx"]).to(dev())
    unit = torch.linspace(0, 1, S).to(dev())
    noise = _t(g["s_noise_" + key]).to(dev()) if stratified else None
    anneal = None
    if step is not None and step < 2000:
        anneal = min(max(step / 2000, 0.2), 1)
    t = ops.sample_t(near_far, idx, S, unit, noise, anneal)
    pos, views = ops.materialise_samples(starts, dirs, idx, t)
    tag = "s" if stratified else "u"
    assert np.array_equal(t.cpu().numpy(), g["%s_t_%s" % (tag, key)])
    assert np.array_equal(pos.cpu().numpy(), g["%s_pos_%s" % (tag, key)])
    if not stratified:
        assert np.array_equal(views.cpu().numpy(), g["u_view_" + key])
    # the one-launch form (what samplers without an opacity model run): the same bits, hence the
    # reference's own values too
    t1, pos1, views1 = ops.sample_materialise(near_far, starts, dirs, idx, S, unit, noise, anneal)
    assert torch.equal(t1, t) and torch.equal(pos1, pos) and torch.equal(views1, views)
    t2, pos2, none = ops.sample_materialise(near_far, starts, dirs, idx, S, unit, noise, anneal, want_views=False)
    assert none is None and torch.equal(t2, t) and torch.equal(pos2, pos)
    empty = idx[:0]
    assert ops.sample_materialise(near_far, starts, dirs, empty, S, unit, None, None)[1].shape == (0, S, 3)


def test_sampling_empty_and_ragged(ops, golden):
    r = golden("raygen")
    near_far = _t(r["near_far_eye2"]).to(dev())
    unit = torch.linspace(0, 1, 7).to(dev())
    empty = torch.zeros((0,), dtype=torch.int64, device=dev())
    assert ops.sample_t(near_far, empty, 7, unit, None, None).shape == (0, 7)
    one = torch.tensor([5], dtype=torch.int64, device=dev())
    t = ops.sample_t(near_far, one, 7, unit, None, None)
    exp = orc.uniform_t(_t(r["near_far_eye2"])[0, 5:6], _t(r["near_far_eye2"])[1, 5:6], 7, None)
    assert np.array_equal(t.cpu().numpy(), exp.numpy(), equal_nan=True)


def test_focus_sampling(ops, golden):
    g = golden("focus")
    s = golden("sampling")
    r = golden("raygen")
    cdf = ops.cdf_build(_t(g["probe_t"]).to(dev()), _t(g["probe_opacity"]).to(dev()))
    np.testing.assert_allclose(cdf.cpu().numpy(), g["probe_cdf"], rtol=0, atol=2e-5)  # 1-exp(-x) near 0 amplifies a 1-ulp expf difference
    S, n_focus = 16, 8
    near_far = _t(r["near_far_eye2"]).to(dev())
    idx = _t(s["idx"]).to(dev())
    unit_u = torch.linspace(0, 1, S - n_focus).to(dev())
    unit_f = torch.linspace(0, 1, n_focus).to(dev())
    cdfs = _t(g["cdfs"]).to(dev())
    # non-stratified: u = linspace(0,1,n) for every ray
    t = torch.empty((len(idx), S), dtype=torch.float32, device=dev())
    ops.sample_t(near_far, idx, S - n_focus, unit_u, None, None, out=t)
    u = unit_f.unsqueeze(0).repeat(len(idx), 1).contiguous()
    ops.focus_sample_merge(near_far, cdfs, idx, u, unit_f, t, n_focus)
    assert np.array_equal(t.cpu().numpy(), g["t_u"])
    # stratified with the reference's own random blocks
    t = torch.empty((len(idx), S), dtype=torch.float32, device=dev())
    ops.sample_t(near_far, idx, S - n_focus, unit_u, _t(g["noise_s"]).to(dev()), None, out=t)
    ops.focus_sample_merge(near_far, cdfs, idx, _t(g["focus_u_s"]).to(dev()), unit_f, t, n_focus)
    assert np.array_equal(t.cpu().numpy(), g["t_s"])
    assert bool((t[:, 1:] >= t[:, :-1]).all())


# ----------------------------------------------------------------------------------- K3
@pytest.mark.parametrize("name,scale,inc", [("positional", math.pi, False),
                                            ("gaussian", math.pi, False),
                                            ("nerf", 1.0, True)])
def test_fourier_encode(ops, golden, name, scale, inc):
    g = golden("models")
    x = _t(g["x"])
    if name == "nerf":
        b, a = _t(g["nerf/pos_encoding"]), None
        exp = orc.nerf_encode(x, b, True)
    else:
        b, a = _t(g[name + "/b_values"]), _t(g[name + "/a_values"])
        exp = orc.fourier_features(x, a, b)
    got = ops.fourier_encode(x.to(dev()), b.contiguous().to(dev()),
                             None if a is None else a.to(dev()), scale, inc)
    # |angle| reaches ~140 rad (positional) / ~60 rad (gaussian, 3-term dot product whose
    # summation order differs from MKL's): one ulp of the angle is 8e-6 / 4e-6
    np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=0, atol=2e-5)


# ----------------------------------------------------------------------------------- K5
def test_composite_against_golden(ops, golden):
    g = golden("composite")
    logits = _t(g["logits"]).to(dev())
    t = _t(g["t"]).to(dev())
    flag = torch.zeros((1,), dtype=torch.int32, device=dev())
    color, alpha, depth = ops.composite_fwd(logits, t, True, flag)
    np.testing.assert_allclose(color.cpu().numpy(), g["color"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(alpha.cpu().numpy(), g["alpha"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(depth.cpu().numpy(), g["depth"])
    assert int(flag.item()) == 0
    # gradient of the reference loss
    gt_c, gt_a = _t(g["gt_color"]).to(dev()), _t(g["gt_alpha"]).to(dev())
    R = color.shape[0]
    d_color = 2 * (color - gt_c) / (3 * R)
    d_alpha = 0.1 * 2 * (alpha - gt_a) / R
    d_logits = ops.composite_bwd(logits, t, d_color.contiguous(), d_alpha.contiguous())
    np.testing.assert_allclose(d_logits.cpu().numpy(), g["dlogits"], rtol=2e-4, atol=2e-8)


@pytest.mark.parametrize("S", [1 + 1, 37, 64, 65, 128, 200, 256])
def test_composite_ragged_sample_counts(ops, S):
    torch.manual_seed(S)
    R = 257
    t = torch.sort(torch.rand(R, S) * 4 + 2, -1)[0]
    logits = torch.randn(R, S, 4) * 2
    logits[:8, :, 3] = -30
    logits[8:16, :, 3] = 25
    ref = logits.clone().requires_grad_(True)
    c, a, d = orc.render(ref, t, True)
    color, alpha, depth = ops.composite_fwd(logits.to(dev()), t.to(dev()), True)
    np.testing.assert_allclose(color.cpu().numpy(), c.detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(alpha.cpu().numpy(), a.detach().numpy(), rtol=1e-5, atol=2e-6)
    same = depth.cpu().numpy() == d.numpy()
    assert same.mean() > 0.99       # argmax ties under 1-ulp weight differences
    gc, ga = torch.rand(R, 3), torch.rand(R)
    orc.mse_loss(c, a, gc, ga, 0.1).backward()
    dc = (2 * (color.cpu() - gc) / (3 * R)).to(dev())
    da = (0.2 * (alpha.cpu() - ga) / R).to(dev())
    dl = ops.composite_bwd(logits.to(dev()), t.to(dev()), dc.contiguous(), da.contiguous())
    np.testing.assert_allclose(dl.cpu().numpy(), ref.grad.numpy(), rtol=5e-4, atol=1e-7)


def test_composite_nan_flag_and_invariants(ops):
    torch.manual_seed(1)
    R, S = 64, 64
    t = torch.sort(torch.rand(R, S) * 4 + 2, -1)[0].to(dev())
    logits = torch.randn(R, S, 4, device=dev())
    flag = torch.zeros((1,), dtype=torch.int32, device=dev())
    color, alpha, _ = ops.composite_fwd(logits, t, False, flag)
    assert int(flag.item()) == 0
    assert float(alpha.max()) <= 1 + 1e-5 and float(alpha.min()) >= 0
    assert float(color.max()) <= 1 + 1e-5
    # permuting rays permutes outputs
    perm = torch.randperm(R, device=dev())
    c2, a2, _ = ops.composite_fwd(logits[perm].contiguous(), t[perm].contiguous(), False)
    assert torch.equal(c2, color[perm]) and torch.equal(a2, alpha[perm])
    logits[3, 5, 1] = float("nan")
    ops.composite_fwd(logits, t, False, flag)
    assert int(flag.item()) == 1


# ----------------------------------------------------------------------------------- K6
def test_mse_loss(ops, golden):
    g = golden("dataset")
    rays = _t(g["full_rays"]).to(dev())
    sums, dc, da = ops.mse_loss(_t(g["pred_color"]).to(dev()), _t(g["pred_alpha"]).to(dev()),
                                _t(g["colors"]).to(dev()), _t(g["alphas"]).to(dev()), rays,
                                1.0 / (3 * len(rays)), 0.1 / len(rays))
    R = len(rays)
    loss = float(sums[0]) / (3 * R) + 0.1 * float(sums[1]) / R
    assert abs(loss - float(g["loss_rgba"])) < 1e-6
    exp_dc = 2 * (g["pred_color"] - g["gt_color"]) / (3 * R)
    np.testing.assert_allclose(dc.cpu().numpy(), exp_dc, rtol=1e-6, atol=1e-9)
    exp_da = 0.2 * (g["pred_alpha"] - g["gt_alpha"]) / R
    np.testing.assert_allclose(da.cpu().numpy(), exp_da, rtol=1e-6, atol=1e-9)
    sums2, _, _ = ops.mse_loss(_t(g["pred_color"]).to(dev()), _t(g["pred_alpha"]).to(dev()),
                               _t(g["colors"]).to(dev()), None, rays, 1.0, 0.0, want_grad=False)
    rgb_only = orc.mse_loss(_t(g["pred_color"]), None, _t(g["colors"])[_t(g["full_rays"])], None)
    assert abs(float(sums2[0]) / (3 * R) - float(rgb_only)) < 1e-6


# ----------------------------------------------------------------------------------- K7
def test_clip_adam_against_golden(ops, golden):
    g = golden("training")
    p = torch.cat([_t(g["adam_init0"]).reshape(-1), _t(g["adam_init1"]).reshape(-1)]).to(dev())
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for it in range(3):
        grads = torch.cat([_t(g["adam_g0_%d" % it]).reshape(-1),
                           _t(g["adam_g1_%d" % it]).reshape(-1)]).to(dev())
        ops.clip_adam(p, grads, m, v, it + 1, orc.lr_decay(5e-4, it, 0.1, 25000),
                      weight_decay=1e-3)
        exp = np.concatenate([g["adam_p0_%d" % it].reshape(-1), g["adam_p1_%d" % it].reshape(-1)])
        np.testing.assert_allclose(p.cpu().numpy(), exp, rtol=2e-6, atol=1e-7)


def test_clip_adam_large_buffer(ops):
    torch.manual_seed(3)
    n = 263428                                 # tiny-NeRF parameter count
    p0 = torch.randn(n) * 0.05
    g0 = torch.randn(n) * 0.3
    p, gr = p0.clone().to(dev()), g0.clone().to(dev())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    norm = torch.zeros((1,), device=dev())
    ops.clip_adam(p, gr, m, v, 1, 5e-4, norm_out=norm)
    # float64 reference of clip_grad_value_ -> clip_grad_norm_ -> Adam (a float32 sum of
    # 263k equal-magnitude squares on one flat CPU tensor is itself only good to ~1e-4)
    g64 = g0.double().clamp(-0.1, 0.1)
    total = float(g64.square().sum().sqrt())
    g64 = g64 * min(1.0, 0.1 / (total + 1e-6))
    m64 = 0.1 * g64
    v64 = 0.001 * g64 * g64
    ref_p = p0.double() - (5e-4 / 0.1) * m64 / (v64.sqrt() / math.sqrt(0.001) + 1e-8)
    assert abs(float(norm) - total) / total < 2e-6
    np.testing.assert_allclose(gr.cpu().numpy(), g64.numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(p.cpu().numpy(), ref_p.numpy(), rtol=1e-5, atol=2e-7)


# ----------------------------------------------------------------------------------- K8
def test_to_image(ops, golden):
    g = golden("dataset")
    local = _t(g["to_image_rays"] - 256).to(dev())
    cols = g["to_image_colors"].copy()
    cols = np.minimum(cols, 1.0)               # u8 conversion of >255 is undefined in C
    img = ops.to_image(_t(cols).to(dev()), local, 16, 16)
    exp = orc.to_image(g["to_image_rays"] - 256, cols, 16, 16)
    assert np.array_equal(img.cpu().numpy(), exp)


# ----------------------------------------------------------------------------------- K4
def _load_fourier(g, name):
    """A fourier_feature_nets_amd model carrying the golden (or formula-fill) weights."""
    import fourier_feature_nets_amd as ffn
    from tests.test_oracle_golden import _fourier_params
    a, b, ws, bs = _fourier_params(g, name)
    channels = [w.shape[0] for w in ws[:-1]]
    model = ffn.FourierFeatureMLP(3, 4, a, b, channels)
    with torch.no_grad():
        for layer, w, bias in zip(model.layers, ws, bs):
            layer.weight.copy_(w)
            layer.bias.copy_(bias)
    return model.to(dev()), (a, b, ws, bs)


def _load_nerf(g, name, skips, inc):
    import fourier_feature_nets_amd as ffn
    from tests.test_oracle_golden import _nerf_params
    p = _nerf_params(g, name)
    n_layers = len([k for k in p if k.startswith("layers.") and k.endswith("weight")])
    ch = p["layers.0.weight"].shape[0]
    fp, fv = p["pos_encoding"].shape[1] // 3, p["view_encoding"].shape[1] // 3
    model = ffn.NeRF(n_layers, ch, math.log2(float(p["pos_encoding"].max())), fp,
                     math.log2(float(p["view_encoding"].max())), fv, skips, inc)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[name + "/keys"]]
    model.load_state_dict({k: v for k, v in p.items()})
    assert torch.equal(model.pos_encoding.data, p["pos_encoding"])
    return model.to(dev()), p


@pytest.mark.parametrize("name", ["mlp", "basic", "positional", "gaussian", "gaussian512"])
def test_fused_mlp_forward_against_golden(golden, name):
    g = golden("models")
    model, _ = _load_fourier(g, name)
    with torch.no_grad():
        y = model(_t(g["x"]).to(dev()))
    tol = 3e-5 if not name.startswith("gaussian") else 1e-4
    np.testing.assert_allclose(y.cpu().numpy(), g[name + "/out"], rtol=tol, atol=tol)


def test_fused_mlp_forward_ragged_sizes(golden):
    g = golden("models")
    model, (a, b, ws, bs) = _load_fourier(g, "positional")
    torch.manual_seed(0)
    for n in [1, 31, 32, 33, 127, 128, 129, 1000]:
        x = torch.rand(n, 3) * 2 - 1
        exp = orc.fourier_mlp_forward(x, a, b, ws, bs)
        with torch.no_grad():
            y = model(x.to(dev()))
        np.testing.assert_allclose(y.cpu().numpy(), exp.numpy(), rtol=3e-5, atol=3e-5)


def _check_grads(g, name, named_params, tol):
    for key, par in named_params:
        if not par.requires_grad:
            continue
        got = par.grad.cpu()
        full = "%s/grad/%s" % (name, key)
        if full in g.files:
            np.testing.assert_allclose(got.numpy(), g[full], rtol=tol, atol=tol, err_msg=key)
        else:
            head = g["%s/gradhead/%s" % (name, key)]
            np.testing.assert_allclose(got.reshape(-1)[:512].numpy(), head, rtol=tol, atol=tol,
                                       err_msg=key)
            total = float(g["%s/gradsum/%s" % (name, key)])
            scale = float(g["%s/gradabs/%s" % (name, key)])
            assert abs(float(got.double().sum()) - total) <= 2e-5 * scale, key


@pytest.mark.parametrize("name", ["mlp", "basic", "positional", "gaussian", "gaussian512"])
def test_fused_mlp_backward_against_golden(golden, name):
    g = golden("models")
    model, _ = _load_fourier(g, name)
    y = model(_t(g["x"]).to(dev()))
    probe = torch.linspace(-1, 1, y.numel()).reshape(y.shape).to(dev())
    (y * probe).sum().backward()
    _check_grads(g, name, model.named_parameters(), 5e-4 if not name.startswith("gaussian") else 2e-3)


def test_wide_mlp_many_blocks_and_ragged(golden):
    """512-wide chain (two waves per block): more blocks than resident pairs, a ragged tail and
    a last pass in which some pairs only keep the barriers company."""
    g = golden("models")
    model, (a, b, ws, bs) = _load_fourier(g, "gaussian512")
    torch.manual_seed(11)
    ref = orc.OracleFourierMLP(a, b, ws, bs)
    for n in (1, 33, 20000 + 19):
        x = torch.rand(n, 3) * 2 - 1
        probe = torch.randn(n, 4) / n
        for par in list(ref.weights) + list(ref.biases):
            par.grad = None
        model.zero_grad()
        exp = ref(x)
        (exp * probe).sum().backward()
        y = model(x.to(dev()))
        np.testing.assert_allclose(y.detach().cpu().numpy(), exp.detach().numpy(), rtol=1e-4, atol=1e-4)
        (y * probe.to(dev())).sum().backward()
        for i, layer in enumerate(model.layers):
            np.testing.assert_allclose(layer.weight.grad.cpu().numpy(), ref.weights[i].grad.numpy(),
                                       rtol=2e-3, atol=2e-6)
            np.testing.assert_allclose(layer.bias.grad.cpu().numpy(), ref.biases[i].grad.numpy(),
                                       rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("name,skips,inc", [("nerf", [4], True), ("nerf_small", [2], False)])
def test_fused_nerf_forward_backward_against_golden(golden, name, skips, inc):
    g = golden("models")
    model, _ = _load_nerf(g, name, skips, inc)
    y = model(_t(g["x"]).to(dev()), _t(g["v"]).to(dev()))
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[name + "/out"], rtol=5e-5, atol=5e-5)
    probe = torch.linspace(-1, 1, y.numel()).reshape(y.shape).to(dev())
    (y * probe).sum().backward()
    _check_grads(g, name, model.named_parameters(), 1e-3)


def test_fused_mlp_backward_many_blocks(golden):
    """Enough samples that every persistent wgrad wave gets work and jobs are split."""
    g = golden("models")
    model, (a, b, ws, bs) = _load_fourier(g, "gaussian")
    torch.manual_seed(4)
    n = 50000 + 17
    x = torch.rand(n, 3) * 2 - 1
    probe = torch.randn(n, 4) / n
    ref = orc.OracleFourierMLP(a, b, ws, bs)
    (ref(x) * probe).sum().backward()
    y = model(x.to(dev()))
    (y * probe.to(dev())).sum().backward()
    for i, layer in enumerate(model.layers):
        np.testing.assert_allclose(layer.weight.grad.cpu().numpy(), ref.weights[i].grad.numpy(),
                                   rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(layer.bias.grad.cpu().numpy(), ref.biases[i].grad.numpy(),
                                   rtol=2e-3, atol=2e-6)


def test_models_refuse_cpu():
    import fourier_feature_nets_amd as ffn
    model = ffn.MLP(3, 4, num_channels=32)
    with pytest.raises(RuntimeError, match="GPU"):
        model(torch.zeros(4, 3))


def test_fused_mlp_empty_batch(golden):
    """Zero samples: (0,4) logits forward, all-zero gradients backward (autograd of the
    reference modules on an empty batch)."""
    g = golden("models")
    model, _ = _load_fourier(g, "mlp")
    y = model(torch.zeros((0, 3), device=dev()))
    assert y.shape == (0, 4)
    y.sum().backward()
    for par in model.parameters():
        if par.requires_grad:
            assert par.grad is not None and float(par.grad.abs().max()) == 0.0


def test_fused_mlp_full_size_properties(golden):
    """At the bench shape (65 536 rays x 64 samples = 4 194 304 samples per launch) the oracle
    is out of reach; size-independent properties instead: a sample's logits do not depend on the
    batch around it (bit-exact), and the gradient is additive over the batch."""
    g = golden("models")
    model, _ = _load_fourier(g, "positional")
    n = 65536 * 64
    gen = torch.Generator(device=dev()).manual_seed(12)
    x = torch.rand((n, 3), generator=gen, device=dev()) * 2 - 1
    probe = torch.randn((n, 4), generator=gen, device=dev()) / n
    pick = torch.randint(0, n, (4096,), generator=gen, device=dev())
    pick = torch.cat([pick, torch.tensor([0, n - 1], device=dev())])
    y = model(x)
    assert torch.equal(y.detach()[pick], model(x[pick].contiguous()).detach())

    def grads_of(lo, hi):
        model.zero_grad()
        (model(x[lo:hi]) * probe[lo:hi]).sum().backward()
        return [p.grad.clone() for p in model.parameters() if p.grad is not None]

    whole = grads_of(0, n)
    cut = n // 2 + 32 * 7 + 5                     # not block aligned
    parts = [a + b for a, b in zip(grads_of(0, cut), grads_of(cut, n))]
    for a, b in zip(whole, parts):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * max(scale, 1e-6)


def test_abi_rejects_bad_arguments_with_a_message():
    """Entry points validate shapes / chains and return an error code + message instead of
    launching (the Python layer turns it into FfnError)."""
    import ctypes
    from fourier_feature_nets_amd import _lib
    from fourier_feature_nets_amd._lib import c_i, c_i64, c_p, FfnError
    from fourier_feature_nets_amd.mlp_engine import FfnMlpChain
    stream = c_p(torch.cuda.current_stream().cuda_stream)
    buf = torch.zeros(64, device=dev())
    ptr = c_p(buf.data_ptr())
    with pytest.raises(FfnError, match="ffn_composite_fwd"):
        _lib.call("ffn_composite_fwd", ptr, ptr, c_i(4), c_i(0), ptr, ptr, c_p(0), c_p(0), stream)
    with pytest.raises(FfnError, match="ffn_mlp_pack"):
        _lib.call("ffn_mlp_pack", ptr, c_i(4), c_i(4), c_i(4), c_i(0), c_p(0), c_p(0), c_i(0), c_i(1),
                  ptr, stream)
    chain = FfnMlpChain()                 # num_steps == 0: not a chain
    with pytest.raises(FfnError, match="bad chain"):
        _lib.call("ffn_mlp_forward", ctypes.byref(chain), ptr, ptr, ptr, c_p(0), c_i64(8), ptr,
                  c_p(0), c_p(0), c_i64(0), c_i64(0), stream)
    chain.num_steps = 1
    chain.step[0].out_tiles = 3           # not a supported tile count
    chain.step[0].act_groups = 4
    with pytest.raises(FfnError, match="bad chain"):
        _lib.call("ffn_mlp_backward_data", ctypes.byref(chain), ptr, ptr, c_i64(8), ptr, ptr,
                  c_i64(0), c_i64(0), stream)
    with pytest.raises(FfnError, match="ffn_occupancy_build"):
        _lib.call("ffn_occupancy_build", ptr, c_i(0), ctypes.c_float(0.1), c_i(0), c_p(0), ptr, stream)
    torch.cuda.synchronize()              # nothing was launched, nothing is poisoned
    assert float(buf.abs().max()) == 0.0


def test_voxels_lookup_against_grid_sample():
    """K10 == F.grid_sample(volume, p/scale, padding_mode="border", align_corners=False) + bias
    (voxels_model.py:35-45), including positions outside the cube."""
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(8)
    for side, scale in ((8, 1.0), (5, 1.5), (32, 0.7)):
        vox = ffn.Voxels(side, scale)
        with torch.no_grad():
            vox.voxels.copy_(torch.randn_like(vox.voxels))
            vox.bias.copy_(torch.randn(1, 4))
        pos = (torch.rand(4099, 3) * 2 - 1) * scale * 1.3
        pos[0] = torch.tensor([scale, -scale, 0.0])
        grid = (pos / scale).reshape(1, -1, 1, 1, 3)
        exp = torch.nn.functional.grid_sample(vox.voxels.detach(), grid, padding_mode="border",
                                              align_corners=False)
        exp = exp.transpose(1, 2).reshape(-1, 4) + vox.bias.detach()
        vox = vox.to(dev())
        with torch.no_grad():
            got = vox(pos.to(dev()))
        # fp32 rounding of the voxel coordinate differs from ATen's by a few ulp of the weights
        np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=2e-5)
        with pytest.raises(NotImplementedError):
            vox(pos.to(dev()))                      # gradients through the volume: out of scope
