"""Round-4 parity tests on the MI355X: layer widths the kernels have no tile count for (run
zero-padded: ``train_nerf.py:28-31`` / ``train_tiny_nerf.py:27-30`` accept any ``--num-channels``,
``nn.Linear`` any width -- ``nerf_model.py:52-74``, ``fourier_feature_models.py:43-51``), biases
beyond the kernels' LDS copy (a 512-wide full NeRF), and the multi-GPU bench fields."""

import contextlib
import io
import math
import os

import numpy as np
import pytest
import torch

from oracle import ffn_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENE = os.path.join(GOLDEN, "scene16.npz")


def dev():
    return torch.device("cuda:0")


def _quiet(fn, *args, **kwargs):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*args, **kwargs)


# ----------------------------------------------------------------------------------- any width
def _make(kind):
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(17)
    if kind == "mlp96":
        return ffn.MLP(3, 4, num_channels=96)
    if kind == "mlp7":
        return ffn.MLP(3, 4, num_layers=2, num_channels=7)
    if kind == "positional384":
        return ffn.PositionalFourierMLP(3, 4, 5.5, num_channels=384)
    if kind == "gaussian200":
        return ffn.GaussianFourierMLP(3, 4, 3.0, num_channels=200, embedding_size=64)
    if kind == "nerf192":
        return ffn.NeRF(8, 192, 9, 10, 3, 4, [4], True)
    if kind == "nerf32":
        return ffn.NeRF(8, 32, 9, 10, 3, 4, [4], True)
    if kind == "nerf512":
        return ffn.NeRF(8, 512, 9, 10, 3, 4, [4], True)
    if kind == "nerf100":
        return ffn.NeRF(4, 100, 5, 6, 2, 3, [2], False)
    raise KeyError(kind)


def _oracle_of(model):
    """The oracle on a copy of ``model``'s (CPU) weights, and (model key, oracle tensor) pairs."""
    if hasattr(model, "opacity_out"):
        params = {k: v.detach().clone() for k, v in model.state_dict().items()}
        ref = orc.OracleNeRF(params, sorted(model.skips), model.include_inputs)
        pairs = [(k, ref.p[k]) for k, v in model.named_parameters() if v.requires_grad]
        return ref, pairs
    ws = [layer.weight.detach().clone() for layer in model.layers]
    bs = [layer.bias.detach().clone() for layer in model.layers]
    a = None if model.a_values is None else model.a_values.detach().clone()
    b = None if model.b_values is None else model.b_values.detach().clone()
    ref = orc.OracleFourierMLP(a, b, ws, bs)
    pairs = []
    for i in range(len(ws)):
        pairs += [("layers.%d.weight" % i, ref.weights[i]), ("layers.%d.bias" % i, ref.biases[i])]
    return ref, pairs


KINDS = ["mlp96", "mlp7", "positional384", "gaussian200", "nerf192", "nerf32", "nerf512", "nerf100"]


@pytest.mark.parametrize("kind", KINDS)
def test_any_layer_width_forward_and_gradients_against_the_oracle(kind):
    """Widths without a tile count of their own -- 96, 7, 384, 200, 192 (hidden_view 96), 32
    (hidden_view 16), 100 (hidden_view 50), and the 512-wide full NeRF whose 7.9k bias floats
    outgrow the kernels' LDS copy -- run zero-padded to the next supported width: logits and
    EVERY weight / bias gradient against the oracle's autograd on the same weights (ragged sample
    counts; several weight-gradient segments per unit at the larger one)."""
    model = _make(kind)
    ref, pairs = _oracle_of(model)
    model = model.to(dev())
    prog = model.program()
    assert any(sp.out_p != sp.out for sp in prog.layers if sp.to_logits is None) or kind == "nerf512"
    if kind == "nerf512":
        assert prog.wide and prog.fwd.bias_floats > 4096
    named = dict(model.named_parameters())
    for n in (45, 9000 + 7):
        torch.manual_seed(n)
        x = torch.rand(n, 3) * 2 - 1
        v = torch.nn.functional.normalize(torch.randn(n, 3), dim=1)
        probe = torch.randn(n, 4) / math.sqrt(n)
        for _, par in pairs:
            par.grad = None
        model.zero_grad()
        exp = ref(x, v) if model.use_view else ref(x)
        (exp * probe).sum().backward()
        args = (x.to(dev()), v.to(dev())) if model.use_view else (x.to(dev()),)
        with torch.no_grad():
            y_inf = model(*args)
        y = model(*args)
        tol = 1e-4 if kind.startswith("gaussian") else 3e-5
        scale = max(1.0, float(exp.abs().max()))
        np.testing.assert_allclose(y_inf.cpu().numpy(), exp.detach().numpy(), rtol=tol, atol=tol * scale)
        np.testing.assert_allclose(y.detach().cpu().numpy(), exp.detach().numpy(), rtol=tol, atol=tol * scale)
        (y * probe.to(dev())).sum().backward()
        for key, want in pairs:
            got, want = named[key].grad.cpu().double(), want.grad.double()
            assert got.shape == want.shape, key
            err = float((got - want).abs().max())
            # 5e-4 of the tensor's scale (the tolerance of the golden gradient tests), relative L2
            # loose enough for an occasional ReLU sign flip of a near-zero pre-activation
            assert err <= 5e-4 * max(float(want.abs().max()), 1e-6) + 1e-7, (kind, key, n, err)
            rel = float((got - want).norm() / max(float(want.norm()), 1e-12))
            assert rel < 2e-3, (kind, key, n, rel)


@pytest.mark.parametrize("kind", ["mlp96", "nerf192", "nerf32", "nerf512", "positional384"])
def test_any_layer_width_optimisation_step_against_the_oracle(kind):
    """One complete `TrainEngine.train_step` (sample -> model -> composite -> loss -> backward ->
    clip -> Adam, ray_caster.py:319-329) of a padded-width model == the oracle's step: loss and
    every updated weight."""
    import fourier_feature_nets_amd as ffn
    model = _make(kind)
    ref, pairs = _oracle_of(model)
    model = model.to(dev())
    S = 16
    train = _quiet(ffn.ImageDataset.load, SCENE, "train", S, True, False, device=dev())
    engine = ffn.TrainEngine(model)
    batch = torch.arange(0, len(train), 3, device=dev())
    loss = float(engine.train_step(train, batch, None, 5e-4))
    engine.check_finite()
    rays = train.ray_ids(batch).cpu()
    smp = train.sampler
    state = {"starts": smp.starts.cpu(), "directions": smp.directions.cpu(), "near_far": smp.near_far.cpu()}
    pos, view, t, _ = orc.sample(state, rays.numpy(), None, S)
    gc, ga = orc.ground_truth(train.colors.cpu(), train.alphas.cpu(), rays)
    before = [par.detach().clone() for _, par in pairs]
    ref_loss = orc.OracleTrainer(ref, 5e-4).step(pos, view, t, gc, ga, 5e-4)
    assert abs(loss - ref_loss) < 5e-6 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    named = dict(model.named_parameters())
    moved = 0.0
    for (key, want), w0 in zip(pairs, before):
        np.testing.assert_allclose(named[key].detach().cpu().numpy(), want.detach().numpy(), rtol=0,
                                   atol=5e-5, err_msg=key)
        moved = max(moved, float((want.detach() - w0).abs().max()))
    assert moved > 1e-4                                        # Adam did take its step


@pytest.mark.parametrize("kind", ["mlp96", "nerf192", "nerf100"])
def test_padded_widths_in_the_split_bf16_mode_and_the_fused_render(kind):
    """The opt-in split-bf16 kernels and the one-launch render consume the same padded chains:
    bf16x3 logits within 2e-4 of the exact ones, bf16x3 gradients within 2e-3 (relative L2) of the
    exact ones, fused render == three-pass render bit for bit."""
    import fourier_feature_nets_amd as ffn
    model = _make(kind).to(dev())
    n = 5000
    torch.manual_seed(5)
    x = (torch.rand(n, 3, device=dev()) * 2 - 1)
    v = torch.nn.functional.normalize(torch.randn(n, 3, device=dev()), dim=1)
    args = (x, v) if model.use_view else (x,)
    with torch.no_grad():
        exact = model(*args)
        model.precision = "bf16x3"
        split = model(*args)
        model.precision = "f32"
    assert float((exact - split).abs().max()) < 2e-4 * max(1.0, float(exact.abs().max()))
    probe = torch.linspace(-1, 1, 4 * n, device=dev()).view(n, 4) / math.sqrt(n)
    grads = {}
    for mode in ("f32", "bf16x3"):
        model.train_precision = mode
        model.zero_grad()
        (model(*args) * probe).sum().backward()
        grads[mode] = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    model.train_precision = "f32"
    for g32, g16 in zip(grads["f32"], grads["bf16x3"]):
        assert float((g32 - g16).norm()) <= 2e-3 * float(g32.norm()) + 1e-7
    # fused render against the three passes
    train = _quiet(ffn.ImageDataset.load, SCENE, "train", 24, True, False, device=dev())
    caster = ffn.Raycaster(model)
    frames = {}
    for fused in (True, False):
        caster.fused_render = fused
        frames[fused] = caster.render_image(train.sampler, 0, 4096).copy()
    assert frames[True].any() and np.array_equal(frames[True], frames[False])
