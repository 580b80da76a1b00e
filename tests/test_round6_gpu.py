"""Round-6 GPU tests (MI355X, all through the C ABI): `FourierFeatureMLP.keep_activations` served
from the activation slab (fourier_feature_models.py:70-75); the one-launch sampling kernel (K2a + K2b,
ray_sampler.py:359-403) in its round-6 form -- 1024-sample chunks, aligned float4 stores -- against the
two launches at chunk edges.

Tolerances: the hidden activations are what the logits tests hold the chain to -- 3e-5 of their
scale (1e-4 for the dense Gaussian B matrix, as in tests/test_kernels_gpu.py)."""

import numpy as np
import pytest
import torch

from oracle import ffn_oracle as orc
from tests import test_kernels_gpu as tk

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


# ----------------------------------------------------------------------------------- keep_activations
@pytest.mark.parametrize("name", ["mlp", "positional", "gaussian", "gaussian512"])
def test_keep_activations_is_the_last_hidden_slab(golden, name):
    """`model.keep_activations = True` makes `forward` leave the output of the last hidden layer in
    `model.activations` as a host numpy array (fourier_feature_models.py:70-75; what
    `Raycaster.render_activations` and the pixel / signal datasets read, ray_caster.py:168-189):
    here a copy of the activation slab of the SAME forward launch -- against the oracle's
    restatement, with and without gradient tracking, ragged batch sizes; logits and gradients are
    the ones of the plain call."""
    model, (a, b, ws, bs) = tk._load_fourier(golden("models"), name)
    torch.manual_seed(5)
    for n in (1000, 33):
        x = torch.rand(n, 3, device=dev()) * 2 - 1
        want = orc.fourier_mlp_last_hidden(x.cpu(), a, b, ws, bs).numpy()
        scale = max(float(np.abs(want).max()), 1.0)
        tol = (1e-4 if name.startswith("gaussian") else 3e-5) * scale
        with torch.no_grad():
            plain = model(x)
        assert model.activations == []
        model.keep_activations = True
        with torch.no_grad():
            kept = model(x)
        assert len(model.activations) == 1 and isinstance(model.activations[0], np.ndarray)
        got = model.activations[0]
        assert got.shape == want.shape and got.dtype == np.float32
        np.testing.assert_allclose(got, want, rtol=0, atol=tol)
        assert float((kept - plain).abs().max()) <= 4e-6 * max(float(plain.abs().max()), 1.0)
        # with gradient tracking: the same record, and the gradients of the plain call
        model.zero_grad()
        out = model(x)
        np.testing.assert_allclose(model.activations[0], want, rtol=0, atol=tol)
        out.square().sum().backward()
        g_keep = [p.grad.clone() for p in model.parameters() if p.grad is not None]
        model.keep_activations = False
        model.zero_grad()
        model(x).square().sum().backward()
        g_plain = [p.grad.clone() for p in model.parameters() if p.grad is not None]
        assert model.activations == []
        assert len(g_keep) == len(g_plain) > 0
        for u, v in zip(g_keep, g_plain):
            assert torch.equal(u, v)


def test_keep_activations_on_a_padded_width_and_an_empty_batch():
    """A hidden layer whose width has no tile count of its own (96 -> 128 channels in the slab):
    the record has the natural 96 columns; an empty batch records an empty array."""
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(3)
    model = ffn.MLP(3, 4, num_layers=3, num_channels=96).to(dev())
    ws = [l.weight.detach().cpu() for l in model.layers]
    bs = [l.bias.detach().cpu() for l in model.layers]
    x = torch.rand(77, 3, device=dev()) * 2 - 1
    model.keep_activations = True
    with torch.no_grad():
        model(x)
    want = orc.fourier_mlp_last_hidden(x.cpu(), None, None, ws, bs).numpy()
    assert model.activations[0].shape == (77, 96)
    np.testing.assert_allclose(model.activations[0], want, rtol=0, atol=3e-5 * max(float(np.abs(want).max()), 1.0))
    with torch.no_grad():
        model(x[:0])
    assert model.activations[0].shape == (0, 96)


# ----------------------------------------------------------------------------------- K2: chunk edges of the one-launch sampler
@pytest.mark.parametrize("rays,samples", [(1, 1), (1, 3), (5, 2), (7, 64), (16, 64), (17, 64), (1023, 1), (1024, 1),
                                          (1025, 1), (341, 3), (342, 3), (100, 100), (33, 128), (9, 257)])
@pytest.mark.parametrize("stratified", [False, True])
def test_one_launch_sampling_equals_the_two_launches_at_chunk_edges(rays, samples, stratified):
    """`ffn_sample_materialise` works on chunks of 1024 consecutive (ray, sample) elements -- one float4 of
    t per thread, three float4 of positions / views per thread from an LDS copy -- so its edges are the
    sample counts that do not divide 1024, a last chunk of any fill, rays that straddle chunks and
    tails that are not a multiple of four: t-values, positions and view directions BIT-identical to
    `ffn_sample_t` + `ffn_materialise_samples` (which the reference goldens pin), with and without
    jitter, annealing and the view output."""
    from fourier_feature_nets_amd import ops
    gen = torch.Generator(device=dev()).manual_seed(rays * 1000 + samples)
    total = 4 * rays + 3
    near = torch.rand(total, device=dev(), generator=gen) * 2 + 1
    near_far = torch.stack([near, near + torch.rand(total, device=dev(), generator=gen) * 3 + 0.1]).contiguous()
    starts = torch.randn(total, 3, device=dev(), generator=gen)
    dirs = torch.nn.functional.normalize(torch.randn(total, 3, device=dev(), generator=gen), dim=1)
    idx = torch.randint(0, total, (rays,), device=dev(), generator=gen)
    unit = torch.linspace(0, 1, samples).to(dev())
    noise = torch.rand(rays, samples, device=dev(), generator=gen) if stratified else None
    for anneal in (None, 0.35):
        t = ops.sample_t(near_far, idx, samples, unit, noise, anneal)
        pos, views = ops.materialise_samples(starts, dirs, idx, t)
        t1, pos1, views1 = ops.sample_materialise(near_far, starts, dirs, idx, samples, unit, noise, anneal)
        assert torch.equal(t1, t) and torch.equal(pos1, pos) and torch.equal(views1, views)
        # without the view output
        t2, pos2, none = ops.sample_materialise(near_far, starts, dirs, idx, samples, unit, noise, anneal, want_views=False)
        assert none is None and torch.equal(t2, t) and torch.equal(pos2, pos)
