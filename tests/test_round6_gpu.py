"""Round-6 GPU tests (MI355X, all through the C ABI): `FourierFeatureMLP.keep_activations` served
from the activation slab (fourier_feature_models.py:70-75), ...

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
