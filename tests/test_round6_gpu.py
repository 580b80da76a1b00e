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


# ----------------------------------------------------------------------------------- bf16x6: the matrix / vector waves organisation
def _x6_buffers(prog, x, n, org, monkeypatch):
    monkeypatch.setenv("FFN_BF16X6_ORG", org)
    saved = torch.zeros((prog.saved_floats(n),), dtype=torch.float32, device=dev())
    logits = prog.forward(x, None, saved, precision="bf16x6")
    with torch.no_grad():
        inference = prog.forward(x, None, None, precision="bf16x6")
    torch.manual_seed(n)
    d_logits = torch.randn(n, 4, device=dev()) / n
    ws = prog.workspace(n)
    ws.dz.zero_()
    flat = torch.zeros((prog.num_grad_floats,), dtype=torch.float32, device=dev())
    prog.backward(d_logits, x, None, saved, flat, precision="bf16x6")
    torch.cuda.synchronize()
    return logits, inference, saved, ws.dz.clone(), flat


@pytest.mark.parametrize("name", ["positional", "gaussian256", "positional8"])
def test_bf16x6_matrix_vector_waves_write_the_bits_of_the_two_waves_per_simd_kernels(golden, name, monkeypatch):
    """The bf16x6 chain kernels of the tiny NeRF / Fourier MLP family run in the matrix-waves / vector-waves
    organisation (csrc/mlp_bf16_mv.hip); `FFN_BF16X6_ORG=ws` keeps the two-waves-per-SIMD kernels
    (csrc/mlp_bf16_ws.hip) that every other chain runs and that `test_round5_gpu.py` holds against the
    exact-f32 kernels and float64.  Forward: the same six partial products per K block in the same order
    per accumulator -- activation slabs, feature slabs and sign masks are BIT-identical; the logits (the
    fused head's partial sums meet in another order) within 2e-7 of their scale; inference == training
    forward.  Backward data: the hidden steps are the same arithmetic, but step 0 -- d(loss)/d(logits)
    through the fused head, four real K rows -- is f32 arithmetic of the vector waves where the
    two-waves-per-SIMD kernels issue six bf16 products: dZ and every gradient within 1e-6 of their largest
    element, the zeros of the ReLU masks in the same places, and a second launch BIT-identical to the
    first.  Sizes: a lone sample, the 32-sample block and the 64-sample
    pass either side, a ragged tail, and more passes than a workgroup gets at once (the next pass's first
    feature segment is generated under the last step of the pass before)."""
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(5)
    if name == "positional8":
        model = ffn.PositionalFourierMLP(3, 4, 5.5, num_layers=8, num_channels=256, embedding_size=256).to(dev())
    elif name == "gaussian256":
        model = ffn.GaussianFourierMLP(3, 4, 3.0, num_layers=4, num_channels=256, embedding_size=256).to(dev())
    else:
        model, _ = tk._load_fourier(golden("models"), name)
    prog = model.program()
    monkeypatch.delenv("FFN_BF16X6_ORG", raising=False)
    assert prog.x6_organisation() == "matrix/vector waves" and prog.x6_organisation(backward=True) == "matrix/vector waves"
    monkeypatch.setenv("FFN_BF16X6_ORG", "ws")
    assert prog.x6_organisation() == "two waves per SIMD"
    for n in (1, 31, 32, 33, 64, 65, 1000, 4097, 70001):
        torch.manual_seed(n)
        x = torch.rand(n, 3, device=dev()) * 2 - 1
        ref = _x6_buffers(prog, x, n, "ws", monkeypatch)
        new = _x6_buffers(prog, x, n, "mv", monkeypatch)
        scale = max(float(ref[0].abs().max()), 1.0)
        assert float((new[0] - ref[0]).abs().max()) <= 2e-7 * scale, (n, "logits")
        assert torch.equal(new[0], new[1]), (n, "inference != training forward")
        assert torch.equal(ref[2].view(torch.int32), new[2].view(torch.int32)), (n, "slabs / masks")
        for what, a, b in (("dZ", ref[3], new[3]), ("gradients", ref[4], new[4])):
            assert float((a - b).abs().max()) <= 1e-6 * max(float(a.abs().max()), 1e-30), (n, what)
        assert torch.equal(ref[3] == 0, new[3] == 0), (n, "dZ: the masks' zeros moved")
        again = _x6_buffers(prog, x, n, "mv", monkeypatch)
        assert torch.equal(again[3].view(torch.int32), new[3].view(torch.int32)) and torch.equal(again[4], new[4]), (n, "repeat")


def test_bf16x6_chains_outside_the_family_keep_the_two_waves_per_simd_kernels(golden):
    """Full NeRF (a 63-channel encoding, a skip connection, a 128-wide view layer) and a plain MLP (three
    raw inputs) are not what the matrix / vector waves kernels cover: `x6_organisation` says so, and
    their launches are the round-5 kernels' (held against the exact kernels by test_round5_gpu.py)."""
    nerf, _ = tk._load_nerf(golden("models"), "nerf", [4], True)
    assert nerf.program().x6_organisation() == "two waves per SIMD"
    assert nerf.program().x6_organisation(backward=True) == "two waves per SIMD"
    for other in ("mlp", "gaussian", "basic"):        # three raw inputs / an encoding that is not 16 j K blocks
        model, _ = tk._load_fourier(golden("models"), other)
        assert model.program().x6_organisation() == "two waves per SIMD", other
