"""Round-3 parity tests on the MI355X: the YCrCb colour space (a6), the ``subsample_index``
filter of ``get_rays`` (a15), the split-bf16 training kernels against the REFERENCE's gradient
goldens, BASELINE config 5's actual combination (512-wide model + empty-space skipping + an
optimiser step) against the masked oracle, micro-batched training steps, and data-parallel
``fit`` (2 ranks == 1 rank, shared-seed permutation, sharded validation)."""

import contextlib
import io
import os
import socket
import subprocess
import sys

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


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _quiet(fn, *args, **kwargs):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*args, **kwargs)


# ----------------------------------------------------------------------------------- a6: YCrCb
def test_ycrcb_kernel_equals_the_oracle_on_every_chroma_pair():
    """K8b (ffn_ycrcb_to_rgb_u8) == the oracle's restatement of OpenCV's 8-bit YCrCb -> RGB,
    bit for bit: all 65 536 (Cr, Cb) pairs at several luma values (saturation on both ends),
    plus the known answers of the primaries.  Parity vs OpenCV itself is unpinned (no cv2)."""
    from fourier_feature_nets_amd import ops
    cr, cb = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    for y in (0, 1, 76, 128, 200, 254, 255):
        ycc = np.stack([np.full_like(cr, y), cr, cb], -1)            # (256,256,3)
        got = ops.ycrcb_to_rgb_u8(_t(ycc.copy()).to(dev())).cpu().numpy()
        assert np.array_equal(got, orc.ycrcb_to_rgb_u8(ycc)), y
    prim = np.array([[[76, 255, 85], [150, 21, 43], [29, 107, 255], [255, 128, 128]]], np.uint8)
    got = ops.ycrcb_to_rgb_u8(_t(prim.copy()).to(dev())).cpu().numpy()[0]
    # (the 8-bit round trip of a primary is within one or two levels, like OpenCV's)
    assert np.abs(got.astype(int) - np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255],
                                              [255, 255, 255]])).max() <= 2
    empty = torch.zeros((0, 3), dtype=torch.uint8, device=dev())
    assert ops.ycrcb_to_rgb_u8(empty).numel() == 0


def test_ycrcb_dataset_ground_truth_and_image_assembly():
    """color_space="YCrCb" end to end: the dataset's ground truth is the u8 image converted like
    cv2.COLOR_RGB2YCrCb and divided by 255 (image_dataset.py:114-118); to_image / render_image
    convert the truncated u8 frame back (ray_sampler.py:193-198, ray_dataset.py:176-181); alpha,
    index maps and the loss path are those of the RGB dataset."""
    import fourier_feature_nets_amd as ffn
    data = np.load(SCENE)
    n_train = int(data["split_counts"][0])
    ycc = _quiet(ffn.ImageDataset.load, SCENE, "train", 16, True, False, device=dev(),
                 color_space="YCrCb")
    rgb = _quiet(ffn.ImageDataset.load, SCENE, "train", 16, True, False, device=dev())
    assert ycc.color_space == "YCrCb" and rgb.color_space == "RGB"
    expect = orc.rgb_to_ycrcb_u8(data["images"][:n_train, ..., :3]).astype(np.float32) / 255
    assert np.array_equal(ycc.colors.cpu().numpy(), expect.reshape(-1, 3))
    assert torch.equal(ycc.alphas, rgb.alphas) and torch.equal(ycc.crop_index, rgb.crop_index)
    assert ycc.subset([0, 1], 8, False, "sub").color_space == "YCrCb"
    # image assembly: predicted colours (YCrCb in [0,1]) -> u8 -> RGB
    smp = ycc.sampler
    cam = 1
    torch.manual_seed(3)
    n_valid = int(smp._valid_for_camera(cam).numel())
    colors = torch.rand(n_valid, 3).numpy()
    as_rgb = smp.to_image(cam, colors, "RGB")
    as_ycc = smp.to_image(cam, colors, "YCrCb")
    assert np.array_equal(as_ycc, orc.ycrcb_to_rgb_u8(as_rgb))
    # (pixels no ray reaches are (0,0,0) in YCrCb = dark green in RGB, like the reference's)
    assert np.array_equal(as_ycc[as_rgb.sum(-1) == 0][:1], orc.ycrcb_to_rgb_u8(np.zeros((1, 3), np.uint8)))
    with pytest.raises(NotImplementedError):
        smp.to_image(cam, colors, "HSV")
    # RayDataset.to_image follows the dataset's own colour space
    index = ycc.index_for_camera(cam)
    per_ray = torch.rand(len(index), 3).numpy()
    assert np.array_equal(ycc.to_image(cam, per_ray), orc.ycrcb_to_rgb_u8(rgb.to_image(cam, per_ray)))
    # Raycaster.render_image: the fused kernel's u8 frame through K8b == converting the RGB frame
    from tests.test_pipeline_gpu import _small_model
    model = _small_model(np.load(os.path.join(GOLDEN, "training.npz")))
    caster = ffn.Raycaster(model)
    frame_rgb = caster.render_image(smp, cam, 64)
    frame_ycc = caster.render_image(smp, cam, 64, "YCrCb")
    assert np.array_equal(frame_ycc, orc.ycrcb_to_rgb_u8(frame_rgb))
    caster.fused_render = False                      # the three-pass render takes the same route
    assert np.array_equal(caster.render_image(smp, cam, 64, "YCrCb"), frame_ycc)


def test_ycrcb_training_step_against_the_oracle():
    """One optimisation step on a YCrCb dataset == the oracle's step on the converted ground
    truth (the colour space only changes what the loss compares against)."""
    import fourier_feature_nets_amd as ffn
    from tests.test_pipeline_gpu import _small_model
    g = np.load(os.path.join(GOLDEN, "training.npz"))
    model = _small_model(g)
    ref = orc.OracleFourierMLP(model.a_values.data.cpu().clone(), model.b_values.data.cpu().clone(),
                               [l.weight.data.cpu().clone() for l in model.layers],
                               [l.bias.data.cpu().clone() for l in model.layers])
    train = _quiet(ffn.ImageDataset.load, SCENE, "train", 16, True, False, device=dev(),
                   color_space="YCrCb")
    engine = ffn.TrainEngine(model)
    batch = torch.arange(0, len(train), 7, device=dev())
    loss = float(engine.train_step(train, batch, None, 5e-4))
    data = np.load(SCENE)
    n_train = int(data["split_counts"][0])
    rays = train.ray_ids(batch).cpu()
    smp = train.sampler
    state = {"starts": smp.starts.cpu(), "directions": smp.directions.cpu(), "near_far": smp.near_far.cpu()}
    pos, view, t, _ = orc.sample(state, rays.numpy(), None, 16)
    colors = _t(orc.rgb_to_ycrcb_u8(data["images"][:n_train, ..., :3]).astype(np.float32) / 255).reshape(-1, 3)
    alphas = _t(data["images"][:n_train, ..., 3].astype(np.float32) / 255).reshape(-1)
    gc, ga = orc.ground_truth(colors, alphas, rays)
    ref_loss = orc.OracleTrainer(ref, 5e-4).step(pos, view, t, gc, ga, 5e-4)
    assert abs(loss - ref_loss) < 2e-6 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    for layer, w in zip(model.layers, ref.weights):
        np.testing.assert_allclose(layer.weight.detach().cpu().numpy(), w.detach().numpy(), rtol=0, atol=3e-5)


# ----------------------------------------------------------------------------------- a15: subsample_index
@pytest.mark.parametrize("mode", ["Full", "Center", "Sparse"])
def test_subsample_index_filter_matches_the_list_filter(mode):
    """``dataset.subsample_index = {pixel ids}`` keeps the rays whose id modulo rays_per_camera
    is in the set, AFTER the mode's index map and BEFORE the validity filter
    (image_dataset.py:364-386) -- checked against that list comprehension, order included, for
    get_rays / ray_ids / epoch_ray_ids, and that clearing the set restores the plain path."""
    import fourier_feature_nets_amd as ffn
    ds = _quiet(ffn.ImageDataset.load, SCENE, "train", 8, True, False, device=dev())
    ds.mode = getattr(ffn.RayDataset.Mode, mode)
    per_cam = ds.sampler.rays_per_camera
    rng = np.random.default_rng(4)
    subset = set(rng.choice(per_cam, per_cam // 3, replace=False).tolist())
    idx = rng.permutation(len(ds))[:min(len(ds), 300)]
    plain = ds.ray_ids(idx).cpu().tolist()
    ds.subsample_index = subset
    assert ds.subsample_index == subset
    index = ds._mode_index()
    mapped = idx.tolist() if index is None else index.cpu().numpy()[idx].tolist()
    bad = ds.sampler.invalid_rays
    expect = [i for i in mapped if i % per_cam in subset]          # image_dataset.py:381-383
    expect = [i for i in expect if i not in bad]                   # ray_sampler.py:283-295
    assert 0 < len(expect) < len(plain)
    assert ds.ray_ids(idx).cpu().tolist() == expect
    samples = ds.get_rays(_t(idx), None)
    assert samples.rays.cpu().tolist() == expect and samples.positions.shape[0] == len(expect)
    # the epoch-at-once filter used by fit gives the same per-batch sets
    order = _t(idx).to(dev())
    rays, bounds = ds.epoch_ray_ids(order, 64)
    for bi, start in enumerate(range(0, len(idx), 64)):
        part = ds.ray_ids(idx[start:start + 64]).cpu().tolist()
        assert rays[bounds[bi]:bounds[bi + 1]].cpu().tolist() == part
    ds.subsample_index = None
    assert ds.ray_ids(idx).cpu().tolist() == plain


# ----------------------------------------------------------------------------------- split-bf16 vs the reference
@pytest.mark.parametrize("name", ["mlp", "basic", "positional", "gaussian", "nerf", "nerf_small"])
def test_split_bf16_gradients_against_the_reference_goldens(golden, name):
    """`train_precision = "bf16x3"` pinned DIRECTLY against the reference's gradients
    (tests/golden/models.npz:*/grad/*, captured from the reference's autograd), not only against
    the f32 twin: every parameter gradient within 1e-3 of its tensor's scale, relative L2 over
    all compared entries <= 1e-4 (exact-f32 kernels: 5e-4 abs; the split products carry 16
    mantissa bits per operand)."""
    from tests.test_kernels_gpu import _load_fourier, _load_nerf
    g = golden("models")
    if name.startswith("nerf"):
        model, _ = _load_nerf(g, name, [4] if name == "nerf" else [2], name == "nerf")
        args = (_t(g["x"]).to(dev()), _t(g["v"]).to(dev()))
    else:
        model, _ = _load_fourier(g, name)
        args = (_t(g["x"]).to(dev()),)
    model.train_precision = "bf16x3"
    y = model(*args)
    probe = torch.linspace(-1, 1, y.numel()).reshape(y.shape).to(dev())
    (y * probe).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[name + "/out"], rtol=2e-4,
                               atol=2e-4 * max(float(np.abs(g[name + "/out"]).max()), 1.0))
    num = den = 0.0
    compared = 0
    for key, par in model.named_parameters():
        if not par.requires_grad:
            continue
        got = par.grad.detach().cpu().double().reshape(-1)
        full = "%s/grad/%s" % (name, key)
        if full in g.files:
            ref = _t(g[full]).double().reshape(-1)
        else:                       # big tensors: the golden keeps the first 512 entries + sums
            ref = _t(g["%s/gradhead/%s" % (name, key)]).double()
            got_all = got
            got = got[:512]
            total = float(g["%s/gradsum/%s" % (name, key)])
            mass = float(g["%s/gradabs/%s" % (name, key)])
            assert abs(float(got_all.sum()) - total) <= 1e-4 * max(mass, 1e-6), key
            assert abs(float(got_all.abs().sum()) - mass) <= 1e-4 * max(mass, 1e-6), key
        scale = max(float(ref.abs().max()), 1e-6)
        assert float((got - ref).abs().max()) <= 1e-3 * scale, (key, float((got - ref).abs().max()), scale)
        num += float(((got - ref) ** 2).sum())
        den += float((ref ** 2).sum())
        compared += 1
    assert compared >= 4
    assert (num / den) ** 0.5 <= 1e-4, (num / den) ** 0.5
