"""Round 5 (needs an MI355X): the f32-ACCURATE split mode ("bf16x6": three bf16 parts per
operand, six bf16 matrix products per f32 product; csrc/mlp_bf16_ws.hip) held to the EXACT
mode's bar -- every reference-golden / oracle test of the exact-f32 kernels re-run in it at the
SAME tolerances (the test bodies themselves, re-entered with FFN_PRECISION=bf16x6), its error
against float64 next to the exact kernels' own, its slabs / masks / dZ against theirs -- and
BASELINE config 3 as a whole against the reference's own `fit` (tests/golden/fit_schedule_nerf.npz:
full NeRF + opacity-guided sampling across the crop removal)."""

import contextlib
import inspect
import io
import os

import numpy as np
import pytest
import torch

from tests import test_kernels_gpu as tk
from tests import test_pipeline_gpu as tp
from tests import test_round4_gpu as t4

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev():
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _quiet(fn, *args, **kwargs):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*args, **kwargs)


# ----------------------------------------------------------------------------------- config 3 through the reference's fit
def _replay_nerf_fit(tmp_path, focus_mode):
    """`Raycaster.fit` on the rig / model / opacity volume of make_fit_schedule_nerf.py, from the
    reference's initial weights, with the reference's generators (np.random for the permutations,
    the CPU torch generator for jitter and focus draws)."""
    import fourier_feature_nets_amd as ffn
    from tests.golden.make_fit_schedule_nerf import (ANNEAL_START, ANNEAL_STEPS, BATCH, CROP_STEPS, NERF,
                                                      NUM_STEPS, REPORT, SAMPLES, SIZE, TRAIN_CAMS, VAL_CAMS,
                                                      VOXEL_SCALE, VOXEL_SIDE)
    from tests.psnr_ensemble import write_npz
    g = np.load(os.path.join(GOLDEN, "fit_schedule_nerf.npz"))
    npz = write_npz(str(tmp_path / "scene.npz"), TRAIN_CAMS, VAL_CAMS, SIZE)
    model = ffn.NeRF(**NERF)
    model.load_state_dict({k[len("init/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("init/")})
    model = model.to(dev())
    opacity = ffn.Voxels(VOXEL_SIDE, VOXEL_SCALE)
    opacity.load_state_dict({"voxels": torch.from_numpy(g["opacity/voxels"]),
                             "bias": torch.from_numpy(g["opacity/bias"])})
    opacity = opacity.to(dev())
    train = _quiet(ffn.ImageDataset.load, npz, "train", SAMPLES, True, True, opacity, BATCH, "RGB",
                   anneal_start=ANNEAL_START, num_anneal_steps=ANNEAL_STEPS, focus_mode=focus_mode)
    val = _quiet(ffn.ImageDataset.load, npz, "val", SAMPLES, True, False, opacity, BATCH, "RGB",
                 focus_mode=focus_mode)
    train.sampler.noise_source = "host"
    torch.manual_seed(777)
    np.random.seed(777)
    caster = ffn.Raycaster(model)
    batches, modes, t_values = [], [], []
    orig_init, orig_step, orig_samples = ffn.TrainEngine.__init__, ffn.TrainEngine.train_step, ffn.TrainEngine._samples

    def recording_init(self, *a, **k):
        orig_init(self, *a, **k)
        self.loss_history = []

    def recording_step(self, dataset, batch, step, lr, rays=None, **kw):
        batches.append(torch.as_tensor(batch).cpu().numpy().astype(np.int64))
        modes.append(int(dataset.mode.value))
        return orig_step(self, dataset, batch, step, lr, rays=rays, **kw)

    def recording_samples(self, sampler, chunk, step):
        out = orig_samples(self, sampler, chunk, step)
        if sampler is train.sampler:
            t_values.append(out[0].detach().reshape(chunk.shape[0], -1).cpu().numpy().copy())
        return out

    ffn.TrainEngine.__init__, ffn.TrainEngine.train_step = recording_init, recording_step
    ffn.TrainEngine._samples = recording_samples
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            log = caster.fit(train, val, BATCH, 5e-4, NUM_STEPS, CROP_STEPS, REPORT, 0.1, 25000, 0.0, [])
    finally:
        ffn.TrainEngine.__init__, ffn.TrainEngine.train_step = orig_init, orig_step
        ffn.TrainEngine._samples = orig_samples
    return dict(g=g, model=model, train=train, val=val, caster=caster, batches=batches, modes=modes,
                t_values=t_values, log=log, stdout=buf.getvalue())


def test_config3_fit_against_the_references_own_fit(tmp_path):
    """BASELINE config 3 as a whole against the REFERENCE's own run (tests/golden/
    fit_schedule_nerf.npz from make_fit_schedule_nerf.py; train_nerf.py:85-141,
    ray_sampler.py:148-173,234-269,301-357): NeRF(4 x 64, skip at 2, view branch) trained through
    `Raycaster.fit` with half of every ray's samples drawn from the CDF of a frozen voxel opacity
    model, 15 optimiser steps across the crop removal, in `focus_mode="table"` (the reference's
    snapshot-at-construction semantics).  CDF tables against the reference's (checksums and every
    997th row of the rays that hit the volume), every training batch EXACTLY, the t-values handed to every training step, losses,
    report lines, final weights."""
    from tests.golden.make_fit_schedule_nerf import CDF_STRIDE, NUM_STEPS
    r = _replay_nerf_fit(tmp_path, "table")
    g, model = r["g"], r["model"]
    for name, ds in (("train", r["train"]), ("val", r["val"])):
        cdfs, valid = ds.sampler.cdfs, ds.sampler.valid != 0
        assert cdfs is not None
        ids = torch.nonzero(valid).flatten()
        assert int(ids.numel()) == int(g[name + "_cdf_sums"][2])
        assert np.array_equal(ids[::CDF_STRIDE].cpu().numpy(), g[name + "_cdf_ids"])
        # Rays that cross near-empty cells have alpha = 1 - exp(-sigma delta) ~ 1e-3 from a
        # cancellation whose LAST BIT depends on the exp at hand (glibc / SLEEF there, ROCm's here:
        # the reference's own value is that noisy): single weights move by ~1e-4 relative, a CDF
        # entry by up to ~2e-4; everything else agrees to rounding (mean |diff| < 2e-5)
        mine, theirs = cdfs[ids[::CDF_STRIDE]].cpu().numpy(), g[name + "_cdf_rows"]
        np.testing.assert_allclose(mine, theirs, atol=5e-4)
        assert float(np.abs(mine - theirs).mean()) < 2e-5
        assert float((np.abs(mine - theirs) > 1e-4).mean()) < 0.03
        c = cdfs[ids].double()
        np.testing.assert_allclose([float(c.sum()), float((c * c).sum())], g[name + "_cdf_sums"][:2], rtol=1e-5)
    assert r["modes"] == g["modes"].tolist() and r["modes"][5] == 2 and r["modes"][6] == 0
    assert len(r["batches"]) == len(g["batches"]) == NUM_STEPS + 1
    for step, (mine, theirs) in enumerate(zip(r["batches"], g["batches"])):
        assert np.array_equal(mine, theirs), step
    offsets = np.concatenate([[0], np.cumsum(g["t_rows"])])
    assert [len(t) for t in r["t_values"]] == g["t_rows"].tolist()
    for step, mine in enumerate(r["t_values"]):
        theirs = g["t_values"][offsets[step]:offsets[step + 1]]
        assert bool((np.diff(mine, axis=1) >= 0).all()), step
        # a CDF within 1e-7 of the reference's moves an inverse-transform sample continuously
        np.testing.assert_allclose(mine, theirs, rtol=2e-4, atol=2e-4, err_msg=str(step))
    losses = [float(x) for x in r["caster"].engine.loss_history]
    np.testing.assert_allclose(losses, g["losses"], rtol=3e-4, atol=1e-7)
    log = r["log"]
    assert [e.step for e in log] == g["log_steps"].tolist()
    np.testing.assert_allclose([e.train_psnr for e in log], g["log_train_psnr"], atol=5e-3)
    np.testing.assert_allclose([e.val_psnr for e in log], g["log_val_psnr"], atol=5e-3)
    mine = [ln for ln in r["stdout"].splitlines() if ln[:7].isdigit() or ln.startswith("Removing")]
    theirs = [ln for ln in str(g["stdout"]).splitlines() if ln[:7].isdigit() or ln.startswith("Removing")]
    assert len(mine) == len(theirs)
    for a, b in zip(mine, theirs):
        if a.startswith("Removing"):
            assert a == b
            continue
        a, b = a.split(), b.split()
        assert a[0] == b[0] and abs(float(a[4]) - float(b[4])) < 5e-3 and abs(float(a[6]) - float(b[6])) < 5e-3
    for key in g.files:
        if key.startswith("final/"):
            got = dict(model.state_dict())[key[len("final/"):]].detach().cpu().numpy()
            # 2e-4 like the tiny model's replay for all but a handful of entries: Adam moves a weight
            # by ~lr per step whatever the size of its gradient, so an entry whose gradient is within
            # rounding of zero (the 2^9-frequency columns of the first layer) may walk the other way
            # in either implementation, by up to 2 lr per step -- under 1 % of a tensor's entries
            diff = np.abs(got - g[key])
            assert float(diff.max()) <= 2 * 5e-4 * (NUM_STEPS + 1), (key, float(diff.max()))
            assert int((diff > 2e-4).sum()) <= max(2, 0.01 * diff.size), (key, int((diff > 2e-4).sum()), diff.size)


def test_config3_live_focus_sampling_draws_the_tables_t_values(tmp_path):
    """`focus_mode="live"` (no CDF table: the opacity model probed per batch) hands the SAME
    t-values to every training step of the same `fit` as `focus_mode="table"`, bit for bit -- and
    so the same losses and weights."""
    a = _replay_nerf_fit(tmp_path, "table")
    b = _replay_nerf_fit(tmp_path, "live")
    assert b["train"].sampler.cdfs is None and a["train"].sampler.cdfs is not None
    assert len(a["t_values"]) == len(b["t_values"]) > 10
    for step, (ta, tb) in enumerate(zip(a["t_values"], b["t_values"])):
        assert np.array_equal(ta, tb), step
    assert [float(x) for x in a["caster"].engine.loss_history] == [float(x) for x in b["caster"].engine.loss_history]
    for (ka, va), (kb, vb) in zip(a["model"].state_dict().items(), b["model"].state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


# ----------------------------------------------------------------------------------- bf16x6: the exact mode's tests, re-entered
def _case(fn, **kwargs):
    label = fn.__name__[len("test_"):] + ("[%s]" % "-".join(str(v) for v in kwargs.values()) if kwargs else "")
    return pytest.param(fn, kwargs, id=label)


# every test of the exact-f32 kernels against REFERENCE GOLDENS or the oracle whose models are
# chains of <= 256 channels (the mode's coverage; 512-wide chains raise in it, tested below)
EXACT_MODE_TESTS = (
    [_case(tk.test_fused_mlp_forward_against_golden, name=n) for n in ("mlp", "basic", "positional", "gaussian")] +
    [_case(tk.test_fused_mlp_forward_ragged_sizes)] +
    [_case(tk.test_fused_mlp_backward_against_golden, name=n) for n in ("mlp", "basic", "positional", "gaussian")] +
    [_case(tk.test_fused_nerf_forward_backward_against_golden, name="nerf", skips=[4], inc=True),
     _case(tk.test_fused_nerf_forward_backward_against_golden, name="nerf_small", skips=[2], inc=False),
     _case(tk.test_fused_mlp_backward_many_blocks),
     _case(tk.test_fused_mlp_empty_batch),
     _case(tp.test_render_and_autograd_match_oracle),
     _case(tp.test_fit_trajectory_matches_reference),
     _case(tp.test_nerf_train_step_matches_oracle),
     _case(tp.test_psnr_after_long_training_matches_oracle)] +
    [_case(t4.test_any_layer_width_forward_and_gradients_against_the_oracle, kind=k)
     for k in ("mlp96", "mlp7", "gaussian200", "nerf192", "nerf32", "nerf100")] +
    [_case(t4.test_any_layer_width_optimisation_step_against_the_oracle, kind=k)
     for k in ("mlp96", "nerf192", "nerf32")] +
    [_case(t4.test_fit_against_the_references_own_fit_small_ensemble),
     _case(t4.test_fit_schedule_across_the_crop_removal),
     _case(test_config3_fit_against_the_references_own_fit)])


# cases whose model has a hidden layer of 256 channels reading 256 channels: a FULL weight-gradient
# unit (four 128 x 128 quadrants), the kind the three-part weight-gradient kernel takes
WGRAD_X6_EXPECTED = {(tk.test_fused_mlp_backward_against_golden, "positional"),
                     (tk.test_fused_nerf_forward_backward_against_golden, "nerf")}


@pytest.mark.parametrize("fn,kwargs", EXACT_MODE_TESTS)
def test_exact_mode_tests_in_the_f32_accurate_split_mode(fn, kwargs, golden, tmp_path, monkeypatch):
    """The reference-golden / oracle tests of the EXACT mode, unchanged -- same bodies, same
    tolerances -- with every model they build in the opt-in bf16x6 mode (FFN_PRECISION is read by
    the model constructors: inference calls, the training forward, backward data AND the weight
    gradients of every full unit run the three-part split kernels; units with fewer than four
    quadrants the exact-f32 kernel that folds them)."""
    import fourier_feature_nets_amd as ffn
    monkeypatch.setenv("FFN_PRECISION", "bf16x6")
    probe = ffn.MLP(3, 4)
    assert probe.precision == probe.train_precision == "bf16x6"
    seen = []
    from fourier_feature_nets_amd import _lib
    real = _lib.call

    def spy(name, *a):
        seen.append(name)
        return real(name, *a)

    monkeypatch.setattr(_lib, "call", spy)      # (ops._call resolves _lib.call per launch)
    params = inspect.signature(fn).parameters
    extra = {}
    if "golden" in params:
        extra["golden"] = golden
    if "tmp_path" in params:
        extra["tmp_path"] = tmp_path
    fn(**kwargs, **extra)
    used = {n for n in seen if n.startswith("ffn_mlp_")}
    if fn is not tk.test_fused_mlp_empty_batch:
        assert used & {"ffn_mlp_forward_bf16x6", "ffn_mlp_forward_bf16x6_train"}, sorted(used)
    # a backward pass through a chain with a full 256 x 256 unit ran the three-part weight gradients
    if (fn, kwargs.get("name")) in WGRAD_X6_EXPECTED:
        assert "ffn_mlp_wgrad_units_bf16x6" in used, sorted(used)
    # no exact-f32 (or bf16x3) chain kernel ran behind the test's back
    assert not used & {"ffn_mlp_forward", "ffn_mlp_backward_data", "ffn_mlp_forward_bf16x3",
                       "ffn_mlp_forward_bf16x3_train", "ffn_mlp_backward_data_bf16x3",
                       "ffn_mlp_wgrad_units_bf16x3"}, sorted(used)


# ----------------------------------------------------------------------------------- bf16x6 against float64 and the exact kernels
@pytest.mark.parametrize("layers", [2, 8])
def test_bf16x6_error_against_float64_next_to_the_exact_kernels(layers):
    """The stop rule's error half as a test: logits and every weight gradient of a raw-input
    256-channel ReLU MLP against float64 -- bf16x6 within 2x the exact-f32 kernels' own error
    (plus 1e-7 of the scale), bf16x3 an order of magnitude outside it."""
    from tests.probe_bf16x6 import error_table
    rows = error_table(dev(), layers, n=4096)["modes"]
    exact, split6, split9, split3 = rows["f32"], rows["bf16x6_6p"], rows["bf16x6_9p"], rows["bf16x3"]
    for key in ("logits_max_abs_err_over_max_abs", "logits_rms_err_over_rms"):
        assert split6[key] <= 2.0 * exact[key] + 1e-7, (key, rows)
        assert split9[key] <= 2.0 * exact[key] + 1e-7, (key, rows)
    # gradients: sums over every sample, whose rounding depends on how the weight-gradient planner
    # cuts the batch into segments -- the exact kernels' own error moves by 2x between two plans of
    # the same batch (measured: a head bias 3.4e-7 / 6.6e-7); over eight seeds the ratio of the
    # worst tensor is 1.5 in the median and 2.8 at most (profiles/r05_bf16x6_probe.json)
    for key in ("worst_tensor_grad_max_abs_err_over_max_abs", "worst_tensor_grad_rms_err_over_rms"):
        assert split6[key] <= 3.0 * exact[key] + 1e-7, (key, rows)
        assert split9[key] <= 3.0 * exact[key] + 1e-7, (key, rows)
    assert split3["logits_rms_err_over_rms"] > 5.0 * split6["logits_rms_err_over_rms"], rows
    assert split6["inference_equals_training_forward"] and split9["inference_equals_training_forward"]


@pytest.mark.parametrize("name", ["positional", "gaussian", "nerf", "nerf_small", "basic", "mlp"])
def test_bf16x6_slabs_masks_and_dz_against_the_exact_kernels(golden, name):
    """The training forward / backward-data kernels of the mode against the exact-f32 ones, buffer
    against buffer (same slab and mask formats): encoding features BIT-identical (the f32 kernels'
    polynomials), activations within 2e-6 of their slab's scale (bf16x3: 2e-5), mask bits equal
    except where a pre-activation is within rounding of zero, dZ within 6e-6 (bf16x3: 6e-5), the
    weight gradients -- exact-f32 units on either side's slabs -- within 1e-5."""
    from tests.test_kernels_gpu import _load_fourier, _load_nerf
    g = golden("models")
    if name.startswith("nerf"):
        model, _ = _load_nerf(g, name, [4] if name == "nerf" else [2], name == "nerf")
    else:
        model, _ = _load_fourier(g, name)
    torch.manual_seed(11)
    n = 1000                                     # ragged: 31.25 blocks
    x = torch.rand(n, 3, device=dev()) * 2 - 1
    views = torch.nn.functional.normalize(torch.randn(n, 3, device=dev()), dim=1) if model.use_view else None
    prog = model.program()
    saved, logits = {}, {}
    for mode in ("f32", "bf16x6"):
        buf = torch.zeros((prog.saved_floats(n),), dtype=torch.float32, device=dev())
        logits[mode] = prog.forward(x, views, buf, precision=mode)
        saved[mode] = buf
    scale = max(float(logits["f32"].abs().max()), 1.0)
    assert float((logits["bf16x6"] - logits["f32"]).abs().max()) <= 4e-6 * scale
    with torch.no_grad():
        assert torch.equal(prog.forward(x, views, None, precision="bf16x6"), logits["bf16x6"])
    acts_e, masks_e = prog._split_saved(saved["f32"], n)
    acts_f, masks_f = prog._split_saved(saved["bf16x6"], n)
    blocks = (n + 31) // 32
    feature_slots = len(prog.enc_slot)
    for slot in range(prog.fwd.num_slots + feature_slots):
        ch, off = prog.fwd.slot_channels[slot], prog.fwd.slot_offset[slot]
        a = acts_e[off * blocks * 32:(off + ch) * blocks * 32]
        b = acts_f[off * blocks * 32:(off + ch) * blocks * 32]
        if slot >= prog.fwd.num_slots:
            assert torch.equal(a, b), ("feature slab", slot)
            continue
        tol = 2e-6 * max(float(a.abs().max()), 1.0)
        assert float((a - b).abs().max()) <= tol, (slot, float((a - b).abs().max()), tol)
    me, mf = masks_e.view(torch.int32), masks_f.view(torch.int32)
    differing = (me ^ mf) != 0
    bits = sum(bin(int(v) & 0xffffffff).count("1") for v in (me ^ mf)[differing].cpu().tolist())
    assert bits <= max(4, int(1e-5 * me.numel() * 32)), bits
    d_logits = torch.randn(n, 4, device=dev()) / n
    ws = prog.workspace(n)
    dz, flat = {}, {}
    for mode in ("f32", "bf16x6"):
        ws.dz.zero_()
        flat[mode] = torch.zeros((prog.num_grad_floats,), dtype=torch.float32, device=dev())
        prog.backward(d_logits, x, views, saved["f32"], flat[mode], precision=mode)
        dz[mode] = ws.dz.clone()
    assert prog.bwd_x6 is not None
    for slot in range(prog.fwd.num_slots):
        ch, off = prog.fwd.slot_channels[slot], prog.fwd.slot_offset[slot]
        a = dz["f32"][off * blocks * 32:(off + ch) * blocks * 32]
        b = dz["bf16x6"][off * blocks * 32:(off + ch) * blocks * 32]
        tol = 6e-6 * float(a.abs().max())
        assert float((a - b).abs().max()) <= tol, (slot, float((a - b).abs().max()), tol)
    assert float((flat["f32"] - flat["bf16x6"]).abs().max()) <= 1e-5 * float(flat["f32"].abs().max())


def test_bf16x6_batch_independence_and_ragged_sizes(golden):
    """Inference in the mode: a sample's logits do not depend on the batch around it (bit for
    bit: the same six products in the same order whatever the block / pass a sample lands in),
    sizes around the 64-sample pass and the 32-sample block, and an empty batch."""
    from tests.test_kernels_gpu import _load_fourier
    model, _ = _load_fourier(golden("models"), "positional")
    model.precision = "bf16x6"
    torch.manual_seed(3)
    x = torch.rand(20000 + 13, 3, device=dev()) * 2 - 1
    with torch.no_grad():
        whole = model(x)
        assert model(x[:0]).shape == (0, 4)
        for n in (1, 31, 32, 33, 63, 64, 65, 127, 129, 4097):
            assert torch.equal(model(x[:n]), whole[:n]), n
        for lo in (64, 96, 1000 * 32):
            assert torch.equal(model(x[lo:lo + 777]), whole[lo:lo + 777]), lo


def test_bf16x6_refuses_what_it_does_not_cover(golden, monkeypatch):
    """512-wide chains have no three-part kernels (their X image does not fit the LDS): a mode
    REQUESTED for such a model (assigned to its attributes) raises, it does not fall back to another
    arithmetic.  (Only the environment-wide default does: next test.)"""
    from tests.test_kernels_gpu import _load_fourier
    monkeypatch.delenv("FFN_PRECISION", raising=False)
    model, _ = _load_fourier(golden("models"), "gaussian512")
    x = torch.rand(100, 3, device=dev())
    model.precision = "bf16x6"
    with torch.no_grad(), pytest.raises(NotImplementedError):
        model(x)
    model.precision = "f32"
    model.train_precision = "bf16x6"
    with pytest.raises(NotImplementedError):
        model(x)


def test_bf16x6_as_the_process_default_runs_uncovered_chains_in_exact_f32(golden, monkeypatch):
    """`FFN_PRECISION=bf16x6` (the environment-wide default, what `pytest --precision bf16x6` sets) is
    "the mode wherever it has kernels": a 512-wide chain built under it runs the EXACT-f32 kernels
    -- logits bit-identical to an exact-mode model -- while a 256-wide one runs the three-part ones."""
    from fourier_feature_nets_amd import _lib
    from tests.test_kernels_gpu import _load_fourier
    exact, _ = _load_fourier(golden("models"), "gaussian512")
    exact.precision = exact.train_precision = "f32"
    exact._precision_is_default = False
    monkeypatch.setenv("FFN_PRECISION", "bf16x6")
    wide, _ = _load_fourier(golden("models"), "gaussian512")
    narrow, _ = _load_fourier(golden("models"), "gaussian")
    assert wide.precision == narrow.precision == "bf16x6" and wide._precision_is_default
    seen = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (seen.append(name), real(name, *a))[1])
    x = torch.rand(100, 3, device=dev())
    with torch.no_grad():
        assert torch.equal(wide(x), exact(x))
    assert "ffn_mlp_forward" in seen and not any("bf16x6" in n for n in seen)
    wide(x).square().sum().backward()          # the training pass falls back the same way
    assert not any("bf16x6" in n for n in seen)
    del seen[:]
    with torch.no_grad():
        narrow(x)
    assert "ffn_mlp_forward_bf16x6" in seen


# ----------------------------------------------------------------------------------- look-ahead sampling
def test_announced_next_batch_gives_the_same_steps(golden):
    """`TrainEngine.train_step(..., lookahead=(dataset, next rays, next step))` -- the next step's
    sampling kernels enqueued under this step's gradient all-reduce in data parallel -- takes
    the SAME optimisation steps as the plain call: stratified + annealed sampling from the device
    generator, bit for bit while every announcement is honoured; an announcement that is not
    (step 3 here) falls back to fresh samples and costs one block of noise, nothing else."""
    import fourier_feature_nets_amd as ffn
    g = golden("training")
    results = []
    for announce in (False, True):
        model = tp._small_model(g)
        train = _quiet(ffn.ImageDataset.load, tp.SCENE, "train", 16, True, True, anneal_start=0.2,
                       num_anneal_steps=8)
        engine = ffn.TrainEngine(model)
        gen = torch.Generator(device=dev()).manual_seed(7)
        batches = [torch.randint(0, len(train), (96,), device=dev(), generator=gen) for _ in range(7)]
        rays = [train.ray_ids(b) for b in batches]
        torch.manual_seed(31)
        losses = []
        for step in range(6):
            ahead = None
            if announce:
                # (step 3 announces a batch that does not come: the engine must notice)
                nxt = rays[step + 1] if step != 3 else rays[0]
                ahead = (train, nxt, step + 1)
            losses.append(float(engine.train_step(train, batches[step], step, 5e-4, rays=rays[step], lookahead=ahead)))
        engine.check_finite()
        results.append((losses, engine.flat.clone()))
    # steps 0..3 are identical bit for bit; from step 4 on the runs differ only through the noise
    # block the unhonoured announcement consumed
    assert results[0][0][:4] == results[1][0][:4]
    assert all(abs(a - b) < 0.05 for a, b in zip(results[0][0][4:], results[1][0][4:]))


def test_announced_next_batch_without_noise_is_bit_identical(golden):
    """The same with a non-stratified sampler (nothing to consume): every step, honoured or not,
    bit for bit -- losses and final weights."""
    import fourier_feature_nets_amd as ffn
    g = golden("training")
    results = []
    for announce in (False, True):
        model = tp._small_model(g)
        train = _quiet(ffn.ImageDataset.load, tp.SCENE, "train", 16, True, False, anneal_start=0.2,
                       num_anneal_steps=8)
        engine = ffn.TrainEngine(model)
        gen = torch.Generator(device=dev()).manual_seed(7)
        batches = [torch.randint(0, len(train), (96,), device=dev(), generator=gen) for _ in range(7)]
        rays = [train.ray_ids(b) for b in batches]
        losses = []
        for step in range(6):
            ahead = (train, rays[step + 1] if step != 3 else rays[0], step + 1) if announce else None
            losses.append(float(engine.train_step(train, batches[step], step, 5e-4, rays=rays[step], lookahead=ahead)))
        results.append((losses, engine.flat.clone()))
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1])


# ----------------------------------------------------------------------------------- 512-wide model as its own opacity model
@pytest.mark.exact_only(reason="compares the one-launch kernels (exact f32) with the multi-launch paths; bf16x3 has kernels for this model, so the multi-launch side computes in it", modes=("bf16x3",))
def test_wide_model_as_its_own_opacity_model_renders_on_the_one_launch_paths(golden):
    """orbit_video.py's configuration (orbit_video.py:66-78: without --opacity-model the radiance
    field itself guides the sampler) for BASELINE config 5's 512-wide Gaussian-feature model: the
    coarse pass runs in ONE launch (the pair-of-waves variant of the fused focus kernel) and the
    frame in one launch of the fused render kernel; the frame equals, byte for byte, the one
    rendered through a sampler with the reference's precomputed CDF table."""
    import fourier_feature_nets_amd as ffn
    from tests.test_round2_gpu import SCENE, _scene_sampler
    model, _ = tk._load_fourier(golden("models"), "gaussian512")
    data = np.load(SCENE)
    cams = _scene_sampler(8).cameras
    frames = {}
    for mode in ("table", "live"):
        smp = _quiet(ffn.RaySampler, data["bounds"], cams, 32, False, model, 64, device=dev(), focus_mode=mode)
        if mode == "live":
            assert smp.cdfs is None and smp._can_fuse_focus(16) and model.program().wide
            seen = []
            from fourier_feature_nets_amd import _lib
            real = _lib.call
            _lib.call = lambda name, *a: (seen.append(name), real(name, *a))[1]
            try:
                frames[mode] = ffn.Raycaster(model).render_image(smp, 1, 64)
            finally:
                _lib.call = real
            assert "ffn_focus_fused" in seen and "ffn_render_fused_fwd" in seen, sorted(set(seen))
            assert "ffn_mlp_forward" not in seen
        else:
            frames[mode] = ffn.Raycaster(model).render_image(smp, 1, 64)
    assert frames["live"].shape == frames["table"].shape and frames["live"].dtype == np.uint8
    assert np.array_equal(frames["live"], frames["table"])
    assert int(frames["live"].max()) > 0


# ----------------------------------------------------------------------------------- forward / backward pairing
def test_backward_checks_the_mask_region_of_its_own_forward(golden):
    """A training forward in exact f32 whose launch was split (full rounds + a short last round on
    the team kernels) keeps the tail blocks' ReLU masks in a region only the f32 backward with the
    same split reads: a backward in another mode on THAT buffer raises; the same buffer filled
    without a split, and an older buffer with its own record, differentiate in either mode."""
    model, _ = tk._load_fourier(golden("models"), "positional")
    prog = model.program()
    n_split = 32 * (prog._resident_waves() + 10)
    if prog._tail_plan(n_split) is None:
        pytest.skip("no tail split at this size on this device")
    n_plain = 1000
    assert prog._tail_plan(n_plain) is None
    grads = torch.empty((prog.num_grad_floats,), dtype=torch.float32, device=dev())
    bufs = {}
    for n in (n_plain, n_split):
        x = torch.rand(n, 3, device=dev()) * 2 - 1
        buf = torch.zeros((prog.saved_floats(n),), dtype=torch.float32, device=dev())
        prog.forward(x, None, buf, precision="f32")
        bufs[n] = (x, buf, torch.randn(n, 4, device=dev()) / n)
    x, buf, dl = bufs[n_plain]            # the OLDER buffer: checked against its own forward
    prog.backward(dl, x, None, buf, grads, precision="bf16x3")
    prog.backward(dl, x, None, buf, grads, precision="f32")
    x, buf, dl = bufs[n_split]
    prog.backward(dl, x, None, buf, grads, precision="f32")
    with pytest.raises(RuntimeError, match="tail split"):
        prog.backward(dl, x, None, buf, grads, precision="bf16x3")
    with pytest.raises(RuntimeError, match="tail split"):
        prog.backward(dl, x, None, buf, grads, precision="bf16x6")
