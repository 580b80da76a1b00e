"""Host-side logic that needs no GPU: chain / job planning for the fused MLP kernels, the
encoding column maps, camera rigs, and the data-parallel arithmetic (2 gloo ranks on CPU, with
the oracle standing in for the kernels)."""

import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import fourier_feature_nets_amd as ffn
from fourier_feature_nets_amd.mlp_engine import EncodingSpec, MlpProgram
from oracle import ffn_oracle as orc


def _plan(model):
    enc, specs = model._chain(torch.device("cpu"))
    return MlpProgram(enc, specs, torch.device("cpu"), planning_only=True)


def test_models_keep_the_reference_state_dict_layout():
    nerf = ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True)
    shapes = {k: tuple(v.shape) for k, v in nerf.state_dict().items()}
    assert list(shapes)[:2] == ["pos_encoding", "view_encoding"]
    assert shapes["layers.0.weight"] == (256, 63) and shapes["layers.4.weight"] == (256, 319)
    assert shapes["hidden_view.weight"] == (128, 283) and shapes["color_out.weight"] == (3, 128)
    assert sum(p.numel() for p in nerf.parameters() if p.requires_grad) == 595844
    tiny = ffn.PositionalFourierMLP(3, 4, 5.5)
    assert tuple(tiny.b_values.shape) == (3, 255)
    assert sum(p.numel() for p in tiny.parameters() if p.requires_grad) == 263428
    assert tiny.use_view is False and nerf.use_view is True
    # frequency tables are bit-identical to the oracle's restatement of the reference
    assert torch.equal(tiny.b_values.data, orc.positional_b_values(5.5, 256, 3))
    assert torch.equal(nerf.pos_encoding.data, orc.axis_frequency_matrix(9, 10))


def test_encoding_internal_order_covers_every_natural_column_once():
    for freq, inc in [(255, False), (30, True), (12, True), (0, False), (256, False), (3, False)]:
        b = None if freq == 0 else torch.zeros(3, freq)
        enc = EncodingSpec(b, None, math.pi, inc, torch.device("cpu"))
        nat = [enc.natural_index(c) for c in range(enc.width)]
        real = [n for n in nat if n >= 0]
        assert sorted(real) == list(range(enc.natural_width))
        assert enc.width % 32 == 0 and enc.width >= enc.natural_width
        # cos at even internal channels, sin at odd ones
        for k in range(freq):
            assert nat[2 * k] == k and nat[2 * k + 1] == freq + k


@pytest.mark.parametrize("make", [
    lambda: ffn.PositionalFourierMLP(3, 4, 5.5),
    lambda: ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True),
    lambda: ffn.MLP(3, 4, num_channels=64),
    lambda: ffn.NeRF(4, 64, 5, 6, 2, 3, [2], False),
    lambda: ffn.GaussianFourierMLP(3, 4, 10.0, num_channels=512),
    # widths the kernels have no tile count for run zero-padded (any nn.Linear width is accepted)
    lambda: ffn.MLP(3, 4, num_channels=96),
    lambda: ffn.NeRF(8, 192, 9, 10, 3, 4, [4], True),
    lambda: ffn.NeRF(8, 32, 9, 10, 3, 4, [4], True),
    lambda: ffn.NeRF(8, 512, 9, 10, 3, 4, [4], True),
    lambda: ffn.PositionalFourierMLP(3, 4, 5.5, num_channels=384),
    lambda: ffn.MLP(3, 4, num_layers=2, num_channels=7),
])
def test_chain_and_wgrad_plans(make):
    model = make()
    prog = _plan(model)
    n_dense = len(prog.layers)
    stepped = [i for i in range(n_dense) if prog.step_of[i] is not None]
    assert prog.fwd.num_steps == len(stepped)
    assert len(stepped) + len(prog.fused_heads) == n_dense
    assert prog.num_grad_floats == sum(p.numel() for p in model._dense_params())
    # forward steps: K groups multiples of 4, every hidden output has exactly one slab
    for k in range(prog.fwd.num_steps):
        st = prog.fwd.step[k]
        assert st.act_groups % 4 == 0 and st.aux_groups % 4 == 0
        assert st.out_tiles in ((2, 4, 8, 16) if prog.wide else (1, 2, 4, 8))
    hidden = [i for i, sp in enumerate(prog.layers) if sp.to_logits is None]
    assert prog.fwd.num_slots == len(hidden)
    # a logits head that reads a hidden layer rides in that layer's epilogue
    for (i, off, channels) in prog.fused_heads:
        producer = prog.fwd.step[prog.step_of[prog.producer_of[i]]]
        assert producer.head_off == off and channels == prog.layers[i].act_in
    saved_by = [prog.fwd.step[k].save_in_slot for k in range(prog.fwd.num_steps)]
    saved_by += [prog.fwd.step[k].save_out_slot for k in range(prog.fwd.num_steps)]
    used = [s for s in saved_by if s >= 0]
    assert len(used) == len(set(used))           # each activation is saved exactly once
    consumed = {prog.slot_of[p] for p in prog.producer_of if p >= 0}
    assert set(used) == consumed
    # backward chain ends by storing dZ of the first layer
    last = prog.bwd.step[prog.bwd.num_steps - 1]
    assert last.save_out_slot == 0
    # every (row, col) of every weight gradient is produced by exactly one reduce job
    for blocks, precision in ((1, "f32"), (7, "f32"), (4096, "f32"), (5, "bf16x3"), (4096, "bf16x3"),
                              (131072, "bf16x3"), (4099, "bf16x3"), (3, "bf16x6"), (4099, "bf16x6"),
                              (131072, "bf16x6")):
        # (the split-bf16 plan has its own unit costs; the bf16x6 plan is TWO launches: units with
        # four quadrants and the heads on the three-part kernel, narrower ones on the exact-f32 one)
        plan = prog._plan_wgrad(blocks, precision)
        segments = [seg for _, segs, _ in plan["launches"] for seg in segs]
        if precision != "bf16x6":
            assert len(plan["launches"]) == 1 and plan["launches"][0][0] == precision
            assert segments == plan["unit_segments"]
        kernel_of = {}
        for kind, segs, starts in plan["launches"]:
            assert len(starts) == 257 and starts[-1] == len(segs) and starts[0] == 0
            for seg in segs:
                assert kernel_of.setdefault(seg.job, kind) == kind
        cover = {}
        for seg in segments:
            cover.setdefault(("u", seg.job), []).append((seg.blk_begin, seg.blk_end))
        assert len(cover) == len(prog.wgrad_units)
        for spans in cover.values():
            spans.sort()
            assert spans[0][0] == 0 and spans[-1][1] == blocks
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        for li, spec in enumerate(prog.layers):
            hits = np.zeros((spec.out, spec.ld), np.int32)
            cmap = prog.col_maps[li].numpy()
            for rj in plan["reduce_jobs"]:
                if rj.w_grad_off != prog.grad_w_off[li]:
                    continue
                for jj in range(rj.n_quads):
                    for q in range(4):
                        col = cmap[rj.k_base + 4 * (rj.n_quad0 + jj) + q]
                        if col < 0:
                            continue
                        if rj.kind == 0:
                            rows = range(rj.m_ch0, min(rj.m_ch0 + 128, spec.out))
                        else:
                            rows = range(rj.lg_n)
                        for row in rows:
                            hits[row, col] += 1
            assert hits.min() == 1 and hits.max() == 1, (li, hits.min(), hits.max())
        # the reducer is told the fold the exact-f32 unit kernel applies to a narrow input window
        # (csrc/wgrad_common.h ffn_wgrad_fold, decided on the UNIT's window); the split-bf16
        # kernel never folds
        expected = []
        for meta in prog.unit_meta:
            if meta.get("head"):
                expected += [1] * sum(1 for wave in range(4) if meta["n_quads"] - 16 * wave > 0)
            else:
                mh, nh = prog._quadrants(meta["m_quads"], meta["n_quads"])
                rule = 4 if meta["n_quads"] <= 8 else (2 if meta["n_quads"] <= 16 else 1)
                folds = precision == "f32" or (precision == "bf16x6" and mh * nh < 4)
                expected += [rule if folds else 1] * (mh * nh)
        assert [rj.n_fold for rj in plan["reduce_jobs"]] == expected
        if precision == "bf16x6":
            for u, meta in enumerate(prog.unit_meta):
                full = not meta.get("head") and prog._quadrants(meta["m_quads"], meta["n_quads"]) == (2, 2)
                assert kernel_of[u] == ("bf16x6" if (full or meta.get("head")) else "f32"), u
        slots = set()
        for rj in plan["reduce_jobs"]:
            mine = set(range(rj.slot_begin, rj.slot_end, rj.slot_stride))
            assert not (mine & slots)
            slots |= mine
        assert max(slots) < plan["slots"]
        # every partial slot a unit's segments write is read by exactly that unit's reduce jobs
        written = {}
        for seg in segments:
            for wave in range(4):
                assert seg.slot + wave not in written
                written[seg.slot + wave] = seg.job
        assert slots <= set(written)       # (a narrow head leaves the partials of its idle waves unread)
        assert len(plan["unit_starts"]) == 257 and plan["unit_starts"][-1] == len(plan["unit_segments"])


def test_tail_plan_of_a_training_launch():
    """Which kernels run a launch's short last round (mlp_engine._tail_plan): four waves per block
    while the remainder is at most one block per CU, wave pairs up to half a round, nothing for
    exact rounds / long remainders / tiny or huge launches; chains with a 64-wide step have no
    quad kernels, 512-wide chains no team tail at all."""
    prog = _plan(ffn.PositionalFourierMLP(3, 4, 5.5))
    prog._waves = 1024                                  # 256 CUs x 4 resident wavefronts
    assert prog.pair_chain_ok and prog.quad_chain_ok
    blocks = lambda k: 32 * k                            # noqa: E731
    assert prog._tail_plan(blocks(3 * 1024 + 98)) == (3072, 4)      # the reference's default batch
    assert prog._tail_plan(blocks(1024 + 256)) == (1024, 4)
    assert prog._tail_plan(blocks(1024 + 257)) == (1024, 2)
    assert prog._tail_plan(blocks(1024 + 512)) == (1024, 2)
    assert prog._tail_plan(blocks(1024 + 513)) is None
    assert prog._tail_plan(blocks(2048)) is None and prog._tail_plan(blocks(700)) is None
    assert prog._tail_plan(blocks(17 * 1024 + 5)) is None
    assert prog._tail_split(blocks(3 * 1024 + 98) - 7) == 3072     # (ragged last block)
    narrow = _plan(ffn.NeRF(4, 64, 5, 6, 2, 3, [2], False))       # hidden_view: 32 channels = 1 tile
    narrow._waves = 1024
    assert not narrow.pair_chain_ok and narrow._tail_plan(blocks(1024 + 98)) is None
    mid = _plan(ffn.MLP(3, 4, num_channels=64))
    mid._waves = 1024
    assert mid.pair_chain_ok and not mid.quad_chain_ok and mid._tail_plan(blocks(1024 + 98)) == (1024, 2)
    wide = _plan(ffn.GaussianFourierMLP(3, 4, 10.0, num_channels=512))
    wide._waves = 1024
    assert wide._tail_plan(blocks(1024 + 98)) is None


def test_unsupported_shapes_raise_not_fall_back():
    # round 5: layers of up to 1024 channels plan (a team of four waves per block, wide == 3):
    # 513..1024 pad to the next of 128 / 256 / 512 / 1024; beyond 1024 raises
    with pytest.raises(NotImplementedError):
        _plan(ffn.GaussianFourierMLP(3, 4, 10.0, num_channels=1025))
    with pytest.raises(NotImplementedError):
        _plan(ffn.NeRF(8, 2048, 9, 10, 3, 4, [4], True))
    big = _plan(ffn.NeRF(8, 1024, 9, 10, 3, 4, [4], True))
    assert big.big and big.fwd.wide == 3 and big.bwd.wide == 3 and big.mask_words == 1024
    assert max(big.fwd.step[k].out_tiles for k in range(big.fwd.num_steps)) == 32
    assert max(big.fwd.step[k].act_groups for k in range(big.fwd.num_steps)) == 128
    assert big.fwd16 is None and big.fwd_x6 is None and not big.pair_chain_ok
    # (its two head blocks -- 4100 + 2052 floats -- lie beyond the kernels' LDS copy: read from L2)
    assert sum(4 + 4 * ch for _, _, ch in big.fused_heads) > 4096
    odd = _plan(ffn.MLP(3, 4, num_channels=768))
    assert odd.big and [sp.out_p for sp in odd.layers] == [1024, 1024, 1024, 4]
    mid = _plan(ffn.NeRF(8, 513, 9, 10, 3, 4, [4], True))
    assert mid.big and mid.layers[0].out_p == 1024
    # everything up to 512 channels plans: padded widths, biases beyond the kernels' LDS copy
    wide = _plan(ffn.NeRF(8, 512, 9, 10, 3, 4, [4], True))
    assert wide.fwd.bias_floats > 4096 and all(off + 4 + 4 * ch <= 4096 for _, off, ch in wide.fused_heads)
    padded = _plan(ffn.MLP(3, 4, num_channels=96))
    assert [sp.out_p for sp in padded.layers] == [128, 128, 128, 4]
    assert [sp.act_in_p for sp in padded.layers] == [0, 128, 128, 128]
    model = ffn.MLP(3, 4, num_channels=32)
    with pytest.raises(RuntimeError, match="GPU"):
        model(torch.zeros(2, 3))
    with pytest.raises(RuntimeError, match="GPU"):
        ffn.RaySampler(np.eye(4, dtype=np.float32) * 2, [], 8, device="cpu")


def test_orbit_cameras_look_at_the_origin():
    cams = ffn.orbit(np.array([0, 1, 0.]), np.array([0, 0, 1.]), 12, 40, ffn.Resolution(64, 48), 4)
    assert len(cams) == 12
    for cam in cams:
        pose = cam.extrinsics
        eye = pose[:3, 3]
        assert abs(np.linalg.norm(eye) - 4) < 1e-5
        np.testing.assert_allclose(pose[:3, 2], -eye / np.linalg.norm(eye), atol=1e-6)
        np.testing.assert_allclose(pose[:3, :3] @ pose[:3, :3].T, np.eye(3), atol=1e-6)
        assert cam.intrinsics[0, 2] == 32 and cam.intrinsics[1, 2] == 24
    # altitude ramps up then down; azimuth makes two turns
    heights = [c.extrinsics[1, 3] for c in cams]
    assert heights[5] > heights[0] and heights[5] > heights[11]
    assert abs(cams[0].fov_y_degrees - 2 * np.degrees(np.arctan(32 / cams[0].intrinsics[1, 1]))) < 1e-4


def test_lr_schedule_and_result_tuples():
    from fourier_feature_nets_amd.utils import learning_rate_at

    class Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0}]

    ffn.exponential_lr_decay(Opt, 5e-4, 25000, 0.1, 25000)
    assert Opt.param_groups[0]["lr"] == Opt.param_groups[1]["lr"] == orc.lr_decay(5e-4, 25000, 0.1, 25000)
    assert learning_rate_at(5e-4, 0, 0.1, 25000) == 5e-4
    res = ffn.RenderResult(torch.zeros(2, 3), torch.zeros(2), None)
    assert res.numpy().depth is None and res.to(torch.float64).color.dtype == torch.float64
    samples = ffn.RaySamples(torch.zeros(4, 2, 3), None, torch.zeros(4, 2), torch.arange(4))
    assert samples.subset([1, 2]).rays.tolist() == [1, 2]
    assert samples.subset(slice(0, 3)).positions.shape == (3, 2, 3)
    assert samples.numpy().view_directions is None


# --------------------------------------------------------------------------------- data parallel
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, out):
    """Each rank: its contiguous shard of the valid-filtered global batch, gradients scaled
    by the GLOBAL ray count, one all-reduce(sum) of the flat gradient buffer, clip AFTER the
    reduction, identical Adam step everywhere -- exactly TrainEngine's recipe."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    b = orc.positional_b_values(3.0, 12, 3)
    dims = [(32, 2 * b.shape[1]), (32, 32), (4, 32)]
    ws = [torch.randn(o, k) / math.sqrt(k) for o, k in dims]
    bs = [torch.randn(o) * 0.1 for o, _ in dims]
    R, S = 37, 8                                       # ragged on purpose: 19 + 18 rays
    pos = torch.rand(R, S, 3) * 2 - 1
    t = torch.sort(torch.rand(R, S) * 3 + 1, -1)[0]
    gt_c, gt_a = torch.rand(R, 3), (torch.rand(R) > 0.5).float()

    class Group:                                       # TrainEngine.shard only needs these
        pass

    engine = ffn.TrainEngine.__new__(ffn.TrainEngine)
    engine.group = dist.group.WORLD
    mine = engine.shard(torch.arange(R))
    model = orc.OracleFourierMLP(torch.ones(b.shape[1]), b, ws, bs)
    logits = model(pos[mine].reshape(-1, 3)).reshape(len(mine), S, 4)
    color, alpha, _ = orc.render(logits, t[mine], False)
    # sums scaled by the global counts, as ffn_mse_loss is called under DP
    loss = ((gt_c[mine] - color).square().sum() / (3 * R)
            + 0.1 * (gt_a[mine] - alpha).square().sum() / R)
    loss.backward()
    # the product's collective: ONE buffer [flat gradients | 2 loss sums] through
    # TrainEngine._all_reduce (gloo groups are reduced through the host)
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    err_c = float((gt_c[mine] - color).square().sum())
    err_a = float((gt_a[mine] - alpha).square().sum())
    engine.reduce_buf = torch.cat([flat, torch.tensor([err_c, err_a])])
    engine._host_staged = True
    engine.collective_events = None
    engine._all_reduce()
    flat = engine.reduce_buf[:-2]
    sums = engine.reduce_buf[-2:]
    total = sums[0] / (3 * R) + 0.1 * sums[1] / R
    grads = []
    offset = 0
    for p in model.parameters():
        grads.append(flat[offset:offset + p.numel()].view_as(p).clone())
        offset += p.numel()
    orc.clip_gradients(grads)
    with torch.no_grad():
        for p, g in zip(model.parameters(), grads):
            orc.adam_update(p, g, torch.zeros_like(p), torch.zeros_like(p), 1, 5e-4)
    if rank == 0:
        torch.save({"loss": float(total), "count": len(mine),
                    "params": [p.detach().clone() for p in model.parameters()],
                    "inputs": (ws, bs, b, pos, t, gt_c, gt_a)}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_step_equals_single_process_step(tmp_path, world):
    """ONE global batch of 37 rays sharded over 2 / 4 gloo ranks (ragged: 19 + 18, 10 + 10 + 10 + 7)
    == the single-process step on the same batch: what `bench.py --scaling strong` times, and --
    with a global batch that grows with the ranks -- what `--scaling weak` times."""
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    blob = torch.load(out)
    ws, bs, b, pos, t, gt_c, gt_a = blob["inputs"]
    assert blob["count"] == -(-37 // world)
    model = orc.OracleFourierMLP(torch.ones(b.shape[1]), b, ws, bs)
    trainer = orc.OracleTrainer(model, 5e-4)
    loss = trainer.step(pos, None, t, gt_c, gt_a, 5e-4)
    assert abs(loss - blob["loss"]) < 1e-6
    for mine, theirs in zip(model.parameters(), blob["params"]):
        np.testing.assert_allclose(mine.detach().numpy(), theirs.numpy(), rtol=1e-5, atol=1e-7)


def test_driver_scripts_keep_the_reference_flags_and_defaults():
    """argparse surface of scripts/{train_nerf,train_tiny_nerf,orbit_video}.py == the
    reference scripts' (captured into tests/golden/cli_defaults.json by make_goldens.py), plus
    the documented extensions."""
    import json
    from scripts import _cli
    with open(os.path.join(os.path.dirname(__file__), "golden", "cli_defaults.json")) as f:
        ref = json.load(f)
    kinds = ("nerf_model", dict(choices=["mlp", "basic", "positional", "gaussian"]))
    mine = {
        "train_nerf": vars(_cli.build_parser("t", _cli.TRAIN_COMMON, _cli.NERF_ONLY)
                           .parse_args(["d.npz", "out"])),
        "train_tiny_nerf": vars(_cli.build_parser("t", _cli.TRAIN_COMMON, _cli.TINY_ONLY,
                                                  positional_extra=[kinds])
                                .parse_args(["d.npz", "positional", "out"])),
        "orbit_video": vars(_cli.build_parser("t", _cli.ORBIT).parse_args(["m.pt", "400", "out"])),
    }
    for name in ref:
        # the extensions: --precision (opt-in split-bf16 kernels) and the opt-in empty-space
        # skipping schedule; the defaults are the exact mode
        assert mine[name].pop("precision") == "f32"
        assert mine[name].pop("focus_mode") == "auto"        # extension: table / live CDFs
        if name != "orbit_video":
            assert mine[name].pop("skip_empty_space") is False
            assert (mine[name].pop("skip_warmup"), mine[name].pop("skip_refresh")) == (1000, 500)
        assert mine[name] == ref[name], name


def test_ellipse_element_matches_opencv_known_answers():
    """The Dilate-mode structuring element (image_dataset.py:92-94 calls
    cv2.getStructuringElement(MORPH_ELLIPSE, ...); OpenCV is not in this image).  Known answers:
    the 3x3, 5x5 and 7x7 ellipses printed in OpenCV's morphology documentation."""
    from fourier_feature_nets_amd.dataset import _ellipse
    assert _ellipse(1).tolist() == [[1]]
    assert _ellipse(3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert _ellipse(5).tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1],
                                    [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    assert _ellipse(7).tolist() == [[0, 0, 0, 1, 0, 0, 0], [0, 1, 1, 1, 1, 1, 0],
                                    [1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1],
                                    [1, 1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1, 0],
                                    [0, 0, 0, 1, 0, 0, 0]]
    for size in (9, 17, 65):         # mirror-symmetric (not transpose-symmetric, like OpenCV's)
        e = _ellipse(size)
        assert (e == e[::-1]).all() and (e == e[:, ::-1]).all()
        assert e[size // 2].all() and e[:, size // 2].all()


def test_wgrad_split_is_balanced_and_plan_buckets_are_tight():
    """Segments of the weight-gradient plan: every worker's cost stays within one block of the
    mean (the first version let rounding leftovers pile up on the last worker), and the block
    count a plan is made for exceeds the real one by at most 1/32."""
    from fourier_feature_nets_amd.mlp_engine import MlpProgram
    costs = [24, 24, 13, 8, 6]
    for blocks in (1, 5, 97, 2813, 131072):
        for workers in (4, 256):
            segs, starts = MlpProgram._split(costs, blocks, workers)
            assert len(starts) == workers + 1 and starts[-1] == len(segs)
            loads = [sum((b1 - b0) * costs[j] for j, b0, b1 in segs[starts[w]:starts[w + 1]])
                     for w in range(workers)]
            total = sum(costs) * blocks
            assert sum(loads) == total
            assert max(loads) <= total / workers + max(costs)
            # contiguous, complete coverage of every job
            cover = {}
            for j, b0, b1 in segs:
                assert b0 < b1
                assert cover.get(j, 0) == b0
                cover[j] = b1
            assert cover == {j: blocks for j in range(len(costs))}
    for n in (1, 31, 32, 33, 2048, 2049, 90001, 4194304, 8388608 - 17):
        blocks = (n + 31) // 32
        planned = MlpProgram.plan_blocks(n)
        assert blocks <= planned <= blocks + max(0, blocks // 32)
        assert MlpProgram.plan_blocks(planned * 32) == planned          # idempotent


def test_host_ycrcb_conversion_equals_the_oracle():
    """The vectorised host conversion used at dataset construction (utils.rgb_to_ycrcb_u8)
    against the oracle's per-pixel restatement, incl. every corner of the colour cube."""
    from fourier_feature_nets_amd.utils import check_color_space, rgb_to_ycrcb_u8
    rng = np.random.default_rng(11)
    corners = np.array([[r, g, b] for r in (0, 255) for g in (0, 255) for b in (0, 255)], np.uint8)
    rgb = np.concatenate([corners, rng.integers(0, 256, (3000, 3), dtype=np.uint8)])
    assert np.array_equal(rgb_to_ycrcb_u8(rgb), orc.rgb_to_ycrcb_u8(rgb))
    image = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)          # keeps the array shape
    assert rgb_to_ycrcb_u8(image).shape == image.shape
    assert check_color_space("YCrCb") == "YCrCb"
    with pytest.raises(NotImplementedError):
        check_color_space("HSV")


def test_sample_cameras_keeps_the_reference_order():
    """`RayDataset.sample_cameras` against the subsets the REFERENCE's own method picked
    (tests/golden/camera_subsets.json, written by make_camera_subsets.py from
    ray_dataset.py:185-216): the reference iterates a Python set, so the ORDER of the subset --
    which decides the pixels `_validate` draws for the psnr_train column -- is not sorted
    (0, 42, 12, 77, 55, 91, 28 on the 100-camera PSNR rig); rings hold many equidistant ties."""
    import json
    from types import SimpleNamespace
    from tests.golden.make_camera_subsets import rig_positions
    with open(os.path.join(os.path.dirname(__file__), "golden", "camera_subsets.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 8
    unsorted = 0
    for case in cases:
        pos = rig_positions(case["rig"], case["cameras"], case["seed"])
        cams = [SimpleNamespace(position=p[None, :]) for p in pos]
        stand_in = SimpleNamespace(num_cameras=case["cameras"], sampler=SimpleNamespace(cameras=cams),
                                   label="x", subset=lambda cameras, *_: [int(c) for c in cameras])
        chosen = ffn.RayDataset.sample_cameras(stand_in, case["pick"], 64, False)
        assert chosen == case["chosen"], (case, chosen)
        unsorted += chosen != sorted(chosen)
    assert unsorted >= 4


def test_bench_batch_plan_weak_and_strong():
    """`bench.py --scaling`: weak keeps the per-GPU batch and grows the global one, strong keeps
    the global batch and shards it; a strong batch that does not divide is refused."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.batch_plan(65536, 1, "weak") == (65536, 65536) == bench.batch_plan(65536, 1, "strong")
    assert bench.batch_plan(65536, 8, "weak") == (524288, 65536)
    assert bench.batch_plan(65536, 8, "strong") == (65536, 8192)
    with pytest.raises(SystemExit):
        bench.batch_plan(1000, 3, "strong")


def test_psnr_ensemble_verdicts_and_stale_scene_files(tmp_path):
    """tests/psnr_ensemble.py, the statistics the PSNR clause rests on: the three-valued verdict
    (fail / pass-resolved / pass-unresolved) with the minimum detectable difference, the
    seed-paired window, `protocol_matches` against a fixture that holds fewer runs than planned --
    and a cached scene file with another camera count is refused, not picked up."""
    import argparse
    import json
    from tests import psnr_ensemble as pe

    def doc(finals, seeds_planned, drift):
        runs = [{"seed": 100 + i, "reports": [{"step": s, "train_psnr": v, "val_psnr": v + drift * s * (1 if i % 2 else -1)}
                                              for s in (0, 10, 20)], "final_val_psnr": v}
                for i, v in enumerate(finals)]
        out = {"protocol": {"seeds": [100 + i for i in range(seeds_planned)], "rays": 1024}, "runs": runs}
        out.update(pe.stats_of(runs))
        return out

    ref_path = str(tmp_path / "ref.json")
    out_path = str(tmp_path / "out.json")

    def verdict(mine, theirs, drift=0.0, planned=None):
        with open(ref_path, "w") as f:
            json.dump(doc(theirs, planned or len(theirs), 0.0), f)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            got = pe.compare(doc(mine, len(mine), drift), ref_path, out_path)["against_reference"]
        return got

    tight = [30.00, 30.01, 29.99, 30.00, 30.01, 29.99]
    a = verdict(tight, tight)
    assert a["resolution"]["verdict"] == "pass-resolved" and a["protocol_matches"]
    assert a["resolution"]["first_report_step_with_a_seed_outside_0p05_db"] is None
    wide = [30.0, 30.6, 29.4, 30.3, 29.7, 30.0]
    b = verdict(wide, wide, drift=0.004)
    assert b["resolution"]["verdict"] == "pass-unresolved"          # 2 s.e. of the means is far above 0.05 dB
    assert b["resolution"]["first_report_step_with_a_seed_outside_0p05_db"] == 20   # 0.004 dB/step leaves at 0.08
    assert b["resolution"]["paired_reports_all_seeds_within_0p05_db_at_steps"] == [0, 10]
    c = verdict([v + 0.2 for v in tight], tight)
    assert c["resolution"]["verdict"] == "fail" and c["verdict"] == "fail"
    # a fixture with fewer runs than its protocol planned still matches on the seeds it carries
    d = verdict(tight, tight[:3], planned=6)
    assert d["protocol_matches"] and len(d["per_seed_delta_db"]) == 3
    # TWO seeds on a side resolve next to nothing: two values that happen to agree estimate a spread
    # of nothing -- the test pools the variance of the two halves; a 0.7 dB gap must not read `fail`
    # (the round-6 config-3 slow protocol: 2 reference seeds against 24 HIP seeds of other difficulty)
    many = [17.0 + 0.5 * ((i * 7) % 5 - 2) for i in range(24)]
    e = verdict(many, [17.7, 17.8], planned=24)
    assert e["resolution"]["verdict"] == "pass-unresolved" and e["verdict"] == "pass"
    assert e["resolution"]["critical_value (Student's t, 97.5 %, n_a + n_b - 2 degrees of freedom; never below 2)"] > 2.0
    # (the means are taken over the two seeds both halves hold)
    assert e["hip_final"]["n"] == 2 and e["hip_final_all_seeds"]["n"] == 24 and e["stderr_of_delta_db"] > 0.3

    args = argparse.Namespace(workdir=str(tmp_path), size=8, cameras=3, val_cameras=2)
    first = pe.scene_path(args)
    assert pe.scene_path(args) == first
    os.replace(first, first.replace("_3_2.npz", "_3_1.npz"))         # a file written for 3 + 2 cameras under the 3 + 1 name
    with pytest.raises(RuntimeError, match="cameras"):
        pe.scene_path(argparse.Namespace(workdir=str(tmp_path), size=8, cameras=3, val_cameras=1))


def test_bench_finds_its_kernels_in_the_committed_traffic_profile():
    """bench.py's `roofline.traffic` comes from the committed PMC passes, looked up by the kernel
    symbol rocprofv3 printed: a template argument added to a kernel must not turn it into null."""
    import argparse
    import bench
    args = argparse.Namespace(rays=65536, samples=64, model="tiny")
    for name in bench.SYMBOL_OF:
        value, source = bench.traffic_of(name, args)
        assert value is not None and value > 1e9, (name, source)
    # another shape than the profiled one has no traffic figure (null, never a wrong one)
    assert bench.traffic_of("wgrad_unit_kernel", argparse.Namespace(rays=1024, samples=64, model="tiny"))[0] is None


def test_psnr_ensemble_single_seed_resolves_nothing(tmp_path):
    """One seed a side has no standard error: the means resolve nothing (never a `fail`)."""
    import contextlib, io, json
    from tests import psnr_ensemble as pe

    def doc(v):
        runs = [{"seed": 1, "reports": [{"step": 0, "train_psnr": v, "val_psnr": v}], "final_val_psnr": v}]
        out = {"protocol": {"seeds": [1]}, "runs": runs}
        out.update(pe.stats_of(runs))
        return out
    ref = str(tmp_path / "ref.json")
    with open(ref, "w") as f:
        json.dump(doc(20.0), f)
    with contextlib.redirect_stdout(io.StringIO()):
        got = pe.compare(doc(20.4), ref, str(tmp_path / "out.json"))["against_reference"]
    assert got["resolution"]["verdict"] == "pass-unresolved" and got["verdict"] == "pass"
    json.dumps(got)      # (no infinity in the document)
