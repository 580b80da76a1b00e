"""Worker bodies of the multi-process GPU tests (spawned with torch.multiprocessing, so they
must live in an importable module).  Every rank drives the PRODUCT's TrainEngine.train_step on
cuda:0 (ranks share the one GPU of the test box) over a gloo group, which the engine reduces
through the host."""

import contextlib
import io
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "golden", "scene16.npz")


def small_model(device):
    import fourier_feature_nets_amd as ffn
    g = np.load(os.path.join(HERE, "golden", "training.npz"))
    torch.manual_seed(0)
    model = ffn.PositionalFourierMLP(3, 4, 5.5, num_channels=64, embedding_size=48)
    sd = {k[len("fit_init/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fit_init/")}
    model.load_state_dict(sd)
    return model.to(device)


def run_steps(group, steps, max_samples=None):
    """`steps` optimisation steps on ragged, non-stratified batches (so that the sharded and the
    unsharded run see the same samples) with annealing active.  Returns (losses, flat weights)."""
    import fourier_feature_nets_amd as ffn
    device = torch.device("cuda:0")
    model = small_model(device)
    with contextlib.redirect_stdout(io.StringIO()):
        train = ffn.ImageDataset.load(SCENE, "train", 16, True, False, anneal_start=0.2,
                                      num_anneal_steps=8, device=device)
    engine = ffn.TrainEngine(model, 0.0, group)
    if max_samples is not None:
        engine.max_samples = max_samples
    losses = []
    for step in range(steps):
        batch = torch.arange(step, len(train), 3, device=device)
        rays = train.ray_ids(batch)
        if rays.numel() % 2 == 0:          # odd on purpose: the two shards differ in size
            batch = batch[:-1]
        losses.append(float(engine.train_step(train, batch, step, 5e-4)))
    engine.check_finite()
    return losses, engine.flat.detach().cpu().clone()


def dp_train_worker(rank, world, port, out_path, steps, max_samples):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        losses, flat = run_steps(dist.group.WORLD, steps, max_samples)
        if rank == 0:
            torch.save({"losses": losses, "flat": flat}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run_fit(group, steps=12, report=4):
    """`Raycaster.fit` for ``steps`` optimisation steps with a report (two validations, a log
    entry) every ``report`` steps, crop curriculum included, on a non-stratified training set so
    that sharded and unsharded runs see the same samples.  Epoch permutations come from ONE seed
    per epoch (np.random on rank 0).  Returns (printed lines, log psnrs, flat weights)."""
    import fourier_feature_nets_amd as ffn
    device = torch.device("cuda:0")
    np.random.seed(5)
    torch.manual_seed(5)
    model = small_model(device)
    with contextlib.redirect_stdout(io.StringIO()):
        train = ffn.ImageDataset.load(SCENE, "train", 16, True, False, anneal_start=0.2,
                                      num_anneal_steps=6, device=device)
        val = ffn.ImageDataset.load(SCENE, "val", 16, True, False, device=device)
    caster = ffn.Raycaster(model)
    caster.process_group = group
    caster.shuffle_source = "seeded"
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        log = caster.fit(train, val, 96, 5e-4, steps, 4, report, 0.1, 25000, 0.0, [])
    psnr = [(e.step, e.train_psnr, e.val_psnr) for e in log]
    return out.getvalue().splitlines(), psnr, caster.engine.flat.detach().cpu().clone()


def dp_fit_worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lines, psnr, flat = run_fit(dist.group.WORLD)
        torch.save({"lines": lines, "psnr": psnr, "flat": flat}, out_path + ".%d" % rank)
        dist.barrier()
    finally:
        dist.destroy_process_group()
