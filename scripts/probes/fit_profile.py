"""Where does Raycaster.fit's wall-clock go at the reference's default batch?  (cProfile, host side)"""
import cProfile, contextlib, io, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import fourier_feature_nets_amd as ffn
from tests.psnr_ensemble import write_npz
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 64
npz = "/tmp/fit_profile_scene.npz"
if not os.path.exists(npz):
    write_npz(npz, 100, 7, 400)
torch.manual_seed(1); np.random.seed(1)
model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
with contextlib.redirect_stdout(io.StringIO()):
    train = ffn.ImageDataset.load(npz, "train", samples, True, True, None, 4096, "RGB", anneal_start=0.2, num_anneal_steps=500, device=dev)
    val = ffn.ImageDataset.load(npz, "val", samples, True, False, None, 4096, "RGB", device=dev)
caster = ffn.Raycaster(model)
pr = cProfile.Profile()
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    caster.fit(train, val, 1024, 5e-4, steps, 250, 250, 0.1, 25000, 0.0, [], True)
    torch.cuda.synchronize()
    pr.disable()
print("fit: %.2f s for %d steps = %.3f ms/step" % (time.time() - t0, steps, 1e3 * (time.time() - t0) / steps))
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
