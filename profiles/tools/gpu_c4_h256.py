"""config 4's step (21 maps, 128 x 256, Mask-3, RENITestLoss, latent Adam) at the reference's shipped width H = 256 and in fp32 at H = 128,
dense / RENI_WEIGHT_SPARSE / RENI_WEIGHT_COMPACT -- the generic kernels (k_reni_main) walk the same device-built lists"""
import os, sys, time
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py"]
import bench, torch
from reni_amd.engine import TrainEngine
from reni_amd.models import RENIAutoDecoder
from reni_amd.utils import get_directions, get_sineweight
dev = torch.device("cuda:0")
D = get_directions(256).to(dev); S = get_sineweight(256).to(dev); P = D.shape[1]
W = S * bench.mask3(256).to(dev)
imgs = torch.rand(21, P, 3, device=dev) * 2 - 1
for dtype, H in (("bf16", 256), ("f32", 128), ("bf16", 128)):
    for mode in (False, True, "pixels"):
        torch.manual_seed(0)
        m = RENIAutoDecoder(21, 36, "SO2", H, 5, 3, True, "tanh", 30.0, 30.0, True)
        with torch.no_grad():
            m.Z.normal_()
        m.set_compute_dtype(dtype).to(dev)
        eng = TrainEngine(m, lr=1e-1, loss_kind="test", alpha=1e-7, beta=1e-4, sparse_weight=mode)
        idx = torch.arange(21, device=dev)
        for _ in range(5):
            t = eng.step(idx, imgs, W, D)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            t = eng.step(idx, imgs, W, D)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{dtype} H={H} sparse_weight={mode!s:7} {dt * 1e3:8.3f} ms/step  loss {float(t[0]):.6f}", flush=True)
