"""G16 (G14's latent-optimisation loop with FiLM conditioning, tests/golden/make_g16_film_trajectory.py) on the HIP kernels: final-image
PSNR (masked-out / kept) and latent cosine against the reference's fp32 run, beside the reference's autocast-bf16 run.
usage: python profiles/tools/gpu_g16_film.py [bf16|f32]     (RENI_NO_PERSIST=1: the generic kernels)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_trajectory import _cos, _psnr  # noqa: E402
from tests.util import load_golden  # noqa: E402
from reni_amd.engine import TrainEngine  # noqa: E402
from reni_amd.film import RENIAutoDecoderFiLM  # noqa: E402
from reni_amd.utils import get_directions, get_sineweight  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
g, f = load_golden("g14_c4_trajectory.npz"), load_golden("g16_film_c4_trajectory.npz")
N, W = g["imgs"].shape[0], int(g["W"])
torch.manual_seed(int(f["seed"]))
m = RENIAutoDecoderFiLM(N, 36, "SO2", 128, 5, 128, 3, 3, "tanh", True)
m.set_compute_dtype(dtype).to(dev)
D = get_directions(W).to(dev)
S = (get_sineweight(W) * torch.from_numpy(g["mask"])).to(dev)
imgs = torch.from_numpy(g["imgs"]).to(dev)
P = D.shape[1]
eng = TrainEngine(m, lr=float(f["lr"]), loss_kind="test", alpha=float(g["alpha"]), beta=float(g["beta"]))
idx = torch.arange(N, device=dev)
tgt = imgs.permute(0, 2, 3, 1).view(N, P, 3)
terms = []
for it in range(int(f["steps"])):
    t = eng.step(idx, tgt, S, D)
    if it in f["rec_at"]:
        terms.append(float(t[0]))
with torch.no_grad():
    img = m(m.Z.data, D).detach().float().cpu().numpy()
Z = m.Z.detach().cpu().numpy()
masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
ref_img, ac_img = f["img_after_200"], f["img_after_200_autocast_bf16"].astype(np.float32)
rel = np.abs(np.array(terms) - f["terms"][:, 0]) / f["terms"][:, 0]
rel_ac = np.abs(f["terms_autocast_bf16"][:, 0] - f["terms"][:, 0]) / f["terms"][:, 0]
print(f"G16 FiLM {dtype} persist={'RENI_NO_PERSIST' not in os.environ}: loss dev max {rel.max():.3e} (autocast {rel_ac.max():.3e}); final image PSNR masked-out / kept "
      f"{_psnr(img, ref_img, masked_out):.2f} / {_psnr(img, ref_img, ~masked_out):.2f} dB (autocast {_psnr(ac_img, ref_img, masked_out):.2f} / {_psnr(ac_img, ref_img, ~masked_out):.2f}); "
      f"latent cos {_cos(Z, f['Z_after_200']):.4f} (autocast {_cos(f['Z_after_200_autocast_bf16'], f['Z_after_200']):.4f})")
