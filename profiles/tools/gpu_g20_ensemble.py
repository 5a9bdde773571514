"""G20 (G14's loop on the concat decoder at 256 features) on the HIP kernels over a perturbation ensemble: 1e-6 x N(0, 1) target noise per
seed, final-image PSNR (masked-out / kept pixels) and latent cosine against the reference's UNPERTURBED fp32 run
(tests/golden/g20_concat256_c4_trajectory.npz), beside the reference's own unperturbed autocast-bf16 run.
usage: python profiles/tools/gpu_g20_ensemble.py [bf16|f32] [n_seeds]      (env RENI_NO_PERSIST etc. apply as usual)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_trajectory import _cos, _psnr  # noqa: E402
from tests.util import load_golden  # noqa: E402


def run(dtype, imgs_cpu, g, f, dev):
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    from reni_amd.utils import get_directions, get_sineweight
    N, W = imgs_cpu.shape[0], int(g["W"])
    torch.manual_seed(int(f["seed"]))
    m = RENIAutoDecoder(N, 36, "SO2", int(f["width"]), 5, 3, True, "tanh", 30.0, 30.0, True)
    m.set_compute_dtype(dtype).to(dev)
    D = get_directions(W).to(dev)
    S = (get_sineweight(W) * torch.from_numpy(g["mask"])).to(dev)
    imgs = imgs_cpu.to(dev)
    P = D.shape[1]
    eng = TrainEngine(m, lr=float(f["lr"]), loss_kind="test", alpha=float(g["alpha"]), beta=float(g["beta"]))
    idx = torch.arange(N, device=dev)
    tgt = imgs.permute(0, 2, 3, 1).view(N, P, 3)
    for _ in range(int(f["steps"])):
        eng.step(idx, tgt, S, D)
    with torch.no_grad():
        img = m(m.Z.data, D).detach().float().cpu().numpy()
    return img, m.Z.detach().cpu().numpy()


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda:0")
    g, f = load_golden("g14_c4_trajectory.npz"), load_golden("g20_concat256_c4_trajectory.npz")
    masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
    ref_img, ref_Z = f["img_after_200"], f["Z_after_200"]
    ac_img = f["img_after_200_autocast_bf16"].astype(np.float32)
    imgs0 = torch.from_numpy(g["imgs"])
    rows = []
    for seed in range(0, n + 1):
        imgs = imgs0 if seed == 0 else imgs0 + 1e-6 * torch.randn(imgs0.shape, generator=torch.Generator().manual_seed(1000 + seed))
        img, Z = run(dtype, imgs, g, f, dev)
        r = (seed, _psnr(img, ref_img, masked_out), _psnr(img, ref_img, ~masked_out), _cos(Z, ref_Z))
        rows.append(r)
        own = "generic" if os.environ.get("RENI_NO_PERSIST") else "persistent"
        extra = ""
        if dtype == "bf16" and f"img_after_200_emulated_{own}" in f:   # the reference's fp32 autograd on THIS kernel's network (make_g20_...py)
            ei = f[f"img_after_200_emulated_{own}"].astype(np.float32)
            extra = " | against the fp32-autograd run on the %s network: %.2f / %.2f dB cos %.4f" % (
                own, _psnr(img, ei, masked_out), _psnr(img, ei, ~masked_out), _cos(Z, f[f"Z_after_200_emulated_{own}"]))
            if seed > 0:
                extra += " | against seed 0 of this kernel: %.2f dB" % _psnr(img, img0, masked_out)
        if seed == 0:
            img0 = img
        print("seed %d: HIP %s %.2f / %.2f dB cos %.4f" % ((r[0], dtype) + r[1:]) + extra, flush=True)
    a = np.array(rows)[:, 1:]
    print("HIP %s over %d runs: masked-out mean %.2f min %.2f max %.2f | kept mean %.2f min %.2f max %.2f | cos mean %.3f" %
          (dtype, len(rows), a[:, 0].mean(), a[:, 0].min(), a[:, 0].max(), a[:, 1].mean(), a[:, 1].min(), a[:, 1].max(), a[:, 2].mean()))
    print("reference under autocast-bf16, unperturbed: %.2f / %.2f dB cos %.4f" %
          (_psnr(ac_img, ref_img, masked_out), _psnr(ac_img, ref_img, ~masked_out), _cos(f["Z_after_200_autocast_bf16"], ref_Z)))


if __name__ == "__main__":
    main()
