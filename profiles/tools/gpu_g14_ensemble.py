"""G14 (config 4 in the small) on the HIP kernels over the perturbation ensemble of tests/golden/make_g14_ensemble.py: the same 1e-6 target
noise per seed, final-image PSNR (masked-out / kept pixels) and latent cosine against the reference's UNPERTURBED fp32 run, beside the
reference's own fp32 and autocast-bf16 runs on the same perturbed targets (tests/golden/g14_ensemble.npz).
usage: python profiles/tools/gpu_g14_ensemble.py [bf16|f32] [n_seeds]      (env RENI_* build / debug switches apply as usual)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_trajectory import _cos, _decoder_sd, _psnr  # noqa: E402
from tests.util import load_golden  # noqa: E402


def run(dtype, imgs_cpu, g, dev):
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    from reni_amd.utils import get_directions, get_sineweight
    N, W = imgs_cpu.shape[0], int(g["W"])
    m = RENIAutoDecoder(N, 36, "SO2", 128, 5, 3, True, "tanh", 30.0, 30.0, True)
    m.load_state_dict({"model." + k: v for k, v in _decoder_sd().items()})
    m.set_compute_dtype(dtype).to(dev)
    D = get_directions(W).to(dev)
    S = (get_sineweight(W) * torch.from_numpy(g["mask"])).to(dev)
    imgs = imgs_cpu.to(dev)
    P = D.shape[1]
    eng = TrainEngine(m, lr=float(g["lr"]), loss_kind="test", alpha=float(g["alpha"]), beta=float(g["beta"]))
    idx = torch.arange(N, device=dev)
    tgt = imgs.permute(0, 2, 3, 1).view(N, P, 3)
    for _ in range(int(g["steps"])):
        eng.step(idx, tgt, S, D)
    with torch.no_grad():
        img = m(m.Z.data, D).detach().float().cpu().numpy()
    return img, m.Z.detach().cpu().numpy()


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda:0")
    g = load_golden("g14_c4_trajectory.npz")
    ens = load_golden("g14_ensemble.npz")
    masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
    ref_img, ref_Z = g["img_after_200"], g["Z_after_200"]
    imgs0 = torch.from_numpy(g["imgs"])
    rows = []
    for seed in range(0, n + 1):
        imgs = imgs0 if seed == 0 else imgs0 + float(ens["noise"]) * torch.randn(imgs0.shape, generator=torch.Generator().manual_seed(int(ens["seed_base"]) + seed))
        img, Z = run(dtype, imgs, g, dev)
        r = (seed, _psnr(img, ref_img, masked_out), _psnr(img, ref_img, ~masked_out), _cos(Z, ref_Z))
        rows.append(r)
        e = ens["rows"][seed - 1] if 1 <= seed <= len(ens["rows"]) else None
        print("seed %d: HIP %s %.2f / %.2f dB cos %.4f" % ((r[0], dtype) + r[1:]) +
              ("" if e is None else " | reference fp32 %.2f / %.2f  autocast-bf16 %.2f / %.2f dB cos %.4f" % (e[1], e[2], e[4], e[5], e[6])), flush=True)
    a = np.array(rows)[:, 1:]
    print("HIP %s over %d runs: masked-out mean %.2f min %.2f max %.2f | kept mean %.2f min %.2f max %.2f | cos mean %.3f" %
          (dtype, len(rows), a[:, 0].mean(), a[:, 0].min(), a[:, 0].max(), a[:, 1].mean(), a[:, 1].min(), a[:, 1].max(), a[:, 2].mean()))
    e = ens["rows"]
    print("reference under autocast-bf16 over %d seeds: masked-out mean %.2f min %.2f max %.2f | kept mean %.2f min %.2f max %.2f | cos mean %.3f" %
          (len(e), e[:, 4].mean(), e[:, 4].min(), e[:, 4].max(), e[:, 5].mean(), e[:, 5].min(), e[:, 5].max(), e[:, 6].mean()))


if __name__ == "__main__":
    main()
