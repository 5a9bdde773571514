"""G14 perturbation ensemble (round 6; VERDICT r05 item 5: is the bf16 image gap systematic or the loop's own chaos?).

Run in the build container only (imports the reference from /root/reference, like make_golden.py):

    python tests/golden/make_g14_ensemble.py [n_seeds]

For seeds 1..n the G14 loop (make_golden.g14: frozen seed-42 config-2 decoder, 3 maps at 64 x 128, the real Mask-3, RENITestLoss(1e-7,
1e-4), Adam(0.1) on the latents from zero, 200 steps) is run again with the TARGET images perturbed by 1e-6 x N(0, 1) noise -- far below
anything an 8-bit or half-float environment map resolves -- once in fp32 and once under torch.autocast(bfloat16), i.e. the reference's
own code in both arithmetics.  Recorded per seed: the PSNR (dB, over the masked-out and over the kept pixels) of the completed maps against
the UNPERTURBED fp32 run's maps (g14_c4_trajectory.npz: img_after_200), and the cosine of the final latents to that run's.
The spread over seeds is the loop's sensitivity to perturbations no arithmetic can avoid; tests/test_gpu_trajectory.py runs the bf16
kernels on the same perturbed targets (rebuilt from the seed) and compares distributions, not one draw.
Output: tests/golden/g14_ensemble.npz (a few hundred bytes of numbers)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs gdown / torchvision, imports the reference)

NOISE = 1e-6


def perturbed(imgs, seed):
    """imgs [N,3,H,W] float32 + NOISE x N(0,1) from torch's CPU generator (the GPU test rebuilds exactly this)"""
    g = torch.Generator().manual_seed(1000 + seed)
    return imgs + NOISE * torch.randn(imgs.shape, generator=g)


def psnr(x, y, sel):
    d = np.asarray(x, np.float64)[:, sel] - np.asarray(y, np.float64)[:, sel]
    return float(10.0 * np.log10(4.0 / np.mean(d * d)))


def cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def main(n):
    g = np.load(os.path.join(HERE, "g14_c4_trajectory.npz"))
    src = mg._c2_decoder()
    ckpt = {"model." + k: v.clone() for k, v in src.state_dict().items()}
    N, W = 3, int(g["W"])
    m = mg.ref.RENIAutoDecoder(N, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, True)
    m.load_state_dict(ckpt)
    imgs = torch.from_numpy(g["imgs"])
    mask = torch.from_numpy(g["mask"])
    D1 = mg.ref_utils.get_directions(W); S1 = mg.ref_utils.get_sineweight(W) * mask
    crit = mg.ref_loss.RENITestLoss(alpha=float(g["alpha"]), beta=float(g["beta"]))
    D = D1.repeat(N, 1, 1); S = S1.repeat(N, 1, 1)
    idx = torch.arange(N)
    masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
    ref_img, ref_Z = g["img_after_200"], g["Z_after_200"]

    def run(t, autocast):
        with torch.no_grad():
            m.Z.zero_()
        opt = torch.optim.Adam([m.Z], lr=float(g["lr"]))
        for _ in range(int(g["steps"])):
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                Z = m.Z[idx, :, :]
                out = m(Z, D)
                opt.zero_grad()
                tl = crit(out.float(), t, S, Z)
            tl[0].backward()
            opt.step()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            img = m(m.Z[idx, :, :], D).float().numpy().copy()
        return img, m.Z.detach().numpy().copy()

    rows = []
    for seed in range(1, n + 1):
        t = perturbed(imgs, seed).permute(0, 2, 3, 1).reshape(N, -1, 3)
        row = [seed]
        for ac in (False, True):
            img, Z = run(t, ac)
            row += [psnr(img, ref_img, masked_out), psnr(img, ref_img, ~masked_out), cos(Z, ref_Z)]
        print("seed %d: fp32 %.2f / %.2f dB cos %.4f | autocast-bf16 %.2f / %.2f dB cos %.4f" % tuple(row), flush=True)
        rows.append(row)
        np.savez_compressed(os.path.join(HERE, "g14_ensemble.npz"), noise=np.float64(NOISE), seed_base=np.int64(1000),
                            cols=np.array(["seed", "f32_psnr_masked_out", "f32_psnr_kept", "f32_cos_Z", "ac_psnr_masked_out", "ac_psnr_kept", "ac_cos_Z"]),
                            rows=np.array(rows, np.float64))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
