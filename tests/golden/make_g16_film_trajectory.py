"""G16: G14's latent-optimisation loop with the reference's DEFAULT conditioning (configs/default.py:9: FiLM) at the bench architecture.

Run in the build container only (imports the reference from /root/reference, like make_golden.py):

    python tests/golden/make_g16_film_trajectory.py

RENIAutoDecoderFiLM(3, 36, "SO2", 128, 5 FiLM layers, mapping network 3 x 128, tanh, fixed_decoder=True) from torch.manual_seed(7) -- the
HIP-side class draws bit-identical weights from the same seed (tests/test_api_cpu.py), so no state dict is stored -- G14's three maps at
64 x 128, the real Mask-3, RENITestLoss(1e-7, 1e-4), Adam(0.1) on the latents from zero, 200 steps: fp32, and the same code under
torch.autocast(bfloat16).  Recorded: the loss 4-tuple every ten steps, the final latents and the completed maps of both runs.
Round 6 (profiles/r06_trajectory.md): does the forward / backward weight-image inconsistency found on the concat path exist on FiLM's?"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

SEED = 7
# python make_g16_film_trajectory.py 256 -> G18: the same loop on the reference's SHIPPED default model (configs/default.py:13-16:
# 256 features, 5 FiLM layers, mapping network 3 x 256) -> g18_film256_c4_trajectory.npz
WIDTH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
OUT = "g16_film_c4_trajectory.npz" if WIDTH == 128 else f"g18_film{WIDTH}_c4_trajectory.npz"


def main():
    g = np.load(os.path.join(HERE, "g14_c4_trajectory.npz"))
    N, W = 3, int(g["W"])
    torch.manual_seed(SEED)
    m = mg.ref.RENIAutoDecoderFiLM(N, 36, "SO2", WIDTH, 5, WIDTH, 3, 3, "tanh", True)
    assert float(m.Z.abs().sum()) == 0.0
    imgs = torch.from_numpy(g["imgs"]); mask = torch.from_numpy(g["mask"])
    D1 = mg.ref_utils.get_directions(W); S1 = mg.ref_utils.get_sineweight(W) * mask
    crit = mg.ref_loss.RENITestLoss(alpha=float(g["alpha"]), beta=float(g["beta"]))
    t = imgs.permute(0, 2, 3, 1).reshape(N, -1, 3)
    D = D1.repeat(N, 1, 1); S = S1.repeat(N, 1, 1)
    idx = torch.arange(N)
    steps = 200

    def run(autocast):
        with torch.no_grad():
            m.Z.zero_()
        opt = torch.optim.Adam([m.Z], lr=1e-1)
        rec_at, terms = [], []
        for it in range(steps):
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                Z = m.Z[idx, :, :]
                out = m(Z, D)
                opt.zero_grad()
                tl = crit(out.float(), t, S, Z)
            tl[0].backward()
            opt.step()
            if it % 10 == 0 or it == steps - 1:
                rec_at.append(it); terms.append([x.item() for x in tl])
                print(autocast, it, terms[-1][0], flush=True)
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            img = m(m.Z[idx, :, :], D).float().numpy().copy()
        return rec_at, np.array(terms), m.Z.detach().numpy().copy(), img

    rec_at, terms, Zf, img = run(False)
    _, terms_ac, Zf_ac, img_ac = run(True)
    np.savez_compressed(os.path.join(HERE, OUT), seed=np.int64(SEED), width=np.int64(WIDTH), rec_at=np.array(rec_at), terms=terms,
                        terms_autocast_bf16=terms_ac, Z_after_200=Zf, Z_after_200_autocast_bf16=Zf_ac, img_after_200=img.astype(np.float32),
                        img_after_200_autocast_bf16=img_ac.astype(np.float16), steps=np.int64(steps), lr=np.float64(1e-1))
    print("saved", OUT, os.path.getsize(os.path.join(HERE, OUT)) / 1024, "KiB")


if __name__ == "__main__":
    main()
