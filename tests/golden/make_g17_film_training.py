"""G17: G15's decoder-training loop with the reference's DEFAULT conditioning (configs/default.py:9: FiLM) at the bench architecture.

Run in the build container only (imports the reference from /root/reference, like make_golden.py):

    python tests/golden/make_g17_film_training.py

RENIAutoDecoderFiLM(8, 36, "SO2", 128, 5 FiLM layers, mapping network 3 x 128, tanh, fixed_decoder=False) from torch.manual_seed(7) -- the
HIP-side class draws bit-identical weights AND latents from the same seed (tests/test_api_cpu.py), so no state dict is stored -- G15's eight
smooth maps at 32 x 64 in batches of 4 in loader order, RENITrainLoss, Adam(1e-5 = configs/default.py:25, the rate the reference trains this model at; at 1e-3 the run diverges in ANY arithmetic:
its own autocast run ends 7x away from its fp32 run) over net + final_layer + mapping_network + Z
(RENI_module.py:80-146), 100 steps: fp32, and the same code under torch.autocast(bfloat16).  Recorded: the loss of every step, the final
latents of both runs, the norm and the head values of every final parameter of the fp32 run."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

SEED = 7
LR = float(os.environ.get("G17_LR", "1e-5"))   # configs/default.py:25: RENI.FIT_DECODER.LR_START
# python make_g17_film_training.py 256 -> G19: the same loop on the reference's SHIPPED default model (configs/default.py:13-20: 256 features,
# mapping network 3 x 256) -> g19_film256_c2_trajectory.npz
WIDTH = int(sys.argv[1]) if len(sys.argv) > 1 else 128
OUT = "g17_film_c2_trajectory.npz" if WIDTH == 128 else f"g19_film{WIDTH}_c2_trajectory.npz"


def main():
    g = np.load(os.path.join(HERE, "g15_c2_trajectory.npz"))
    N, B, W, steps = g["imgs"].shape[0], int(g["B"]), int(g["W"]), int(g["steps"])
    imgs_all = torch.from_numpy(g["imgs"])
    D1 = mg.ref_utils.get_directions(W); S1 = mg.ref_utils.get_sineweight(W)
    D = D1.repeat(B, 1, 1); S = S1.repeat(B, 1, 1)
    crit = mg.ref_loss.RENITrainLoss()

    def run(autocast):
        torch.manual_seed(SEED)
        m = mg.ref.RENIAutoDecoderFiLM(N, 36, "SO2", WIDTH, 5, WIDTH, 3, 3, "tanh", False)
        Z0 = m.Z.detach().numpy().copy()
        opt = torch.optim.Adam(m.parameters(), lr=LR)
        losses = []
        for it in range(steps):
            idx = torch.arange(B) + (it % (N // B)) * B
            t = imgs_all[idx].permute(0, 2, 3, 1).reshape(B, -1, 3)
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                out = m(m.Z[idx, :, :], D)
                loss = crit(out.float(), t, S)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
            if it % 10 == 0:
                print(autocast, it, losses[-1], flush=True)
        return m, Z0, np.array(losses)

    m_ac, _, losses_ac = run(True)
    m, Z0, losses = run(False)
    arrs = {}
    for k, p in m.named_parameters():
        if k == "Z":
            continue
        v = p.detach().numpy()
        arrs["fn." + k] = np.float64(np.linalg.norm(v.astype(np.float64)))
        arrs["fh." + k] = v.reshape(-1)[:32].copy()
    np.savez_compressed(os.path.join(HERE, OUT), seed=np.int64(SEED), width=np.int64(WIDTH), Z0_abs_sum=np.float64(np.abs(Z0).sum()),
                        losses=losses, losses_autocast_bf16=losses_ac, Z_final=m.Z.detach().numpy(),
                        Z_final_autocast_bf16=m_ac.Z.detach().numpy(), steps=np.int64(steps), lr=np.float64(LR), **arrs)
    dev = np.abs(losses_ac - losses) / losses
    print("autocast vs fp32: early", dev[:40].max(), "late", dev[40:].max())


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
