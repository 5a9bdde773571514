"""CPU diagnostic (build container or any box; no GPU): G20's latent-optimisation loop in plain fp32 torch autograd on the concat decoder at 256
features with its hidden weights rounded as bf16(W x scale) / scale for several scales (scale 1 = the generic bf16 kernels' network and
the reference under autocast's weights; omega / 2 pi = the persistent kernels' images; the rest arbitrary), head bf16(W_out), sine outputs
rounded to bf16 (straight-through).  Prints where each loop's completed maps end relative to the fp32 network's: the spread is what "which
2^-9 perturbation of the weights" is worth (profiles/r06_trajectory.md section 8).   usage: python profiles/tools/cpu_g20_weight_rounding_lottery.py"""
import sys, math, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle import reni_oracle as O
from tests.util import load_golden
from reni_amd.models import RENIAutoDecoder
torch.set_num_threads(8)
g = load_golden("g14_c4_trajectory.npz"); f = load_golden("g20_concat256_c4_trajectory.npz")
W, N, L = int(g["W"]), 3, 5
D1 = O.get_directions(W); S1 = O.get_sineweight(W) * torch.from_numpy(g["mask"])
imgs = torch.from_numpy(g["imgs"]); P = D1.shape[1]
T = imgs.permute(0, 2, 3, 1).reshape(N, P, 3)
alpha, beta = float(g["alpha"]), float(g["beta"])
torch.manual_seed(int(f["seed"]))
m = RENIAutoDecoder(N, 36, "SO2", 256, 5, 3, True, "tanh", 30.0, 30.0, True)
sd = {k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith("net.")}
Ws = [sd[f"net.{i}.linear.weight"] for i in range(L + 1)] + [sd[f"net.{L + 1}.weight"]]
bs = [sd[f"net.{i}.linear.bias"] for i in range(L + 1)] + [sd[f"net.{L + 1}.bias"]]
masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
ref_img, ref_Z = f["img_after_200"], f["Z_after_200"]
def psnr(x, y, sel):
    d = np.asarray(x, np.float64)[:, sel] - np.asarray(y, np.float64)[:, sel]
    return float(10 * np.log10(4.0 / np.mean(d * d)))
def run(s, act_round=True, head_round=True):
    Wq = list(Ws)
    if s is not None:
        st = torch.tensor(s, dtype=torch.float32)
        for i in range(1, L + 1):
            Wq[i] = (Ws[i] * st).bfloat16().float() / st
        if head_round: Wq[L + 1] = Ws[L + 1].bfloat16().float()
    Z = torch.zeros(N, 36, 3, requires_grad=True)
    opt = torch.optim.Adam([Z], lr=0.1)
    Dx = D1.expand(N, P, 3); Sx = S1.expand(N, P, 3)
    def fwd(Zv):
        h = O.encode("SO2", Zv, Dx)
        for i in range(L + 2):
            a = torch.nn.functional.linear(h, Wq[i], bs[i])
            if i <= L:
                h = torch.sin(30.0 * a)
                if s is not None and act_round:
                    h = h + (h.detach().bfloat16().float() - h.detach())
            else:
                return torch.tanh(a)
    for it in range(200):
        opt.zero_grad()
        out = fwd(Z)
        O.test_loss(out, T, Sx, Z, alpha, beta)[0].backward()
        opt.step()
    with torch.no_grad():
        img = fwd(Z).numpy()
    zc = float((Z.detach().numpy().ravel() @ ref_Z.ravel()) / (np.linalg.norm(Z.detach().numpy()) * np.linalg.norm(ref_Z)))
    return psnr(img, ref_img, masked_out), psnr(img, ref_img, ~masked_out), zc
for name, s in [("fp32 (no rounding)", None), ("generic: bf16(W)", 1.0), ("persistent: bf16(W omega/2pi)", 30.0 * 0.15915494309189535)] + [(f"scale {x}", x) for x in (1.37, 2.9, 3.3, 6.1, 0.77, 5.0)]:
    r = run(s)
    print(f"{name:32s} masked-out {r[0]:.2f} dB  kept {r[1]:.2f} dB  cos {r[2]:.4f}", flush=True)
