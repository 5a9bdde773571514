"""dA / dfilm / dparams of a FiLM backward call at H = 256: k_reni_wide256<2, FILM> against the generic chain and the fp32 kernels, per piece."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from reni_amd.film import RENIAutoDecoderFiLM  # noqa: E402
from reni_amd.utils import get_directions, get_sineweight  # noqa: E402

dev = torch.device("cuda:0")
B, W, nF = 3, 64, 3
D, S = get_directions(W).to(dev), get_sineweight(W).to(dev)
T = (torch.rand(B, D.shape[1], 3, generator=torch.Generator().manual_seed(11)) * 2 - 1).to(dev)
res = {}
for name, env, dtype in (("wide", None, "bf16"), ("generic", "1", "bf16"), ("f32", None, "f32")):
    if env:
        os.environ["RENI_NO_PERSIST"] = env
    else:
        os.environ.pop("RENI_NO_PERSIST", None)
    torch.manual_seed(3)
    m = RENIAutoDecoderFiLM(B, 36, "SO2", 256, nF, 64, 2, 3, "tanh", False)
    with torch.no_grad():
        m.Z.normal_(generator=torch.Generator().manual_seed(4)); m.Z.mul_(0.5)
    m.set_compute_dtype(dtype).to(dev)
    A, film = m._glue(m.Z.detach())
    plan, flat = m._plan(), m._flat_params()
    terms, dA, dfilm, dparams, _ = plan.film_forward_loss_backward(A.detach(), film.detach(), D, flat, T, S)
    res[name] = (terms.cpu().numpy(), dA.cpu().numpy(), dfilm.cpu().numpy(), dparams.cpu().numpy())
os.environ.pop("RENI_NO_PERSIST", None)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
f = res["f32"]
for name in ("wide", "generic"):
    r = res[name]
    L = r[2].shape[1]
    print(name, "loss", r[0][0], "(f32", f[0][0], ") dA", rel(r[1], f[1]),
          "dfreq per layer", [round(rel(r[2][:, l, 0], f[2][:, l, 0]), 4) for l in range(L)],
          "dphase per layer", [round(rel(r[2][:, l, 1], f[2][:, l, 1]), 4) for l in range(L)], "dparams", rel(r[3], f[3]))
    print("   ratio dfreq wide/f32 (median over elements), per layer:", [float(np.median(r[2][:, l, 0] / f[2][:, l, 0])) for l in range(L)],
          " dphase:", [float(np.median(r[2][:, l, 1] / f[2][:, l, 1])) for l in range(L)])
