"""G20 / G14 diagnostic: the SYSTEMATIC part of each bf16 kernel's latent-gradient error near the reference's optimum.
At K latents Z_k = Z* + 1e-3 N(0, 1) around the reference's final latents the gradient of RENITestLoss is taken by the fp32 kernels (truth),
the persistent bf16 kernels and the generic bf16 kernels; e = g - g_fp32.  |mean_k e| is what does not average out (a bias shifts the
stationary point), mean_k |e| the total.  usage: python profiles/tools/gpu_g20_gradbias.py [128|256] [K]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_trajectory import _decoder_sd  # noqa: E402
from tests.util import load_golden  # noqa: E402
from reni_amd.models import RENIAutoDecoder  # noqa: E402
from reni_amd.utils import get_directions, get_sineweight  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
g = load_golden("g14_c4_trajectory.npz")
f = load_golden("g20_concat256_c4_trajectory.npz") if width == 256 else g
W, N = int(g["W"]), 3
D = get_directions(W).to(dev)
S = (get_sineweight(W) * torch.from_numpy(g["mask"])).to(dev)
imgs = torch.from_numpy(g["imgs"]).to(dev)
P = D.shape[1]
T = imgs.permute(0, 2, 3, 1).reshape(N, P, 3)
Zs = [torch.from_numpy(f["Z_after_200"]) + (1e-3 * torch.randn(N, 36, 3, generator=torch.Generator().manual_seed(k)) if k else 0) for k in range(K)]
res = {}
for name, env, dtype in (("f32", None, "f32"), ("persistent", None, "bf16"), ("generic", "1", "bf16")):
    if env:
        os.environ["RENI_NO_PERSIST"] = env
    else:
        os.environ.pop("RENI_NO_PERSIST", None)
    if width == 256:
        torch.manual_seed(int(f["seed"]))
        m = RENIAutoDecoder(N, 36, "SO2", 256, 5, 3, True, "tanh", 30.0, 30.0, True)
    else:
        m = RENIAutoDecoder(N, 36, "SO2", 128, 5, 3, True, "tanh", 30.0, 30.0, True)
        m.load_state_dict({"model." + k: v for k, v in _decoder_sd().items()})
    m.set_compute_dtype(dtype).to(dev)
    plan, fp = m._plan(), m._flat_params()
    out = []
    for Z in Zs:
        lt, dZ, _, _ = plan.forward_loss_backward(Z.to(dev), D, fp, T, S, loss_kind="test", alpha=float(g["alpha"]), beta=float(g["beta"]), need_dw=False)
        out.append(dZ.double().cpu().numpy())
    res[name] = np.array(out)   # [K, N, 36, 3]
g32 = res["f32"]
print(f"width {width}: |g_fp32| per latent set: mean {np.mean([np.linalg.norm(x) for x in g32]):.3e}")
for name in ("persistent", "generic"):
    e = res[name] - g32
    tot = np.mean([np.linalg.norm(x) for x in e]); bias = np.linalg.norm(e.mean(0))
    print(f"  {name:10s}: mean_k |e| {tot:.3e}   |mean_k e| (bias) {bias:.3e}   bias / total {bias / tot:.2f}")
ep, eg = (res["persistent"] - g32).mean(0).ravel(), (res["generic"] - g32).mean(0).ravel()
print(f"  cosine between the two kernels' bias vectors: {ep @ eg / (np.linalg.norm(ep) * np.linalg.norm(eg)):.3f}")
