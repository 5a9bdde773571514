"""G20 / G14 diagnostic: is each bf16 kernel's latent gradient the gradient OF ITS OWN FORWARD NETWORK?
For each kernel an fp64 emulation of the network its forward pass evaluates is differentiated by autograd (straight-through at the bf16
rounding of activations): generic = hidden / head weights bf16(W); persistent = hidden weights bf16(W omega / 2 pi) 2 pi / omega, head
bf16(W).  rel-L2 distance of each kernel's dZ to each emulation's dZ at a few latents.  A kernel that is consistent sits much closer to
ITS emulation than to the fp32 network.   usage: python profiles/tools/gpu_g20_consistency.py [128|256]"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reni_oracle as O  # noqa: E402  (diagnostic only)
from tests.test_gpu_trajectory import _decoder_sd  # noqa: E402
from tests.util import load_golden  # noqa: E402
from reni_amd.models import RENIAutoDecoder  # noqa: E402
from reni_amd.utils import get_directions, get_sineweight  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
g = load_golden("g14_c4_trajectory.npz")
f = load_golden("g20_concat256_c4_trajectory.npz") if width == 256 else g
W, N, L = int(g["W"]), 3, 5
D1 = get_directions(W); S1 = get_sineweight(W) * torch.from_numpy(g["mask"])
imgs = torch.from_numpy(g["imgs"])
P = D1.shape[1]
T = imgs.permute(0, 2, 3, 1).reshape(N, P, 3)
alpha, beta = float(g["alpha"]), float(g["beta"])


def model(dtype):
    if width == 256:
        torch.manual_seed(int(f["seed"]))
        m = RENIAutoDecoder(N, 36, "SO2", 256, 5, 3, True, "tanh", 30.0, 30.0, True)
    else:
        m = RENIAutoDecoder(N, 36, "SO2", 128, 5, 3, True, "tanh", 30.0, 30.0, True)
        m.load_state_dict({"model." + k: v for k, v in _decoder_sd().items()})
    return m


sd = {k: v.detach().clone() for k, v in model("f32").state_dict().items() if k.startswith("net.")}
Ws = [sd[f"net.{i}.linear.weight"] for i in range(L + 1)] + [sd[f"net.{L + 1}.weight"]]
bs = [sd[f"net.{i}.linear.bias"] for i in range(L + 1)] + [sd[f"net.{L + 1}.bias"]]
s32 = torch.tensor(30.0 * 0.15915494309189535, dtype=torch.float32)


def emu(Z, mode, act_round=True):
    """mode: fp32 (no rounding at all) | generic | persistent"""
    Zr = Z.double().clone().requires_grad_(True)
    x = O.encode("SO2", Zr, D1.double().expand(N, P, 3))
    h = x
    for i in range(L + 2):
        Wi = Ws[i]
        if mode != "fp32" and i >= 1:
            if mode == "persistent" and i <= L:
                Wi = ((Wi * s32).bfloat16().float().double() / s32.double())
            else:
                Wi = Wi.bfloat16().float().double()
        a = torch.nn.functional.linear(h, Wi.double(), bs[i].double())
        if i <= L:
            h = torch.sin(30.0 * a)
            if mode != "fp32" and act_round:
                h = h + (h.detach().float().bfloat16().double() - h.detach())
        else:
            out = torch.tanh(a)
    terms = O.test_loss(out, T.double(), S1.double().expand(N, P, 3), Zr, alpha, beta)
    terms[0].backward()
    return Zr.grad.detach().numpy()


def kernels(Z):
    res = {}
    for name, env, dtype in (("f32", None, "f32"), ("persistent", None, "bf16"), ("generic", "1", "bf16")):
        if env:
            os.environ["RENI_NO_PERSIST"] = env
        else:
            os.environ.pop("RENI_NO_PERSIST", None)
        m = model(dtype)
        m.set_compute_dtype(dtype).to(dev)
        lt, dZ, _, _ = m._plan().forward_loss_backward(Z.to(dev), D1.to(dev), m._flat_params(), T.to(dev), S1.to(dev), loss_kind="test", alpha=alpha, beta=beta, need_dw=False)
        res[name] = dZ.double().cpu().numpy()
    return res


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


Zstar = torch.from_numpy(f["Z_after_200"])
for label, Z in (("Z*", Zstar), ("0.5 Z*", 0.5 * Zstar), ("Z* + 0.05 N(0,1)", Zstar + 0.05 * torch.randn(N, 36, 3, generator=torch.Generator().manual_seed(1)))):
    k = kernels(Z)
    e = {m: emu(Z, m) for m in ("fp32", "generic", "persistent")}
    e_nr = {m: emu(Z, m, act_round=False) for m in ("generic", "persistent")}
    print(f"width {width}, latents {label}: |dZ| fp32 network {np.linalg.norm(e['fp32']):.3e}")
    print(f"   fp32 kernels vs fp32 emulation: {rel(k['f32'], e['fp32']):.2e}")
    for kn in ("persistent", "generic"):
        print(f"   {kn:10s} kernel vs: fp32 network {rel(k[kn], e['fp32']):.3e} | generic emulation {rel(k[kn], e['generic']):.3e} | persistent emulation {rel(k[kn], e['persistent']):.3e}"
              f" | (weights only, no activation rounding: generic {rel(k[kn], e_nr['generic']):.3e}, persistent {rel(k[kn], e_nr['persistent']):.3e})")
    print(f"   emulations against the fp32 network: generic {rel(e['generic'], e['fp32']):.3e}, persistent {rel(e['persistent'], e['fp32']):.3e}")

# ---- is the kernel's deviation from ITS network's exact gradient a bias or noise?  K latents around Z*
K = 12
dev_p, dev_g = [], []
for k in range(K):
    Z = Zstar + 1e-3 * torch.randn(N, 36, 3, generator=torch.Generator().manual_seed(100 + k))
    kk = kernels(Z)
    dev_p.append(kk["persistent"] - emu(Z, "persistent"))
    dev_g.append(kk["generic"] - emu(Z, "generic"))
for name, d in (("persistent", np.array(dev_p)), ("generic", np.array(dev_g))):
    tot = np.mean([np.linalg.norm(x) for x in d]); bias = np.linalg.norm(d.mean(0))
    print(f"width {width}: {name:10s} kernel minus the exact gradient of its own network, {K} latents around Z*: mean |e| {tot:.3e}, |mean e| {bias:.3e}, ratio {bias / tot:.2f}")
