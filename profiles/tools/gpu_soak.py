"""Diagnostic (not collected by pytest): run-to-run bit-equality of long optimisation loops on the persistent kernels --
300 training steps (k_reni_train_bf16<128,true>) and 300 latent-only steps (k_reni_train_bf16<128,false>), each twice; and
(round 2) FiLM on the persistent kernels, the H = 256 fragment streams (bf16 concat / FiLM, fp32 concat)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from reni_amd.engine import TrainEngine
from reni_amd.models import RENIAutoDecoder
from reni_amd.utils import get_directions, get_sineweight

dev = torch.device("cuda:0")
D = get_directions(256).to(dev); S = get_sineweight(256).to(dev); P = D.shape[1]


def run(frozen, B, steps, loss_kind, make=None, dtype="bf16", grid=None):
    global D, S, P
    if grid is not None:
        D = get_directions(grid).to(dev); S = get_sineweight(grid).to(dev); P = D.shape[1]
    torch.manual_seed(0)
    m = make(B) if make else RENIAutoDecoder(B, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, frozen)
    if frozen:
        with torch.no_grad():
            m.Z.normal_(generator=torch.Generator().manual_seed(1))
    m.set_compute_dtype(dtype).to(dev)
    T = torch.rand(B, P, 3, generator=torch.Generator().manual_seed(2)).to(dev) * 2 - 1
    eng = TrainEngine(m, lr=1e-3 if not frozen else 1e-2, loss_kind=loss_kind, alpha=1e-7, beta=1e-4 if loss_kind == "test" else 0.0)
    idx = torch.arange(B, device=dev)
    losses = []
    for s in range(steps):
        terms = eng.step(idx, T, S, D)
        if s % 50 == 0 or s == steps - 1:
            losses.append(float(terms[0]))
    torch.cuda.synchronize()
    lat = (m.Z if hasattr(m, "Z") else m.mu).detach().clone()
    return losses, lat, m._flat_params().detach().clone()


for name, frozen, B, kind in (("training (C2 shape, 16 images)", False, 16, "mse"), ("latent-only, MSE (C4 shape, 21 images)", True, 21, "mse"),
                              ("latent-only, RENITestLoss with cosine term", True, 21, "test")):
    a = run(frozen, B, 300, kind)
    b = run(frozen, B, 300, kind)
    same = torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[0] == b[0]
    finite = bool(torch.isfinite(a[1]).all() and torch.isfinite(a[2]).all())
    print(f"{name}: losses {['%.5f' % x for x in a[0]]}  bit-identical runs: {same}  finite: {finite}  decreasing: {a[0][-1] < a[0][0]}")
    assert same and finite and a[0][-1] < a[0][0]
from reni_amd.film import RENIAutoDecoderFiLM
extra = (
    ("FiLM 5x128 on the persistent kernels, 16 images", lambda B: RENIAutoDecoderFiLM(B, 36, "SO2", 128, 5, 128, 3, 3, "tanh", False), "bf16", 256, 16, 150),
    ("concat 5x256 ND49 bf16 (fragment stream), 8 images", lambda B: RENIAutoDecoder(B, 49, "SO2", 256, 5, 3, True, "tanh", 30, 30, False), "bf16", 256, 8, 60),
    ("FiLM 5x256 bf16 (fragment stream, image runs), 8 images", lambda B: RENIAutoDecoderFiLM(B, 49, "SO2", 256, 5, 256, 3, 3, "tanh", False), "bf16", 256, 8, 60),
    ("concat 5x256 ND49 fp32 (fp32 fragment stream), 4 images at 64x128", lambda B: RENIAutoDecoder(B, 49, "SO2", 256, 5, 3, True, "tanh", 30, 30, False), "f32", 128, 4, 40),
)
for name, make, dtype, grid, B, steps in extra:
    a = run(False, B, steps, "mse", make, dtype, grid)
    b = run(False, B, steps, "mse", make, dtype, grid)
    same = torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[0] == b[0]
    finite = bool(torch.isfinite(a[1]).all() and torch.isfinite(a[2]).all())
    print(f"{name}: losses {['%.5f' % x for x in a[0]]}  bit-identical runs: {same}  finite: {finite}  decreasing: {a[0][-1] < a[0][0]}", flush=True)
    assert same and finite and a[0][-1] < a[0][0]
print("soak ok")
