"""Per-step latent-gradient error of the bf16 kernels along the reference's G14 trajectory (frozen decoder, masked RENITestLoss),
beside the error of the reference's own arithmetic under torch.autocast(bfloat16) at the same latents.
usage: python profiles/tools/gpu_traj_diag.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan

g4 = dict(np.load(os.path.join(ROOT, "tests/golden/g4_c2shape.npz")))
g = dict(np.load(os.path.join(ROOT, "tests/golden/g14_c4_trajectory.npz")))
spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
params = {k[3:]: torch.from_numpy(v) for k, v in g4.items() if k.startswith("sd.net.")}
dev = torch.device("cuda:0")
N, W = 3, int(g["W"])
D = O.get_directions(W); S = O.get_sineweight(W) * torch.from_numpy(g["mask"])
T = torch.from_numpy(g["imgs"]).permute(0, 2, 3, 1).reshape(N, -1, 3)
fp = flat_params(spec, params).to(dev)
plans = {d: make_plan(spec, d) for d in ("f32", "bf16")}
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
cos = lambda a, b: float((a.ravel() @ b.ravel()) / np.linalg.norm(a) / np.linalg.norm(b))
for name, Z in [("Z=0", np.zeros((N, 36, 3), np.float32))] + [(f"after {k}", g[f"Z_after_{k}"]) for k in (20, 100, 200)]:
    Zt = torch.from_numpy(Z)
    ref = O.fwd_loss_bwd(spec, params, Zt, D.expand(N, -1, 3), T, S.expand(N, -1, 3), "test", 1e-7, 1e-4, need_dw=False)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ac = O.fwd_loss_bwd(spec, params, Zt, D.expand(N, -1, 3), T, S.expand(N, -1, 3), "test", 1e-7, 1e-4, need_dw=False)
    r = ref["dZ"].numpy()
    out = {}
    for d, plan in plans.items():
        lt, dZ, _, _ = plan.forward_loss_backward(Zt.to(dev), D.to(dev), fp, T.to(dev), S.to(dev), loss_kind="test", alpha=1e-7, beta=1e-4, need_dw=False)
        out[d] = dZ.cpu().numpy()
    a = ac["dZ"].float().numpy()
    print(f"{name:10s} |dZ| {np.linalg.norm(r):.3e}  hip f32 rel {rel(out['f32'], r):.2e}  hip bf16 rel {rel(out['bf16'], r):.2e} cos {cos(out['bf16'], r):.6f}"
          f"  | reference autocast-bf16 rel {rel(a, r):.2e} cos {cos(a, r):.6f}")
    # per-component sign agreement (what Adam's first steps see)
    print(f"{'':10s} sign flips vs reference: hip bf16 {(np.sign(out['bf16']) != np.sign(r)).mean():.4f}  autocast {(np.sign(a) != np.sign(r)).mean():.4f}")
