"""diagnostic: k_reni_wide256 run to run, dense vs RENI_WEIGHT_SPARSE, per image"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem
dev = torch.device("cuda:0")
spec = O.DecoderSpec(9, "SO2", 256, int(os.environ.get("L", "3")), 3, True, "tanh")
B = 4
params, Z, D, W, T = random_problem(spec, B, 0, seed=17, grid_w=128)
P = D.shape[1]
m = torch.zeros(B, 64, 128, 1)
m[0, 10:46, 40:83] = 1.0
m[1] = 1.0
m[2, 5:9, 100:128] = 1.0
m[2, 0, 0] = 1.0
Wm = (W.view(1, 64, 128, 3) * m).reshape(B, P, 3)
plan = make_plan(spec, "bf16")
fp = flat_params(spec, params).to(dev)
Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), Wm.to(dev)
def run(mode):
    lt, dZ, _, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, loss_kind="test", alpha=1e-7, beta=1e-4, need_dw=False, sparse_weight=mode)
    return lt.cpu(), dZ.cpu()
res = {}
for mode in (False, True, "pixels"):
    rs = [run(mode) for _ in range(4)]
    same = [torch.equal(rs[0][1], r[1]) and torch.equal(rs[0][0], r[0]) for r in rs]
    print(mode, "run-to-run equal:", same, "loss", rs[0][0].tolist())
    res[mode] = rs[0]
for mode in (True, "pixels"):
    for k in range(B):
        d = (res[mode][1][k] - res[False][1][k]).abs().max()
        print(mode, "image", k, "max |dZ - dense|", float(d), "rel", float((res[mode][1][k] - res[False][1][k]).norm() / (res[False][1][k].norm() + 1e-30)))
ref = O.fwd_loss_bwd(spec, params, Z, D.expand(B, -1, 3), T, Wm, "test", 1e-7, 1e-4, need_dw=False)
print("dense vs oracle rel_l2", O.rel_l2(res[False][1].numpy(), ref["dZ"].numpy()), "loss", float(res[False][0][0]), float(ref["loss_terms"][0]))
