"""Cycle trace of one workgroup of the persistent training kernel (needs a -DRENI_TRACE build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem

dev = torch.device("cuda:0")
spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
B = 64
params, Z, D, W, T = random_problem(spec, B, 0, seed=2, grid_w=256)
plan = make_plan(spec, "bf16")
fp = flat_params(spec, params).to(dev)
Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
tr = torch.zeros(512, dtype=torch.int64, device=dev)
plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
torch.cuda.synchronize()
os.environ["RENI_TRACE_PTR"] = str(tr.data_ptr())
plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
torch.cuda.synchronize()
t = tr.cpu().view(-1, 2).tolist()
t = [x for x in t if x[0] != 0]
prev = None
tile_start = None
for tag, clk in t:
    if tag == 1:
        tile_start = clk
    d = (clk - prev) if prev is not None else 0
    print(f"tag {tag:3d}  +{d:8d}  (tile +{clk - tile_start if tile_start else 0:8d})")
    prev = clk
