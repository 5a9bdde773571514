"""Ablation timing of the fused kernel (profiling experiments; numerics are wrong under a mask)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem
from reni_amd import ops

dev = torch.device("cuda:0")
spec = O.DecoderSpec(int(os.environ.get("ABL_ND", "36")), "SO2", int(os.environ.get("ABL_H", "128")), 5, 3, True, "tanh")
B = int(os.environ.get("ABL_B", "64"))
dtype = os.environ.get("ABL_DTYPE", "bf16")
params, Z, D, W, T = random_problem(spec, B, 0, seed=2, grid_w=256)
P = D.shape[1]
plan = make_plan(spec, dtype)
fp = flat_params(spec, params).to(dev)
Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
masks = [int(x) for x in sys.argv[1:]] or [0, 1, 3, 7, 128, 8, 32, 128 + 8, 128 + 8 + 32]
names = {256: "atomF32", 512: "atomI64", 1024: "oneBuffer", 1: "noflush", 2: "nodWgemm", 4: "notwrite", 8: "nostashstore", 32: "nostage", 128: "nodWphase"}
ops.profile_enable(True)
for need_dw in (True, False):
    for m in masks:
        if not need_dw and (m & (1 | 2 | 4 | 128)):
            continue
        os.environ["RENI_DEBUG_MASK"] = str(m)
        for _ in range(2):
            plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, need_dw=need_dw)
        torch.cuda.synchronize()
        ops.profile_read(True)
        for _ in range(5):
            plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, need_dw=need_dw)
        torch.cuda.synchronize()
        ms, n = ops.profile_read(True)
        label = "+".join(v for k, v in names.items() if m & k) or "full"
        print(f"need_dw={need_dw} mask={m:4d} {label:40s} main kernel {ms/n:8.3f} ms  {B*P/(ms/n)/1e3:8.1f} Msamples/s")
os.environ["RENI_DEBUG_MASK"] = "0"
