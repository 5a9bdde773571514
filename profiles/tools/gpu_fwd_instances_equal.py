"""Diagnostic: do the forward-only (statistics) instance and the backward-call instances of the persistent bf16 kernels produce the SAME
output values?  (The cosine term's coefficients come from the former's outputs and are applied to the latter's.)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oracle import reni_oracle as O
from tests.util import random_problem, make_plan, flat_params
dev = torch.device("cuda:0")
for H, L in ((128, 5), (128, 3), (256, 5)):
    spec = O.DecoderSpec(36, "SO2", H, L, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, 5, 0, seed=77, grid_w=64)
    plan = make_plan(spec, "bf16"); fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
    o_f = plan.forward(Zd, Dd, fp)
    res = {}
    for name, kw in (("frozen", dict(need_dw=False)), ("training", dict(need_dw=True))):
        _, _, _, o = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, want_out=True, **kw)
        res[name] = o
        print(f"H {H} L {L}: forward instance vs {name} instance: bit-equal {torch.equal(o_f, o)}, max |diff| {float((o_f - o).abs().max()):.3e}")
