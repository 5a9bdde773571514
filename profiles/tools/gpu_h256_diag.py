"""Diagnostic (not collected by pytest): H = 256 gradients vs the oracle, run-to-run determinism."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem, unflatten
dev = torch.device("cuda:0")
for dt in ("f32", "bf16"):
    for (eq, nd, L) in (("SO3", 9, 0), ("SO3", 9, 1), ("SO2", 36, 5)):
        spec = O.DecoderSpec(nd, eq, 256, L, 3, True, "tanh")
        B, P = 3, 333
        params, Z, D, W, T = random_problem(spec, B, P, seed=7)
        plan = make_plan(spec, dt)
        fp = flat_params(spec, params).to(dev)
        ref = O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, W.expand(B, P, 3))
        runs = []
        for it in range(3):
            lt, dZ, dp, _ = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev))
            runs.append((dZ.cpu(), dp.cpu()))
        det = max(float((runs[0][1] - r[1]).abs().max()) for r in runs[1:])
        g = unflatten(spec, runs[0][1])
        errs = {k: O.rel_l2(g[k], ref["grads"][k]) for k in g}
        worst = max(errs, key=errs.get)
        print(dt, eq, nd, L, "run-to-run", det, "dZ", O.rel_l2(runs[0][0], ref["dZ"]), "worst", worst, errs[worst], flush=True)
