"""Diagnostic (not collected by pytest): one fuzz case in bf16 and fp32 beside the oracle (is a failure a bug or the
conditioning of a one-sample problem?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import reni_oracle as O
from tests.test_gpu_fuzz import _cases
from tests.util import flat_params, make_plan
sel = dict(H=128, L=6, eq="SO3", nd=1, B=1, P=1, act="exp", film=False)
c = [x for x in _cases(600) if all(x[k] == v for k, v in sel.items())][0]
print(c)
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(c["seed"])
B, P, nd, H, L = c["B"], c["P"], c["nd"], c["H"], c["L"]
Z = torch.randn(B, nd, 3, generator=gen) * 0.6
D = torch.nn.functional.normalize(torch.randn(B if c["per_image"] else 1, P, 3, generator=gen), dim=-1)
S = torch.rand(1, P, 3, generator=gen) + 0.1
T = torch.rand(B, P, 3, generator=gen) * 2 - 1
spec = O.DecoderSpec(nd, c["eq"], H, L, 3, True, c["act"])
params = O.init_params(spec, gen)
ref = O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, S.expand(B, P, 3))
print("ref out", ref["out"].flatten().tolist(), "ref dZ", ref["dZ"].flatten().tolist())
for dt in ("f32", "bf16"):
    for env in ("0", "1"):
        os.environ["RENI_NO_PERSIST"] = env
        plan = make_plan(spec, dt)
        fp = flat_params(spec, params).to(dev)
        lt, dZ, dp, out = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), S.to(dev), need_dw=True, want_out=True)
        print(dt, "no_persist", env, "out", out.flatten().tolist(), "dZ", dZ.flatten().tolist(), "rel", O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()))
