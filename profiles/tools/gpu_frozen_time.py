"""Frozen-decoder instance alone at config 4's size (21 x 32 768 directions, dense, RENITestLoss): HIP-event time of the main pass and of the
statistics pass over 30 calls -- no check of the values (for the timing-only RENI_EXP ablation builds, whose results are wrong by design)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reni_oracle as O  # noqa: E402
from reni_amd import ops  # noqa: E402
from tests.util import flat_params, make_plan, random_problem  # noqa: E402

dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
spec = O.DecoderSpec(36, "SO2", H, 5, 3, True, "tanh")
B = 21
params, Z, D, W, T = random_problem(spec, B, 0, seed=2, grid_w=256)
plan = make_plan(spec, "bf16")
fp = flat_params(spec, params).to(dev)
Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
for _ in range(20):
    plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, loss_kind="test", alpha=1e-7, beta=1e-4, need_dw=False)
torch.cuda.synchronize()
ops.profile_enable(True)
ops.profile_read(reset=True, kind=ops.PROF_ALL)
for _ in range(30):
    plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, loss_kind="test", alpha=1e-7, beta=1e-4, need_dw=False)
torch.cuda.synchronize()
ms, n = ops.profile_read(reset=False, kind=ops.PROF_FWD_BWD)
lo, hi = ops.profile_minmax(ops.PROF_FWD_BWD)
sm, sn = ops.profile_read(reset=False, kind=ops.PROF_STATS)
print("frozen H=%d: main pass avg %.4f ms (min %.4f max %.4f)  statistics pass avg %.4f ms" % (H, ms / max(n, 1), lo, hi, sm / max(sn, 1)))
