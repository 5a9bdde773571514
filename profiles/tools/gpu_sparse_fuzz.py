"""diagnostic: RENI_WEIGHT_SPARSE / RENI_WEIGHT_COMPACT against the dense call on random masks and batches, thousands of calls in one
process -- dense == tiles bit for bit, pixels bit-identical run to run and within rounding of dense"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from oracle import reni_oracle as O
from tests.util import random_problem, make_plan, flat_params
dev = torch.device("cuda:0")
HID = int(sys.argv[3]) if len(sys.argv) > 3 else 128   # (256: the sparse / compact walks of k_reni_wide256<1>)
spec = O.DecoderSpec(36, "SO2", HID, 5, 3, True, "tanh")
plan = make_plan(spec, "bf16")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for it in range(N):
    B = int(torch.randint(1, 9, (1,), generator=g))
    gw = [64, 128, 256][int(torch.randint(0, 3, (1,), generator=g))]
    params, Z, D, W, T = random_problem(spec, B, 0, seed=1000 + it, grid_w=gw)
    Hh, Ww = gw // 2, gw
    P = Hh * Ww
    m = torch.zeros(B, Hh, Ww, 1)
    for b in range(B):
        kind = int(torch.randint(0, 6, (1,), generator=g))
        if kind == 0:
            m[b] = 1.0
        elif kind == 1:
            pass
        else:
            r0 = int(torch.randint(0, Hh - 1, (1,), generator=g)); r1 = int(torch.randint(r0 + 1, Hh + 1, (1,), generator=g))
            c0 = int(torch.randint(0, Ww - 1, (1,), generator=g)); c1 = int(torch.randint(c0 + 1, Ww + 1, (1,), generator=g))
            m[b, r0:r1, c0:c1] = 1.0
            if kind == 2:
                m[b, 0, 0] = 1.0
            if kind == 3:
                m[b] = m[b] * (torch.rand(Hh, Ww, 1, generator=g) < 0.3)
    Wm = (W.view(1, Hh, Ww, 3) * m).reshape(B, P, 3).to(dev)
    fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td = Z.to(dev), D.to(dev), T.to(dev)

    def run(mode):
        lt, dZ, _, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wm, loss_kind="test", alpha=1e-7, beta=1e-4, need_dw=False, sparse_weight=mode)
        return lt, dZ
    d0, d1, s0, p0, p1 = run(False), run(False), run(True), run("pixels"), run("pixels")
    torch.cuda.synchronize()
    msg = []
    if not (torch.equal(d0[0], d1[0]) and torch.equal(d0[1], d1[1])): msg.append("dense run-to-run")
    if not (torch.equal(s0[0], d0[0]) and torch.equal(s0[1], d0[1])): msg.append("tiles != dense")
    if not (torch.equal(p0[0], p1[0]) and torch.equal(p0[1], p1[1])): msg.append("pixels run-to-run")
    den = d0[1].norm(dim=(1, 2)).clamp_min(1e-20)
    rel = ((p0[1] - d0[1]).norm(dim=(1, 2)) / den).max()
    if not (float(rel) <= 5e-6) or not torch.isfinite(p0[1]).all(): msg.append(f"pixels vs dense rel {float(rel):.3e}")
    if msg:
        bad += 1
        print("iter", it, "B", B, "grid", gw, msg, flush=True)
print("hidden", HID, "iterations", N, "bad", bad, "seconds", round(time.time() - t0, 1))
