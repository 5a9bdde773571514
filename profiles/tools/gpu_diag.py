"""Stage-by-stage GPU diagnostics (prints errors instead of asserting).  Run on the GPU box:
    python profiles/tools/gpu_diag.py
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem, unflatten
from reni_amd import ops


def main():
    dev = torch.device("cuda:0")
    print("device:", torch.cuda.get_device_name(0))
    print("layout probes (mismatch counts, want [0, 0]):", ops.selftest_layouts())
    for dtype in ("f32", "bf16"):
        for (eq, nd, H, L, lll, act) in [("SO2", 9, 64, 0, True, None), ("SO2", 9, 64, 1, True, None),
                                          ("SO2", 9, 64, 3, True, "tanh"), ("SO3", 9, 64, 2, True, "tanh"),
                                          ("None", 5, 32, 2, False, None), ("SO2", 36, 128, 5, True, "tanh")]:
            spec = O.DecoderSpec(nd, eq, H, L, 3, lll, act)
            B, P = 3, 200
            params, Z, D, W, T = random_problem(spec, B, P, seed=1)
            plan = make_plan(spec, dtype)
            fp = flat_params(spec, params).to(dev)
            ref = O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, W.expand(B, P, 3))
            out = plan.forward(Z.to(dev), D.to(dev), fp).cpu()
            e_out = float((out - ref["out"]).abs().max())
            lt, dZ, dp, out2 = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev), want_out=True)
            torch.cuda.synchronize()
            e_out2 = float((out2.cpu() - ref["out"]).abs().max())
            e_loss = abs(float(lt[0]) - ref["loss_terms"][0]) / abs(ref["loss_terms"][0])
            e_dz = O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy())
            gp = unflatten(spec, dp.cpu())
            errs = {k.replace("net.", "").replace("linear.", ""): O.rel_l2(gp[k].numpy(), ref["grads"][k].numpy()) for k in gp}
            worst = max(errs.values())
            print(f"[{dtype}] {eq} nd={nd} H={H} L={L} lin={lll} act={act}: out {e_out:.2e} out(bwd) {e_out2:.2e} "
                  f"loss {e_loss:.2e} dZ {e_dz:.2e} dW worst {worst:.2e}")
            if worst > (1e-4 if dtype == "f32" else 5e-2):
                print("     per-tensor:", {k: f"{v:.1e}" for k, v in errs.items()})
    # quick timing at config-2 shape
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    for dtype in ("bf16", "f32"):
        B = 8
        params, Z, D, W, T = random_problem(spec, B, 0, seed=2, grid_w=256)
        P = D.shape[1]
        plan = make_plan(spec, dtype)
        fp = flat_params(spec, params).to(dev)
        Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
        for need_dw in (True, False):
            for _ in range(2):
                plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, need_dw=need_dw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, need_dw=need_dw)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"[{dtype}] fwd+bwd need_dw={need_dw} B={B} P={P}: {dt*1e3:.3f} ms  {B*P/dt/1e6:.1f} Msamples/s  info={plan.launch_info(B,P)}")
        for _ in range(2):
            plan.forward(Zd, Dd, fp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            plan.forward(Zd, Dd, fp)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"[{dtype}] fwd only B={B} P={P}: {dt*1e3:.3f} ms  {B*P/dt/1e6:.1f} Msamples/s")


if __name__ == "__main__":
    main()
