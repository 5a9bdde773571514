"""Parity margins of the persistent bf16 kernels against the float64 oracle at the config-2 architecture (prints, does not assert):
weights as initialised and scaled up (sine arguments beyond one revolution).  usage: gpu_parity_numbers.py [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem, unflatten

dev = torch.device("cuda:0")
spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for scale in (1.0, 2.0, 4.0):
    params, Z, D, W, T = random_problem(spec, 3, P, seed=5)
    params = {k: (v * scale if k.endswith("linear.weight") and not k.startswith("net.0.") else v) for k, v in params.items()}
    ref = O.fwd_loss_bwd(spec, params, Z, D.expand(3, P, 3), T, W.expand(3, P, 3))
    for need_dw in (True, False):
        plan = make_plan(spec, "bf16")
        fp = flat_params(spec, params).to(dev)
        lt, dZ, dp, out = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev), want_out=True, need_dw=need_dw)
        torch.cuda.synchronize()
        e_out = float((out.cpu() - ref["out"]).abs().max())
        e_loss = abs(float(lt[0]) - ref["loss_terms"][0]) / abs(ref["loss_terms"][0])
        e_dz = O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy())
        msg = f"scale {scale} need_dw={need_dw}: out {e_out:.2e} loss {e_loss:.2e} dZ {e_dz:.2e}"
        if need_dw:
            gp = unflatten(spec, dp.cpu())
            errs = {k.replace("net.", "").replace("linear.", ""): O.rel_l2(gp[k].numpy(), ref["grads"][k].numpy()) for k in gp}
            msg += " dW " + " ".join(f"{k}:{v:.1e}" for k, v in errs.items())
        print(msg)
