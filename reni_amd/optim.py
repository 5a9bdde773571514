"""Fused Adam over flat fp32 buffers (reni_adam_step in libreni_hip.so).

Semantics = ``torch.optim.Adam(params, lr)`` with the default betas (0.9, 0.999) and eps 1e-8,
which is what the reference always uses (src/lightning/RENI_module.py:192 ignores the configured
betas -- SURVEY.md Appendix B3).  It is a ``torch.optim.Optimizer`` so the reference's LR
schedulers (ExponentialLR / StepLR, RENI_module.py:197-222) drive it unchanged.
"""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, grad_scale=grad_scale)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous parameters")
                ops.adam_step(p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"],
                              group["betas"], group["eps"], group["grad_scale"])
        return loss
