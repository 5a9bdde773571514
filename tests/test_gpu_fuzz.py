"""Seeded shape fuzzing of the fused paths against the CPU oracle (pytest -m gpu): every compiled width, 0..7 hidden
layers, tiny / ragged direction counts, per-image direction sets, both conditionings, trainable and frozen decoders,
fp32 and bf16 -- the combinations the hand-picked cases do not enumerate (stream path with L = 1, persistent kernel with
L = 1..5 and ragged tiles, single-sample problems, ...)."""
import os

import numpy as np
import pytest
import torch

from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, unflatten

pytestmark = pytest.mark.gpu

TOL = {"f32": dict(loss=5e-6, grad=3e-5), "bf16": dict(loss=3e-3, grad=3.5e-2)}


def _cases(n=64):
    rng = np.random.RandomState(20261001)
    out = []
    for i in range(n):
        H = int(rng.choice([32, 64, 128, 128, 256]))
        c = dict(H=H, L=int(rng.randint(0, 8)) if H <= 128 else int(rng.randint(0, 4)), eq=str(rng.choice(["SO2", "SO3", "None"])),
                 nd=int(rng.choice([1, 2, 5, 9])), B=int(rng.randint(1, 5)), P=int(rng.choice([1, 7, 31, 33, 127, 129, 300])),
                 dtype="bf16" if rng.rand() < 0.6 else "f32", film=bool(rng.rand() < 0.4), frozen=bool(rng.rand() < 0.3),
                 per_image=bool(rng.rand() < 0.3), act=rng.choice(["tanh", "none", "exp"]).item(), seed=1000 + i)
        if c["film"] and c["eq"] == "None":
            c["eq"] = "SO2"
        if c["act"] == "none":
            c["act"] = None
        out.append(c)
    return out


@pytest.mark.parametrize("c", _cases(int(os.environ.get("RENI_FUZZ_CASES", "64"))), ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k != "seed"))
def test_fuzz_against_oracle(c):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(c["seed"])
    B, P, nd, H, L = c["B"], c["P"], c["nd"], c["H"], c["L"]
    Z = torch.randn(B, nd, 3, generator=gen) * 0.6
    D = torch.nn.functional.normalize(torch.randn(B if c["per_image"] else 1, P, 3, generator=gen), dim=-1)
    S = torch.rand(1, P, 3, generator=gen) + 0.1
    T = torch.rand(B, P, 3, generator=gen) * 2 - 1
    tol = dict(TOL[c["dtype"]])
    # Outside the shipped configurations' regime -- more than five hidden layers (bf16 rounding compounds per layer), or a handful of
    # samples (no averaging over directions behind the bf16 rounding of each) -- the bf16 gradient tolerance is not a constant but
    # what THE REFERENCE'S OWN ARITHMETIC gives on the same problem with its linear layers in bf16 (autocast), x 1.5, never below the
    # 3e-2 of SURVEY 8c (the constants 8e-2 and (L + 2) / 6 that stood here asserted less than they seemed to).
    own_bf16 = c["dtype"] == "bf16" and (L > 5 or B * P < 256) and B * P >= 8

    def bf16_bound(ref, run_ref):
        with torch.autocast("cpu", dtype=torch.bfloat16):
            rb = run_ref()
        errs = [O.rel_l2(rb["dZ"].float().numpy(), ref["dZ"].numpy())]
        errs += [O.rel_l2(rb["grads"][k].float().numpy(), v.numpy()) for k, v in ref["grads"].items() if float(v.abs().max()) > 0]
        return max(TOL["bf16"]["grad"], 1.5 * max(errs))

    BF16_CAP = 0.15  # a derived bound above this asserts nothing: such a case is checked through the fp32 kernels instead

    if c["dtype"] == "bf16" and B * P < 8:
        # A gradient from fewer than eight samples through up to eight sine layers is ill-conditioned: on the one-sample
        # case of a 600-case run fp32 itself kept 3.5 digits (3e-4) and bf16 none (profiles/tools/gpu_fuzz_one.py).  A bf16 gradient
        # comparison would assert nothing there, so these shapes (launch geometry, ragged single tile, masking) are
        # checked through the fp32 kernels instead, at the precision fp32 keeps on them.
        c = dict(c, dtype="f32")
        tol = dict(loss=5e-6, grad=1e-3)
    if c["film"]:
        from reni_amd.film import RENIAutoDecoderFiLM
        spec = O.FilmSpec(nd, c["eq"], H, L + 1, 12, 1, 3, c["act"])
        params = O.film_init_params(spec, gen)
        ref = O.film_fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, S)
        if own_bf16:
            tol["grad"] = bf16_bound(ref, lambda: O.film_fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, S))
            if tol["grad"] > BF16_CAP:   # ill-conditioned for ANY bf16 arithmetic: the shape is checked at fp32 precision
                c = dict(c, dtype="f32")
                tol = dict(loss=5e-6, grad=1e-3)
        m = RENIAutoDecoderFiLM(B, nd, c["eq"], H, L + 1, 12, 1, 3, c["act"], c["frozen"])
        m.load_state_dict({"model." + k: v for k, v in params.items()}, strict=False)
        m.set_compute_dtype(c["dtype"]).to(dev)
        Zd = Z.to(dev).requires_grad_(True)
        terms = m.fused_loss(Zd, D.to(dev), T.to(dev), S.to(dev))
        terms[0].backward()
        assert abs(float(terms[0].detach()) - ref["terms"][0]) <= tol["loss"] * abs(ref["terms"][0]) + 1e-12
        assert O.rel_l2(Zd.grad.cpu().numpy(), ref["dZ"].numpy()) <= tol["grad"]
        if not c["frozen"]:
            got = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None and k != "Z"}
            for k, v in ref["grads"].items():
                if float(v.abs().max()) > 0:
                    assert O.rel_l2(got[k].numpy(), v.numpy()) <= tol["grad"], k
    else:
        spec = O.DecoderSpec(nd, c["eq"], H, L, 3, True, c["act"])
        params = O.init_params(spec, gen)
        ref = O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, S.expand(B, P, 3))
        if own_bf16:
            tol["grad"] = bf16_bound(ref, lambda: O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, S.expand(B, P, 3)))
            if tol["grad"] > BF16_CAP:   # ill-conditioned for ANY bf16 arithmetic: the shape is checked at fp32 precision
                c = dict(c, dtype="f32")
                tol = dict(loss=5e-6, grad=1e-3)
        plan = make_plan(spec, c["dtype"])
        fp = flat_params(spec, params).to(dev)
        lt, dZ, dp, out = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), S.to(dev), need_dw=not c["frozen"], want_out=True)
        assert float((out.cpu() - ref["out"]).abs().max()) <= (1e-5 if c["dtype"] == "f32" else 8e-3)
        assert abs(float(lt[0]) - ref["loss_terms"][0]) <= tol["loss"] * abs(ref["loss_terms"][0]) + 1e-12
        assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()) <= tol["grad"]
        if not c["frozen"]:
            gp = unflatten(spec, dp.cpu())
            for k in gp:
                assert O.rel_l2(gp[k].numpy(), ref["grads"][k].numpy()) <= tol["grad"], k
