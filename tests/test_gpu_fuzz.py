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

    if B * P < 8:
        # A gradient from fewer than eight samples through up to eight sine layers is ill-conditioned: on the one-sample
        # case of a 600-case run fp32 itself kept 3.5 digits (3e-4) and bf16 none (profiles/tools/gpu_fuzz_one.py).  A bf16 gradient
        # comparison would assert nothing there, so these shapes (launch geometry, ragged single tile, masking) are
        # checked through the fp32 kernels instead, at the precision fp32 keeps on them.  (Round 6, 1 500 cases: two fp32 cases with
        # ONE sample per image sat at 3.2e-5 / 3.7e-5 against the 3e-5 of well-posed problems -- the same bound for them.)
        c = dict(c, dtype="f32")
        tol = dict(loss=5e-6, grad=1e-3)
    if c["film"]:
        from reni_amd.film import RENIAutoDecoderFiLM
        spec = O.FilmSpec(nd, c["eq"], H, L + 1, 12, 1, 3, c["act"])
        params = O.film_init_params(spec, gen)
        ref = O.film_fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, S)
        # FiLM angles reach tens of revolutions (freq = 15 f + 30 on every layer): the latent gradient of a small problem can be
        # ill-conditioned far beyond SURVEY 8c's constants, whatever computes it (round 6, 1 500 cases).  Two measured floors:
        #  bf16: ANY kernel with bf16 MFMA operands rounds the hidden and head weights to bf16 -- the deviation of the ORACLE'S OWN fp32
        #        result under exactly that rounding (first layer exact: the kernels split it hi + lo) is the floor; x 2 for the
        #        activations' rounding (median 2.7 %, worst 57 % over the FiLM cases; the reference under autocast: O(1));
        #  fp32: the oracle's own fp32-against-fp64 error (up to 8e-5 where the constant is 3e-5), x 2.5.
        if c["dtype"] == "bf16":
            pw = {k: (v.bfloat16().float() if (k.endswith("weight") and ((k.startswith("net.") and not k.startswith("net.0.")) or k.startswith("final_layer."))) else v)
                  for k, v in params.items()}
            e_w = O.rel_l2(O.film_fwd_loss_bwd(spec, pw, Z, D.expand(B, P, 3), T, S)["dZ"].numpy(), ref["dZ"].numpy())
            tol["grad"] = max(tol["grad"], 2.0 * e_w)
            if tol["grad"] > BF16_CAP:
                c = dict(c, dtype="f32")
                tol = dict(TOL["f32"])
        if own_bf16 and c["dtype"] == "bf16":
            tol["grad"] = max(tol["grad"], bf16_bound(ref, lambda: O.film_fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, S)))
            if tol["grad"] > BF16_CAP:   # ill-conditioned for ANY bf16 arithmetic: the shape is checked at fp32 precision
                c = dict(c, dtype="f32")
                tol = dict(loss=5e-6, grad=1e-3)
        if c["dtype"] == "f32":
            r64 = O.film_fwd_loss_bwd(spec, {k: v.double() for k, v in params.items()}, Z.double(), D.expand(B, P, 3).double(), T.double(), S.double())
            tol["grad"] = max(tol["grad"], 2.5 * O.rel_l2(ref["dZ"].numpy(), r64["dZ"].numpy()))
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
