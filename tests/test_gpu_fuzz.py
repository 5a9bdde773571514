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


def _cases_medium(n=16):
    """Problems of many tiles (round 6): up to nine images of up to ~4 000 directions, ragged counts around the 128-sample tile and the
    tile-pair / workgroup-range boundaries of the persistent kernels (H = 128: k_reni_train_bf16 + k_reni_l0_ring / k_reni_dw1; H = 256:
    k_reni_wide256 + k_dw_frag), the latent sizes of the shipped configurations.  The small-problem generator above stops at three tiles
    per image: the idle half of an odd tile count's last pair in k_reni_wide256<2, FILM> was found by it only at 600 cases."""
    rng = np.random.RandomState(20261003)
    out = []
    for i in range(n):
        H = int(rng.choice([128, 128, 256, 256, 64]))
        c = dict(H=H, L=int(rng.randint(1, 6)), eq=str(rng.choice(["SO2", "SO2", "SO3", "None"])), nd=int(rng.choice([2, 9, 36])),
                 B=int(rng.randint(1, 10)), P=int(rng.choice([128, 129, 255, 256, 257, 384, 640, 1000, 1024, 1152, 2047, 2048, 2304, 4097])),
                 dtype="bf16" if rng.rand() < 0.8 else "f32", film=bool(rng.rand() < 0.4), frozen=bool(rng.rand() < 0.35),
                 per_image=bool(rng.rand() < 0.2), act=rng.choice(["tanh", "none", "exp"]).item(), seed=50000 + i)
        if c["film"] and c["eq"] == "None":
            c["eq"] = "SO2"
        if c["act"] == "none":
            c["act"] = None
        out.append(c)
    return out


@pytest.mark.parametrize("c", _cases(int(os.environ.get("RENI_FUZZ_CASES", "64"))), ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k != "seed"))
def test_fuzz_against_oracle(c):
    _run_case(c)


@pytest.mark.parametrize("c", _cases_medium(int(os.environ.get("RENI_FUZZ_MEDIUM", "16"))), ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k != "seed"))
def test_fuzz_many_tiles_against_oracle(c):
    _run_case(c)


def _run_case(c):
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


@pytest.mark.parametrize("c", _cases_medium(int(os.environ.get("RENI_FUZZ_FORWARD", "16"))), ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k not in ("seed", "frozen")))
def test_fuzz_forward_against_oracle(c):
    """The INFERENCE entry points (reni_forward / reni_film_model_forward: the forward instances of the persistent kernels -- k_reni_wide256<0>,
    <0, FILM>, the H = 128 forward form -- which the fused calls above do not run) on the many-tiles problems.  fp32: 2e-5 of the
    output's scale.  bf16: 8e-3 of it (SURVEY 8c's 5e-3 is for tanh outputs of the shipped shapes), or twice what the oracle's own output
    moves by when its hidden and head weights are rounded to bf16 -- the floor of any kernel with bf16 MFMA operands -- if that is more."""
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(c["seed"] + 7)
    B, P, nd, H, L = c["B"], c["P"], c["nd"], c["H"], c["L"]
    Z = torch.randn(B, nd, 3, generator=gen) * 0.6
    D = torch.nn.functional.normalize(torch.randn(B if c["per_image"] else 1, P, 3, generator=gen), dim=-1)

    def rounded(params, film):
        def hit(k):
            if not k.endswith("weight"):
                return False
            return ((k.startswith("net.") and not k.startswith("net.0.")) or k.startswith("final_layer.")) if film else not k.startswith("net.0.")
        return {k: (v.bfloat16().float() if hit(k) else v) for k, v in params.items()}

    if c["film"]:
        from reni_amd.film import RENIAutoDecoderFiLM
        spec = O.FilmSpec(nd, c["eq"], H, L + 1, 12, 1, 3, c["act"])
        params = O.film_init_params(spec, gen)
        ref = O.film_forward(spec, params, Z, D)
        ref_w = O.film_forward(spec, rounded(params, True), Z, D)
        m = RENIAutoDecoderFiLM(B, nd, c["eq"], H, L + 1, 12, 1, 3, c["act"], True)
        m.load_state_dict({"model." + k: v for k, v in params.items()}, strict=False)
        m.set_compute_dtype(c["dtype"]).to(dev)
        with torch.no_grad():
            out = m(Z.to(dev), D.expand(B, P, 3).contiguous().to(dev)).float().cpu()
    else:
        spec = O.DecoderSpec(nd, c["eq"], H, L, 3, True, c["act"])
        params = O.init_params(spec, gen)
        ref = O.reni_forward(spec, params, Z, D.expand(B, P, 3))
        ref_w = O.reni_forward(spec, rounded(params, False), Z, D.expand(B, P, 3))
        plan = make_plan(spec, c["dtype"])
        out = plan.forward(Z.to(dev), D.to(dev), flat_params(spec, params).to(dev)).float().cpu()
    scale = max(1.0, float(ref.abs().max()))
    err = float((out - ref).abs().max())
    if c["dtype"] == "f32":
        assert err <= 2e-5 * scale, (err, scale)
    else:
        floor = float((ref_w - ref).abs().max())
        assert err <= max(8e-3 * scale, 2.0 * floor), (err, scale, floor)


@pytest.mark.parametrize("c", _cases_medium(int(os.environ.get("RENI_FUZZ_LOSSES", "16"))), ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k != "seed"))
def test_fuzz_test_loss_and_upstream_gradients(c):
    """The two other ways into the backward kernels, on the many-tiles problems: (even seeds) RENITestLoss with a LIVE cosine term and
    prior (loss_functions.py:60-71; beta = 0.05, alpha = 1e-3: the statistics pass of the forward instances, the per-image coefficients,
    the prior's 2 alpha Z) through the fused call; (odd seeds) the module's autograd path -- model(Z, D) then out.backward(dout) with an
    arbitrary upstream gradient (reni_forward + reni_backward / reni_film_model_backward: MainArgs::dout, loss_kind 2).  Bounds: fp32
    3e-5 (or 2.5 x the oracle's own fp32-against-fp64 error); bf16 3.5e-2, or twice what the oracle's own gradient moves by when its
    hidden and head weights are rounded to bf16, capped at 0.15 (above: the case runs on the fp32 kernels)."""
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(c["seed"] + 13)
    B, P, nd, H, L = c["B"], c["P"], c["nd"], c["H"], c["L"]
    Z = torch.randn(B, nd, 3, generator=gen) * 0.6
    D = torch.nn.functional.normalize(torch.randn(B if c["per_image"] else 1, P, 3, generator=gen), dim=-1).expand(B, P, 3).contiguous()
    S = (torch.rand(1, P, 3, generator=gen) + 0.1).expand(B, P, 3).contiguous()
    T = torch.rand(B, P, 3, generator=gen) * 2 - 1
    dout = torch.randn(B, P, 3, generator=gen) / (3.0 * P)
    upstream = bool(c["seed"] & 1)
    alpha, beta = 1e-3, 0.05
    film = c["film"]
    if film:
        spec = O.FilmSpec(nd, c["eq"], H, L + 1, 12, 1, 3, c["act"])
        params = O.film_init_params(spec, gen)
        fwd = lambda p, z, d: O.film_forward(spec, p, z, d)  # noqa: E731
        hit = lambda k: k.endswith("weight") and ((k.startswith("net.") and not k.startswith("net.0.")) or k.startswith("final_layer."))  # noqa: E731
    else:
        spec = O.DecoderSpec(nd, c["eq"], H, L, 3, True, c["act"])
        params = O.init_params(spec, gen)
        fwd = lambda p, z, d: O.reni_forward(spec, p, z, d)  # noqa: E731
        hit = lambda k: k.endswith("weight") and not k.startswith("net.0.")  # noqa: E731

    def oracle(p, dt=torch.float32):
        ps = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in p.items()}
        Zr = Z.detach().to(dt).clone().requires_grad_(True)
        out = fwd(ps, Zr, D.to(dt))
        if upstream:
            loss = (out * dout.to(dt)).sum()
            terms = (loss,)
        else:
            terms = O.test_loss(out, T.to(dt), S.to(dt), Zr, alpha, beta)
        terms[0].backward()
        return [float(t.detach()) for t in terms], Zr.grad.detach().float(), {k: (v.grad.detach().float() if v.grad is not None else torch.zeros_like(v).float()) for k, v in ps.items()}

    terms_ref, dZ_ref, g_ref = oracle(params)
    dtype = c["dtype"]
    tol = dict(TOL[dtype])
    tol_k = {}   # per parameter: the same floor, from the same rounded-weights run (the cosine term's head / first-layer gradients sit above dZ's)
    if dtype == "bf16":
        _, dZ_w, g_w = oracle({k: (v.bfloat16().float() if hit(k) else v) for k, v in params.items()})
        tol["grad"] = max(tol["grad"], 2.0 * O.rel_l2(dZ_w.numpy(), dZ_ref.numpy()))
        # (x 3 for parameters: the rounded-weights run leaves out the activations' rounding, which the first layer's gradient sees through
        #  every layer -- two test-loss cases of 600 sat at 4.3-5.0 % on BOTH kernel families with a weights-only floor of 2.1-2.4 %)
        #  Beside it 2 x the reference's own error on the same problem under autocast(bfloat16) (the small-problem generator: 1.5 x): with the
        #  cosine term live every bf16 arithmetic loses digits in the decoder's gradient -- autocast 3.9 % / 8.9 % on those two cases' dW_0)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            _, _, g_ac = oracle(params)
        # ONE bound for every parameter -- the largest of the per-parameter floors: with the cosine term live the dominant error is the
        # term's per-image coefficient (o . t) / |o|^2, a small difference of large sums when output and target are nearly orthogonal (as
        # random targets are), and a coefficient's error shows at the same relative size in EVERY parameter's gradient (one case of 600:
        # 4.8 % from head bias to first layer on the persistent kernels, 1.4-1.9 % on the generic ones, autocast 0.5-4.1 % --
        # profiles/tools/gpu_fuzz_losses_one.py 50196)
        worst = max([max(3.0 * O.rel_l2(g_w[k].numpy(), v.numpy()), 2.0 * O.rel_l2(g_ac[k].numpy(), v.numpy()))
                     for k, v in g_ref.items() if float(v.abs().max()) > 0] + [0.0])
        tol_k = {k: min(0.15, worst) for k in g_ref}
        if tol["grad"] > 0.15:
            dtype, tol, tol_k = "f32", dict(TOL["f32"]), {}
    if dtype == "f32":
        _, dZ_64, _ = oracle(params, torch.float64)
        tol["grad"] = max(tol["grad"], 2.5 * O.rel_l2(dZ_ref.numpy(), dZ_64.numpy()))

    if film:
        from reni_amd.film import RENIAutoDecoderFiLM
        m = RENIAutoDecoderFiLM(B, nd, c["eq"], H, L + 1, 12, 1, 3, c["act"], c["frozen"])
        m.load_state_dict({"model." + k: v for k, v in params.items()}, strict=False)
    else:
        from reni_amd.models import RENIAutoDecoder
        m = RENIAutoDecoder(B, nd, c["eq"], H, L, 3, True, c["act"], 30.0, 30.0, c["frozen"])
        sd = {"model." + k: v for k, v in params.items()}
        sd["model.Z"] = torch.zeros(B, nd, 3)
        m.load_state_dict(sd)
    m.set_compute_dtype(dtype).to(dev)
    Zd = Z.to(dev).requires_grad_(True)
    if upstream:
        out = m(Zd, D.to(dev))
        out.backward(dout.to(dev))
        assert float((out.detach().float().cpu() - fwd(params, Z, D)).abs().max()) <= (2e-5 if dtype == "f32" else 2e-2) * max(1.0, float(fwd(params, Z, D).abs().max()))
    else:
        terms = m.fused_loss(Zd, D.to(dev), T.to(dev), S.to(dev), loss_kind="test", alpha=alpha, beta=beta)
        terms[0].backward()
        got = [float(t.detach()) for t in terms]
        for i in (0, 1, 2, 3):
            assert abs(got[i] - terms_ref[i]) <= tol["loss"] * abs(terms_ref[i]) + (1e-7 if dtype == "f32" else 2e-4 * beta * B) , (i, got, terms_ref)
    assert O.rel_l2(Zd.grad.cpu().numpy(), dZ_ref.numpy()) <= tol["grad"], ("dZ", tol)
    if not c["frozen"]:
        got_g = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None and k != "Z"}
        for k, v in g_ref.items():
            if float(v.abs().max()) > 0:
                assert O.rel_l2(got_g[k].numpy(), v.numpy()) <= max(tol["grad"], tol_k.get(k, 0.0), 2e-4 if dtype == "f32" else 0.0), (k, tol, tol_k.get(k))


def _cases_engine(n=12):
    rng = np.random.RandomState(20261004)
    out = []
    for i in range(n):
        H = int(rng.choice([128, 128, 256, 64]))
        out.append(dict(H=H, L=int(rng.randint(1, 6)), eq=str(rng.choice(["SO2", "SO2", "SO3", "None"])), nd=int(rng.choice([2, 9, 36])),
                        B=int(rng.randint(1, 7)), W=int(rng.choice([16, 32, 64])), dtype="bf16" if rng.rand() < 0.8 else "f32",
                        loss=str(rng.choice(["mse", "test"])), seed=70000 + i))
    return out


@pytest.mark.parametrize("c", _cases_engine(int(os.environ.get("RENI_FUZZ_ENGINE", "12"))), ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k != "seed"))
def test_fuzz_fused_training_step_equals_three_calls(c):
    """reni_train_step_rows (ONE call: gradients, the fused Adam over decoder + latent table, the next batch's prologue staged behind the
    backward pass) against the three calls it replaces (reni_forward_loss_backward_rows -> reni_adam_step2), on random shapes: every
    width and depth, one to six images of 128 .. 2 048 directions, both losses, announced and unannounced next batches.  Same kernels in
    the same order of sums: loss terms of every step, parameters, latents and all four Adam moments must be BIT-EQUAL."""
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    from reni_amd.utils import get_directions, get_sineweight
    dev = torch.device("cuda:0")
    B, N, W = c["B"], 3 * c["B"], c["W"]
    D, S = get_directions(W).to(dev), get_sineweight(W).to(dev)
    P = D.shape[1]
    T = torch.stack([torch.rand(P, 3, generator=torch.Generator().manual_seed(c["seed"] + 7 * i)) * 2 - 1 for i in range(N)]).to(dev)
    batches = [torch.arange(B, device=dev) + o for o in (0, B, 2 * B, B, 0)]
    res = {}
    for name, kw in (("fused", dict()), ("three_calls", dict(fused_step=False))):
        torch.manual_seed(c["seed"])
        m = RENIAutoDecoder(N, c["nd"], c["eq"], c["H"], c["L"], 3, True, "tanh", 30.0, 30.0, False)
        m.set_compute_dtype(c["dtype"]).to(dev)
        e = TrainEngine(m, lr=1e-3, loss_kind=c["loss"], alpha=1e-4, beta=1e-2, **kw)
        terms = []
        for k, idx in enumerate(batches):
            nxt = batches[k + 1] if (k + 1 < len(batches) and k != 2) else None   # (step 2 does not announce: the next call runs its own prologue)
            terms.append(e.step(idx, T[idx], S, D, next_idx=nxt).clone())
        torch.cuda.synchronize()
        res[name] = [t.detach().cpu() for t in (torch.stack(terms), m._flat_params(), m.Z.data, e.m_dec, e.v_dec, e.m_lat, e.v_lat)]
        if name == "fused":
            assert e._stage is not None, "the fused step did not run"
    for a, b, what in zip(res["fused"], res["three_calls"], ("terms", "params", "Z", "m_dec", "v_dec", "m_lat", "v_lat")):
        assert torch.isfinite(a).all(), what
        assert torch.equal(a, b), (what, float((a - b).abs().max()))


@pytest.mark.parametrize("c", _cases_engine(int(os.environ.get("RENI_FUZZ_LATENT", "12"))), ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k != "seed"))
def test_fuzz_fused_latent_step_equals_two_calls(c):
    """reni_latent_step_rows[_cached] (the FIT_LATENT iteration as ONE call, examples.ipynb cell 4) against forward_loss_backward_rows +
    adam_rows_step on random shapes and random rectangular masks, with the weight dense, RENI_WEIGHT_SPARSE and RENI_WEIGHT_COMPACT: per
    mode the two engines must be BIT-EQUAL (loss terms of every step, latents, both moments); sparse must equal dense bit for bit, the
    packed form to rounding."""
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    from reni_amd.utils import get_directions, get_sineweight
    dev = torch.device("cuda:0")
    N, W = max(c["B"], 2), c["W"]
    rng = np.random.RandomState(c["seed"])
    D = get_directions(W).to(dev)
    P = D.shape[1]
    mask = torch.zeros(W // 2, W, 1)
    r0 = rng.randint(0, W // 2 - 1); r1 = rng.randint(r0 + 1, W // 2 + 1); c0 = rng.randint(0, W - 1); c1 = rng.randint(c0 + 1, W + 1)
    mask[r0:r1, c0:c1] = 1.0
    if rng.rand() < 0.3:
        mask[0, 0] = 1.0    # pixel 0 kept: the cosine term is live
    S = (get_sineweight(W).view(W // 2, W, 3) * mask).reshape(1, P, 3).to(dev)
    T = torch.stack([torch.rand(P, 3, generator=torch.Generator().manual_seed(c["seed"] + 7 * i)) * 2 - 1 for i in range(N)]).to(dev)
    idx = torch.arange(N, device=dev)
    res = {}
    for mode in (False, True, "pixels"):
        for name, kw in (("fused", dict()), ("two_calls", dict(fused_step=False))):
            torch.manual_seed(c["seed"])
            m = RENIAutoDecoder(N, c["nd"], c["eq"], c["H"], c["L"], 3, True, "tanh", 30.0, 30.0, True)
            with torch.no_grad():
                m.Z.normal_(generator=torch.Generator().manual_seed(c["seed"] + 1)).mul_(0.3)
            m.set_compute_dtype(c["dtype"]).to(dev)
            e = TrainEngine(m, lr=1e-2, loss_kind="test", alpha=1e-4, beta=1e-2, sparse_weight=mode, **kw)
            terms = [e.step(idx, T, S, D).clone() for _ in range(4)]
            torch.cuda.synchronize()
            res[(mode, name)] = [t.detach().cpu() for t in (torch.stack(terms), m.Z.data, e.m_lat, e.v_lat)]
        for a, b, what in zip(res[(mode, "fused")], res[(mode, "two_calls")], ("terms", "Z", "m_lat", "v_lat")):
            assert torch.isfinite(a).all(), (mode, what)
            assert torch.equal(a, b), (mode, what, float((a - b).abs().max()))
    for a, b, what in zip(res[(False, "fused")], res[(True, "fused")], ("terms", "Z", "m_lat", "v_lat")):
        assert torch.equal(a, b), ("sparse != dense", what, float((a - b).abs().max()))
    zd, zp = res[(False, "fused")][1], res[("pixels", "fused")][1]
    # (re-associated sums: last-bit differences in dZ, which Adam's normalisation turns into up to lr-sized differences of components whose
    #  gradient is rounding noise -- 3.6e-4 was the largest of 300 cases after four steps of 1e-2)
    assert float((zd - zp).abs().max()) <= 2e-3 * max(1.0, float(zd.abs().max())), float((zd - zp).abs().max())
