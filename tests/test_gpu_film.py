"""GPU parity tests of the FiLM-conditioned variants (pytest -m gpu): module API -> per-image torch glue ->
reni_film_* (C ABI) against the goldens generated from the reference (tests/golden/g11_film_*.npz) and the
CPU oracle.  Tolerances as for the Cond-by-Concat family (tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from oracle import reni_oracle as O
from tests.util import load_golden, sd_from

pytestmark = pytest.mark.gpu

TOL = {"f32": dict(out=1e-5, loss=2e-6, grad=2e-5), "bf16": dict(out=5e-3, loss=2e-3, grad=3e-2)}
ACT = {0: None, 1: "tanh", 2: "exp"}
EQ = {1: "SO2", 2: "SO3"}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _model_from_golden(g, dev, dtype="f32", fixed=False):
    from reni_amd.film import RENIAutoDecoderFiLM, RENIVariationalAutoDecoderFiLM
    eq, nd, H, nF, mf, ml, act = [int(x) for x in g["cfg"]]
    vad = "sd.mu" in g
    cls = RENIVariationalAutoDecoderFiLM if vad else RENIAutoDecoderFiLM
    m = cls(2, nd, EQ[eq], H, nF, mf, ml, 3, ACT[act], fixed)
    m.load_state_dict({"model." + k: v for k, v in sd_from(g).items()})
    m.set_compute_dtype(dtype)
    return m.to(dev), O.FilmSpec(nd, EQ[eq], H, nF, mf, ml, 3, ACT[act])


def _grads(m):
    return {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None and k not in ("Z", "mu", "log_var")}


@pytest.mark.parametrize("tag", ["so2_ad", "so3_vad", "so2_one"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_film_golden_fused_training_step(dev, tag, dtype):
    """fused forward + RENITrainLoss + backward: out, loss, dZ and the gradient of EVERY parameter (SIREN, head,
    mapping network) against the reference's autograd."""
    g = load_golden(f"g11_film_{tag}.npz")
    m, spec = _model_from_golden(g, dev, dtype)
    W = int(g["W"])
    D = O.get_directions(W).to(dev)
    S = O.get_sineweight(W).to(dev)
    T = torch.from_numpy(g["target"]).to(dev)
    Z = torch.from_numpy(g["Z"]).to(dev).requires_grad_(True)
    tol = TOL[dtype]
    with torch.no_grad():
        out = m(Z.detach(), D)
    assert float((out.cpu() - torch.from_numpy(g["out"])).abs().max()) <= tol["out"]
    terms = m.fused_loss(Z, D, T, S)
    terms[0].backward()
    assert abs(float(terms[0]) - float(g["loss"])) <= tol["loss"] * abs(float(g["loss"]))
    assert O.rel_l2(Z.grad.cpu().numpy(), g["dZ"]) <= tol["grad"]
    got = _grads(m)
    want = {k[2:]: v for k, v in g.items() if k.startswith("g.")}
    assert sorted(got) == sorted(want)
    for k in want:
        assert O.rel_l2(got[k].numpy(), want[k]) <= tol["grad"], k


@pytest.mark.parametrize("tag", ["so2_ad", "so3_vad"])
def test_film_golden_generic_autograd_and_test_loss(dev, tag):
    """model(Z, D) followed by a torch-side loss (reni_film_backward) and the fused RENITestLoss path, fp32."""
    from reni_amd.loss_functions import RENITestLoss, RENITrainLoss
    g = load_golden(f"g11_film_{tag}.npz")
    m, spec = _model_from_golden(g, dev)
    W = int(g["W"])
    D = O.get_directions(W).to(dev).repeat(2, 1, 1)
    S = O.get_sineweight(W).to(dev).repeat(2, 1, 1)
    T = torch.from_numpy(g["target"]).to(dev)
    Z = torch.from_numpy(g["Z"]).to(dev).requires_grad_(True)
    loss = RENITrainLoss()(m(Z, D), T, S)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    assert O.rel_l2(Z.grad.cpu().numpy(), g["dZ"]) <= 2e-5
    got = _grads(m)
    for k, v in g.items():
        if k.startswith("g."):
            assert O.rel_l2(got[k[2:]].numpy(), v) <= 2e-5, k
    # RENITestLoss(alpha, beta): (loss, mse, prior, cosine) and the latent gradient
    Z2 = torch.from_numpy(g["Z"]).to(dev).requires_grad_(True)
    t_fused = m.fused_loss(Z2, D, T, S, loss_kind="test", alpha=1e-3, beta=1e-1)
    t_fused[0].backward()
    assert np.abs(t_fused.detach().cpu().numpy() - g["test_terms"]).max() <= 2e-6 * abs(g["test_terms"][0])
    assert O.rel_l2(Z2.grad.cpu().numpy(), g["test_dZ"]) <= 2e-5
    # the criterion's fused form as RENI.training_step calls it with a mask configured (sparse_weight=True): accepted by the FiLM
    # models too (no effect there: their kernels evaluate every tile) -- the same numbers
    Z4 = torch.from_numpy(g["Z"]).to(dev).requires_grad_(True)
    t_crit = RENITestLoss(alpha=1e-3, beta=1e-1).fused(m, Z4, D, T, S, sparse_weight=True)
    t_crit[0].backward()
    assert torch.equal(torch.stack([x.detach() for x in t_crit]), t_fused.detach()) and torch.equal(Z4.grad, Z2.grad)
    Z3 = torch.from_numpy(g["Z"]).to(dev).requires_grad_(True)
    t_torch = RENITestLoss(alpha=1e-3, beta=1e-1)(m(Z3, D), T, S, Z3)
    t_torch[0].backward()
    assert O.rel_l2(Z3.grad.cpu().numpy(), g["test_dZ"]) <= 2e-5


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [
    dict(eq="SO2", nd=5, H=32, nF=2, P=77, B=3),
    dict(eq="SO3", nd=9, H=64, nF=4, P=333, B=2),
    dict(eq="SO2", nd=9, H=128, nF=3, P=200, B=2),
    dict(eq="SO2", nd=36, H=128, nF=5, P=128, B=1),
    dict(eq="SO3", nd=9, H=256, nF=3, P=150, B=2),
], ids=lambda c: f"{c['eq']}-nd{c['nd']}-H{c['H']}-nF{c['nF']}")
def test_film_random_problems_vs_oracle(dev, cfg, dtype):
    """every compiled width, ragged tiles, per-image direction sets, frozen and trainable decoders"""
    from reni_amd.film import RENIAutoDecoderFiLM
    spec = O.FilmSpec(cfg["nd"], cfg["eq"], cfg["H"], cfg["nF"], 24, 2, 3, "tanh")
    gen = torch.Generator().manual_seed(5)
    params = O.film_init_params(spec, gen)
    B, P = cfg["B"], cfg["P"]
    Z = torch.randn(B, cfg["nd"], 3, generator=gen) * 0.5
    D = torch.nn.functional.normalize(torch.randn(B, P, 3, generator=gen), dim=-1)
    S = torch.rand(1, P, 3, generator=gen)
    T = torch.rand(B, P, 3, generator=gen) * 2 - 1
    ref = O.film_fwd_loss_bwd(spec, params, Z, D, T, S)
    tol = TOL[dtype]
    for fixed in (False, True):
        m = RENIAutoDecoderFiLM(B, cfg["nd"], cfg["eq"], cfg["H"], cfg["nF"], 24, 2, 3, "tanh", fixed)
        m.load_state_dict({"model." + k: v for k, v in params.items()}, strict=False)
        m.set_compute_dtype(dtype).to(dev)
        Zd = Z.to(dev).requires_grad_(True)
        terms = m.fused_loss(Zd, D.to(dev), T.to(dev), S.to(dev))
        terms[0].backward()
        assert abs(float(terms[0]) - ref["terms"][0]) <= tol["loss"] * abs(ref["terms"][0])
        assert O.rel_l2(Zd.grad.cpu().numpy(), ref["dZ"].numpy()) <= tol["grad"], ("dZ", fixed)
        got = _grads(m)
        if fixed:
            assert not got  # frozen decoder: only the latent receives a gradient
        else:
            for k, v in ref["grads"].items():
                assert O.rel_l2(got[k].numpy(), v.numpy()) <= tol["grad"], k


def test_film_full_size_properties(dev):
    """BASELINE-sized FiLM problem (128x256 grid, ND = 36, 5 x 128): additivity of the loss and of every gradient
    over a split of the batch, bf16 vs fp32 agreement, run-to-run determinism."""
    from reni_amd.film import RENIAutoDecoderFiLM
    torch.manual_seed(3)
    m = RENIAutoDecoderFiLM(4, 36, "SO2", 128, 5, 128, 3, 3, "tanh", False).to(dev)
    D = O.get_directions(256).to(dev)
    S = O.get_sineweight(256).to(dev)
    T = (torch.rand(4, D.shape[1], 3, generator=torch.Generator().manual_seed(9)) * 2 - 1).to(dev)
    idx = torch.arange(4, device=dev)

    def step(sel, dtype):
        m.set_compute_dtype(dtype)
        m.zero_grad(set_to_none=True)
        terms = m.fused_loss(m.Z[sel], D, T[sel], S)
        terms[0].backward()
        flat = torch.cat([p.grad.reshape(-1) for k, p in m.named_parameters() if k != "Z"])
        return float(terms[0]), flat.clone(), m.Z.grad.clone()

    la, ga, za = step(idx, "f32")
    lb, gb, zb = step(idx, "f32")
    assert la == lb and torch.equal(ga, gb) and torch.equal(za, zb)
    l0, g0, z0 = step(idx[:2], "f32")
    l1, g1, z1 = step(idx[2:], "f32")
    assert abs(l0 + l1 - la) <= 2e-6 * abs(la)
    assert O.rel_l2((g0 + g1).cpu().numpy(), ga.cpu().numpy()) <= 2e-5
    assert O.rel_l2((z0 + z1).cpu().numpy(), za.cpu().numpy()) <= 2e-5
    lh, gh, zh = step(idx, "bf16")
    assert abs(lh - la) <= 2e-3 * abs(la)
    assert O.rel_l2(gh.cpu().numpy(), ga.cpu().numpy()) <= 3e-2
    assert O.rel_l2(zh.cpu().numpy(), za.cpu().numpy()) <= 3e-2


@pytest.mark.parametrize("H,nF,persist", [(128, 3, True), (128, 3, False), (256, 2, False)])
def test_film_stream_runs_many_small_images(dev, H, nF, persist, monkeypatch):
    """Image-run bookkeeping of both bf16 FiLM paths (the persistent kernels at H = 128; the operand-stream path, which
    RENI_NO_PERSIST selects at H = 128 and H = 256 always uses): more one-tile images than workgroups, so a workgroup's
    range covers several images (several runs per workgroup) -- gradients of every parameter and of every latent
    against the oracle."""
    from reni_amd.film import RENIAutoDecoderFiLM
    if not persist:
        monkeypatch.setenv("RENI_NO_PERSIST", "1")
    B, P, nd = 700, 96, 4
    spec = O.FilmSpec(nd, "SO2", H, nF, 16, 1, 3, "tanh")
    gen = torch.Generator().manual_seed(11)
    params = O.film_init_params(spec, gen)
    Z = torch.randn(B, nd, 3, generator=gen) * 0.5
    D = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=gen), dim=-1)
    S = torch.rand(1, P, 3, generator=gen)
    T = torch.rand(B, P, 3, generator=gen) * 2 - 1
    ref = O.film_fwd_loss_bwd(spec, params, Z, D, T, S)
    m = RENIAutoDecoderFiLM(B, nd, "SO2", H, nF, 16, 1, 3, "tanh", False)
    m.load_state_dict({"model." + k: v for k, v in params.items()}, strict=False)
    m.set_compute_dtype("bf16").to(dev)
    Zd = Z.to(dev).requires_grad_(True)
    terms = m.fused_loss(Zd, D.to(dev), T.to(dev), S.to(dev))
    terms[0].backward()
    assert abs(float(terms[0].detach()) - ref["terms"][0]) <= 2e-3 * abs(ref["terms"][0])
    assert O.rel_l2(Zd.grad.cpu().numpy(), ref["dZ"].numpy()) <= 3e-2
    got = _grads(m)
    for k, v in ref["grads"].items():
        assert O.rel_l2(got[k].numpy(), v.numpy()) <= 3e-2, k


def test_film_core_entry_points_with_caller_owned_glue(dev):
    """reni_film_forward / reni_film_forward_loss_backward (the per-sample core, for callers that own the per-image
    glue): A, film from the torch restatement (film.py::_glue), dA / dfilm carried back by torch autograd -- the latent
    gradient and the mapping network's gradients must equal the reference's (G11)."""
    g = load_golden("g11_film_so2_ad.npz")
    m, spec = _model_from_golden(g, dev)
    W = int(g["W"])
    D = O.get_directions(W).to(dev)
    S = O.get_sineweight(W).to(dev)
    T = torch.from_numpy(g["target"]).to(dev)
    Z = torch.from_numpy(g["Z"]).to(dev).requires_grad_(True)
    A, film = m._glue(Z)
    plan, flat = m._plan(), m._flat_params()
    out = plan.film_forward(A.detach(), film.detach(), D, flat)
    assert float((out.cpu() - torch.from_numpy(g["out"])).abs().max()) <= 1e-5
    terms, dA, dfilm, dparams, _ = plan.film_forward_loss_backward(A.detach(), film.detach(), D, flat, T, S)
    assert abs(float(terms[0]) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    torch.autograd.backward([A, film], [dA, dfilm])
    assert O.rel_l2(Z.grad.cpu().numpy(), g["dZ"]) <= 2e-5
    for k, p in m.mapping_network.named_parameters():
        assert O.rel_l2(p.grad.cpu().numpy(), g["g.mapping_network." + k]) <= 2e-5, k
    # hidden / head slots of the flat gradient come from the kernels; the first layer's slot is the caller's (zero)
    n_first = m.net[0].layer.weight.numel() + m.net[0].layer.bias.numel()
    assert float(dparams[:n_first].abs().max()) == 0.0
    got = dparams[n_first:n_first + m.net[1].layer.weight.numel()].view_as(m.net[1].layer.weight)
    assert O.rel_l2(got.cpu().numpy(), g["g.net.1.layer.weight"]) <= 2e-5


@pytest.mark.parametrize("fixed", [False, True])
def test_film_persistent_runs_cut_inside_workgroup_ranges(dev, fixed, monkeypatch):
    """FiLM on the persistent kernels (k_reni_train_bf16<.., FILM>, k_reni_dw1<.., FILM>): 5 images of 129 tiles on 256
    workgroups, so contiguous tile ranges of 2-3 tiles start and end in the middle of images and several cross an image
    boundary (the accumulators leave the registers there).  Compared with the fp32 kernels (oracle-checked above) and
    with the operand-stream path on the same inputs; bit-identical from run to run."""
    from reni_amd.film import RENIAutoDecoderFiLM
    B, P, nd = 5, 128 * 128 + 77, 6
    torch.manual_seed(17)
    m = RENIAutoDecoderFiLM(B, nd, "SO2", 128, 4, 32, 2, 3, "tanh", fixed).to(dev)
    with torch.no_grad():
        m.Z.normal_(0, 0.5)
    gen = torch.Generator().manual_seed(23)
    D = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=gen), dim=-1).to(dev)
    S = (torch.rand(1, P, 3, generator=gen) + 0.1).to(dev)
    T = (torch.rand(B, P, 3, generator=gen) * 2 - 1).to(dev)

    def step(dtype):
        m.set_compute_dtype(dtype)
        m.zero_grad(set_to_none=True)
        Zd = m.Z.detach().clone().requires_grad_(True)
        terms = m.fused_loss(Zd, D, T, S)
        terms[0].backward()
        g = [p.grad.reshape(-1) for k, p in m.named_parameters() if k != "Z" and p.grad is not None]
        return float(terms[0].detach()), Zd.grad.clone(), (torch.cat(g).clone() if g else None)

    l32, z32, g32 = step("f32")
    lp, zp, gp = step("bf16")
    lq, zq, gq = step("bf16")
    assert lp == lq and torch.equal(zp, zq) and (gp is None or torch.equal(gp, gq))
    monkeypatch.setenv("RENI_NO_PERSIST", "1")
    m._plans.clear()  # (the selector is read when a plan is created)
    ls, zs, gs = step("bf16")
    for l_, z_, g_ in ((lp, zp, gp), (ls, zs, gs)):
        assert abs(l_ - l32) <= 3e-3 * abs(l32)
        assert O.rel_l2(z_.cpu().numpy(), z32.cpu().numpy()) <= 3e-2
        if not fixed:
            assert O.rel_l2(g_.cpu().numpy(), g32.cpu().numpy()) <= 3e-2
    assert (gp is None) == fixed


def test_film_train_engine_step_equals_autograd_plus_adam(dev):
    """TrainEngine on a FiLM model (one library call + fused Adam over the flat [net | final_layer | mapping_network] buffer
    and the latent table) == the same model stepped through fused_loss -> autograd -> torch.optim.Adam over every
    parameter (RENI_module.py:168-192: Adam over all parameters, dense over the latent table)."""
    from reni_amd.engine import TrainEngine
    from reni_amd.film import RENIAutoDecoderFiLM
    N, P, nd = 6, 300, 5
    gen = torch.Generator().manual_seed(31)
    D = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=gen), dim=-1).to(dev)
    S = (torch.rand(1, P, 3, generator=gen) + 0.1).to(dev)
    T = (torch.rand(N, P, 3, generator=gen) * 2 - 1).to(dev)
    models = []
    for _ in range(2):
        torch.manual_seed(7)
        models.append(RENIAutoDecoderFiLM(N, nd, "SO2", 64, 3, 16, 2, 3, "tanh", False).set_compute_dtype("f32").to(dev))
    a, b = models
    eng = TrainEngine(a, lr=1e-2)
    opt = torch.optim.Adam(b.parameters(), lr=1e-2)
    batches = [torch.tensor([0, 3, 4], device=dev), torch.tensor([5, 1, 1], device=dev)]  # (a repeated row accumulates)
    for idx in batches:
        ta = eng.step(idx, T[idx], S, D)
        opt.zero_grad(set_to_none=True)
        tb = b.fused_loss(b.Z[idx], D, T[idx], S)
        tb[0].backward()
        opt.step()
        assert abs(float(ta[0]) - float(tb[0].detach())) <= 1e-6 * abs(float(tb[0].detach()))
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb)
    for k in sa:
        assert float((sa[k] - sb[k]).abs().max()) <= 2e-6 + 1e-4 * float(sb[k].abs().max()) * 1e-2, k


@pytest.mark.parametrize("B,W,nF", [(4, 256, 5), (3, 64, 5), (5, 32, 3), (2, 128, 6), (3, 16, 5)])
def test_film_forward_at_the_shipped_width_on_the_wide_kernel(dev, monkeypatch, B, W, nF):
    """Round 6 (second half): the forward pass of the reference's DEFAULT model at its shipped width -- FiLM, 256
    features (configs/default.py:9,13; RENI.py:508-519, 565-586) -- runs on k_reni_wide256<0, FILM>: per-image (freq, phase) tables in
    LDS, several images per call (the tables change inside a workgroup's walk), odd pair counts, 2..5 hidden FiLM layers.  Against the
    generic kernel (RENI_NO_PERSIST: other bits, same arithmetic class) and the fp32 kernels, whose parity with the reference is pinned by
    G11: the two bf16 kernels must be as close to fp32 as each other."""
    from reni_amd.film import RENIAutoDecoderFiLM
    from reni_amd.utils import get_directions
    D = get_directions(W).to(dev)
    outs = {}
    for name, env, dtype in (("wide", None, "bf16"), ("generic", "1", "bf16"), ("f32", None, "f32")):
        if env:
            monkeypatch.setenv("RENI_NO_PERSIST", env)
        else:
            monkeypatch.delenv("RENI_NO_PERSIST", raising=False)
        torch.manual_seed(3)
        m = RENIAutoDecoderFiLM(B, 49, "SO2", 256, nF, 256, 3, 3, "tanh", True)
        with torch.no_grad():
            m.Z.normal_(generator=torch.Generator().manual_seed(4))
        m.set_compute_dtype(dtype).to(dev)
        with torch.no_grad():
            outs[name] = m(torch.arange(B, device=dev), D).float().cpu()
    monkeypatch.delenv("RENI_NO_PERSIST", raising=False)
    assert torch.isfinite(outs["wide"]).all()
    assert not torch.equal(outs["wide"], outs["generic"]), "the persistent instance did not run (bit-identical to RENI_NO_PERSIST)"
    e_w = float((outs["wide"] - outs["f32"]).abs().max()); e_g = float((outs["generic"] - outs["f32"]).abs().max())
    r_w = float((outs["wide"] - outs["f32"]).pow(2).mean().sqrt()); r_g = float((outs["generic"] - outs["f32"]).pow(2).mean().sqrt())
    assert e_w <= 5e-3 and e_w <= 1.5 * e_g + 2e-4 and r_w <= 1.2 * r_g + 2e-5, (e_w, e_g, r_w, r_g)
    assert float((outs["wide"] - outs["generic"]).abs().max()) <= 2.5e-3


@pytest.mark.parametrize("fixed", [False, True])
@pytest.mark.parametrize("B,W,nF", [(3, 64, 5), (5, 32, 3), (2, 32, 6), (3, 16, 4), (5, 16, 5)])
def test_film_backward_at_the_shipped_width_on_the_wide_chain(dev, monkeypatch, B, W, nF, fixed):
    """Round 6: every FiLM backward call at 256 features with up to four hidden FiLM layers -- the reference's default model
    (configs/default.py:9,13), trainable or frozen decoder -- runs its chain on k_reni_wide256<2, FILM> in front of the fragment stream's
    consumers (k_dw_frag<256, true>: weight gradients, d(freq), d(phase); k_wide_head_dw).  Loss, dZ and every parameter's gradient
    (SIREN, head, mapping network) against the fp32 kernels (pinned to the reference by G11 and the oracle tests) and against the generic
    bf16 chain (RENI_NO_PERSIST: other bits -- proof that the wide form ran -- and the same error class).  nF = 6 (five hidden layers:
    the tables do not fit) stays on the generic chain: there the two runs must be bit-identical.  W = 16: one 128-sample tile per image
    and an ODD number of tiles -- the last pair's second group has no tile of its own (its unconditional stream stores must carry its
    neighbour's words: the first version of the instance wrote zeros over them, found by tests/test_gpu_fuzz.py at 600 cases)."""
    from reni_amd.film import RENIAutoDecoderFiLM
    from reni_amd.utils import get_directions, get_sineweight
    D, S = get_directions(W).to(dev), get_sineweight(W).to(dev)
    T = (torch.rand(B, D.shape[1], 3, generator=torch.Generator().manual_seed(11)) * 2 - 1).to(dev)
    res = {}
    for name, env, dtype in (("wide", None, "bf16"), ("generic", "1", "bf16"), ("f32", None, "f32")):
        if env:
            monkeypatch.setenv("RENI_NO_PERSIST", env)
        else:
            monkeypatch.delenv("RENI_NO_PERSIST", raising=False)
        torch.manual_seed(3)
        m = RENIAutoDecoderFiLM(B, 36, "SO2", 256, nF, 64, 2, 3, "tanh", fixed)
        with torch.no_grad():
            m.Z.normal_(generator=torch.Generator().manual_seed(4))
            m.Z.mul_(0.5)
        m.set_compute_dtype(dtype).to(dev)
        Zd = m.Z.detach().clone().requires_grad_(True)
        terms = m.fused_loss(Zd, D, T, S)
        terms[0].backward()
        res[name] = (float(terms[0]), Zd.grad.detach().cpu(), _grads(m))
    monkeypatch.delenv("RENI_NO_PERSIST", raising=False)
    lw, zw, gw = res["wide"]; lg, zg, gg = res["generic"]; lf, zf, gf = res["f32"]
    if nF - 1 <= 4:
        assert not torch.equal(zw, zg), "the wide chain did not run (bit-identical to RENI_NO_PERSIST)"
    else:
        assert torch.equal(zw, zg) and lw == lg
    assert abs(lw - lf) <= 5e-3 * abs(lf) and torch.isfinite(zw).all()
    e_w, e_g = O.rel_l2(zw.numpy(), zf.numpy()), O.rel_l2(zg.numpy(), zf.numpy())
    assert e_w <= 3e-2 and e_w <= 1.5 * e_g + 2e-3, ("dZ", e_w, e_g)
    assert bool(gw) == (not fixed)
    for k, v in gf.items():
        ew, eg = O.rel_l2(gw[k].numpy(), v.numpy()), O.rel_l2(gg[k].numpy(), v.numpy())
        assert ew <= 5e-2 and ew <= 1.5 * eg + 5e-3, (k, ew, eg)
