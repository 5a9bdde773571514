"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against the CPU oracle
and the golden vectors generated from the reference.

Tolerances:
  fp32 kernels : output <= 1e-5 abs, loss <= 1e-6 rel (2e-6 where sums are long), grads <= 1e-5 rel-L2
  bf16 kernels : output <= 5e-3 abs, grads <= 3e-2 rel-L2   (the oracle's own bf16 run: 1.5e-3 / 1.8e-2)
"""
import numpy as np
import pytest
import torch

from oracle import reni_oracle as O
from tests.util import flat_params, load_golden, make_plan, random_problem, sd_from, unflatten

pytestmark = pytest.mark.gpu

TOL = {"f32": dict(out=1e-5, loss=2e-6, grad=1e-5), "bf16": dict(out=5e-3, loss=2e-3, grad=3e-2)}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _check(plan, spec, params, Z, D, W, T, dev, dtype, loss_kind="mse", alpha=0.0, beta=0.0, ref=None):
    B, P = Z.shape[0], D.shape[1]
    if ref is None:
        ref = O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, W.expand(B, P, 3), loss_kind, alpha, beta)
    fp = flat_params(spec, params).to(dev)
    lt, dZ, dp, out = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev), loss_kind=loss_kind,
                                                 alpha=alpha, beta=beta, want_out=True)
    tol = TOL[dtype]
    assert float((out.cpu() - ref["out"]).abs().max()) <= tol["out"]
    lt = lt.cpu().numpy()
    for i in range(4):
        assert abs(lt[i] - ref["loss_terms"][i]) <= tol["loss"] * max(abs(ref["loss_terms"][0]), 1e-30), (i, lt, ref["loss_terms"])
    assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()) <= tol["grad"]
    gp = unflatten(spec, dp.cpu())
    for k in gp:
        assert O.rel_l2(gp[k].numpy(), ref["grads"][k].numpy()) <= tol["grad"], k
    return ref


def test_mfma_layout_probes():
    from reni_amd import ops
    assert ops.selftest_layouts() == [0, 0]


def test_c1_forward_golden_through_module_api(dev):
    """BASELINE config 1 (32x64, ND=9, 3x64, no output activation) against the reference's output."""
    from reni_amd.models import RENIAutoDecoder
    g = load_golden("g3_c1_forward.npz")
    m = RENIAutoDecoder(1, 9, "SO2", 64, 3, 3, True, None, 30, 30, False)
    m.load_state_dict({"model." + k: v for k, v in sd_from(g).items()})
    m.to(dev)
    D = O.get_directions(64).to(dev)
    with torch.no_grad():
        out = m(0, D)
    assert out.shape == (1, 2048, 3)
    assert float((out.cpu() - torch.from_numpy(g["out"])).abs().max()) <= 1e-5


@pytest.mark.parametrize("name,eq,lll,act", [
    ("g4_small.npz", "SO2", True, "tanh"), ("g4_small_so3.npz", "SO3", True, "tanh"),
    ("g4_small_none.npz", "None", True, None), ("g4_small_sinehead.npz", "SO2", False, None)])
def test_g4_small_golden_fwd_bwd_f32(dev, name, eq, lll, act):
    g = load_golden(name)
    sd = sd_from(g)
    params = {k: v for k, v in sd.items() if k != "Z"}
    spec = O.DecoderSpec(9, eq, 64, 3, 3, lll, act)
    W = int(g["W"])
    ref = {"out": torch.from_numpy(g["out"]), "loss_terms": (float(g["loss"]), float(g["loss"]), 0.0, 0.0),
           "dZ": torch.from_numpy(g["dZ"]), "grads": {k: torch.from_numpy(g["g." + k]) for k in params}}
    _check(make_plan(spec, "f32"), spec, params, torch.from_numpy(g["Z"]), O.get_directions(W), O.get_sineweight(W),
           torch.from_numpy(g["target"]), dev, "f32", ref=ref)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_g4_c2_shape_golden(dev, dtype):
    """ND=36, H=128, L=5 (BASELINE config 2's architecture) at 32x64, against the reference."""
    g = load_golden("g4_c2shape.npz")
    sd = sd_from(g)
    params = {k: v for k, v in sd.items() if k != "Z"}
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    W = int(g["W"])
    plan = make_plan(spec, dtype)
    fp = flat_params(spec, params).to(dev)
    Z = torch.from_numpy(g["Z"]).to(dev)
    lt, dZ, dp, out = plan.forward_loss_backward(Z, O.get_directions(W).to(dev), fp, torch.from_numpy(g["target"]).to(dev),
                                                 O.get_sineweight(W).to(dev), want_out=True)
    tol = TOL[dtype]
    assert float((out.cpu()[:, :256] - torch.from_numpy(g["out_head"])).abs().max()) <= tol["out"]
    assert abs(float(lt[0]) - float(g["loss"])) <= tol["loss"] * float(g["loss"])
    assert O.rel_l2(dZ.cpu().numpy(), g["dZ"]) <= tol["grad"]
    gp = unflatten(spec, dp.cpu())
    for k in gp:
        n = float(np.linalg.norm(gp[k].numpy().astype(np.float64)))
        assert abs(n / float(g["gn." + k]) - 1) <= tol["grad"], k
        assert np.abs(gp[k].numpy().reshape(-1)[:32] - g["gh." + k]).max() <= 3 * tol["grad"] * float(g["gn." + k]) / np.sqrt(gp[k].numel()) + tol["grad"] * np.abs(g["gh." + k]).max(), k


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [
    dict(eq="SO2", nd=9, H=64, L=0), dict(eq="SO2", nd=36, H=128, L=5), dict(eq="SO3", nd=12, H=128, L=2),
    dict(eq="None", nd=5, H=32, L=2), dict(eq="SO2", nd=49, H=128, L=5, act="exp"),
    dict(eq="SO2", nd=36, H=256, L=5), dict(eq="SO2", nd=9, H=128, L=7), dict(eq="SO3", nd=9, H=256, L=1)])
def test_random_problems_vs_oracle(dev, dtype, cfg):
    """ragged P (not a multiple of the 128-sample tile), arbitrary (non-grid) directions."""
    spec = O.DecoderSpec(cfg["nd"], cfg["eq"], cfg["H"], cfg["L"], 3, True, cfg.get("act", "tanh"))
    params, Z, D, W, T = random_problem(spec, 3, 333, seed=7)
    if cfg.get("act") == "exp":  # keep exp() outputs O(1)
        params = {k: (v * 0.2 if k.endswith(f"{cfg['L'] + 1}.weight") else v) for k, v in params.items()}
    _check(make_plan(spec, dtype), spec, params, Z, D, W, T, dev, dtype)


def test_per_image_directions_and_single_sample(dev):
    spec = O.DecoderSpec(9, "SO2", 64, 2, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, 2, 130, seed=3, per_image_dirs=True)
    _check(make_plan(spec, "f32"), spec, params, Z, D, W, T, dev, "f32")
    params, Z, D, W, T = random_problem(spec, 1, 1, seed=4)
    _check(make_plan(spec, "f32"), spec, params, Z, D, W, T, dev, "f32")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_test_loss_with_mask_and_cosine(dev, dtype):
    """RENITestLoss (MSE + alpha |Z|^2 + beta cosine) with a masked weight; frozen decoder (dZ only)."""
    spec = O.DecoderSpec(9, "SO2", 64, 3, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, 3, 0, seed=5, grid_w=32)
    mask = (torch.rand(1, D.shape[1], 1, generator=torch.Generator().manual_seed(1)) > 0.6).float().expand(1, -1, 3)
    Wm = W * mask
    B, P = 3, D.shape[1]
    ref = O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, Wm.expand(B, P, 3), "test", 1e-3, 1e-1, need_dw=False)
    plan = make_plan(spec, dtype)
    fp = flat_params(spec, params).to(dev)
    lt, dZ, dp, _ = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), Wm.to(dev), loss_kind="test",
                                               alpha=1e-3, beta=1e-1, need_dw=False)
    assert dp is None
    lt = lt.cpu().numpy()
    tol = TOL[dtype]
    for i in range(4):
        assert abs(lt[i] - ref["loss_terms"][i]) <= tol["loss"] * abs(ref["loss_terms"][0]), (i, lt, ref["loss_terms"])
    assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()) <= tol["grad"]
    # with gradients for the decoder as well
    _check(plan, spec, params, Z, D, Wm, T, dev, dtype, "test", 1e-3, 1e-1)


@pytest.mark.parametrize("L", [1, 2, 4, 5])
def test_cosine_loss_on_the_persistent_kernels(dev, L):
    """RENITestLoss with the cosine term at H = 128 in bf16: the forward-only statistics instance (odd and even layer
    counts take different weight-prefetch paths), then the frozen instance (dZ only) and the training instance."""
    spec = O.DecoderSpec(9, "SO2", 128, L, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, 3, 700, seed=50 + L)
    mask = (torch.rand(1, D.shape[1], 1, generator=torch.Generator().manual_seed(2)) > 0.5).float().expand(1, -1, 3)
    Wm = W * mask
    B, P = 3, D.shape[1]
    plan = make_plan(spec, "bf16")
    fp = flat_params(spec, params).to(dev)
    tol = TOL["bf16"]
    for need_dw in (False, True):
        ref = O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, Wm.expand(B, P, 3), "test", 1e-3, 1e-1, need_dw=need_dw)
        lt, dZ, dp, _ = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), Wm.to(dev), loss_kind="test",
                                                   alpha=1e-3, beta=1e-1, need_dw=need_dw)
        lt = lt.cpu().numpy()
        for i in range(4):
            assert abs(lt[i] - ref["loss_terms"][i]) <= tol["loss"] * abs(ref["loss_terms"][0]), (i, lt, ref["loss_terms"])
        assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()) <= tol["grad"]
        if need_dw:
            gp = unflatten(spec, dp.cpu())
            for k in gp:
                t = tol["grad"]
                if gp[k].numel() <= 3 or k == "net.%d.weight" % (L + 1):
                    # The head bias gradient is three sums of signed per-sample terms that largely cancel, so the bf16 error of
                    # the OUTPUTS shows in it magnified (the kernel sums the terms themselves in fp32).  Its bound is what the
                    # reference's own arithmetic gives on this problem when its linear layers run in bf16 (autocast), x 1.5.
                    # (Round 6: the head WEIGHT gradient -- 3 x 128 sums of the same signed terms times bf16 activations -- gets the
                    # same derived bound: at L = 4 it sits at 2.8-3.1e-2 depending on the rounding realisation of d loss / d y.)
                    with torch.autocast("cpu", dtype=torch.bfloat16):
                        rb = O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, Wm.expand(B, P, 3), "test", 1e-3, 1e-1)
                    # (capped: where the reference's own bf16 run is off by more than 10 %, 1.5 x that asserts nothing -- ADVICE r03)
                    t = min(max(t, 1.5 * O.rel_l2(rb["grads"][k].float().numpy(), ref["grads"][k].numpy())), 0.15)
                e = O.rel_l2(gp[k].numpy(), ref["grads"][k].numpy())
                assert e <= t, (k, e, t)


@pytest.mark.parametrize("need_dw", [False, True])
def test_generic_kernels_when_the_persistent_ones_are_disabled(dev, monkeypatch, need_dw):
    """H = 128, L <= 5 in bf16 is served by the persistent kernels (training, frozen, forward-only instances); the generic
    chain + operand-stream path behind them (RENI_NO_PERSIST=1; also what L > 5 and FiLM use) must stay in parity too."""
    monkeypatch.setenv("RENI_NO_PERSIST", "1")
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, 3, 333, seed=11)
    plan = make_plan(spec, "bf16")
    fp = flat_params(spec, params).to(dev)
    ref = O.fwd_loss_bwd(spec, params, Z, D.expand(3, -1, 3), T, W.expand(3, -1, 3), need_dw=need_dw)
    out = plan.forward(Z.to(dev), D.to(dev), fp)
    assert float((out.cpu() - ref["out"]).abs().max()) <= TOL["bf16"]["out"]
    lt, dZ, dp, _ = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev), need_dw=need_dw)
    assert abs(float(lt[0]) - ref["loss_terms"][0]) <= TOL["bf16"]["loss"] * abs(ref["loss_terms"][0])
    assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()) <= TOL["bf16"]["grad"]
    if need_dw:
        gp = unflatten(spec, dp.cpu())
        for k in gp:
            assert O.rel_l2(gp[k].numpy(), ref["grads"][k].numpy()) <= TOL["bf16"]["grad"], k


def test_strided_channel_planar_target(dev):
    """targets arrive as the permute+view of [B,3,H,W] images (RENI_module.py:83-84), uncopied."""
    spec = O.DecoderSpec(9, "SO2", 64, 1, 3, True, "tanh")
    params, Z, D, W, _ = random_problem(spec, 2, 0, seed=6, grid_w=32)
    imgs = torch.rand(2, 3, 16, 32, generator=torch.Generator().manual_seed(2)) * 2 - 1
    T = imgs.permute(0, 2, 3, 1).reshape(2, -1, 3)
    ref = O.fwd_loss_bwd(spec, params, Z, D.expand(2, -1, 3), T, W.expand(2, -1, 3))
    imgs_d = imgs.to(dev)
    Tv = imgs_d.permute(0, 2, 3, 1).view(2, -1, 3)
    assert not Tv.is_contiguous()
    plan = make_plan(spec, "f32")
    lt, dZ, dp, _ = plan.forward_loss_backward(Z.to(dev), D.to(dev), flat_params(spec, params).to(dev), Tv, W.to(dev))
    assert abs(float(lt[0]) - ref["loss_terms"][0]) <= 2e-6 * ref["loss_terms"][0]
    assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()) <= 1e-5


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_generic_autograd_backward(dev, dtype):
    """model(Z, D) used as an ordinary differentiable op with an arbitrary downstream loss."""
    from reni_amd.models import RENIAutoDecoder
    spec = O.DecoderSpec(9, "SO3", 64, 2, 3, True, "tanh")
    torch.manual_seed(11)
    m = RENIAutoDecoder(3, 9, "SO3", 64, 2, 3, True, "tanh", 30, 30, False).set_compute_dtype(dtype)
    params = {k: v.detach().clone() for k, v in m.net.state_dict().items()}
    params = {"net." + k: v for k, v in params.items()}
    Z0 = m.Z.detach().clone()
    D = torch.nn.functional.normalize(torch.randn(3, 77, 3), dim=-1)
    R = torch.randn(3, 77, 3)
    # oracle
    Zr = Z0.clone().requires_grad_(True)
    ps = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    (O.reni_forward(spec, ps, Zr, D) * R).sum().backward()
    # product
    m.to(dev)
    out = m(torch.tensor([0, 1, 2], device=dev), D.to(dev))
    (out * R.to(dev)).sum().backward()
    tol = TOL[dtype]["grad"]
    assert O.rel_l2(m.Z.grad.cpu().numpy(), Zr.grad.numpy()) <= tol
    for k, p in m.net.named_parameters():
        assert O.rel_l2(p.grad.cpu().numpy(), ps["net." + k].grad.numpy()) <= tol, k


def test_run_to_run_bit_identical(dev):
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, 4, 0, seed=8, grid_w=64)
    plan = make_plan(spec, "bf16")
    args = (Z.to(dev), D.to(dev), flat_params(spec, params).to(dev), T.to(dev), W.to(dev))
    a = plan.forward_loss_backward(*args, want_out=True)
    a = [x.clone() for x in a]
    b = plan.forward_loss_backward(*args, want_out=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_fused_adam_matches_torch(dev):
    from reni_amd.optim import FusedAdam
    torch.manual_seed(0)
    p0 = torch.randn(1000, device=dev)
    pa = torch.nn.Parameter(p0.clone()); pb = torch.nn.Parameter(p0.clone())
    oa = FusedAdam([pa], lr=1e-2); ob = torch.optim.Adam([pb], lr=1e-2)
    for s in range(5):
        g = torch.randn(1000, device=dev) * (10.0 ** (s - 2))
        pa.grad = g.clone(); pb.grad = g.clone()
        oa.step(); ob.step()
    assert float((pa - pb).abs().max()) <= 1e-6


def test_wide_stream_path_more_tiles_than_workgroups(dev):
    """H = 256 bf16 (operand stream + k_dw_stream): 300 one-tile images on 256 CUs -- contiguous record ranges of
    unequal length, ragged tiles -- every gradient against the oracle, and run-to-run bit-equality."""
    spec = O.DecoderSpec(4, "SO3", 256, 2, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, 300, 100, seed=21)
    plan = make_plan(spec, "bf16")
    _check(plan, spec, params, Z, D, W, T, dev, "bf16")
    fp = flat_params(spec, params).to(dev)
    a = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev))
    b = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev))
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [dict(H=128, L=3, w0=20.0, wh=45.0), dict(H=128, L=4, w0=45.0, wh=12.0), dict(H=128, L=5, w0=7.5, wh=30.0),
                                 dict(H=256, L=3, w0=20.0, wh=45.0), dict(H=64, L=2, w0=12.0, wh=40.0)],
                         ids=lambda c: f"H{c['H']}-L{c['L']}-w{c['w0']:g}-{c['wh']:g}")
def test_unequal_omegas_vs_oracle(dev, dtype, cfg):
    """FIRST_OMEGA_0 != HIDDEN_OMEGA_0 (the reference's config keys, RENI.py:128-178; every shipped config sets both to 30, and so did every
    test until round 6).  On the persistent bf16 kernels the omegas enter through the packed images' scales, the constant on d loss / d y
    and the per-layer constants of the weight gradients (c_l = (omega_hidden / omega_first) (8 / 2 pi)^l): a
    training call (every parameter's gradient) and a frozen-decoder call (dZ), with the cosine term, against the fp64 oracle."""
    spec = O.DecoderSpec(9, "SO2", cfg["H"], cfg["L"], 3, True, "tanh", cfg["w0"], cfg["wh"])
    params, Z, D, W, T = random_problem(spec, 3, 0, seed=61, grid_w=64)
    plan = make_plan(spec, dtype)
    _check(plan, spec, params, Z, D, W, T, dev, dtype)
    B, P = 3, D.shape[1]
    ref = O.fwd_loss_bwd(spec, params, Z, D.expand(B, P, 3), T, W.expand(B, P, 3), "test", 1e-3, 1e-1, need_dw=False)
    lt, dZ, dp, _ = plan.forward_loss_backward(Z.to(dev), D.to(dev), flat_params(spec, params).to(dev), T.to(dev), W.to(dev), loss_kind="test",
                                               alpha=1e-3, beta=1e-1, need_dw=False)
    assert dp is None
    tol = TOL[dtype]
    assert abs(float(lt[0]) - ref["loss_terms"][0]) <= tol["loss"] * abs(ref["loss_terms"][0])
    assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()) <= tol["grad"]


def test_env_switch_zero_means_off(dev, monkeypatch):
    """RENI_NO_PERSIST=0 (and "") must leave the persistent kernels selected: a switch is on when set to anything BUT "" or "0"
    (reni_plan_create reads it once; reni_path_info reports what is in force)."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    seen = {}
    for val in (None, "0", "", "1"):
        if val is None:
            monkeypatch.delenv("RENI_NO_PERSIST", raising=False)
        else:
            monkeypatch.setenv("RENI_NO_PERSIST", val)
        info = make_plan(spec, "bf16").path_info(4, 8192)
        seen[val] = (info["persistent_kernels"], tuple(info["env_overrides"]))
    monkeypatch.delenv("RENI_NO_PERSIST", raising=False)
    assert seen[None] == seen["0"] == seen[""] and seen[None][0] and not seen["1"][0], seen
