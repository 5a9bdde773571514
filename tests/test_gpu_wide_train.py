"""Round 5: the training form of k_reni_wide256 (MODE 2) -- the H = 256 bf16 chain of a call WITH weight gradients feeds k_dw_frag's
fragment stream in place of k_reni_main's FRAG form, and k_wide_head_dw forms the head's gradient from the g_y stream
(reference: src/models/RENI.py:86-87,132-178 and the autograd backward through every layer, at the width configs/default.py:13 ships)."""
import os

import pytest
import torch

from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem, unflatten

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda:0")


def _generic_plan(spec, dtype):
    os.environ["RENI_NO_PERSIST"] = "1"  # (read once, at plan creation)
    try:
        return make_plan(spec, dtype)
    finally:
        del os.environ["RENI_NO_PERSIST"]


@pytest.mark.parametrize("B,gw,P,eq,act,loss", [(3, 64, 0, "SO2", "tanh", "mse"), (2, 32, 0, "SO2", "exp", "test"), (5, None, 200, "None", "tanh", "mse"),
                                                (300, None, 128, "SO2", "tanh", "mse")])
def test_wide_training_form_against_the_oracle_and_the_generic_chain(dev, B, gw, P, eq, act, loss):
    """Loss, dZ and every dW / db against the float64 oracle at the bf16 tolerance and against the generic kernels (RENI_NO_PERSIST)
    at a fraction of it; run-to-run bit-equal.  Shapes: whole tiles, ragged images (P = 200: a partly filled last tile), more tiles
    than workgroups; the cosine-statistics loss (its forward statistics pass is MODE 0 of the same kernel)."""
    spec = O.DecoderSpec(9, eq, 256, 5, 3, True, act)
    params, Z, D, W, T = random_problem(spec, B, P, seed=11 + B, grid_w=gw, per_image_dirs=gw is None)
    if act == "exp":
        T = T.abs()
    plan = make_plan(spec, "bf16")
    P = D.shape[1]
    info = plan.path_info(B, P)
    assert info["persistent_kernels"] and info["fragment_stream"] == "bf16" and info["dw1_kernel"] == "none", info
    old = _generic_plan(spec, "bf16")
    oinfo = old.path_info(B, P)
    assert not oinfo["persistent_kernels"] and "RENI_NO_PERSIST" in oinfo["env_overrides"], oinfo
    fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
    kw = dict(loss_kind=loss, alpha=1e-3 if loss == "test" else 0.0, beta=1e-2 if loss == "test" else 0.0)
    lt, dZ, dp, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, **kw)
    lt, dZ, dp = lt.clone(), dZ.clone(), dp.clone()
    lt2, dZ2, dp2, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, **kw)
    assert torch.equal(dZ, dZ2) and torch.equal(dp, dp2) and torch.equal(lt, lt2)
    lo, dZo, dpo, _ = old.forward_loss_backward(Zd, Dd, fp, Td, Wd, **kw)
    assert abs(float(lt[0]) - float(lo[0])) <= 2e-3 * abs(float(lo[0]))
    assert O.rel_l2(dZ.cpu().numpy(), dZo.cpu().numpy()) <= 2e-2
    assert O.rel_l2(dp.cpu().numpy(), dpo.cpu().numpy()) <= 2e-2
    ref = O.factored_fwd_bwd(spec, {k: v.numpy() for k, v in params.items()}, Z.numpy(), D.numpy(), T.numpy(), W.numpy(),
                             loss_kind=loss, alpha=kw["alpha"], beta=kw["beta"])
    assert abs(float(lt[0]) - ref["loss_terms"][0]) <= 3e-3 * abs(ref["loss_terms"][0])
    assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"]) <= 3e-2
    gp, go = unflatten(spec, dp.cpu()), unflatten(spec, dpo.cpu())
    for k in gp:
        e, eo = O.rel_l2(gp[k].numpy(), ref["grads"][k]), O.rel_l2(go[k].numpy(), ref["grads"][k])
        assert e <= 3e-2, (k, e, eo)


def test_wide_training_form_in_image_chunks(dev):
    """A workspace cap that forces image chunks (run_backward_chunked): the chunked sum equals the one-pass gradients to fp32 rounding."""
    spec = O.DecoderSpec(9, "SO2", 256, 5, 3, True, "tanh")
    B = 6
    params, Z, D, W, T = random_problem(spec, B, 0, seed=3, grid_w=64)
    P = D.shape[1]
    plan = make_plan(spec, "bf16")
    os.environ["RENI_FRAG_WS_CAP_MB"] = "16"
    try:
        small = make_plan(spec, "bf16")
    finally:
        del os.environ["RENI_FRAG_WS_CAP_MB"]
    assert small.path_info(B, P)["images_per_chunk"] < B == plan.path_info(B, P)["images_per_chunk"]
    fp = flat_params(spec, params).to(dev)
    a = [t.clone() for t in plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev))[:3]]
    b = [t.clone() for t in small.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev))[:3]]
    assert abs(float(a[0][0]) - float(b[0][0])) <= 1e-5 * abs(float(a[0][0]))
    assert torch.equal(a[1], b[1])  # an image's dZ does not depend on its neighbours
    assert O.rel_l2(b[2].cpu().numpy(), a[2].cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("B,P", [(1, 128), (3, 384), (1, 100)])
def test_wide_kernel_with_an_odd_number_of_tiles(dev, B, P):
    """k_reni_wide256 works on two tiles per workgroup: with an odd tile count (one tile; nine; one ragged tile) the last pair's second
    group is idle -- forward, frozen-decoder and training forms against the oracle, and the image's result independent of the pairing
    (a batch of 2 B images gives every image the bits it has in a batch of B: its tiles then sit in other groups / pairs)."""
    spec = O.DecoderSpec(9, "SO2", 256, 5, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, 2 * B, P, seed=40 + P, per_image_dirs=True)
    plan = make_plan(spec, "bf16")
    fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
    pn = {k: v.numpy() for k, v in params.items()}
    ref = O.factored_fwd_bwd(spec, pn, Z[:B].numpy(), D[:B].numpy(), T[:B].numpy(), W.numpy())
    out = plan.forward(Zd[:B], Dd[:B], fp).clone()                                             # MODE 0
    assert (out.cpu().numpy() - ref["out"]).__abs__().max() <= 2e-2
    assert torch.equal(plan.forward(Zd, Dd, fp)[:B], out)
    for need_dw in (False, True):                                                              # MODE 1, MODE 2
        lt, dZ, dp, _ = plan.forward_loss_backward(Zd[:B], Dd[:B], fp, Td[:B], Wd, need_dw=need_dw)
        lt, dZ = lt.clone(), dZ.clone()
        dp = dp.clone() if need_dw else None
        assert abs(float(lt[0]) - ref["loss_terms"][0]) <= 3e-3 * abs(ref["loss_terms"][0])
        assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"]) <= 3e-2
        if need_dw:
            gp = unflatten(spec, dp.cpu())
            for k in gp:
                assert O.rel_l2(gp[k].numpy(), ref["grads"][k]) <= 3e-2, k
        _, dZ2, _, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, need_dw=need_dw)
        assert torch.equal(dZ2[:B], dZ)   # an image's gradient does not depend on which group / pair its tiles land in
