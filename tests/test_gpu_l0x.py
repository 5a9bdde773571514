"""Round 5: the L0X split of the config-2 training path -- k_reni_train_bf16<.., SPEC, L0X> ends a tile with layer 2's step and
k_reni_l0_ring takes layer 1's dX GEMM, the first SineLayer's derivative, the layer-0 dA and dW_1 from the g_1 stream
(reference: the autograd backward of src/models/RENI.py:86-87,132-178 through net[0] and net[1])."""
import os

import pytest
import torch

from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem, unflatten

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda:0")


def _plan_without_l0x(spec, dtype):
    os.environ["RENI_NO_L0X"] = "1"  # (read once, at plan creation)
    try:
        return make_plan(spec, dtype)
    finally:
        del os.environ["RENI_NO_L0X"]


@pytest.mark.parametrize("B,gw,P", [(8, 256, 0), (3, 64, 0), (70, 32, 0), (300, None, 128)])
def test_l0x_split_against_the_oracle_and_round_4s_kernels(dev, B, gw, P):
    """Config-2 architecture (ND = 36, SO2, 5 x 128, tanh, WeightedMSE): loss, dZ and every dW / db of the L0X path against the float64
    oracle at the bf16 tolerance, and against the in-kernel layer-0 backward (RENI_NO_L0X) at a fraction of it -- the two differ only
    in the layer-0 cosine (fp32 here, from an fp16 phase there) and the order of the sums.  Shapes: many tiles per workgroup with ranges
    crossing images; fewer tiles than workgroups; four-tile images; one-tile images (an image run per tile, per-image directions)."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, B, P, seed=5 + B, grid_w=gw, per_image_dirs=gw is None)
    plan = make_plan(spec, "bf16")
    P = D.shape[1]
    info = plan.path_info(B, P)
    assert info["dw1_kernel"] == "k_reni_l0_ring", info
    old = _plan_without_l0x(spec, "bf16")
    oinfo = old.path_info(B, P)
    assert oinfo["dw1_kernel"] == "k_reni_dw1_ring" and "RENI_NO_L0X" in oinfo["env_overrides"], oinfo
    fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
    lt, dZ, dp, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
    lt, dZ, dp = lt.clone(), dZ.clone(), dp.clone()
    lt2, dZ2, dp2, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
    assert torch.equal(dZ, dZ2) and torch.equal(dp, dp2) and torch.equal(lt, lt2)  # run-to-run bit-equality
    lo, dZo, dpo, _ = old.forward_loss_backward(Zd, Dd, fp, Td, Wd)
    assert abs(float(lt[0]) - float(lo[0])) <= 1e-6 * abs(float(lo[0]))  # the forward pass is the same code
    assert O.rel_l2(dZ.cpu().numpy(), dZo.cpu().numpy()) <= 2e-3
    assert O.rel_l2(dp.cpu().numpy(), dpo.cpu().numpy()) <= 2e-3
    ref = O.factored_fwd_bwd(spec, {k: v.numpy() for k, v in params.items()}, Z.numpy(), D.numpy(), T.numpy(), W.numpy())
    assert abs(float(lt[0]) - ref["loss_terms"][0]) <= 2e-3 * abs(ref["loss_terms"][0])
    assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"]) <= 3e-2
    gp = unflatten(spec, dp.cpu())
    for k in gp:  # (the first layer -- through dA -- and layer 1 are the two pieces k_reni_l0_ring owns)
        assert O.rel_l2(gp[k].numpy(), ref["grads"][k]) <= 3e-2, (k, O.rel_l2(gp[k].numpy(), ref["grads"][k]))


def test_l0x_rows_by_lds_dma_in_both_layouts(dev):
    """The L0X instance fetches a tile's target / weight rows by LDS-DMA: [P][3] interleaved rows and the channel-planar view
    RENI.training_step passes (`imgs.permute(0, 2, 3, 1).view(B, -1, 3)` of a [B, 3, H, W] batch, RENI_module.py:83-84) give the
    same bits; a layout it cannot fetch (rows not 16-byte aligned) takes round 4's kernels and agrees to the bf16 tolerance."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    B = 5
    params, Z, D, W, T = random_problem(spec, B, 0, seed=77, grid_w=128)
    plan = make_plan(spec, "bf16")
    fp = flat_params(spec, params).to(dev)
    Zd, Dd = Z.to(dev), D.to(dev)
    Ti, Wi = T.to(dev), W.expand(B, -1, 3).contiguous().to(dev)                 # interleaved [B, P, 3]
    Tp = Ti.permute(0, 2, 1).contiguous().permute(0, 2, 1)                       # planar: strides (3 P, 1, P)
    Wp = Wi.permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert Tp.stride() == (3 * T.shape[1], 1, T.shape[1]) and torch.equal(Tp, Ti)
    r_i = [t.clone() for t in plan.forward_loss_backward(Zd, Dd, fp, Ti, Wi)[:3]]
    r_p = [t.clone() for t in plan.forward_loss_backward(Zd, Dd, fp, Tp, Wp)[:3]]
    r_m = [t.clone() for t in plan.forward_loss_backward(Zd, Dd, fp, Tp, Wi)[:3]]
    for a, b, c in zip(r_i, r_p, r_m):
        assert torch.equal(a, b) and torch.equal(a, c)
    # an unaligned view: one float of padding in front of every image
    pad = torch.zeros(B, T.shape[1] * 3 + 1, device=dev)
    pad[:, 1:] = Ti.reshape(B, -1)
    Tu = pad[:, 1:].view(B, -1, 3)
    assert Tu.data_ptr() % 16 != 0
    r_u = plan.forward_loss_backward(Zd, Dd, fp, Tu, Wi)[:3]
    assert abs(float(r_u[0][0]) - float(r_i[0][0])) <= 1e-6 * abs(float(r_i[0][0]))
    assert O.rel_l2(r_u[1].cpu().numpy(), r_i[1].cpu().numpy()) <= 2e-3 and O.rel_l2(r_u[2].cpu().numpy(), r_i[2].cpu().numpy()) <= 2e-3


@pytest.mark.parametrize("H,eq", [(128, "SO2"), (256, "SO2")])
@pytest.mark.parametrize("poison", [0xFF, 0x7F])
def test_training_paths_with_a_poisoned_workspace(dev, H, eq, poison):
    """Every byte of the plan's workspace overwritten (0xFF: NaN floats, -1 integers; 0x7F: NaN floats, huge integers) before the
    call: the training step's kernels -- the L0X instance, k_reni_l0_ring with its image-run records and g_1 stream, the fused tails
    (H = 128); k_reni_wide256's training and frozen forms with their per-tile stash, fragment and g_y streams, k_dw_frag,
    k_wide_head_dw (H = 256) -- must give the clean run's bits.  A reader of a word the call did not write fails here deterministically.
    Shapes: ranges crossing images with a ragged tail of tiles (B = 5 images of 32 tiles over 256 workgroups -> one tile per workgroup,
    fewer tiles than workgroups), and many tiles per workgroup."""
    spec = O.DecoderSpec(36, eq, H, 5, 3, True, "tanh")
    for B, gw in ((5, 64), (3, 256) if H == 128 else (2, 128)):
        params, Z, D, W, T = random_problem(spec, B, 0, seed=3 + B, grid_w=gw)
        plan = make_plan(spec, "bf16")
        fp = flat_params(spec, params).to(dev)
        Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)

        def run(dirty, need_dw):
            if dirty:
                for ws in plan._ws.values():
                    ws.fill_(poison)
            lt, dZ, dp, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, need_dw=need_dw)
            return lt.cpu(), dZ.cpu(), (dp.cpu() if need_dw else None)

        for need_dw in (True, False):
            ref = run(False, need_dw)
            assert plan._ws, "the plan keeps its workspace"
            got = run(True, need_dw)
            assert torch.isfinite(got[0]).all() and torch.isfinite(got[1]).all()
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (B, need_dw)
            if need_dw:
                assert torch.isfinite(got[2]).all() and torch.equal(got[2], ref[2]), (B, float((got[2] - ref[2]).abs().max()))
