"""Run-to-run bit-equality soak of the frozen-decoder persistent instances.

Round 4 saw ONE failure in seven runs of the GPU suite that never reproduced (open questions); round 5 found a real run-to-run
nondeterminism of exactly that class -- a transcendental's result read one instruction too early behind inline asm in k_reni_l0_ring.
tests/test_build_audit.py now audits that hazard class on every instance of every translation unit (tests/isa_audit.py: trans_to_valu);
this is the measurement beside the audit: the shapes of test_sparse_weight_leaves_out_only_exact_zeros -- block mask, no mask, all-zero
weight, one live pixel, block + pixel 0 (cosine term live) -- through the dense, RENI_WEIGHT_SPARSE and RENI_WEIGHT_COMPACT forms of the
bf16 frozen instance and its statistics pass, FIFTY times each, every result compared bit for bit with the first."""
import pytest
import torch

from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem

pytestmark = pytest.mark.gpu

ITERS = 50


def _problem(dev, hidden=128):
    spec = O.DecoderSpec(36, "SO2", hidden, 5, 3, True, "tanh")
    B = 5
    params, Z, D, W, T = random_problem(spec, B, 0, seed=91, grid_w=128)      # 64 x 128 = 8 192 directions = 64 tiles per image
    P = D.shape[1]
    m = torch.zeros(B, 64, 128, 1)
    m[0, 10:46, 40:83] = 1.0
    m[1] = 1.0
    m[3, 33, 77] = 1.0
    m[4, 10:46, 40:83] = 1.0
    m[4, 0, 0] = 1.0
    Wm = (W.view(1, 64, 128, 3) * m).reshape(B, P, 3)
    return spec, flat_params(spec, params).to(dev), Z.to(dev), D.to(dev), T.to(dev), Wm.to(dev)


@pytest.mark.parametrize("hidden", [128, 256])
def test_frozen_instances_are_bit_identical_over_fifty_runs(hidden):
    dev = torch.device("cuda:0")
    spec, fp, Z, D, T, Wm = _problem(dev, hidden)
    plan = make_plan(spec, "bf16")
    first, bad = {}, []
    for it in range(ITERS):
        for mode in (False, True, "pixels"):
            lt, dZ, _, _ = plan.forward_loss_backward(Z, D, fp, T, Wm, loss_kind="test", alpha=1e-7, beta=1e-4, need_dw=False, sparse_weight=mode)
            got = (lt.cpu(), dZ.cpu())
            if it == 0:
                first[mode] = got
                assert torch.isfinite(got[1]).all() and float(got[1].abs().max()) > 0
            elif not (torch.equal(got[0], first[mode][0]) and torch.equal(got[1], first[mode][1])):
                bad.append((it, mode, float((got[1] - first[mode][1]).abs().max())))
        # (something else on the chip between the repeats: the training form over the same data, whose kernels leave other LDS / register state)
        if it % 10 == 5:
            plan.forward_loss_backward(Z, D, fp, T, Wm[:, :, :], need_dw=True)
    assert not bad, bad[:10]
    # the sparse form leaves out exact zeros only: equal to the dense form, as in test_sparse_weight_leaves_out_only_exact_zeros
    assert torch.equal(first[True][0], first[False][0]) and torch.equal(first[True][1], first[False][1])


def test_training_step_l0x_is_bit_identical_over_fifty_runs():
    """The same for the training form (k_reni_train_bf16<128,true,L0X> + k_reni_l0_ring + the fused tails): 50 calls on one batch."""
    dev = torch.device("cuda:0")
    spec, fp, Z, D, T, Wm = _problem(dev)
    plan = make_plan(spec, "bf16")
    W1 = O.get_sineweight(128).to(dev)
    ref = None
    for it in range(ITERS):
        lt, dZ, dp, _ = plan.forward_loss_backward(Z, D, fp, T, W1, need_dw=True)
        got = (lt.cpu(), dZ.cpu(), dp.cpu())
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), (it, [float((a - b).abs().max()) for a, b in zip(got, ref)])
