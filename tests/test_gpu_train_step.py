"""reni_train_step_rows (the fused training step: fwd + loss + bwd, Adam over decoder + latent table, the next batch's prologue) against
the two calls it replaces (reni_forward_loss_backward_rows + reni_adam_step2): the SAME kernels on the same data in another order,
so every buffer must come out bit-equal -- parameters, latents, both Adam moments, the loss terms of every step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(dtype, H, L, N, fused, seed=0):
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    torch.manual_seed(seed)
    m = RENIAutoDecoder(N, 9, "SO2", H, L, 3, True, "tanh", 30.0, 30.0, False)
    m.set_compute_dtype(dtype).to("cuda:0")
    return m, TrainEngine(m, lr=1e-3, fused_step=fused)


def _data(N, W, dev):
    from reni_amd.utils import get_directions, get_sineweight
    D, S = get_directions(W).to(dev), get_sineweight(W).to(dev)
    T = torch.stack([torch.rand(D.shape[1], 3, generator=torch.Generator().manual_seed(300 + i)) * 2 - 1 for i in range(N)]).to(dev)
    return D, S, T


def _state(m, e):
    return [t.detach().clone() for t in (m._flat_params(), m.Z.data, e.m_dec, e.v_dec, e.m_lat, e.v_lat)]


@pytest.mark.parametrize("dtype,H,L,W,B,announce", [
    ("bf16", 128, 5, 256, 8, True),    # persistent kernels, 2 048 tiles = 8 per workgroup: forked (Adam + prologue beside k_reni_dw1)
    ("bf16", 128, 5, 64, 4, True),     # persistent kernels, 64 tiles: one stream
    ("bf16", 128, 3, 64, 4, False),    # nothing announced: every call runs its own prologue
    ("f32", 64, 3, 32, 4, True),       # generic kernels: the same work in sequence
    ("bf16", 256, 2, 32, 2, True),     # H = 256 (fragment stream)
])
def test_fused_step_is_bit_equal_to_the_two_calls(dtype, H, L, W, B, announce):
    dev = torch.device("cuda:0")
    N = 3 * B
    D, S, T = _data(N, W, dev)
    batches = [torch.arange(B, device=dev) + o for o in (0, B, 2 * B, B, 0, 2 * B)]
    batches[3] = torch.tensor(list(range(N - 1, N - 1 - B, -1)), device=dev)   # (not a contiguous run of rows)
    res = {}
    for fused in (False, True):
        m, e = _engine(dtype, H, L, N, fused)
        terms = []
        for k, idx in enumerate(batches):
            nxt = batches[k + 1] if (announce and k + 1 < len(batches)) else None
            terms.append(e.step(idx, T[idx], S, D, next_idx=nxt).clone())
        torch.cuda.synchronize()
        res[fused] = (torch.stack(terms), _state(m, e), e)
    assert res[True][2]._stage is not None and res[False][2]._stage is None      # the fused path really ran
    assert torch.equal(res[False][0], res[True][0]), (res[False][0], res[True][0])
    for a, b, name in zip(res[False][1], res[True][1], ("params", "Z", "m_dec", "v_dec", "m_lat", "v_lat")):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))


@pytest.mark.parametrize("dtype,H,L", [("bf16", 128, 5), ("f32", 64, 3), ("bf16", 256, 2), ("bf16", 128, 4)])
def test_a_batch_other_than_the_staged_one_skips_the_step_loudly(dtype, H, L):
    """The library compares the batch a staged prologue was run for with the batch the next call brings, on the device: a mismatch
    the host-side identity check cannot see (same tensor, other contents) yields a NaN loss and NaN latent gradients AND leaves
    parameters, latent table and Adam moments exactly as they were -- on every path (persistent + k_reni_l0_ring, persistent round-4
    kernels (even L), generic fp32, H = 256): the step is skipped, not applied with garbage."""
    dev = torch.device("cuda:0")
    B, N = 4, 12
    D, S, T = _data(N, 64, dev)
    m, e = _engine(dtype, H, L, N, True)
    idx = torch.arange(B, device=dev)
    nxt = torch.arange(B, device=dev) + B
    e.step(idx, T[idx], S, D, next_idx=nxt)
    before = _state(m, e)
    nxt[1] = 11                                  # the announced batch is changed in place behind the engine's back
    t = e.step(nxt, T[nxt], S, D)
    assert bool(torch.isnan(t).all())
    for a, b, name in zip(before, _state(m, e), ("params", "Z", "m_dec", "v_dec", "m_lat", "v_lat")):
        assert torch.equal(a, b), name           # nothing moved, nothing is NaN
    t3 = e.step(idx, T[idx], S, D)               # and the engine carries on (its own prologue: nothing was announced)
    assert bool(torch.isfinite(t3).all())
    # announcing nothing, or another tensor: the call simply runs its own prologue
    m2, e2 = _engine(dtype, H, L, N, True)
    e2.step(idx, T[idx], S, D, next_idx=nxt)
    other = torch.tensor([2, 5, 7, 9], device=dev)
    t2 = e2.step(other, T[other], S, D)
    assert bool(torch.isfinite(t2).all())


@pytest.mark.parametrize("dtype,H,L,W", [("bf16", 128, 5, 64), ("f32", 64, 3, 32)])
def test_another_call_on_the_plan_between_two_announced_steps_drops_the_stage(dtype, H, L, W):
    """The staged prologue lives in the plan's workspace: a validation forward at another batch size, a fused loss, anything that takes
    the workspace between two announced steps overwrites or abandons it.  The plan counts workspace hand-outs and train_step drops the
    stage when the count moved: the interleaved run is bit-equal to the un-fused engine's."""
    dev = torch.device("cuda:0")
    B = 4
    N = 3 * B
    D, S, T = _data(N, W, dev)
    batches = [torch.arange(B, device=dev) + o for o in (0, B, 2 * B, 0)]
    res = {}
    for fused in (False, True):
        m, e = _engine(dtype, H, L, N, fused)
        terms = []
        for k, idx in enumerate(batches):
            nxt = batches[k + 1] if k + 1 < len(batches) else None
            terms.append(e.step(idx, T[idx], S, D, next_idx=nxt).clone())
            with torch.no_grad():                # a validation forward of 7 images on the same plan, between the announced steps
                out = m(m.Z.data[:7], D.expand(7, -1, -1)) if k != 1 else m(m.Z.data[:2], D.expand(2, -1, -1))
            assert bool(torch.isfinite(out).all())
        torch.cuda.synchronize()
        res[fused] = (torch.stack(terms), _state(m, e))
    assert torch.equal(res[False][0], res[True][0]), (res[False][0], res[True][0])
    for a, b, name in zip(res[False][1], res[True][1], ("params", "Z", "m_dec", "v_dec", "m_lat", "v_lat")):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))


@pytest.mark.parametrize("dtype,H,L,W,sparse", [
    ("bf16", 128, 5, 128, False), ("bf16", 128, 5, 128, True), ("bf16", 128, 5, 128, "pixels"),
    ("f32", 64, 3, 64, True), ("bf16", 256, 2, 64, "pixels"),
])
def test_latent_step_is_bit_equal_to_the_two_calls(dtype, H, L, W, sparse):
    """reni_latent_step_rows (one FIT_LATENT iteration, frozen decoder: RENITestLoss gradient of the batch's latent rows, then the
    dense Adam step on the table) against the two calls it replaces, through TrainEngine (fused_step on / off): loss terms of every
    step, the table and both moments bit-equal -- with a masked weight and each RENI_WEIGHT_* mode."""
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    dev = torch.device("cuda:0")
    N, B = 6, 4
    D, S, T = _data(N, W, dev)
    mask = torch.zeros(W // 2, W, 1)
    mask[W // 8: W // 3, W // 4: (2 * W) // 3] = 1.0
    Wm = S * mask.reshape(1, -1, 1).to(dev)
    batches = [torch.arange(B, device=dev), torch.tensor([5, 1, 4, 2], device=dev), torch.arange(B, device=dev) + 2]
    res = {}
    for fused in (False, True):
        torch.manual_seed(3)
        m = RENIAutoDecoder(N, 9, "SO2", H, L, 3, True, "tanh", 30.0, 30.0, True)
        with torch.no_grad():
            m.Z.normal_(generator=torch.Generator().manual_seed(5))
        m.set_compute_dtype(dtype).to(dev)
        e = TrainEngine(m, lr=1e-1, loss_kind="test", alpha=1e-7, beta=1e-4, sparse_weight=sparse, fused_step=fused)
        terms = [e.step(idx, T[idx], Wm, D).clone() for idx in batches]
        torch.cuda.synchronize()
        res[fused] = (torch.stack(terms), [t.detach().clone() for t in (m.Z.data, e.m_lat, e.v_lat)])
    assert torch.equal(res[False][0], res[True][0]), (res[False][0], res[True][0])
    for a, b, name in zip(res[False][1], res[True][1], ("Z", "m_lat", "v_lat")):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    assert torch.isfinite(res[True][0]).all() and float(res[True][1][1].abs().max()) > 0


@pytest.mark.parametrize("sparse", [True, "pixels"])
def test_cached_weight_lists_follow_the_weight(sparse):
    """reni_weight_lists_build / reni_latent_step_rows_cached: the lists are built once per mask and reused -- and
    rebuilt when the weight tensor is modified in place (its version counter is part of the cache key) or replaced: every step bit-equal
    to the entry point that rebuilds them in every call, with fewer launches."""
    from reni_amd import ops
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    dev = torch.device("cuda:0")
    N, B, W = 6, 4, 128
    D, S, T = _data(N, W, dev)
    mask = torch.zeros(W // 2, W, 1)
    mask[W // 8: W // 3, W // 4: (2 * W) // 3] = 1.0
    Wm = (S * mask.reshape(1, -1, 1).to(dev)).clone()
    idx = torch.arange(B, device=dev)
    res, launches = {}, {}
    for cached in (False, True):
        torch.manual_seed(3)
        m = RENIAutoDecoder(N, 9, "SO2", 128, 5, 3, True, "tanh", 30.0, 30.0, True)
        with torch.no_grad():
            m.Z.normal_(generator=torch.Generator().manual_seed(5))
        m.set_compute_dtype("bf16").to(dev)
        e = TrainEngine(m, lr=1e-1, loss_kind="test", alpha=1e-7, beta=1e-4, sparse_weight=sparse)
        plan = m._plan()
        w = Wm.clone()
        terms = []
        for k in range(6):
            if k == 3:
                w[:, : w.shape[1] // 2] = 0.0      # the mask changes IN PLACE: the cached lists must follow
            if k == 5:
                w = (S * 1.0).expand(1, -1, 3).clone()   # another tensor, no mask at all
            ops.launch_count(reset=True)
            t, _ = plan.latent_step(m.Z.data, idx, D, m._flat_params(), T[idx], w, e.m_lat, e.v_lat, k + 1, 1e-1, loss_kind="test",
                                    alpha=1e-7, beta=1e-4, sparse_weight=sparse, cache_lists=cached)
            launches.setdefault(cached, []).append(ops.launch_count(reset=True))
            terms.append(t.clone())
        torch.cuda.synchronize()
        res[cached] = (torch.stack(terms), m.Z.data.clone())
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    n_build = 3 if sparse == "pixels" else 2
    # steps 1, 2, 4 reuse the lists (no list-building launch); steps 0, 3, 5 build them once (the same launches as the rebuilding call).
    # While pixel 0 is masked (steps 0..4) the cosine term is a constant and the build says so: the statistics instance and
    # k_stats_image -- two launches that would visit nothing -- are left out as well (RENI_WEIGHT_COS_CONSTANT).
    for k in (1, 2, 4):
        assert launches[True][k] == launches[False][k] - n_build - 2, (k, launches)
    for k in (0, 3):
        assert launches[True][k] == launches[False][k] - 2, (k, launches)
    assert launches[True][5] == launches[False][5], launches


def test_cached_lists_built_for_another_shape_are_refused():
    """the lists' layout is a function of (B, P); reni_latent_step_rows_cached checks -- on the host, against the library's
    record of what reni_weight_lists_build built where -- that the buffer it is handed was built for this call's B, P and mode, and
    refuses anything else before a kernel reads it."""
    import ctypes
    from reni_amd import _lib
    from reni_amd.models import RENIAutoDecoder
    dev = torch.device("cuda:0")
    N, W = 6, 64
    D, S, T = _data(N, W, dev)
    m = RENIAutoDecoder(N, 9, "SO2", 128, 5, 3, True, "tanh", 30.0, 30.0, True)
    m.set_compute_dtype("bf16").to(dev)
    plan = m._plan()
    P = D.shape[1]
    w4 = S.expand(4, P, 3).contiguous()
    lptr, _ = plan.weight_lists(4, P, w4, _lib.WEIGHT_SPARSE)
    ml, vl = torch.zeros_like(m.Z.data), torch.zeros_like(m.Z.data)

    def cached(B, mode, ptr):
        idx = torch.arange(B, device=dev)
        w = S.expand(B, P, 3).contiguous()
        ts = (ctypes.c_int64 * 3)(*T[idx].stride()); wst = (ctypes.c_int64 * 3)(*w.stride())
        lt = torch.empty(4, device=dev); dZ = torch.empty(B, 9, 3, device=dev)
        ws = plan.workspace(B, P, _lib.NEED_DZ | mode, dev)
        wp, wn = plan._aligned_ptr(ws)
        tgt = T[idx].contiguous()
        return plan.lib.reni_latent_step_rows_cached(plan._h, B, P, m.Z.data.data_ptr(), N, idx.data_ptr(), D.data_ptr(), 0, m._flat_params().data_ptr(),
                                                     tgt.data_ptr(), ts, w.data_ptr(), wst, _lib.LOSS_TEST, 1e-7, 1e-4, mode, ptr, ml.data_ptr(),
                                                     vl.data_ptr(), 0.1, 0.9, 0.999, 1e-8, 1, lt.data_ptr(), dZ.data_ptr(), wp, wn,
                                                     torch.cuda.current_stream(dev).cuda_stream)

    assert cached(4, _lib.WEIGHT_SPARSE, lptr) == 0
    assert cached(3, _lib.WEIGHT_SPARSE, lptr) == -1 and b"built for B=4" in plan.lib.reni_last_error()      # RENI_EINVAL: another B
    assert cached(4, _lib.WEIGHT_COMPACT, lptr) == -1 and b"mode" in plan.lib.reni_last_error()              # another mode
    assert cached(4, _lib.WEIGHT_SPARSE, lptr + 256) == -1 and b"no reni_weight_lists_build" in plan.lib.reni_last_error()
    torch.cuda.synchronize()
