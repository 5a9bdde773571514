"""SURVEY section 8(e) on the GPU: two ranks (gloo, both on cuda:0 -- RCCL needs a GPU per rank, the exchange step is
the same all-reduce) each run TrainEngine.step on their own images; the decoder they end up with equals the one a single
process gets from the union batch with every gradient scaled 1/W (the reference's DDP mean, run.py:97)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_IMG, B_RANK, STEPS, LR = 4, 2, 2, 1e-2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(n_rows, rows, dev):
    """Same decoder on every rank (seed 0); latent row i of the full table = generator(100 + i)."""
    from reni_amd.models import RENIAutoDecoder
    torch.manual_seed(0)
    full = RENIAutoDecoder(N_IMG, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, False)  # (the latent table is drawn before the
    m = RENIAutoDecoder(n_rows, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, False)    # weights: same table size, same weights)
    m.net.load_state_dict(full.net.state_dict())
    with torch.no_grad():
        for k, r in enumerate(rows):
            m.Z[k] = torch.randn(9, 3, generator=torch.Generator().manual_seed(100 + r))
    return m.set_compute_dtype("f32").to(dev)


def _data(rows, dev):
    from reni_amd.utils import get_directions, get_sineweight
    D, S = get_directions(32).to(dev), get_sineweight(32).to(dev)
    T = torch.stack([torch.rand(D.shape[1], 3, generator=torch.Generator().manual_seed(200 + r)) * 2 - 1 for r in rows]).to(dev)
    return D, S, T


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception:  # surface the worker's traceback in the parent's assertion
        import traceback
        q.put((rank, traceback.format_exc(), None, None))


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), RENI_SHARE_GPU="1", RENI_DIST_BACKEND="gloo")
    from reni_amd import dist as rdist
    from reni_amd.engine import TrainEngine
    rdist.init_from_env()
    dev = torch.device("cuda:0")
    rows = rdist.owned_indices(N_IMG, rank, world)  # image i lives on rank i mod W
    m = _model(len(rows), rows, dev)
    D, S, T = _data(rows, dev)
    eng = TrainEngine(m, lr=LR)
    idx = torch.arange(len(rows), device=dev)
    for _ in range(STEPS):
        eng.step(idx, T, S, D)
    torch.cuda.synchronize()
    q.put((rank, m._flat_params().detach().cpu().numpy(), m.Z.detach().cpu().numpy(), rows))
    torch.distributed.destroy_process_group()


def _overlap_worker(rank, world, port, q):
    """TrainEngine(overlap_comm=True) against the plain exchange, on the persistent bf16 kernels (where the library's 'layers >= 2
    are final' event sits in FRONT of k_reni_dw1) and on the generic fp32 path (where it is recorded at the end)."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), RENI_SHARE_GPU="1", RENI_DIST_BACKEND="gloo")
        from reni_amd import dist as rdist
        from reni_amd.engine import TrainEngine
        from reni_amd.models import RENIAutoDecoder
        rdist.init_from_env()
        dev = torch.device("cuda:0")
        rows = rdist.owned_indices(N_IMG, rank, world)
        D, S, T = _data(rows, dev)
        out = {}
        for dtype, H, L in (("bf16", 128, 3), ("f32", 64, 3)):
            for overlap in (False, True):
                torch.manual_seed(0)
                m = RENIAutoDecoder(len(rows), 9, "SO2", H, L, 3, True, "tanh", 30, 30, False)
                with torch.no_grad():
                    for k, r in enumerate(rows):
                        m.Z[k] = torch.randn(9, 3, generator=torch.Generator().manual_seed(100 + r))
                m.set_compute_dtype(dtype).to(dev)
                eng = TrainEngine(m, lr=LR, overlap_comm=overlap)
                assert eng.overlap_comm == overlap
                eng.time_comm(True)
                idx = torch.arange(len(rows), device=dev)
                for _ in range(3):
                    eng.step(idx, T, S, D)
                us = eng.time_comm(False)
                assert us is not None and us > 0
                torch.cuda.synchronize()
                out[(dtype, overlap)] = (m._flat_params().detach().cpu().numpy(), m.Z.detach().cpu().numpy())
        q.put((rank, out, None, None))
        torch.distributed.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None, None))


def test_overlapped_exchange_is_bit_equal_to_the_plain_one():
    """the all-reduce of layers >= 2 + head started behind reni_set_grad_ready_event on a communication
    stream, the rest behind the call -- two slices of one buffer, so decoder and latents must equal the one-collective step's bit
    for bit, on both ranks, for the persistent and the generic path."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    for dtype in ("bf16", "f32"):
        for r in res:
            a, b = r[1][(dtype, False)], r[1][(dtype, True)]
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), (dtype, r[0])
        assert (res[0][1][(dtype, True)][0] == res[1][1][(dtype, True)][0]).all()   # both ranks: the same decoder


def test_two_rank_step_equals_one_rank_step_on_the_union_batch():
    from reni_amd import ops
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    res = [(r[0], torch.from_numpy(r[1]), torch.from_numpy(r[2]), r[3]) for r in res]
    assert torch.equal(res[0][1], res[1][1]), "ranks ended with different decoders"
    # single process, union batch, every gradient scaled 1/W
    dev = torch.device("cuda:0")
    rows = list(range(N_IMG))
    m = _model(N_IMG, rows, dev)
    D, S, T = _data(rows, dev)
    flat, lat = m._flat_params(), m.Z.data
    md, vd, ml, vl = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(lat), torch.zeros_like(lat)
    idx = torch.arange(N_IMG, device=dev)
    plan = m._plan()
    for t in range(1, STEPS + 1):
        _, dZ, dp, _ = plan.forward_loss_backward(lat[idx], D, flat, T, S, need_dw=True, need_dz=True)
        ops.adam_rows_step(lat, dZ, idx, ml, vl, t, LR, grad_scale=0.5)
        ops.adam_step(flat, dp, md, vd, t, LR, grad_scale=0.5)
    ref = flat.detach().cpu()
    # Two steps: after one the update is lr * sign(g) (identical to 1e-9); Adam's early steps divide by a tiny sqrt(v), so
    # rounding-level gradient differences grow a thousandfold per further step (measured: median 6e-9 after two steps, 6e-6
    # after three) -- the comparison is meaningful only this early.
    d = (res[0][1] - ref).abs()
    scale = float(ref.abs().max())
    qs = [float(torch.quantile(d, q)) for q in (0.5, 0.999)]
    assert qs[0] <= 1e-6 * scale and qs[1] <= 1e-4 * scale, (qs, float(d.max()), scale)
    assert float(d.max()) <= 1.1 * LR * STEPS
    for _, _, Zr, rws in res:
        assert float((Zr - lat[torch.tensor(rws, device=dev)].cpu()).abs().max()) <= 2e-5


LINE_MAX = 6144  # bench.py's contract line (a 24 KB line did not parse on the driver's side)


def _run_bench(extra_env, *args, with_also=False):
    """-> the contract line (the LAST line of stdout, alone on it, <= LINE_MAX bytes); with_also: (line, {name: sub-record}) -- the
    sub-records are the `also <name> {json}` lines in front of it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = r.stdout.splitlines()
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1 and out[-1] == lines[0], r.stdout[-2000:]  # rank 0 alone prints, exactly one JSON line, and it is the last
    assert len(lines[0]) <= LINE_MAX, len(lines[0])
    line = json.loads(lines[0])
    if not with_also:
        return line
    also = {}
    for ln in out:
        if ln.startswith("also "):
            _, name, rec = ln.split(" ", 2)
            also[name] = json.loads(rec)
    return line, also


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's N > 1 form is torchrun, but the command must
    also stand alone): two rank processes are started before anything touches the GPU.  Here both share cuda:0 over
    gloo.  At N > 1 the step defaults to the fused data-parallel call on the library's own RCCL communicator (VERDICT r05 item 2:
    the step N = 1 runs, with the exchange inside it); two ranks on ONE GPU is the case RCCL refuses, and then the line must say so
    (`exchange.comm_fallback`) and name the three-call path it fell back to."""
    line = _run_bench({"RENI_SHARE_GPU": "1", "RENI_DIST_BACKEND": "gloo"}, "--gpus", "2", "--steps", "2", "--warmup", "1",
                      "--batch", "8", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["dist_backend"] == "gloo"
    assert line["config"]["global_batch_images"] == 16 and line["value"] > 0
    assert line["roofline"]["kernel_launches"] == 2
    if line["step_call"].startswith("reni_train_step_rows_dp"):      # RCCL initialised (one GPU per rank)
        assert "comm_fallback" not in line["exchange"] and line["exchange"]["kind"].startswith("inside reni_train_step_rows_dp")
    else:                                                             # it could not: loudly, on the line
        assert line["exchange"]["comm_fallback"] and line["step_call"].startswith("reni_forward_loss_backward_rows + all-reduce")
        assert line["exchange"]["kind"].startswith("torch.distributed all_reduce (gloo)")
    # the opt-out
    t = _run_bench({"RENI_SHARE_GPU": "1", "RENI_DIST_BACKEND": "gloo"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
                   "--no-cpu-baseline", "--comm", "torch")
    assert "comm_fallback" not in t["exchange"] and t["step_call"].startswith("reni_forward_loss_backward_rows + all-reduce")


def test_bench_default_line_carries_every_baseline_config_and_the_capi_exchange():
    """bench.py's default output: the config-2 headline on ONE short contract line (last), and in front of it one `also` line per
    sub-record -- config 4, config 5, the FiLM step, the shipped batch / schedule / width -- measured in the same process, each with its
    own roofline; the contract line keeps [value, ms_per_step, frac_step] of each.  `--comm capi` at N = 1 runs the fused data-parallel
    step on a one-rank communicator (the same call path as N > 1)."""
    line, also = _run_bench({}, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", with_also=True)
    curric = {"c2_curric_16x32", "c2_curric_32x64", "c2_curric_64x128"}   # configs/experiment.yaml:29-34, B = 100
    assert set(also) == {"c4", "c4_dense", "c4_pixels", "c4_f32", "c5", "film", "c2_b100", "c2_h256", "c4_h256", "c4_h256_dense", "fwd_h256", "film_h256"} | curric, {k: v.get("error") for k, v in also.items()}
    assert all("error" not in v for v in also.values()), {k: v.get("error") for k, v in also.items()}
    assert set(line["also"]) == set(also)
    for k, v in also.items():   # the contract line's triple is the sub-record's (to the printed digits)
        val, ms, fs = line["also"][k]
        assert abs(val - v["value"]) <= 1e-4 * val and abs(ms - v["ms_per_step"]) <= 1e-4 * ms and abs(fs - v["roofline"]["frac_step"]) <= 1e-4 * fs
    assert also["c2_b100"]["images_per_gpu_per_step"] == 100
    # config 4 twice: with RENI_WEIGHT_SPARSE (what RENI.training_step passes with a mask; Mask-3: 148 of 256 tiles per image carry
    # weight, pixel 0 is masked -> no statistics pass) and dense; the sparse record's roofline counts visited tiles only
    sp, de = also["c4"]["weight_sparsity"], also["c4_dense"]["weight_sparsity"]
    assert sp["flag"] == "RENI_WEIGHT_SPARSE" and abs(sp["tiles_visited"] - 148 / 256) < 1e-6 and not sp["cosine_term_live"]
    assert de["flag"] == "off" and de["tiles_visited"] == 1.0 and abs(de["pixels_with_weight"] - 0.188) < 2e-3
    # (the kernels' own times, HIP events: a 0.2-0.3 ms step is host-paced on a busy host, its wall time is not a stable comparison)
    assert also["c4"]["roofline"]["kernel_avg_ms"] < 0.8 * also["c4_dense"]["roofline"]["kernel_avg_ms"]
    assert "stats_pass_avg_ms" in also["c4_dense"]["roofline"]
    px = also["c4_pixels"]["weight_sparsity"]
    assert px["flag"] == "RENI_WEIGHT_COMPACT" and abs(px["tiles_visited"] - 49 / 256) < 1e-6
    assert also["c4_pixels"]["roofline"]["kernel_avg_ms"] < 0.8 * also["c4"]["roofline"]["kernel_avg_ms"]
    for c, flop in (("c4", 348448), ("c4_dense", 348448), ("c4_pixels", 348448), ("c5", 177860), ("film", 424480), ("c2_h256", 2028320), ("film_h256", 1635104)):
        r = also[c]
        assert r["value"] > 0 and r["ms_per_step"] > 0 and r["roofline"]["flop_per_sample"] == flop and r["roofline"]["kernel_avg_ms"] > 0
        assert 0 < r["roofline"]["frac_step"] <= r["roofline"]["frac"] * 1.0001 and r["ms_per_step_mean"] >= r["ms_per_step"]
    # `frac` divides by EVERY kernel that performs the step's FLOPs: at H = 256 the chain, k_dw_frag and k_wide_head_dw
    h256 = also["c2_h256"]["roofline"]
    assert [k["kernel"] for k in h256["kernels"]] == ["k_reni_wide256<2>", "k_dw_frag + k_wide_head_dw"]
    assert h256["work_kernels_ms_per_step"] > 1.3 * h256["kernel_avg_ms"] and h256["kernels"][1]["ms_per_step"] > 0
    for c in curric:
        r = also[c]
        assert r["images_per_gpu_per_step"] == 100 and r["launches_per_step"] >= 2 and r["paths"]["env_overrides"] == []
    assert also["c5"]["so3"]["value"] > 0            # both invariances
    assert line["roofline"]["flop_per_sample"] == 522784
    # ONE definition of the headline: the contract's W + K window, first; the sustained-clock re-run is a side field
    assert line["steps"] == 3 and line["warmup"] == 1 and line["sustained"]["steps"] == 3 and "from_idle" not in line
    assert line["config"]["paths"]["dw1_kernel"] == "k_reni_l0_ring" and line["config"]["paths"]["env_overrides"] == []
    assert line["step_call"].startswith("reni_train_step_rows (one call")
    # the fractions side by side, the two kernels of the backward pass listed with their own times; `frac` over BOTH kernels' time
    rf = line["roofline"]
    assert rf["kernel"] == "k_reni_train_bf16<128,true,L0X>" and 0 < rf["frac_step"] < rf["frac"] and rf["kernel_min_ms"] <= rf["kernel_avg_ms"] <= rf["kernel_max_ms"]
    assert [k["kernel"] for k in rf["kernels"]] == ["k_reni_train_bf16<128,true,L0X>", "k_reni_l0_ring"] and rf["kernels"][1]["avg_ms"] > 0
    both = rf["kernel_avg_ms"] + rf["kernels"][1]["avg_ms"]
    assert abs(rf["work_kernels_ms_per_step"] - both) <= 1e-3 * both
    want = 522784 * 64 * 32768 / (both * 1e-3) / 1e12 / 2500.0
    assert abs(rf["frac"] - want) <= 2e-3 * want
    f32 = also["c4_f32"]
    assert f32["dtype"] == "f32" and f32["roofline"]["peak"] == 157.3 and f32["roofline"]["flop_per_sample"] == 348448
    assert also["c4"]["value_kind"] == "effective_samples_per_s" and also["c4_dense"]["value_kind"] == "samples_per_s"
    assert abs(also["c4"]["visited_samples_per_s"] - also["c4"]["value"] * 148 / 256) <= 1e-4 * also["c4"]["value"]
    assert line["launches_per_step"] == int(line["launches_per_step"]) and 2 <= line["launches_per_step"] <= 16
    forced = _run_bench({"RENI_DW1_OLD": "1"}, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-also")
    assert forced["config"]["paths"]["dw1_kernel"] == "k_reni_dw1" and forced["config"]["paths"]["env_overrides"] == ["RENI_DW1_OLD"]
    capi = _run_bench({}, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--comm", "capi")
    assert "also" not in capi and capi["step_call"].startswith("reni_train_step_rows_dp") and capi["value"] > 0
    ex = capi["exchange"]   # (the library's own event pair around its all-reduce, not the bracket of the whole call)
    assert ex["kind"].startswith("inside reni_train_step_rows_dp") and 0 < ex["avg_us_on_compute_stream"] < 0.5e3 * capi["ms_per_step"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
def test_bench_two_ranks_over_rccl():
    """The exchange step on the real transport: backend "nccl" (= RCCL) on two GPUs, self-launched."""
    line = _run_bench({}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["dist_backend"] == "nccl"
    assert line["step_call"].startswith("reni_train_step_rows_dp") and "comm_fallback" not in line["exchange"]
    one = _run_bench({}, "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert line["value"] > 1.2 * one["value"]  # weak scaling: two ranks do twice the work per step


# ---- reni_allreduce_grads: the library's own RCCL exchange step ----------------------------------
def _rccl_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        os.environ.pop("RENI_SHARE_GPU", None)
        os.environ.pop("RENI_DIST_BACKEND", None)
        from reni_amd import dist as rdist
        from reni_amd.engine import TrainEngine
        rdist.init_from_env()  # world 1: no process group at all; world 2: nccl, one GPU per rank
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        comm = rdist.RcclComm(rank, world)
        x = torch.randn(100003, device=dev, generator=torch.Generator(device=dev).manual_seed(5 + rank))
        want = sum(torch.randn(100003, device=dev, generator=torch.Generator(device=dev).manual_seed(5 + r)) for r in range(world)) * 0.25
        comm.allreduce_(x, 0.25)
        assert float((x - want).abs().max()) <= 1e-6
        rows = rdist.owned_indices(N_IMG, rank, world)
        out = {}
        for native in (True, False):
            if world > 1 and not native:
                continue
            m = _model(len(rows), rows, dev)
            D, S, T = _data(rows, dev)
            eng = TrainEngine(m, lr=LR, comm=comm if native else None)
            idx = torch.arange(len(rows), device=dev)
            for _ in range(STEPS):
                eng.step(idx, T, S, D)
            torch.cuda.synchronize()
            out[native] = (m._flat_params().detach().cpu().numpy(), m.Z.detach().cpu().numpy())
        comm.close()
        q.put((rank, out, rows))
        if world > 1:
            torch.distributed.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None))


def _spawn_rccl(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    return res


def test_rccl_comm_of_one_rank_runs_the_exchange_step():
    """reni_rccl_unique_id / reni_rccl_comm_create / reni_allreduce_grads / reni_rccl_comm_destroy on the one GPU this box
    has: a one-rank communicator (the sum over one rank, then the scale kernel), and TrainEngine stepping through it gives
    bit-for-bit the decoder and latents of the fused single-process update."""
    (rank, out, rows), = _spawn_rccl(1)
    assert (out[True][0] == out[False][0]).all() and (out[True][1] == out[False][1]).all()


def _dp_step_worker(rank, world, port, q, cfg):
    """reni_train_step_rows_dp through TrainEngine(comm=RcclComm()): every variant of the step on the same data."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        os.environ.pop("RENI_SHARE_GPU", None)
        os.environ.pop("RENI_DIST_BACKEND", None)
        from reni_amd import dist as rdist
        from reni_amd.engine import TrainEngine
        from reni_amd.models import RENIAutoDecoder
        from reni_amd.utils import get_directions, get_sineweight
        rdist.init_from_env()
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        comm = rdist.RcclComm(rank, world)
        dtype, H, L, W, B = cfg
        N = 3 * B
        D, S = get_directions(W).to(dev), get_sineweight(W).to(dev)
        T = torch.stack([torch.rand(D.shape[1], 3, generator=torch.Generator().manual_seed(300 + 7 * rank + i)) * 2 - 1 for i in range(N)]).to(dev)
        batches = [torch.arange(B, device=dev) + o for o in (0, B, 2 * B, B)]
        out = {}
        variants = {"one_process": dict(), "dp": dict(comm=comm), "dp_overlap": dict(comm=comm, overlap_comm=True),
                    "three_calls": dict(comm=comm, fused_step=False)}
        if world > 1:
            variants.pop("one_process")
        for name, kw in variants.items():
            torch.manual_seed(0)
            m = RENIAutoDecoder(N, 9, "SO2", H, L, 3, True, "tanh", 30.0, 30.0, False)
            m.set_compute_dtype(dtype).to(dev)
            e = TrainEngine(m, lr=1e-3, **kw)
            terms = []
            for k, idx in enumerate(batches):
                nxt = batches[k + 1] if k + 1 < len(batches) else None
                terms.append(e.step(idx, T[idx], S, D, next_idx=nxt).clone())
            torch.cuda.synchronize()
            out[name] = [t.detach().cpu().numpy() for t in (torch.stack(terms), m._flat_params(), m.Z.data, e.m_dec, e.v_dec, e.m_lat, e.v_lat)]  # (numpy: plain pickles)
            out[name + ".fused"] = e._stage is not None
        comm.close()
        q.put((rank, out, None))
        if world > 1:
            torch.distributed.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None))


def _spawn_dp(world, cfg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_step_worker, args=(r, world, port, q, cfg)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    return res


@pytest.mark.parametrize("cfg", [("bf16", 128, 5, 128, 8), ("bf16", 128, 5, 32, 4), ("f32", 64, 3, 32, 4), ("bf16", 128, 4, 64, 4)])
def test_the_data_parallel_step_is_the_same_step(cfg):
    """reni_train_step_rows_dp -- the fused training step with the RCCL exchange INSIDE the call -- with a one-rank
    communicator is bit-equal to reni_train_step_rows (loss terms of every step, parameters, latents, all four Adam moments), with and
    without the early slice on the library's stream, and equal to the three-call path (forward_loss_backward_rows -> reni_allreduce_grads
    -> adam_step2) it replaces for comm= engines.  Paths: persistent + k_reni_l0_ring with the forked / one-stream tail, generic fp32,
    persistent round-4 kernels (even L)."""
    (rank, out, _), = _spawn_dp(1, cfg)
    assert out["one_process.fused"] and out["dp.fused"] and out["dp_overlap.fused"] and not out["three_calls.fused"]
    names = ("terms", "params", "Z", "m_dec", "v_dec", "m_lat", "v_lat")
    for var in ("dp", "dp_overlap", "three_calls"):
        for a, b, name in zip(out["one_process"], out[var], names):
            assert (a == b).all(), (var, name, float(abs(a - b).max()))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
def test_the_data_parallel_step_on_two_gpus():
    """Two GPUs, config 3's step: both ranks end with the same decoder; the fused DP step, its overlapped form and the three-call
    path agree to the order of RCCL's sums (bit-equal between the two fused forms: the same collective split)."""
    res = _spawn_dp(2, ("bf16", 128, 5, 128, 8))
    for var in ("dp", "dp_overlap", "three_calls"):
        assert (res[0][1][var][1] == res[1][1][var][1]).all(), var      # the replicas stay replicas
    p_dp, p_3 = res[0][1]["dp"][1], res[0][1]["three_calls"][1]
    assert float(abs(p_dp - p_3).max()) <= 1e-5 * float(abs(p_3).max())
    assert float(abs(res[0][1]["dp_overlap"][1] - p_dp).max()) <= 1e-5 * float(abs(p_dp).max())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
def test_rccl_comm_two_ranks_equal_the_union_batch():
    """Two GPUs: the engine's exchange step through reni_allreduce_grads; both ranks end with the same decoder, equal to
    the single-process step on the union batch with every gradient scaled 1/2."""
    from reni_amd import ops
    res = _spawn_rccl(2)
    f0, f1 = torch.from_numpy(res[0][1][True][0]), torch.from_numpy(res[1][1][True][0])
    assert torch.equal(f0, f1)
    dev = torch.device("cuda:0")
    rows = list(range(N_IMG))
    m = _model(N_IMG, rows, dev)
    D, S, T = _data(rows, dev)
    flat, lat = m._flat_params(), m.Z.data
    md, vd, ml, vl = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(lat), torch.zeros_like(lat)
    idx = torch.arange(N_IMG, device=dev)
    plan = m._plan()
    for t in range(1, STEPS + 1):
        _, dZ, dp, _ = plan.forward_loss_backward(lat[idx], D, flat, T, S, need_dw=True, need_dz=True)
        ops.adam_rows_step(lat, dZ, idx, ml, vl, t, LR, grad_scale=0.5)
        ops.adam_step(flat, dp, md, vd, t, LR, grad_scale=0.5)
    d = (f0 - flat.detach().cpu()).abs()
    scale = float(flat.abs().max())
    assert float(torch.quantile(d, 0.5)) <= 1e-6 * scale and float(torch.quantile(d, 0.999)) <= 1e-4 * scale


# ---- the FiLM TrainEngine under data parallelism: [net | final_layer | mapping_network] is ONE flat buffer and one all-reduce -------
def _film_model(n_rows, rows, dev):
    from reni_amd.film import RENIAutoDecoderFiLM
    torch.manual_seed(0)
    full = RENIAutoDecoderFiLM(N_IMG, 6, "SO2", 64, 3, 16, 2, 3, "tanh", False)  # (the latent table is drawn before the weights:
    m = RENIAutoDecoderFiLM(n_rows, 6, "SO2", 64, 3, 16, 2, 3, "tanh", False)    # same table size -> same weights on every rank)
    for name in ("net", "final_layer", "mapping_network"):
        getattr(m, name).load_state_dict(getattr(full, name).state_dict())
    with torch.no_grad():
        for k, r in enumerate(rows):
            m.Z[k] = torch.randn(6, 3, generator=torch.Generator().manual_seed(100 + r))
    return m.set_compute_dtype("f32").to(dev)


def _film_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), RENI_SHARE_GPU="1", RENI_DIST_BACKEND="gloo")
        from reni_amd import dist as rdist
        from reni_amd.engine import TrainEngine
        rdist.init_from_env()
        dev = torch.device("cuda:0")
        rows = rdist.owned_indices(N_IMG, rank, world)
        m = _film_model(len(rows), rows, dev)   # (seed 0 draws the same decoder + mapping network on every rank)
        D, S, T = _data(rows, dev)
        eng = TrainEngine(m, lr=LR)
        idx = torch.arange(len(rows), device=dev)
        for _ in range(STEPS):
            eng.step(idx, T, S, D)
        torch.cuda.synchronize()
        q.put((rank, m._all_flat().detach().cpu().numpy(), m.Z.detach().cpu().numpy(), rows))
        torch.distributed.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None, None))


def test_film_two_rank_step_equals_one_rank_step_on_the_union_batch():
    """As the concat test above, for a FiLM model: decoder, head AND mapping network are one flat buffer, all-reduced once."""
    from reni_amd import ops
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_film_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    f0, f1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(f0, f1), "ranks ended with different decoders / mapping networks"
    dev = torch.device("cuda:0")
    rows = list(range(N_IMG))
    m = _film_model(N_IMG, rows, dev)
    D, S, T = _data(rows, dev)
    flat, lat = m._all_flat(), m.Z.data
    n = m._plan().n_params
    md, vd, ml, vl = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(lat), torch.zeros_like(lat)
    idx = torch.arange(N_IMG, device=dev)
    plan = m._plan()
    for t in range(1, STEPS + 1):
        _, dZ, dp, _, _ = plan.film_model_forward_loss_backward(lat[idx], D, flat[:n], flat[n:], T, S, need_dw=True)
        ops.adam_rows_step(lat, dZ, idx, ml, vl, t, LR, grad_scale=0.5)
        ops.adam_step(flat, dp._base, md, vd, t, LR, grad_scale=0.5)
    d = (f0 - flat.detach().cpu()).abs()
    scale = float(flat.abs().max())
    assert float(torch.quantile(d, 0.5)) <= 1e-6 * scale and float(torch.quantile(d, 0.999)) <= 2e-4 * scale, (float(d.max()), scale)
    assert float(d.max()) <= 1.1 * LR * STEPS
