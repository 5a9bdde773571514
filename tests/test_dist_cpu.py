"""world_size-2 gloo tests of the data-parallel helpers (CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from reni_amd import dist as rdist
from reni_amd import trainer


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
        self.Z = torch.nn.Parameter(torch.zeros(6, 2, 3))


class _FakeFilmModel(torch.nn.Module):
    """The FiLM variants' trainable parts outside ``net`` (reni_amd/film.py): final_layer, mapping_network."""
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(4, 3))
        self.final_layer = torch.nn.Linear(3, 3)
        self.mapping_network = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 6))
        self.mu = torch.nn.Parameter(torch.zeros(5, 2, 3))
        self.log_var = torch.nn.Parameter(torch.zeros(5, 2, 3))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    flat = torch.full((5,), float(rank + 1))
    rdist.allreduce_mean_(flat)
    ok = torch.allclose(flat, torch.full((5,), 1.5))
    torch.manual_seed(0)
    m = _FakeModel()
    if rank == 1:
        with torch.no_grad():
            for p in m.net.parameters():
                p.add_(1.0)
    for p in m.net.parameters():
        rdist.broadcast_(p.data, 0)
    torch.manual_seed(0)
    ref = _FakeModel()
    ok = ok and all(torch.equal(a, b) for a, b in zip(m.net.parameters(), ref.net.parameters()))
    # decoder grads: mean over ranks; latent grads: owner's grad / world
    for i, p in enumerate(m.net.parameters()):
        p.grad = torch.full_like(p, float(rank * 2 + i))
    m.Z.grad = torch.zeros_like(m.Z)
    own = rdist.owned_indices(6, rank, world)
    m.Z.grad[own] = 4.0
    trainer.sync_decoder_grads(m)
    for i, p in enumerate(m.net.parameters()):
        ok = ok and torch.allclose(p.grad, torch.full_like(p, 1.0 + i))
    ok = ok and torch.allclose(m.Z.grad[own], torch.full((len(own), 2, 3), 2.0))
    ok = ok and own == list(range(rank, 6, 2))
    v = rdist.allreduce_mean_scalars(torch.tensor([float(rank)]))
    ok = ok and abs(float(v) - 0.5) < 1e-6
    # FiLM: every trainable non-latent parameter is synchronised, also with a missing gradient (empty batch on a rank)
    torch.manual_seed(1)
    f = _FakeFilmModel()
    names = [n for n, _ in f.named_parameters() if n not in trainer.LATENT_NAMES]
    ok = ok and len(trainer._decoder_params(f)) == len(names) == 8
    for i, (n, p) in enumerate(f.named_parameters()):
        if n in trainer.LATENT_NAMES:
            continue
        p.grad = None if rank == 1 else torch.full_like(p, 2.0 + i)
    f.mu.grad = torch.full_like(f.mu, 6.0)
    trainer.sync_decoder_grads(f)
    for i, (n, p) in enumerate(f.named_parameters()):
        if n not in trainer.LATENT_NAMES:
            ok = ok and torch.allclose(p.grad, torch.full_like(p, (2.0 + i) / 2))
    ok = ok and torch.allclose(f.mu.grad, torch.full_like(f.mu, 3.0)) and f.log_var.grad is None
    # uneven ownership (5 images, 2 ranks, batch 2): both ranks run the same number of steps
    nb = rdist.epoch_batches(5, 2, rank, world)
    ok = ok and len(nb) == rdist.steps_per_epoch(5, 2, world) == 2
    ok = ok and nb == ([[0, 2], [4]] if rank == 0 else [[1, 3], []])
    ok = ok and rdist.steps_per_epoch(201, 100, 2) == 2 and rdist.steps_per_epoch(615, 100, 8) == 1
    # latent merge: rows trained by their owner end up on every rank, moments included
    with torch.no_grad():
        f.mu.zero_()
        f.mu[rank::world] = 10.0 + rank
        f.mu[(1 - rank)::world] = -1.0  # stale values of rows this rank does not own
    opt = torch.optim.Adam([f.mu], lr=1e-3)
    opt.state[f.mu] = {"step": torch.tensor(1.0), "exp_avg": f.mu.detach().clone() * 2, "exp_avg_sq": f.mu.detach().clone() * 3}
    trainer.gather_latents(f, opt)
    want = torch.tensor([10.0, 11.0, 10.0, 11.0, 10.0]).view(5, 1, 1).expand(5, 2, 3)
    ok = ok and torch.equal(f.mu.detach(), want) and torch.equal(opt.state[f.mu]["exp_avg"], want * 2)
    ok = ok and torch.equal(opt.state[f.mu]["exp_avg_sq"], want * 3)
    shard = torch.full((len(rdist.owned_indices(5, rank, world)), 2), float(rank))
    full = rdist.gather_shards(shard, 5)
    ok = ok and torch.equal(full[:, 0], torch.tensor([0.0, 1.0, 0.0, 1.0, 0.0]))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gloo_world2_grad_sync():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_ownership_and_step_counts_of_config_3():
    """BASELINE config 3's partition (615 images over 8 ranks, 64 images per rank and step): every image has exactly one owner
    (`i % world`, what DistributedSampler(shuffle=False) gives the reference's un-shuffled loader, run.py:97-110), the owners'
    shares differ by at most one, and EVERY rank runs the same number of steps per epoch -- so the per-step all-reduce pairs up."""
    from reni_amd import dist as rdist
    n, world, B = 615, 8, 64
    owned = [rdist.owned_indices(n, r, world) for r in range(world)]
    assert sorted(i for o in owned for i in o) == list(range(n))
    assert {len(o) for o in owned} <= {76, 77} and all(i % world == r for r, o in enumerate(owned) for i in o)
    steps = rdist.steps_per_epoch(n, B, world)
    assert steps == 2  # ceil(77 / 64)
    for r in range(world):
        bs = rdist.epoch_batches(n, B, r, world)
        assert len(bs) == steps and [i for b in bs for i in b] == owned[r]
        assert len(bs[0]) == 64 and len(bs[1]) in (12, 13)
    # bench.py's per-GPU batch: 64 <= the smallest share, so every rank can draw a full batch at every N in {1, 2, 4, 8}
    for w in (1, 2, 4, 8):
        assert min(len(rdist.owned_indices(n, r, w)) for r in range(w)) >= 64


# ---- world-8 TrainEngine on the CPU: config 3's code path end to end -----------------------------------
# The HIP kernels cannot run here, so the engine's two device calls are replaced BY THE TEST with the oracle (test infrastructure:
# the product has no such path): the fused fwd+loss+bwd by oracle.fwd_loss_bwd, the fused Adam by the formula of torch.optim.Adam.
# Everything else is the product's: ownership, the gather of the batch's rows, ONE all-reduce(sum) of the flat gradient, the 1 / W on
# Adam's grad_scale, the dense latent Adam, short last batches.
N8, B8, W8 = 615, 64, 8
SPEC8 = dict(ndims=2, equivariance="SO2", hidden_features=8, hidden_layers=1)


class _OraclePlan:
    conditioning = "concat"

    def __init__(self):
        from oracle import reni_oracle as O
        self.O = O
        self.spec = O.DecoderSpec(SPEC8["ndims"], SPEC8["equivariance"], SPEC8["hidden_features"], SPEC8["hidden_layers"], 3, True, "tanh")
        self.hidden_features, self.hidden_layers = SPEC8["hidden_features"], SPEC8["hidden_layers"]

    def forward_loss_backward(self, Z, D, flat, target, weight, loss_kind="mse", alpha=0.0, beta=0.0, need_dw=True, need_dz=True,
                              want_out=False, idx=None):
        from tests.util import unflatten
        params = unflatten(self.spec, flat.detach().clone())
        Zb = Z[idx] if idx is not None else Z
        B, P = Zb.shape[0], D.shape[1]
        r = self.O.fwd_loss_bwd(self.spec, params, Zb, D.expand(B, P, 3), target, weight.expand(B, P, 3), loss_kind, alpha, beta)
        dp = torch.cat([r["grads"][k].reshape(-1) for k in self.spec.param_keys()])
        return torch.tensor(r["loss_terms"]), r["dZ"], dp, None


class _TorchAdamOps:
    """ops.adam_step2 / adam_rows_step restated with torch on the CPU (torch.optim.Adam's update; dense over the latent table)."""
    @staticmethod
    def _adam(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(m / (1 - b1 ** t), (v / (1 - b2 ** t)).sqrt() + eps, value=-lr)

    @classmethod
    def adam_step2(cls, p, g, m, v, table, g_rows, idx, tm, tv, step, lr, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        cls._adam(p, g * grad_scale, m, v, step, lr)
        cls.adam_rows_step(table, g_rows, idx, tm, tv, step, lr, grad_scale=grad_scale)

    @classmethod
    def adam_rows_step(cls, table, g_rows, idx, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        dense = torch.zeros_like(table).index_add_(0, idx, g_rows * grad_scale)
        cls._adam(table, dense, m, v, step, lr)


class _TinyModel:
    """What TrainEngine touches of a RENIAutoDecoder: the flat decoder buffer, the latent table, the plan."""
    fixed_decoder = False

    def __init__(self, rows):
        from oracle import reni_oracle as O
        from tests.util import flat_params
        plan = _OraclePlan()
        self._p = plan
        self._flat = flat_params(plan.spec, O.init_params(plan.spec, torch.Generator().manual_seed(8))).clone()
        self.Z = torch.nn.Parameter(torch.stack([torch.randn(SPEC8["ndims"], 3, generator=torch.Generator().manual_seed(1000 + r)) for r in rows]))

    def _plan(self):
        return self._p

    def _flat_params(self):
        return self._flat


def _data8(rows):
    from oracle import reni_oracle as O
    D, S = O.get_directions(8), O.get_sineweight(8)            # 4 x 8 = 32 directions
    T = torch.stack([torch.rand(D.shape[1], 3, generator=torch.Generator().manual_seed(5000 + r)) * 2 - 1 for r in rows])
    return D, S, T


def _engine8_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        torch.set_num_threads(1)
        rdist.init_from_env(backend="gloo")
        from reni_amd import engine
        engine.ops = _TorchAdamOps                      # (the test's stand-ins for the two device calls, see above)
        rows = rdist.owned_indices(N8, rank, world)
        m = _TinyModel(rows)
        D, S, T = _data8(rows)
        eng = engine.TrainEngine(m, lr=1e-2)
        assert eng.world == world
        local = {r: k for k, r in enumerate(rows)}
        batches = rdist.epoch_batches(N8, B8, rank, world)
        for b in batches:                               # one epoch: 64 images, then the short rest (12 or 13)
            idx = torch.tensor([local[i] for i in b])
            eng.step(idx, T[idx], S, D)
        q.put((rank, m._flat_params().numpy().copy(), m.Z.detach().numpy().copy(), [len(b) for b in batches]))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc(), None, None))


def test_world8_train_engine_epoch_equals_one_process_on_the_union_batches():
    """BASELINE config 3's partition through TrainEngine.step on 8 gloo ranks: 615 images, owners i % 8 with 76-77 images each, two
    steps per epoch (64, then 12-13 images per rank).  Every rank ends with the same decoder, and decoder + latents equal one process
    stepping on the UNION of the ranks' batches with every gradient scaled 1/8 (the reference's DDP mean, run.py:97)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine8_worker, args=(r, W8, port, q)) for r in range(W8)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    assert all(r[3][0] == 64 and r[3][1] in (12, 13) and len(r[3]) == 2 for r in res)
    for r in res[1:]:
        assert (r[1] == res[0][1]).all(), "ranks ended with different decoders"
    # one process: the union batch of every step, gradients / 8
    rows = list(range(N8))
    m = _TinyModel(rows)
    D, S, T = _data8(rows)
    flat, lat = m._flat_params(), m.Z.data
    md, vd, ml, vl = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(lat), torch.zeros_like(lat)
    per_rank = [rdist.epoch_batches(N8, B8, r, W8) for r in range(W8)]
    for t in range(2):
        idx = torch.tensor(sorted(i for r in range(W8) for i in per_rank[r][t]))
        _, dZ, dp, _ = m._plan().forward_loss_backward(lat, D, flat, T[idx], S, idx=idx)
        _TorchAdamOps.adam_step2(flat, dp, md, vd, lat, dZ, idx, ml, vl, t + 1, 1e-2, grad_scale=1.0 / W8)
    assert float((torch.from_numpy(res[0][1]) - flat).abs().max()) <= 2e-5 * float(flat.abs().max())
    for r in res:
        own = rdist.owned_indices(N8, r[0], W8)
        assert float((torch.from_numpy(r[2]) - lat[own]).abs().max()) <= 2e-5
