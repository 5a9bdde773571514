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
