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
