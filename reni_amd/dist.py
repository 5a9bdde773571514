"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference wraps the LightningModule in DDP (run.py:97): every step all-reduces (mean) the
gradient of every trainable parameter, i.e. the decoder AND the whole latent table.  Here:

* images (their latent rows, Adam moments and target pixels) are owned by rank ``i % world``
  (what Lightning's DistributedSampler(shuffle=False) over the reference's un-shuffled DataLoader
  produces);
* ONE all-reduce(sum) of the flat decoder-gradient buffer per step, then scale 1/world;
* latent rows need no communication: non-owners would contribute exact zeros, so the owner's
  gradient x 1/world is the reference's all-reduced value.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = os.environ.get("RENI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("RENI_SHARE_GPU"):  # test hook: several ranks on ONE GPU (gloo only; RCCL needs one GPU per rank)
        local = 0
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def owned_indices(n_items, rank, world):
    """Indices of the images a rank owns (round-robin, as DistributedSampler(shuffle=False)).  No padding:
    DistributedSampler pads the short ranks with wrapped-around duplicates so that every rank draws the same count;
    a duplicate would put one latent row on two owners, so here the COUNT of steps is equalised instead
    (``epoch_batches``) and a rank that has run out of images joins the step's collective with a zero gradient."""
    return list(range(rank, n_items, world))


def steps_per_epoch(n_items, batch_size, world):
    """Optimiser steps every rank runs per epoch: the longest rank's, ceil(ceil(n / world) / batch_size)."""
    longest = (n_items + world - 1) // world
    return (longest + batch_size - 1) // batch_size


def epoch_batches(n_items, batch_size, rank, world):
    """This rank's index lists for one epoch -- exactly ``steps_per_epoch`` of them on EVERY rank (the trailing ones may
    be short or empty), so the per-step gradient all-reduce and the epoch-end metric all-reduce always pair up."""
    own = owned_indices(n_items, rank, world)
    return [own[s * batch_size:(s + 1) * batch_size] for s in range(steps_per_epoch(n_items, batch_size, world))]


def merge_owned_rows_(table: torch.Tensor, rank=None, world=None):
    """After training, every rank holds trained values only in the rows it owns (``i % world == rank``); the others are
    still at their initial values there.  Sum of the owner-masked tables = the full trained table on every rank: what
    the reference's DDP keeps at all times by all-reducing the whole latent table.  In place."""
    w = world_size() if world is None else world
    if w == 1:
        return table
    r = dist.get_rank() if rank is None else rank
    mask = torch.zeros(table.shape[0], dtype=table.dtype, device=table.device)
    mask[r::w] = 1
    merged = table * mask.view(-1, *([1] * (table.dim() - 1)))
    dist.all_reduce(merged, op=dist.ReduceOp.SUM)
    table.copy_(merged)
    return table


def gather_shards(shard: torch.Tensor, n_items: int, rank=None, world=None):
    """Full ``[n_items, ...]`` table from per-rank shards holding rows ``rank, rank + world, ...`` (TrainEngine /
    bench.py keep only the owned rows resident)."""
    w = world_size() if world is None else world
    if w == 1:
        return shard.clone()
    r = dist.get_rank() if rank is None else rank
    full = torch.zeros((n_items,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    full[r::w] = shard
    dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full


def allreduce_mean_(flat: torch.Tensor):
    """In-place mean all-reduce of one flat buffer (decoder gradients): one collective per step."""
    w = world_size()
    if w == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / w)
    return flat


class RcclComm:
    """An RCCL communicator owned by libreni_hip.so, for the one exchange step of the data-parallel path
    (``reni_allreduce_grads``: in-place sum of the flat decoder gradient + scale, on the caller's stream -- SURVEY 8 (b)
    item 7).  Rank 0 draws the unique id; it reaches the other ranks through the existing torch.distributed group (any
    backend -- it is 128 bytes, once).  One process per GPU; the device must be current when this is constructed."""

    def __init__(self, rank=None, world=None):
        import ctypes
        from . import _lib
        self._lib, self._check = _lib.load(), _lib.check
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.world = world if world is not None else world_size()
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            self._check(self._lib.reni_rccl_unique_id(buf))
        if self.world > 1:
            box = [buf.raw]
            dist.broadcast_object_list(box, src=0)
            buf = ctypes.create_string_buffer(box[0], 128)
        self._comm = ctypes.c_void_p()
        self._check(self._lib.reni_rccl_comm_create(buf, self.world, self.rank, ctypes.byref(self._comm)))

    def allreduce_(self, flat: torch.Tensor, scale: float = 1.0):
        """flat <- scale * sum over ranks of flat, in place, on the current stream"""
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        with torch.cuda.device(flat.device):
            self._check(self._lib.reni_allreduce_grads(self._comm, flat.data_ptr(), flat.numel(), float(scale),
                                                       torch.cuda.current_stream(flat.device).cuda_stream))
        return flat

    def close(self):
        if getattr(self, "_comm", None) is not None and self._comm:
            self._check(self._lib.reni_rccl_comm_destroy(self._comm))
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def broadcast_(flat: torch.Tensor, src=0):
    """Replicate rank 0's decoder parameters (what DDP does at wrap time, run.py:110)."""
    if world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def allreduce_mean_scalars(values: torch.Tensor):
    """Epoch-end metric sync (self.log_dict(..., sync_dist=True), RENI_module.py:156-163)."""
    w = world_size()
    if w > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
        values = values / w
    return values
