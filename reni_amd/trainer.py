"""Minimal fit loop standing in for ``pl.Trainer.fit`` (run.py:99-110) on the fused path.

Per batch: ``training_step`` -> ``loss.backward()`` -> (data-parallel: ONE all-reduce of the flat
decoder gradient, latent gradients x 1/world) -> ``optimizer.step()``; per epoch: scheduler step,
``training_epoch_end`` metric sync, multi-resolution curriculum.  One process per GPU.
"""
from __future__ import annotations

import torch

from . import dist as rdist


LATENT_NAMES = ("Z", "mu", "log_var")


def _decoder_params(model):
    """Every trainable parameter that is NOT a latent table: the SIREN, and for the FiLM variants also ``final_layer``
    and ``mapping_network`` (reference: DDP all-reduces every trainable parameter, run.py:97)."""
    return [p for n, p in model.named_parameters() if n not in LATENT_NAMES and p.requires_grad]


def _latent_params(model):
    return [(n, getattr(model, n)) for n in LATENT_NAMES if isinstance(getattr(model, n, None), torch.nn.Parameter)]


def sync_decoder_grads(model):
    """All-reduce(mean) the decoder gradient as one flat buffer and scale latent grads by 1/world
    (DDP semantics of run.py:97, see reni_amd/dist.py)."""
    w = rdist.world_size()
    if w == 1:
        return
    ps = _decoder_params(model)
    for p in ps:  # a rank whose batch was empty (fewer owned images) still joins the collective, with zeros
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    if ps:
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        rdist.allreduce_mean_(flat)
        o = 0
        for p in ps:
            n = p.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n
    for _, p in _latent_params(model):
        if p.grad is not None:
            p.grad.mul_(1.0 / w)


def gather_latents(model, optimizer=None, rank=None, world=None):
    """Merge the owner-trained latent rows (and, given the optimiser, their Adam moments) into the full table on every
    rank.  Called at the end of ``fit`` and before any checkpoint, so that ``state_dict()`` taken on rank 0 equals the
    one-process run's (the reference's DDP keeps the full table identical on every rank).  ``rank`` / ``world``: the
    ownership rule ``i % world == rank`` the rows were trained under (default: the process group's)."""
    w = rdist.world_size() if world is None else world
    if w == 1:
        return
    with torch.no_grad():
        for _, p in _latent_params(model):
            rdist.merge_owned_rows_(p.data, rank, world)
            st = optimizer.state.get(p) if optimizer is not None else None
            if st:
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st:
                        rdist.merge_owned_rows_(st[k], rank, world)


def fit(module, max_epochs=None, device=None, batches=None, rank=0, world=1):
    """Run ``module`` (a reni_amd.lightning_module.RENI) for ``max_epochs`` epochs.

    ``batches``: optional explicit list of index lists per epoch (tests); otherwise the module's
    dataloader restricted to the images this rank owns."""
    device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu")
    module.setup()
    module.on_fit_start()
    module.to(device)
    if world > 1:
        assert len(module.dataset) >= world, "fewer images than ranks: a rank would have nothing to report at epoch end"
        for p in _decoder_params(module.model):
            rdist.broadcast_(p.data, 0)
    cfg = module.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    epochs = max_epochs if max_epochs is not None else module.epochs
    history = []
    for epoch in range(epochs):
        module.current_epoch = epoch
        outs = []
        if batches is not None:
            it = []
            if world > 1:  # the merge at the end keeps row i from rank i % world only: an explicit batch must respect that
                for idx in batches:
                    bad = [i for i in idx if i % world != rank]
                    assert not bad, f"rank {rank} of {world} was given images {bad} it does not own (i % world == rank)"
            for idx in batches:
                imgs = torch.stack([module.dataset[i][0] for i in idx])
                it.append((imgs, torch.tensor(idx)))
        else:
            # the same number of steps on every rank (a short rank's trailing batches are short or empty)
            it = []
            for idx in rdist.epoch_batches(len(module.dataset), module.batch_size, rank, world):
                imgs = torch.stack([module.dataset[i][0] for i in idx]) if idx else None
                it.append((imgs, torch.tensor(idx, dtype=torch.long)))
        for bi, (imgs, idx) in enumerate(it):
            opt.zero_grad(set_to_none=True)
            if imgs is not None:
                out = module.training_step((imgs.to(device), idx.to(device)), bi)
                out["loss"].backward()
                outs.append({k: v.detach() for k, v in out.items()})
            else:  # no image left on this rank: the dense latent Adam still takes its (zero-gradient, momentum-only) step
                for _, p in _latent_params(module.model):
                    if p.requires_grad:
                        p.grad = torch.zeros_like(p)
            sync_decoder_grads(module.model)
            opt.step()
            module.global_step += 1
        module.training_epoch_end(outs)
        history.append({k: float(torch.stack([o[k] for o in outs]).mean()) for k in outs[0]} if outs else {})
        if sched is not None:
            sched.step()
        module.maybe_double_resolution()
    gather_latents(module.model, opt, rank if world > 1 else None, world if world > 1 else None)
    return history
