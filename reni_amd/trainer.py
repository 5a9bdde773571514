"""Minimal fit loop standing in for ``pl.Trainer.fit`` (run.py:99-110) on the fused path.

Per batch: ``training_step`` -> ``loss.backward()`` -> (data-parallel: ONE all-reduce of the flat
decoder gradient, latent gradients x 1/world) -> ``optimizer.step()``; per epoch: scheduler step,
``training_epoch_end`` metric sync, multi-resolution curriculum.  One process per GPU.
"""
from __future__ import annotations

import torch

from . import dist as rdist


def _decoder_params(model):
    return [p for p in model.net.parameters() if p.requires_grad]


def sync_decoder_grads(model):
    """All-reduce(mean) the decoder gradient as one flat buffer and scale latent grads by 1/world
    (DDP semantics of run.py:97, see reni_amd/dist.py)."""
    w = rdist.world_size()
    if w == 1:
        return
    ps = [p for p in _decoder_params(model) if p.grad is not None]
    if ps:
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        rdist.allreduce_mean_(flat)
        o = 0
        for p in ps:
            n = p.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n
    for name in ("Z", "mu", "log_var"):
        p = getattr(model, name, None)
        if p is not None and p.grad is not None:
            p.grad.mul_(1.0 / w)


def fit(module, max_epochs=None, device=None, batches=None, rank=0, world=1):
    """Run ``module`` (a reni_amd.lightning_module.RENI) for ``max_epochs`` epochs.

    ``batches``: optional explicit list of index lists per epoch (tests); otherwise the module's
    dataloader restricted to the images this rank owns."""
    device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu")
    module.setup()
    module.on_fit_start()
    module.to(device)
    if world > 1:
        for p in module.model.net.parameters():
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
            for idx in batches:
                imgs = torch.stack([module.dataset[i][0] for i in idx])
                it.append((imgs, torch.tensor(idx)))
        else:
            own = rdist.owned_indices(len(module.dataset), rank, world)
            bs = module.batch_size
            it = []
            for s in range(0, len(own), bs):
                idx = own[s:s + bs]
                it.append((torch.stack([module.dataset[i][0] for i in idx]), torch.tensor(idx)))
        for bi, (imgs, idx) in enumerate(it):
            out = module.training_step((imgs.to(device), idx.to(device)), bi)
            opt.zero_grad(set_to_none=True)
            out["loss"].backward()
            sync_decoder_grads(module.model)
            opt.step()
            module.global_step += 1
            outs.append({k: v.detach() for k, v in out.items()})
        module.training_epoch_end(outs)
        history.append({k: float(torch.stack([o[k] for o in outs]).mean()) for k in outs[0]})
        if sched is not None:
            sched.step()
        module.maybe_double_resolution()
    return history
