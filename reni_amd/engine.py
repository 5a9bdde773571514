"""Flat-buffer training engine: the fast equivalent of one Lightning training iteration
(training_step -> backward -> DDP all-reduce -> Adam.step; RENI_module.py:80-146,192, run.py:97)
for the FIT_DECODER / AutoDecoder case, without per-parameter Python overhead.

One call to ``step`` = fused forward+loss+backward (one kernel launch) -> ONE RCCL all-reduce of the
flat decoder gradient (mean) -> fused Adam on the flat decoder buffer and on this rank's latent
rows.  The model object keeps owning the parameters (its nn.Parameters are views of the same flat
buffer), so ``state_dict()`` stays valid at any time.
"""
from __future__ import annotations

import torch

from . import dist as rdist
from . import ops


class TrainEngine:
    def __init__(self, model, lr: float, loss_kind: str = "mse", alpha: float = 0.0, beta: float = 0.0, comm=None):
        """comm: an optional ``dist.RcclComm``; the decoder-gradient all-reduce then goes through the library's own
        ``reni_allreduce_grads`` on the compute stream instead of torch.distributed's nccl backend (the same RCCL ring
        either way)."""
        self.comm = comm
        self.model = model
        self.plan = model._plan()
        # FiLM-conditioned models (RENI.py:522-858): the optimised buffer is [net | final_layer | mapping_network]
        self.film = self.plan.conditioning == "film"
        self.flat = model._all_flat() if self.film else model._flat_params()
        assert self.flat.is_cuda, "move the model to the GPU first"
        self.latent = model.Z if hasattr(model, "Z") else model.mu
        self.lr = lr
        self.loss_kind, self.alpha, self.beta = loss_kind, alpha, beta
        self.train_decoder = not model.fixed_decoder
        self.m_dec = torch.zeros_like(self.flat)
        self.v_dec = torch.zeros_like(self.flat)
        self.m_lat = torch.zeros_like(self.latent.data)
        self.v_lat = torch.zeros_like(self.latent.data)
        self.t = 0
        self.world = rdist.world_size()

    def step(self, idx: torch.Tensor, target: torch.Tensor, weight: torch.Tensor, directions: torch.Tensor):
        """idx: rows of this rank's latent table in the batch; target/weight: strided [B,P,3] views.
        Returns the device tensor (loss, mse, prior, cosine) of this rank's batch."""
        if self.film:  # mapping network + fused core + glue backward in one library call (reni_film_model_*)
            n = self.plan.n_params
            terms, dZ, dparams, _, _ = self.plan.film_model_forward_loss_backward(
                self.latent.data[idx], directions, self.flat[:n], self.flat[n:], target, weight, loss_kind=self.loss_kind,
                alpha=self.alpha, beta=self.beta, need_dw=self.train_decoder)
            if dparams is not None:
                dparams = dparams._base  # [d params | d map_params]: one buffer, laid out like self.flat
        else:
            # (the batch's latent rows are gathered inside the prologue kernel: no separate Z[idx] gather)
            terms, dZ, dparams, _ = self.plan.forward_loss_backward(
                self.latent.data, directions, self.flat, target, weight, loss_kind=self.loss_kind, alpha=self.alpha,
                beta=self.beta, need_dw=self.train_decoder, need_dz=True, idx=idx)
        self.t += 1
        inv_w = 1.0 / self.world
        # ONE exchange step: the in-place sum of the flat decoder gradient over the ranks (run.py:97's DDP all-reduce; the 1 / world
        # of its mean rides on Adam's grad_scale).  Either the library's own reni_allreduce_grads on this stream (comm=) or
        # torch.distributed's nccl backend, which orders itself against this stream on the device -- no host wait either way.
        # Latent rows need no communication (dist.py).  FIT_LATENT (frozen decoder): no collective at all.
        if self.train_decoder and self.world > 1:
            if self.comm is not None:
                self.comm.allreduce_(dparams, 1.0)
            else:
                torch.distributed.all_reduce(dparams, op=torch.distributed.ReduceOp.SUM)
        elif self.train_decoder and self.comm is not None:  # (a one-rank communicator: the same call path, for tests)
            self.comm.allreduce_(dparams, 1.0)
        # dense Adam over the whole (owned) latent table, as the reference does (rows outside the batch have zero gradient but
        # still move by momentum -- SURVEY.md Appendix B9); decoder and latent table in ONE launch (reni_adam_step2), as at N = 1
        if self.train_decoder:
            ops.adam_step2(self.flat, dparams, self.m_dec, self.v_dec, self.latent.data, dZ, idx, self.m_lat, self.v_lat,
                           self.t, self.lr, grad_scale=inv_w)
        else:
            ops.adam_rows_step(self.latent.data, dZ, idx, self.m_lat, self.v_lat, self.t, self.lr, grad_scale=inv_w)
        return terms
