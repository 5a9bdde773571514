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

import os

from . import _lib, _roctx
from . import dist as rdist
from . import ops

if os.environ.get("RENI_ROCTX"):
    _roctx.enabled = True


class TrainEngine:
    def __init__(self, model, lr: float, loss_kind: str = "mse", alpha: float = 0.0, beta: float = 0.0, comm=None,
                 overlap_comm: bool = False, fused_step: bool = True, sparse_weight=False):
        """comm: an optional ``dist.RcclComm``; the decoder-gradient all-reduce then goes through the library's own
        ``reni_allreduce_grads`` on the compute stream instead of torch.distributed's nccl backend (the same RCCL ring
        either way).
        overlap_comm: start the all-reduce of layers >= 2 + head (28 % of the gradient at config 2) on a communication stream as
        soon as their partial reduction has run -- i.e. beside k_reni_dw1 and the latent tails, ~0.13 ms before the fused call's
        work ends (``reni_set_grad_ready_event``) -- and all-reduce the rest (first layer, layer 1) behind the call as before.
        Same sums, element for element (tests/test_gpu_dist.py); what it hides on xGMI is unmeasured (no multi-GPU box so far),
        so it is off by default."""
        self.comm = comm
        # sparse_weight: the step's weight carries an inpainting mask (RENI_module.py:92-94): True = RENI_WEIGHT_SPARSE (bit-equal),
        # "pixels" = RENI_WEIGHT_COMPACT (equal to rounding), see ops.Plan.forward_loss_backward
        self.sparse_weight = sparse_weight
        self.overlap_comm = overlap_comm
        # fused_step: one process, trainable concat decoder -> the whole step is ONE library call (reni_train_step_rows: Adam and the
        # next batch's prologue run beside the backward pass's last kernel; pass the next batch to step(..., next_idx=) to stage it)
        self.fused_step = fused_step
        self._stage = None
        self._comm_stream = None
        self._ev_rest = self._ev_comm_done = None
        self._comm_ev = None        # [(start, end)] torch events around the exchange step while time_comm is on
        self.model = model
        self.plan = model._plan()
        # FiLM-conditioned models (RENI.py:522-858): the optimised buffer is [net | final_layer | mapping_network]
        self.film = self.plan.conditioning == "film"
        self.flat = model._all_flat() if self.film else model._flat_params()
        # (no device check here: every op the step calls raises RENILibraryError for a CPU tensor -- there is no CPU fallback)
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
        if (overlap_comm or comm is not None) and not self.flat.is_cuda:  # (fail HERE, not inside torch.cuda.Stream or RCCL)
            raise _lib.RENILibraryError("TrainEngine(comm= / overlap_comm=) needs the model on a GPU device; there is no CPU fallback")
        # the fused data-parallel step exchanges inside the library: overlap_comm then means its early slice on the library's stream
        self.overlap_dp = bool(overlap_comm and comm is not None and fused_step)
        if self.overlap_dp:
            self.overlap_comm = False
        if self.overlap_comm and not self.film and self.train_decoder:
            dev = self.flat.device
            self._comm_stream = torch.cuda.Stream(dev)
            self._ev_rest, self._ev_comm_done = torch.cuda.Event(), torch.cuda.Event()
            with torch.cuda.device(dev):
                self._ev_rest.record()  # (materialises the hipEvent_t behind the handle)
            H = self.plan.hidden_features
            self._n_head = self.plan.n_first + (H * H + H if self.plan.hidden_layers >= 1 else 0)  # [0, n_head): final only at the end
        else:
            self.overlap_comm = False

    def time_comm(self, on: bool):
        """Bracket every exchange step with events on the compute stream (what the step actually waits for).  time_comm(False) returns
        the average in microseconds (None: no exchange was timed) and stops timing.  On the fused data-parallel step
        (reni_train_step_rows_dp) the exchange is inside the library call: its duration is the library's own event pair
        (ops.PROF_COMM, recorded while ops.profile_enable() is on), never the bracket of the whole call."""
        if on:
            self._comm_ev = []
            return None
        evs, self._comm_ev = self._comm_ev, None
        if not evs:
            if self.comm is not None and self._stage is not None:  # the fused data-parallel step: the library's own event pairs
                ms, n = ops.profile_read(reset=False, kind=ops.PROF_COMM)
                return 1e3 * ms / n if n else None
            return None
        torch.cuda.synchronize(self.flat.device)
        return 1e3 * sum(a.elapsed_time(b) for a, b in evs) / len(evs)

    def _allreduce(self, buf):
        if self.comm is not None:
            self.comm.allreduce_(buf, 1.0)
        else:
            torch.distributed.all_reduce(buf, op=torch.distributed.ReduceOp.SUM)

    def step(self, idx: torch.Tensor, target: torch.Tensor, weight: torch.Tensor, directions: torch.Tensor, next_idx=None):
        """idx: rows of this rank's latent table in the batch; target/weight: strided [B,P,3] views.
        next_idx: the NEXT step's idx, if the caller knows it (a loader that is one batch ahead does): the fused step stages that
        batch's prologue behind this step's backward pass; the next call must then be made with exactly that idx (checked here).
        Returns the device tensor (loss, mse, prior, cosine) of this rank's batch."""
        # the fused step: one process, or -- with the library's own communicator (comm=RcclComm()) -- one rank of a data-parallel job:
        # the SAME call with the exchange inside it (reni_train_step_rows_dp).  torch.distributed's collective cannot be put inside a
        # library call: comm=None at world > 1 keeps the three-call path below.
        if (self.fused_step and not self.film and self.train_decoder and (self.comm is not None or self.world == 1)
                and hasattr(self.plan, "train_step")):
            import ctypes
            if self._stage is None:
                self._stage = {"state": ctypes.c_uint32(0), "expect": None, "shape": None}
            st = self._stage
            shape = (int(idx.numel()), int(directions.shape[-2]))
            # (host-side check by identity only -- comparing contents would cost a device synchronisation per step; the library
            # compares the staged batch's indices with this call's ON THE DEVICE and poisons the step with NaN if they differ)
            if st["state"].value & 1 and (st["shape"] != shape or st["expect"] != (idx.data_ptr(), idx.numel())):
                st["state"].value = 0   # another batch than the one announced: this call runs its own prologue
            self.t += 1
            _roctx.push("reni.step.fused")
            nxt = next_idx.contiguous() if next_idx is not None and int(next_idx.numel()) == shape[0] else None
            # (the exchange is inside the call: nothing to bracket from here -- with ops.profile_enable() the library times its own
            # all-reduce, ops.profile_read(kind=ops.PROF_COMM), which time_comm(False) returns for this path)
            terms, _, _ = self.plan.train_step(self.latent.data, idx.contiguous(), directions, self.flat, target, weight, self.m_dec,
                                               self.v_dec, self.m_lat, self.v_lat, self.t, self.lr, st["state"], idx_next=nxt,
                                               loss_kind=self.loss_kind, alpha=self.alpha, beta=self.beta,
                                               grad_scale=1.0 / self.world, comm=self.comm, overlap=self.overlap_dp)
            st["expect"], st["shape"] = ((nxt.data_ptr(), nxt.numel()) if nxt is not None else None), shape
            _roctx.pop()
            return terms
        if (self.fused_step and not self.film and not self.train_decoder and self.world == 1 and self.comm is None
                and hasattr(self.plan, "latent_step")):
            # a frozen concat decoder, one process: the FIT_LATENT iteration as ONE library call (reni_latent_step_rows)
            self.t += 1
            _roctx.push("reni.step.latent")
            terms, _ = self.plan.latent_step(self.latent.data, idx.contiguous(), directions, self.flat, target, weight, self.m_lat,
                                             self.v_lat, self.t, self.lr, loss_kind=self.loss_kind, alpha=self.alpha, beta=self.beta,
                                             sparse_weight=self.sparse_weight)
            _roctx.pop()
            return terms
        _roctx.push("reni.step.fwd_bwd")
        if self.film:  # mapping network + fused core + glue backward in one library call (reni_film_model_*)
            n = self.plan.n_params
            terms, dZ, dparams, _, _ = self.plan.film_model_forward_loss_backward(
                self.latent.data[idx], directions, self.flat[:n], self.flat[n:], target, weight, loss_kind=self.loss_kind,
                alpha=self.alpha, beta=self.beta, need_dw=self.train_decoder)
            if dparams is not None:
                dparams = dparams._base  # [d params | d map_params]: one buffer, laid out like self.flat
        else:
            exchange = self.train_decoder and (self.world > 1 or self.comm is not None)
            hook = self.overlap_comm and exchange
            if hook:
                _lib.check(_lib.load().reni_set_grad_ready_event(self._ev_rest.cuda_event))
            try:
                # (the batch's latent rows are gathered inside the prologue kernel: no separate Z[idx] gather)
                terms, dZ, dparams, _ = self.plan.forward_loss_backward(
                    self.latent.data, directions, self.flat, target, weight, loss_kind=self.loss_kind, alpha=self.alpha,
                    beta=self.beta, need_dw=self.train_decoder, need_dz=True, idx=idx,
                    **({"sparse_weight": self.sparse_weight} if self.sparse_weight else {}))
            finally:
                if hook:
                    _lib.check(_lib.load().reni_set_grad_ready_event(None))
        _roctx.pop()
        self.t += 1
        inv_w = 1.0 / self.world
        # ONE exchange step: the in-place sum of the flat decoder gradient over the ranks (run.py:97's DDP all-reduce; the 1 / world
        # of its mean rides on Adam's grad_scale).  Either the library's own reni_allreduce_grads on this stream (comm=) or
        # torch.distributed's nccl backend, which orders itself against this stream on the device -- no host wait either way.
        # Latent rows need no communication (dist.py).  FIT_LATENT (frozen decoder): no collective at all.
        if self.train_decoder and (self.world > 1 or self.comm is not None):  # (a one-rank communicator: the same call path, for tests)
            _roctx.push("reni.step.exchange")
            timed = self._comm_ev is not None and self.flat.is_cuda
            if timed or self.overlap_comm:
                cur = torch.cuda.current_stream(self.flat.device)
            if timed:
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(cur)
            if self.overlap_comm and not self.film:
                # layers >= 2 + head: behind the library's event, on the communication stream (it runs beside k_reni_dw1); the rest
                # (first layer 68 %, layer 1) here, behind the whole call.  Two slices of one buffer: the same sums as one call.
                n0 = self._n_head
                self._comm_stream.wait_event(self._ev_rest)
                with torch.cuda.stream(self._comm_stream):
                    self._allreduce(dparams[n0:])
                    self._ev_comm_done.record(self._comm_stream)
                self._allreduce(dparams[:n0])
                cur.wait_event(self._ev_comm_done)
            else:
                self._allreduce(dparams)
            if timed:
                eb.record(cur)
                self._comm_ev.append((ea, eb))
            _roctx.pop()
        # dense Adam over the whole (owned) latent table, as the reference does (rows outside the batch have zero gradient but
        # still move by momentum -- SURVEY.md Appendix B9); decoder and latent table in ONE launch (reni_adam_step2), as at N = 1
        _roctx.push("reni.step.adam")
        if self.train_decoder:
            ops.adam_step2(self.flat, dparams, self.m_dec, self.v_dec, self.latent.data, dZ, idx, self.m_lat, self.v_lat,
                           self.t, self.lr, grad_scale=inv_w)
        else:
            ops.adam_rows_step(self.latent.data, dZ, idx, self.m_lat, self.v_lat, self.t, self.lr, grad_scale=inv_w)
        _roctx.pop()
        return terms
