"""Host-side mirror of the reference's ``src/models/RENI.py`` module API (Cond-by-Concat family).

Same class names, constructor arguments, attributes, ``state_dict`` keys, ``forward(x, directions)``
dispatch, ``sample_latent`` and ``load_state_dict`` remapping as the reference
(/root/reference/src/models/RENI.py:63-399, 861-905) -- but ``forward`` does not build the
concatenated encoding and does not run ``nn.Linear``: it calls the fused HIP kernels of
``libreni_hip.so`` through the C ABI (include/reni_hip.h).  There is no CPU fallback: calling a
model with CPU tensors raises.

The ``nn.Linear`` modules inside ``net`` exist to own the parameters under the reference's key
names (``net.{i}.linear.weight`` ...).  Their storage is one flat fp32 buffer in the reference's
state-dict order, which is exactly the ``params`` argument of the C ABI.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from . import ops

# --------------------------------------------------------------------------------------------
# parameter-holder layers (same names / init as the reference)
# --------------------------------------------------------------------------------------------


class SineLayer(nn.Module):
    """sin(omega_0 * linear(x)) with the SIREN initialisation of src/models/RENI.py:63-87.

    Inside a RENI decoder this module only owns ``linear``'s parameters; the fused kernels do the
    arithmetic.  ``forward`` is kept for callers that evaluate a layer on a hand-built tensor."""

    def __init__(self, in_features, out_features, bias=True, is_first=False, omega_0=30):
        super().__init__()
        self.omega_0 = omega_0
        self.is_first = is_first
        self.in_features = in_features
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        self.init_weights()

    def init_weights(self):
        with torch.no_grad():
            if self.is_first:
                bound = 1 / self.in_features  # RENI.py:79
            else:
                bound = np.sqrt(6 / self.in_features) / self.omega_0  # RENI.py:81-84
            self.linear.weight.uniform_(-bound, bound)

    def forward(self, input):
        return torch.sin(self.omega_0 * self.linear(input))


class Exp(nn.Module):
    """exp output activation.  (The reference's concat models reference a non-existent ``nn.Exp``,
    RENI.py:173-174 -- SURVEY.md Appendix B1; torch.exp semantics are used here.)"""

    def forward(self, x):
        return torch.exp(x)


def in_features_for(equivariance: str, ndims: int) -> int:
    """RENI.py:118-126."""
    if equivariance == "None":
        return ndims * 3 + ndims
    if equivariance == "SO2":
        return 2 * ndims + ndims * ndims + 2
    if equivariance == "SO3":
        return ndims + ndims * ndims
    raise ValueError(f"unknown equivariance {equivariance!r}")


def _build_net(in_features, hidden_features, hidden_layers, out_features, last_layer_linear,
               output_activation, first_omega_0, hidden_omega_0) -> nn.Sequential:
    """Layer stack in the reference's construction order (RENI.py:132-178), so that a given torch
    seed yields the same initial weights as the reference."""
    net: List[nn.Module] = [SineLayer(in_features, hidden_features, is_first=True, omega_0=first_omega_0)]
    for _ in range(hidden_layers):
        net.append(SineLayer(hidden_features, hidden_features, is_first=False, omega_0=hidden_omega_0))
    if last_layer_linear:
        final_linear = nn.Linear(hidden_features, out_features)
        with torch.no_grad():
            bound = np.sqrt(6 / hidden_features) / hidden_omega_0
            final_linear.weight.uniform_(-bound, bound)
        net.append(final_linear)
    else:
        net.append(SineLayer(hidden_features, out_features, is_first=False, omega_0=hidden_omega_0))
    if output_activation == "exp":
        net.append(Exp())
    elif output_activation == "tanh":
        net.append(nn.Tanh())
    return nn.Sequential(*net)


# --------------------------------------------------------------------------------------------
# autograd glue
# --------------------------------------------------------------------------------------------


class _DecodeFn(torch.autograd.Function):
    """out = model(Z, D).  backward = reni_backward (forward recomputed inside the fused kernel)."""

    @staticmethod
    def forward(ctx, model, Z, D, *params):
        flat = model._flat_params()
        plan = model._plan()
        out = plan.forward(Z, D, flat)
        ctx.model = model
        ctx.n_params = len(params)
        ctx.need_dw = any(p.requires_grad for p in params)
        ctx.save_for_backward(Z, D, flat.detach().clone() if ctx.need_dw else flat.detach())
        return out

    @staticmethod
    def backward(ctx, dout):
        Z, D, flat = ctx.saved_tensors
        model = ctx.model
        need_dz = ctx.needs_input_grad[1]
        dZ, dparams = model._plan().backward(Z, D, flat, dout, need_dw=ctx.need_dw, need_dz=need_dz)
        grads = model._split_flat(dparams) if ctx.need_dw else [None] * ctx.n_params
        return (None, dZ, None, *grads)


class _FusedLossFn(torch.autograd.Function):
    """(loss, mse, prior, cosine) = criterion(model(Z, D), target, weight[, Z]) with the gradients
    of ``loss`` produced in the same fused kernel launch; backward only rescales them."""

    @staticmethod
    def forward(ctx, model, loss_kind, alpha, beta, sparse_weight, target, weight, Z, D, *params):
        flat = model._flat_params()
        need_dw = any(p.requires_grad for p in params)
        need_dz = Z.requires_grad
        terms, dZ, dparams, _ = model._plan().forward_loss_backward(
            Z, D, flat, target, weight, loss_kind=loss_kind, alpha=alpha, beta=beta,
            need_dw=need_dw, need_dz=need_dz, sparse_weight=sparse_weight)
        ctx.model = model
        ctx.n_params = len(params)
        ctx.need_dw, ctx.need_dz = need_dw, need_dz
        ctx.dZ, ctx.dparams = dZ, dparams
        return terms

    @staticmethod
    def backward(ctx, gterms):
        # only terms[0] (the total loss) carries the fused gradient
        s = gterms[0]
        dZ = ctx.dZ * s if ctx.need_dz else None
        if ctx.need_dw:
            grads = ctx.model._split_flat(ctx.dparams * s)
        else:
            grads = [None] * ctx.n_params
        return (None, None, None, None, None, None, None, dZ, None, *grads)


# --------------------------------------------------------------------------------------------
# shared decoder base
# --------------------------------------------------------------------------------------------


class _RENIConcatBase(nn.Module):
    #: arithmetic of the dense layers: "f32" (exact fp32 MFMA) or "bf16" (bf16 MFMA, fp32 accumulate)
    compute_dtype = "f32"

    def _init_common(self, dataset_size, ndims, equivariance, hidden_features, hidden_layers,
                     out_features, last_layer_linear, output_activation, first_omega_0,
                     hidden_omega_0, fixed_decoder):
        self.dataset_size = dataset_size
        self.ndims = ndims
        self.equivariance = equivariance
        self.hidden_features = hidden_features
        self.hidden_layers = hidden_layers
        self.out_features = out_features
        self.last_layer_linear = last_layer_linear
        self.output_activation = output_activation
        self.first_omega_0 = first_omega_0
        self.hidden_omega_0 = hidden_omega_0
        self.fixed_decoder = fixed_decoder
        self.in_features = in_features_for(equivariance, ndims)
        from . import encodings
        self.InvariantRepresentation = {"None": encodings.NoInvariance, "SO2": encodings.SO2InvariantRepresentation,
                                        "SO3": encodings.SO3InvariantRepresentation}[equivariance]
        self._plans = {}
        self._flat = None

    def _finish_net(self):
        self.net = _build_net(self.in_features, self.hidden_features, self.hidden_layers, self.out_features,
                              self.last_layer_linear, self.output_activation, self.first_omega_0,
                              self.hidden_omega_0)
        if self.fixed_decoder:  # RENI.py:180-182
            for param in self.net.parameters():
                param.requires_grad = False
        self._reflatten()

    # ---- flat parameter storage --------------------------------------------------------
    def _net_params(self) -> List[nn.Parameter]:
        return list(self.net.parameters())

    def _reflatten(self):
        """Re-point every decoder parameter at a slice of one flat fp32 buffer (reference order)."""
        ps = self._net_params()
        if not ps:
            return
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for p in ps])
            o = 0
            for p in ps:
                n = p.numel()
                p.data = flat[o:o + n].view(p.shape)
                o += n
        self._flat = flat

    def _flat_params(self) -> torch.Tensor:
        ps = self._net_params()
        flat = self._flat
        ok = flat is not None and flat.device == ps[0].device and flat.dtype == torch.float32
        if ok:
            base, o = flat.data_ptr(), 0
            for p in ps:
                if p.data_ptr() != base + 4 * o or p.dtype != torch.float32:
                    ok = False
                    break
                o += p.numel()
        if not ok:
            self._reflatten()
        return self._flat

    def _split_flat(self, flat_grad: torch.Tensor):
        out, o = [], 0
        for p in self._net_params():
            n = p.numel()
            out.append(flat_grad[o:o + n].view(p.shape) if p.requires_grad else None)
            o += n
        return out

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._reflatten()
        return r

    def _plan(self) -> ops.Plan:
        key = self.compute_dtype
        plan = self._plans.get(key)
        if plan is None:
            plan = ops.Plan(self.equivariance, self.ndims, self.hidden_features, self.hidden_layers,
                            self.out_features, self.last_layer_linear, self.output_activation,
                            self.first_omega_0, self.hidden_omega_0, key)
            assert plan.n_params == sum(p.numel() for p in self._net_params())
            self._plans[key] = plan
        return plan

    def set_compute_dtype(self, dtype: str):
        assert dtype in ("f32", "bf16")
        self.compute_dtype = dtype
        return self

    # ---- the hot path --------------------------------------------------------------------
    def decode(self, Z: torch.Tensor, directions: torch.Tensor) -> torch.Tensor:
        """[B,ND,3] x [B|1,P,3] -> [B,P,3]; what ``InvariantRepresentation`` + ``self.net`` compute
        in the reference (RENI.py:232-233)."""
        if Z.shape[0] != directions.shape[0] and directions.shape[0] != 1:
            raise AssertionError("latent batch and directions batch differ")  # RENI.py:213,220
        return _DecodeFn.apply(self, Z, directions, *self._net_params())

    def fused_loss(self, Z, directions, target, weight, loss_kind="mse", alpha=0.0, beta=0.0, sparse_weight=False):
        """criterion(model(Z, D), target, weight[, Z]) as ONE fused forward+loss+backward launch.
        Returns the 4-vector (loss, mse, prior, cosine); ``.backward()`` on element 0 delivers the
        gradients the kernel already computed."""
        return _FusedLossFn.apply(self, loss_kind, float(alpha), float(beta), sparse_weight, target, weight, Z, directions,
                                  *self._net_params())

    # ---- checkpoint remap (RENI.py:190-203 / 347-360) -----------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True):
        new_state_dict = {k[6:]: v for k, v in state_dict.items() if k.startswith("model.")}
        if self.fixed_decoder:
            net_sd = {k[4:]: v for k, v in new_state_dict.items() if k.startswith("net.")}
            r = self.net.load_state_dict(net_sd, strict=strict)
        else:
            r = super().load_state_dict(new_state_dict, strict=strict)
        self._flat_params()
        return r


# --------------------------------------------------------------------------------------------
# public model classes
# --------------------------------------------------------------------------------------------


class RENIAutoDecoder(_RENIConcatBase):
    """Mirror of src/models/RENI.py:90-233."""

    def __init__(self, dataset_size, ndims, equivariance, hidden_features, hidden_layers, out_features,
                 last_layer_linear, output_activation, first_omega_0, hidden_omega_0, fixed_decoder):
        super().__init__()
        self._init_common(dataset_size, ndims, equivariance, hidden_features, hidden_layers, out_features,
                          last_layer_linear, output_activation, first_omega_0, hidden_omega_0, fixed_decoder)
        self.init_latent_codes(self.dataset_size, self.ndims, fixed_decoder=fixed_decoder)  # before net: RNG order
        self._finish_net()

    def init_latent_codes(self, dataset_size, ndims, fixed_decoder=False):
        if fixed_decoder:
            self.Z = nn.Parameter(torch.zeros(dataset_size, ndims, 3))
        else:
            self.Z = nn.Parameter(torch.randn((dataset_size, ndims, 3)))

    def forward(self, x, directions):
        """x: int | list[int] | 1-D index tensor | [B,ND,3] latent tensor (RENI.py:205-233)."""
        if isinstance(x, bool):
            raise NotImplementedError("x must be an int, a list of ints or a torch.Tensor")
        if isinstance(x, int):
            assert len([x]) == directions.shape[0]
            Z = self.Z[[x], :, :]
        elif isinstance(x, list):
            assert len(x) == directions.shape[0]
            Z = self.Z[x, :, :]
        elif isinstance(x, torch.Tensor):
            Z = self.Z[x, :, :] if len(x.shape) == 1 else x
        else:
            raise NotImplementedError(
                "x must be either an int (idx), torch.Tensor (idxs or latent codes) or a list of ints (idxs)")
        return self.decode(Z, directions)


class RENIVariationalAutoDecoder(_RENIConcatBase):
    """Mirror of src/models/RENI.py:236-399."""

    def __init__(self, dataset_size, ndims, equivariance, hidden_features, hidden_layers, out_features,
                 last_layer_linear, output_activation, first_omega_0, hidden_omega_0, fixed_decoder):
        super().__init__()
        self._init_common(dataset_size, ndims, equivariance, hidden_features, hidden_layers, out_features,
                          last_layer_linear, output_activation, first_omega_0, hidden_omega_0, fixed_decoder)
        self.init_latent_codes(self.dataset_size, self.ndims, self.fixed_decoder)
        self._finish_net()

    def sample_latent(self, idx):
        """Reparameterised sample (RENI.py:329-335); stays in host torch so the RNG stream is torch's."""
        mu = self.mu[idx, :, :]
        log_var = self.log_var[idx, :, :]
        std = torch.exp(0.5 * log_var)
        eps = torch.randn_like(std)
        sample = mu + (eps * std)
        return sample, mu, log_var

    def init_latent_codes(self, dataset_size, ndims, fixed_decoder=True):
        # log_var is drawn BEFORE mu (RENI.py:338-345)
        self.log_var = torch.nn.Parameter(torch.normal(-5, 1, size=(dataset_size, ndims, 3)))
        if fixed_decoder:
            self.mu = nn.Parameter(torch.zeros(dataset_size, ndims, 3))
            self.log_var.requires_grad = False
        else:
            self.mu = nn.Parameter(torch.randn((dataset_size, ndims, 3)))

    def _latent_for(self, idx):
        if self.fixed_decoder:
            return self.mu[idx, :, :]
        Z, _, _ = self.sample_latent(idx)
        return Z

    def forward(self, x, directions):
        """RENI.py:362-399: index inputs sample a latent unless the decoder is frozen."""
        if isinstance(x, bool):
            raise NotImplementedError("x must be an int, a list of ints or a torch.Tensor")
        if isinstance(x, int):
            assert len([x]) == directions.shape[0]
            Z = self._latent_for([x])
        elif isinstance(x, list):
            assert len(x) == directions.shape[0]
            Z = self._latent_for(x)
        elif isinstance(x, torch.Tensor):
            Z = self._latent_for(x) if len(x.shape) == 1 else x
        else:
            raise NotImplementedError(
                "x must be either an int (idx), torch.Tensor (idxs or latent codes) or a list of ints (idxs)")
        return self.decode(Z, directions)


def get_model(config, dataset_size, task):
    """config -> model (src/models/RENI.py:861-933).  ``config`` is any object exposing
    ``config.RENI.<KEY>`` attributes (yacs CfgNode, SimpleNamespace, ...)."""
    r = config.RENI
    fixed_decoder = True if task in ["FIT_LATENT", "FIT_INVERSE"] else False  # RENI.py:874
    if r.CONDITIONING == "Cond-by-Concat":
        cls = {"AutoDecoder": RENIAutoDecoder, "VariationalAutoDecoder": RENIVariationalAutoDecoder}.get(r.MODEL_TYPE)
        if cls is None:
            return None  # the reference falls through and returns None for unknown types
        model = cls(dataset_size, r.LATENT_DIMENSION, r.EQUIVARIANCE, r.HIDDEN_FEATURES, r.HIDDEN_LAYERS,
                    r.OUT_FEATURES, r.LAST_LAYER_LINEAR, r.OUTPUT_ACTIVATION, r.FIRST_OMEGA_0, r.HIDDEN_OMEGA_0,
                    fixed_decoder)
        dtype = getattr(r, "COMPUTE_DTYPE", None)
        if dtype:
            model.set_compute_dtype(dtype)
        return model
    if r.CONDITIONING == "FiLM":
        from .film import RENIAutoDecoderFiLM, RENIVariationalAutoDecoderFiLM
        cls = {"AutoDecoder": RENIAutoDecoderFiLM, "VariationalAutoDecoder": RENIVariationalAutoDecoderFiLM}.get(r.MODEL_TYPE)
        if cls is None:
            return None
        model = cls(dataset_size, r.LATENT_DIMENSION, r.EQUIVARIANCE, r.HIDDEN_FEATURES, r.HIDDEN_LAYERS,
                    r.MAPPING_FEATURES, r.MAPPING_LAYERS, r.OUT_FEATURES, r.OUTPUT_ACTIVATION, fixed_decoder)
        dtype = getattr(r, "COMPUTE_DTYPE", None)
        if dtype:
            model.set_compute_dtype(dtype)
        return model
    return None
