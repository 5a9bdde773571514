"""FiLM-conditioned RENI variants: host-side mirror of ``src/models/RENI.py:407-858``.

Same class names, constructor arguments, attributes, ``state_dict`` keys (``net.{i}.layer.*``,
``final_layer.*``, ``mapping_network.network.{2i}.*``, ``Z`` | ``mu``, ``log_var``), ``forward(x, directions)``
dispatch, ``sample_latent``, ``load_state_dict`` remapping and RNG consumption order as the reference, so a
torch seed yields the reference's initial weights.

Where the work runs: everything in ``libreni_hip.so`` through ``reni_film_model_*`` (include/reni_hip.h) -- the
per-SAMPLE FiLM SIREN, loss and backward in the fused kernels, and the per-IMAGE glue in small HIP kernels:

  - the mapping network, evaluated ONCE per image (the reference evaluates it on every pixel of rows it just
    ``repeat``-ed, RENI.py:413-447; the values are identical), ``freq = 15 f + 30``;
  - the first FiLM layer folded into a per-image affine map of the direction: its input
    ``[|d_xz|, d_y, D_xz Z_xz^T]`` (SO2, RENI.py:441) / ``D Z^T`` (SO3, RENI.py:410) is linear in
    ``(dx, dy, dz, r)``, hence ``freq_0 (W_0 x + b_0) + phase_0 = A_b (dx, dy, dz, r, 1)``;
  - their backward: gradients of the mapping network, of the first layer and of the latent codes.

No CPU fallback.  ``_glue`` below restates the per-image math in differentiable torch ops; it is what the
``reni_film_forward`` / ``reni_film_backward`` core entry points expect from a caller that owns the glue, and the
tests use it to check the HIP glue.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch
from torch import nn

from . import ops
from .models import Exp, _RENIConcatBase

# --------------------------------------------------------------------------------------------
# parameter holders / initialisers (names and RNG use as in the reference)
# --------------------------------------------------------------------------------------------


def kaiming_leaky_init(m):  # RENI.py:455-460
    classname = m.__class__.__name__
    if classname.find("Linear") != -1:
        torch.nn.init.kaiming_normal_(m.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")


def frequency_init(freq):  # RENI.py:463-472
    def init(m):
        with torch.no_grad():
            if isinstance(m, nn.Linear):
                num_input = m.weight.size(-1)
                m.weight.uniform_(-np.sqrt(6 / num_input) / freq, np.sqrt(6 / num_input) / freq)

    return init


def first_layer_film_sine_init(m):  # RENI.py:475-479
    with torch.no_grad():
        if isinstance(m, nn.Linear):
            num_input = m.weight.size(-1)
            m.weight.uniform_(-1 / num_input, 1 / num_input)


class CustomMappingNetwork(nn.Module):
    """Latent invariants -> (frequencies, phase_shifts) (RENI.py:482-505)."""

    def __init__(self, in_features, map_hidden_layers, map_hidden_dim, map_output_dim):
        super().__init__()
        network: List[nn.Module] = []
        for _ in range(map_hidden_layers):
            network.append(nn.Linear(in_features, map_hidden_dim))
            network.append(nn.LeakyReLU(0.2, inplace=True))
            in_features = map_hidden_dim
        network.append(nn.Linear(map_hidden_dim, map_output_dim))
        self.network = nn.Sequential(*network)
        self.network.apply(kaiming_leaky_init)
        with torch.no_grad():
            self.network[-1].weight *= 0.25

    def forward(self, z):
        frequencies_offsets = self.network(z)
        half = torch.div(frequencies_offsets.shape[-1], 2, rounding_mode="floor")
        return frequencies_offsets[..., :half], frequencies_offsets[..., half:]


class FiLMLayer(nn.Module):
    """sin(freq * linear(x) + phase_shift) (RENI.py:508-519).  Inside a RENI decoder this module only owns
    ``layer``'s parameters; ``forward`` is kept for callers that evaluate a layer on a hand-built tensor."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.layer = nn.Linear(input_dim, hidden_dim)

    def forward(self, x, freq, phase_shift):
        x = self.layer(x)
        return torch.sin(freq.expand_as(x) * x + phase_shift.expand_as(x))


# --------------------------------------------------------------------------------------------
# autograd glue around the per-sample kernels
# --------------------------------------------------------------------------------------------


class _FilmDecodeFn(torch.autograd.Function):
    """out = model(Z, D).  backward = reni_film_model_backward (forward recomputed inside the fused kernel)."""

    @staticmethod
    def forward(ctx, model, Z, D, n_net, *params):
        flat, mflat = model._flat_params(), model._map_flat()
        out = model._plan().film_model_forward(Z, D, flat, mflat)
        ctx.model, ctx.n_net = model, n_net
        ctx.need_dw = any(p.requires_grad for p in params)
        ctx.save_for_backward(Z, D, flat.detach().clone() if ctx.need_dw else flat.detach(), mflat)
        return out

    @staticmethod
    def backward(ctx, dout):
        Z, D, flat, mflat = ctx.saved_tensors
        dZ, dparams, dmap = ctx.model._plan().film_model_backward(Z, D, flat, mflat, dout, need_dw=ctx.need_dw)
        grads = ctx.model._split_grads(dparams, dmap) if ctx.need_dw else [None] * (ctx.n_net + len(ctx.model._map_params()))
        return (None, dZ, None, None, *grads)


class _FilmFusedLossFn(torch.autograd.Function):
    """(loss, mse, prior, cosine) = criterion(model(Z, D), target, weight[, Z]) with every gradient produced by the same
    call; backward only rescales them."""

    @staticmethod
    def forward(ctx, model, loss_kind, alpha, beta, target, weight, Z, D, n_net, *params):
        need_dw = any(p.requires_grad for p in params)
        terms, dZ, dparams, dmap, _ = model._plan().film_model_forward_loss_backward(
            Z, D, model._flat_params(), model._map_flat(), target, weight, loss_kind=loss_kind, alpha=alpha, beta=beta,
            need_dw=need_dw)
        ctx.model, ctx.n_net, ctx.need_dw = model, n_net, need_dw
        ctx.dZ, ctx.dparams, ctx.dmap = dZ, dparams, dmap
        return terms

    @staticmethod
    def backward(ctx, gterms):
        s = gterms[0]  # only the total carries the fused gradient
        if ctx.need_dw:
            grads = ctx.model._split_grads(ctx.dparams * s, ctx.dmap * s)
        else:
            grads = [None] * (ctx.n_net + len(ctx.model._map_params()))
        return (None, None, None, None, None, None, ctx.dZ * s, None, None, *grads)


# --------------------------------------------------------------------------------------------
# shared base
# --------------------------------------------------------------------------------------------


class _RENIFiLMBase(_RENIConcatBase):
    def _init_film(self, dataset_size, ndims, equivariance, siren_hidden_features, siren_hidden_layers,
                   mapping_network_features, mapping_network_layers, out_features, output_activation, fixed_decoder):
        self.dataset_size = dataset_size
        self.ndims = ndims
        self.equivariance = equivariance
        self.siren_hidden_features = siren_hidden_features
        self.siren_hidden_layers = siren_hidden_layers
        self.mapping_network_features = mapping_network_features
        self.mapping_network_layers = mapping_network_layers
        self.out_features = out_features
        self.output_activation = output_activation
        self.fixed_decoder = fixed_decoder
        from . import encodings
        if self.equivariance == "None":  # RENI.py:549-552 (in_features / mn_in_features as the reference sets them)
            self.InvariantRepresentation = encodings.NoInvarianceFiLM
            self.in_features = self.ndims * 3
            self.mn_in_features = self.ndims
        elif self.equivariance == "SO2":
            self.InvariantRepresentation = encodings.SO2InvariantRepresentationFiLM
            self.in_features = 2 + self.ndims
            self.mn_in_features = self.ndims * self.ndims + self.ndims
        elif self.equivariance == "SO3":
            self.InvariantRepresentation = encodings.SO3InvariantRepresentationFiLM
            self.in_features = self.ndims
            self.mn_in_features = self.ndims * self.ndims
        else:
            raise ValueError(f"unknown equivariance {equivariance!r}")
        self._plans = {}
        self._flat = None

    def _finish_film(self):
        """Everything after init_latent_codes, in the reference's order (RENI.py:563-596)."""
        H = self.siren_hidden_features
        self.net = nn.ModuleList()
        self.net.append(FiLMLayer(self.in_features, H))
        for _ in range(self.siren_hidden_layers - 1):
            self.net.append(FiLMLayer(H, H))
        self.final_layer = nn.Linear(H, self.out_features)
        self.mapping_network = CustomMappingNetwork(self.mn_in_features, self.mapping_network_layers,
                                                    self.mapping_network_features, len(self.net) * H * 2)
        self.net.apply(frequency_init(25))
        self.final_layer.apply(frequency_init(25))
        self.net[0].apply(first_layer_film_sine_init)
        if self.output_activation == "exp":
            self.final_activation = Exp()
        elif self.output_activation == "tanh":
            self.final_activation = nn.Tanh()
        else:
            self.final_activation = nn.Identity()
        if self.fixed_decoder:
            for module in (self.net, self.final_layer, self.mapping_network):
                for param in module.parameters():
                    param.requires_grad = False
        self._reflatten()

    # flat storage: net.* then final_layer.* (the `params` layout of the reni_film_* entry points)
    def _net_params(self) -> List[nn.Parameter]:
        return list(self.net.parameters()) + list(self.final_layer.parameters())

    def _plan(self) -> ops.Plan:
        key = self.compute_dtype
        plan = self._plans.get(key)
        if plan is None:
            if self.equivariance == "None":
                raise NotImplementedError(
                    "FiLM with equivariance 'None' cannot run in the reference either: in_features / mn_in_features "
                    "are swapped relative to NoInvarianceFiLM's outputs (RENI.py:449-452, 549-552)")
            plan = ops.Plan(self.equivariance, self.ndims, self.siren_hidden_features, self.siren_hidden_layers - 1,
                            self.out_features, True, self.output_activation, 1.0, 1.0, key, conditioning="film",
                            mapping_layers=self.mapping_network_layers, mapping_features=self.mapping_network_features)
            assert plan.n_params == sum(p.numel() for p in self._net_params())
            assert plan.n_map_params == sum(p.numel() for p in self._map_params())
            self._plans[key] = plan
        return plan

    # mapping-network parameters in state-dict order = the `map_params` layout of reni_film_model_*
    def _map_params(self) -> List[nn.Parameter]:
        return list(self.mapping_network.parameters())

    # One flat fp32 buffer holds [net.* | final_layer.* | mapping_network.*]: the `params` and `map_params` arguments of the
    # reni_film_model_* entry points are its two halves (no per-call concatenation), and an optimiser can step it whole.
    def _reflatten(self):
        ps = self._net_params() + (self._map_params() if hasattr(self, "mapping_network") else [])
        if not ps:
            return
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for p in ps])
            o = 0
            for p in ps:
                n = p.numel()
                p.data = flat[o:o + n].view(p.shape)
                o += n
        self._flat_all = flat
        self._flat = flat[:sum(p.numel() for p in self._net_params())]

    def _all_flat(self) -> torch.Tensor:
        """[params | map_params], re-flattened if a parameter was re-pointed (``.to(device)``, ``load_state_dict``)"""
        flat = self._flat_params()
        base, o = self._flat_all.data_ptr(), flat.numel()
        ok = self._flat_all.device == flat.device and flat.data_ptr() == base
        for p in self._map_params():
            if not ok or p.data_ptr() != base + 4 * o or p.dtype != torch.float32:
                ok = False
                break
            o += p.numel()
        if not ok:
            self._reflatten()
        return self._flat_all

    def _map_flat(self) -> torch.Tensor:
        return self._all_flat()[self._flat.numel():]

    def _split_grads(self, dparams: torch.Tensor, dmap: torch.Tensor):
        out = self._split_flat(dparams)
        o = 0
        for p in self._map_params():
            n = p.numel()
            out.append(dmap[o:o + n].view(p.shape) if p.requires_grad else None)
            o += n
        return out

    # ---- per-image glue restated in torch (reference for the HIP glue; see the module docstring) ----------------------------------------------
    def _glue(self, Z: torch.Tensor):
        """Z [B,ND,3] -> (A [B,H,8], film [B,L,2,H]); see the module docstring."""
        B, H, nF = Z.shape[0], self.siren_hidden_features, len(self.net)
        W0, b0 = self.net[0].layer.weight, self.net[0].layer.bias
        if self.equivariance == "SO2":
            Z_xz = torch.stack((Z[:, :, 0], Z[:, :, 2]), -1)
            G = torch.bmm(Z_xz, torch.transpose(Z_xz, 1, 2))
            m = torch.cat((G.flatten(start_dim=1), Z[:, :, 1]), 1)           # RENI.py:429-447, one row per image
            W_ip = W0[:, 2:]
            cols = (Z[:, :, 0] @ W_ip.t(), W0[:, 1].expand(B, H), Z[:, :, 2] @ W_ip.t(), W0[:, 0].expand(B, H))
        else:
            G = Z @ torch.transpose(Z, 1, 2)
            m = G.flatten(start_dim=1)                                         # RENI.py:407-415
            cols = (Z[:, :, 0] @ W0.t(), Z[:, :, 1] @ W0.t(), Z[:, :, 2] @ W0.t(), torch.zeros(B, H, device=Z.device, dtype=Z.dtype))
        frequencies, phase_shifts = self.mapping_network(m)
        frequencies = frequencies * 15 + 30                                     # RENI.py:666
        freq = frequencies.reshape(B, nF, H)
        phase = phase_shifts.reshape(B, nF, H)
        f0, p0 = freq[:, 0], phase[:, 0]
        A = torch.stack([f0 * c for c in cols] + [f0 * b0.expand(B, H) + p0], -1)  # [B,H,5]: dx, dy, dz, r, 1
        A = torch.nn.functional.pad(A, (0, 3))
        film = torch.stack((freq[:, 1:], phase[:, 1:]), 2)                      # [B,L,2,H]
        return A.contiguous(), film.contiguous()

    # ---- the hot path ----------------------------------------------------------------------------
    def decode(self, Z: torch.Tensor, directions: torch.Tensor) -> torch.Tensor:
        if Z.shape[0] != directions.shape[0] and directions.shape[0] != 1:
            raise AssertionError("latent batch and directions batch differ")
        ops._require_cuda(Z, directions)
        net = self._net_params()
        return _FilmDecodeFn.apply(self, Z, directions, len(net), *net, *self._map_params())

    def fused_loss(self, Z, directions, target, weight, loss_kind="mse", alpha=0.0, beta=0.0, sparse_weight=False):
        """criterion(model(Z, D), target, weight[, Z]) as ONE library call (mapping network, fused forward+loss+
        backward, glue backward); returns (loss, mse, prior, cosine); ``.backward()`` on element 0 delivers the
        gradients the call already computed.  sparse_weight (a masked weight, see ops.Plan.forward_loss_backward) is accepted
        for the concat models' signature and has no effect here: the FiLM kernels evaluate every tile."""
        ops._require_cuda(Z, directions, target, weight)
        net = self._net_params()
        return _FilmFusedLossFn.apply(self, loss_kind, float(alpha), float(beta), target, weight, Z, directions, len(net),
                                      *net, *self._map_params())

    def forward_with_frequencies_phase_shifts(self, x, frequencies, phase_shifts):
        """The reference's per-pixel evaluation (RENI.py:665-676) for callers that pass hand-built tensors; the
        product path is ``decode``."""
        frequencies = frequencies * 15 + 30
        H = self.siren_hidden_features
        for index, layer in enumerate(self.net):
            x = layer(x, frequencies[..., index * H:(index + 1) * H], phase_shifts[..., index * H:(index + 1) * H])
        return self.final_activation(self.final_layer(x))

    # ---- checkpoint remap (RENI.py:606-626 / 765-785) ------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True):
        new_state_dict = {k[6:]: v for k, v in state_dict.items() if k.startswith("model.")}
        if self.fixed_decoder:
            net_sd = {k[4:]: v for k, v in new_state_dict.items() if k.startswith("net.")}
            map_sd = {k[16:]: v for k, v in new_state_dict.items() if k.startswith("mapping_network.")}
            r = self.net.load_state_dict(net_sd, strict=strict)
            self.mapping_network.load_state_dict(map_sd, strict=strict)
            dev = self.final_layer.weight.device
            self.final_layer.weight = nn.Parameter(new_state_dict["final_layer.weight"].to(dev), requires_grad=False)
            self.final_layer.bias = nn.Parameter(new_state_dict["final_layer.bias"].to(dev), requires_grad=False)
        else:
            r = nn.Module.load_state_dict(self, new_state_dict, strict=strict)
        self._flat_params()
        return r


# --------------------------------------------------------------------------------------------
# public model classes
# --------------------------------------------------------------------------------------------


class RENIAutoDecoderFiLM(_RENIFiLMBase):
    """Mirror of src/models/RENI.py:522-676."""

    def __init__(self, dataset_size, ndims, equivariance, siren_hidden_features, siren_hidden_layers,
                 mapping_network_features, mapping_network_layers, out_features, output_activation, fixed_decoder):
        nn.Module.__init__(self)
        self._init_film(dataset_size, ndims, equivariance, siren_hidden_features, siren_hidden_layers,
                        mapping_network_features, mapping_network_layers, out_features, output_activation, fixed_decoder)
        self.init_latent_codes(self.dataset_size, self.ndims, self.fixed_decoder)
        self._finish_film()

    def init_latent_codes(self, dataset_size, ndims, fixed_decoder=False):
        if fixed_decoder:
            self.Z = nn.Parameter(torch.zeros(dataset_size, ndims, 3))
        else:
            self.Z = nn.Parameter(torch.randn((dataset_size, ndims, 3)))

    def forward(self, x, directions):
        """x: int | list[int] | 1-D index tensor | [B,ND,3] latent tensor (RENI.py:628-663)."""
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


class RENIVariationalAutoDecoderFiLM(_RENIFiLMBase):
    """Mirror of src/models/RENI.py:679-858."""

    def __init__(self, dataset_size, ndims, equivariance, siren_hidden_features, siren_hidden_layers,
                 mapping_network_features, mapping_network_layers, out_features, output_activation, fixed_decoder):
        nn.Module.__init__(self)
        self._init_film(dataset_size, ndims, equivariance, siren_hidden_features, siren_hidden_layers,
                        mapping_network_features, mapping_network_layers, out_features, output_activation, fixed_decoder)
        self.init_latent_codes(self.dataset_size, self.ndims, self.fixed_decoder)
        self._finish_film()

    def sample_latent(self, idx):  # RENI.py:746-752
        mu = self.mu[idx, :, :]
        log_var = self.log_var[idx, :, :]
        std = torch.exp(0.5 * log_var)
        eps = torch.randn_like(std)
        sample = mu + (eps * std)
        return sample, mu, log_var

    def init_latent_codes(self, dataset_size, ndims, fixed_decoder=True):  # RENI.py:754-762
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
        """RENI.py:787-846: index inputs sample a latent unless the decoder is frozen."""
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
