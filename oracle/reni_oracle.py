"""CPU oracle for the RENI forward/training hot path.

*** TEST INFRASTRUCTURE ONLY ***  Nothing under ``reni_amd/`` may import this module.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and only as the checker / the timed CPU baseline -- never as the product path.

Pinning status: PINNED.  Every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by importing the reference's own ``src/models/RENI.py`` and
``src/utils/loss_functions.py`` in the build container (generator: ``tests/golden/make_golden.py``,
outputs committed as ``tests/golden/*.npz``).  The reference ships no tests of its own
(SURVEY.md section 4), so those generated vectors plus the rotation-invariance identities are the
only pins that exist.

Two independent restatements live here:

* "reference-shaped" (plain torch, autograd): the same op sequence as the reference --
  materialised concatenated encoding, ``linear`` + ``sin`` per layer, autograd backward.
  This is what ``bench.py`` times as ``cpu_baseline`` (kind = "port").
* "factored" (numpy, float64, hand-derived backward): the algebra the HIP kernels implement --
  the per-image constant columns of the first layer folded into a per-image affine map of the
  direction, and the backward pass written out by hand.  It checks the
  derivation independently of autograd.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# grids  (src/utils/utils.py:46-78)
# --------------------------------------------------------------------------------------


def _pixel_centres(sidelen: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """u (width axis) and v (height axis) pixel-centre coordinates, fp32, same op order as
    src/utils/utils.py:50-51 so the result is bit-identical."""
    half = sidelen // 2
    u = (torch.linspace(1, sidelen, steps=sidelen) - 0.5) / half
    v = (torch.linspace(1, half, steps=half) - 0.5) / half
    return u, v


def get_directions(sidelen: int) -> torch.Tensor:
    """Equirectangular unit directions, [1, sidelen/2*sidelen, 3] (src/utils/utils.py:46-65).
    Row-major over (row j, col i); d = (sin(phi) sin(theta), cos(phi), -sin(phi) cos(theta))."""
    u, v = _pixel_centres(sidelen)
    half = sidelen // 2
    theta = (np.pi * (u - 1)).repeat(half)  # col index varies fastest
    phi = (np.pi * v).repeat_interleave(sidelen)
    d = torch.stack(
        (torch.sin(phi) * torch.sin(theta), torch.cos(phi), -torch.sin(phi) * torch.cos(theta)), -1
    )
    return d.unsqueeze(0)


def get_sineweight(sidelen: int) -> torch.Tensor:
    """sin(polar angle) per pixel replicated to 3 channels, [1, P, 3] (src/utils/utils.py:68-78)."""
    _, v = _pixel_centres(sidelen)
    phi = (np.pi * v).repeat_interleave(sidelen)
    return torch.sin(phi).unsqueeze(1).repeat(1, 3).unsqueeze(0)


def get_mask_from_array(mask_hw_c: np.ndarray, sidelen: int) -> torch.Tensor:
    """Nearest-neighbour resize of a uint8 mask image [Hs, Ws, C] to (sidelen/2, sidelen) and
    flatten to [1, P, 3] in {0,1} (src/utils/utils.py:81-91; torchvision is absent, so the
    resize is restated with the floor(dst*scale) source-index rule torch's 'nearest' uses)."""
    m = torch.from_numpy(np.asarray(mask_hw_c)).float() / 255.0
    if m.ndim == 2:
        m = m.unsqueeze(-1)
    if m.shape[-1] == 1:
        m = m.repeat(1, 1, 3)
    hs, ws = m.shape[0], m.shape[1]
    ht, wt = sidelen // 2, sidelen
    ri = torch.clamp((torch.arange(ht).float() * (hs / ht)).floor().long(), max=hs - 1)
    ci = torch.clamp((torch.arange(wt).float() * (ws / wt)).floor().long(), max=ws - 1)
    out = m[ri][:, ci][..., :3]
    return out.reshape(-1, 3).unsqueeze(0)


# --------------------------------------------------------------------------------------
# invariant encodings  (src/models/RENI.py:23-60)
# --------------------------------------------------------------------------------------


def in_features(equivariance: str, ndims: int) -> int:
    """Width of the concatenated MLP input (src/models/RENI.py:118-126)."""
    if equivariance == "None":
        return 4 * ndims
    if equivariance == "SO2":
        return 2 * ndims + ndims * ndims + 2
    if equivariance == "SO3":
        return ndims + ndims * ndims
    raise ValueError(equivariance)


def encode(equivariance: str, Z: torch.Tensor, D: torch.Tensor) -> torch.Tensor:
    """Materialised per-sample MLP input [B, P, F_in]; column order as the reference's cat."""
    B, P = D.shape[0], D.shape[1]
    if equivariance == "SO3":  # RENI.py:23-28 : [D Z^T | vec(Z Z^T)]
        ip = D @ Z.transpose(1, 2)
        G = Z @ Z.transpose(1, 2)
        return torch.cat((ip, G.reshape(B, 1, -1).expand(B, P, -1)), 2)
    if equivariance == "None":  # RENI.py:56-60 : [D Z^T | vec(Z)]
        ip = D @ Z.transpose(1, 2)
        return torch.cat((ip, Z.reshape(B, 1, -1).expand(B, P, -1)), 2)
    if equivariance == "SO2":  # RENI.py:31-53 : [D_xz Z_xz^T | vec(Z_xz Z_xz^T) | |d_xz| | Z_y | d_y]
        Zxz = Z[:, :, [0, 2]]
        Dxz = D[:, :, [0, 2]]
        G = Zxz @ Zxz.transpose(1, 2)
        ip = Dxz @ Zxz.transpose(1, 2)
        r = torch.sqrt(D[:, :, 0] ** 2 + D[:, :, 2] ** 2).unsqueeze(2)
        zy = Z[:, :, 1].unsqueeze(1).expand(B, P, -1)
        dy = D[:, :, 1].unsqueeze(2)
        return torch.cat((ip, G.reshape(B, 1, -1).expand(B, P, -1), r, zy, dy), 2)
    raise ValueError(equivariance)


# --------------------------------------------------------------------------------------
# decoder description + reference-shaped forward  (src/models/RENI.py:63-87,132-178)
# --------------------------------------------------------------------------------------


class DecoderSpec:
    """Hyper-parameters of the conditional SIREN (constructor args of RENIAutoDecoder,
    src/models/RENI.py:91-116)."""

    def __init__(self, ndims, equivariance="SO2", hidden_features=128, hidden_layers=5,
                 out_features=3, last_layer_linear=True, output_activation="tanh",
                 first_omega_0=30.0, hidden_omega_0=30.0):
        self.ndims = ndims
        self.equivariance = equivariance
        self.hidden_features = hidden_features
        self.hidden_layers = hidden_layers
        self.out_features = out_features
        self.last_layer_linear = last_layer_linear
        self.output_activation = output_activation
        self.first_omega_0 = float(first_omega_0)
        self.hidden_omega_0 = float(hidden_omega_0)
        self.in_features = in_features(equivariance, ndims)

    def param_keys(self) -> List[str]:
        """state_dict keys of ``net`` in order (verified there)."""
        L = self.hidden_layers
        keys = []
        for i in range(L + 1):
            keys += [f"net.{i}.linear.weight", f"net.{i}.linear.bias"]
        if self.last_layer_linear:
            keys += [f"net.{L + 1}.weight", f"net.{L + 1}.bias"]
        else:
            keys += [f"net.{L + 1}.linear.weight", f"net.{L + 1}.linear.bias"]
        return keys

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        H, L = self.hidden_features, self.hidden_layers
        keys = self.param_keys()
        shapes = {keys[0]: (H, self.in_features), keys[1]: (H,)}
        for i in range(1, L + 1):
            shapes[keys[2 * i]] = (H, H)
            shapes[keys[2 * i + 1]] = (H,)
        shapes[keys[-2]] = (self.out_features, H)
        shapes[keys[-1]] = (self.out_features,)
        return shapes


def init_params(spec: DecoderSpec, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """SIREN initialisation with the reference's distributions (src/models/RENI.py:76-84,153-160):
    first layer W ~ U(+-1/in); other layers and the linear head W ~ U(+-sqrt(6/in)/omega);
    biases keep nn.Linear's default U(+-1/sqrt(in)).  (The RNG *stream* differs from the reference
    because nn.Linear draws and discards a kaiming weight first; seed-level parity is provided by
    the golden state_dicts instead.)"""
    shapes = spec.param_shapes()
    keys = spec.param_keys()
    out = {}
    for n in range(len(keys) // 2):
        wk, bk = keys[2 * n], keys[2 * n + 1]
        fan_in = shapes[wk][1]
        if n == 0:
            bound = 1.0 / fan_in
        else:
            bound = math.sqrt(6.0 / fan_in) / spec.hidden_omega_0
        out[wk] = (torch.rand(shapes[wk], generator=generator) * 2 - 1) * bound
        out[bk] = (torch.rand(shapes[bk], generator=generator) * 2 - 1) / math.sqrt(fan_in)
    return out


def decoder_forward(spec: DecoderSpec, params: Dict[str, torch.Tensor], x: torch.Tensor,
                    return_preacts: bool = False):
    """x [B,P,F_in] -> out [B,P,3]: sin(omega(Wx+b)) per SineLayer (RENI.py:86-87), then the head
    (linear RENI.py:153-162 or a sine layer :164-171) and the output activation (:173-176;
    "exp" is given torch.exp semantics)."""
    keys = spec.param_keys()
    L = spec.hidden_layers
    pre = []
    h = x
    for i in range(L + 1):
        a = torch.nn.functional.linear(h, params[keys[2 * i]], params[keys[2 * i + 1]])
        pre.append(a)
        omega = spec.first_omega_0 if i == 0 else spec.hidden_omega_0
        h = torch.sin(omega * a)
    y = torch.nn.functional.linear(h, params[keys[-2]], params[keys[-1]])
    if not spec.last_layer_linear:
        y = torch.sin(spec.hidden_omega_0 * y)
    if spec.output_activation == "tanh":
        y = torch.tanh(y)
    elif spec.output_activation == "exp":
        y = torch.exp(y)
    if return_preacts:
        return y, pre
    return y


def reni_forward(spec: DecoderSpec, params: Dict[str, torch.Tensor], Z: torch.Tensor,
                 D: torch.Tensor) -> torch.Tensor:
    """model(Z, D) for a 3-D latent tensor (src/models/RENI.py:225-233)."""
    return decoder_forward(spec, params, encode(spec.equivariance, Z, D))


# --------------------------------------------------------------------------------------
# losses  (src/utils/loss_functions.py)
# --------------------------------------------------------------------------------------


def weighted_mse(out, target, weight):
    """sum over batch of mean over (P*3) of weight*(out-target)^2 (loss_functions.py:6-13)."""
    return (((out - target) ** 2) * weight).reshape(out.shape[0], -1).mean(1).sum(0)


def kld(mu, log_var, z_dims=1):
    """loss_functions.py:16-22."""
    k = -0.5 * (1 + log_var - mu.pow(2) - log_var.exp()).reshape(mu.shape[0], -1).sum(1)
    return (k / z_dims).sum(0)


def weighted_cosine(out, target, weight):
    """loss_functions.py:25-32 -- cosine similarity is taken over the PIXEL axis (dim=1) per
    channel, and multiplied by the weight of pixel 0; eps=1e-20."""
    cs = torch.nn.functional.cosine_similarity(out, target, dim=1, eps=1e-20)  # [B,3]
    return (1 - (cs * weight[:, 0]).mean(1)).sum(0)


def train_loss(out, target, weight):
    """RENITrainLoss (loss_functions.py:39-45)."""
    return weighted_mse(out, target, weight)


def vad_train_loss(out, target, weight, mu, log_var, beta, z_dims):
    """RENIVADTrainLoss (loss_functions.py:47-58)."""
    m = weighted_mse(out, target, weight)
    k = beta * kld(mu, log_var, z_dims)
    return m + k, m, k


def test_loss(out, target, weight, Z, alpha, beta):
    """RENITestLoss (loss_functions.py:60-71) -> (loss, mse, prior, cosine)."""
    m = weighted_mse(out, target, weight)
    p = alpha * torch.pow(Z, 2).sum()
    c = beta * weighted_cosine(out, target, weight)
    return m + p + c, m, p, c


test_loss.__test__ = False  # not a pytest test


# --------------------------------------------------------------------------------------
# one full reference-shaped step: forward + loss + autograd backward
# --------------------------------------------------------------------------------------


def fwd_loss_bwd(spec: DecoderSpec, params: Dict[str, torch.Tensor], Z: torch.Tensor,
                 D: torch.Tensor, target: torch.Tensor, weight: torch.Tensor,
                 loss_kind: str = "mse", alpha: float = 0.0, beta: float = 0.0,
                 need_dw: bool = True):
    """Reference-shaped fwd+loss+bwd.  Returns dict(out, loss_terms, dZ, grads{key: tensor})."""
    Zr = Z.detach().clone().requires_grad_(True)
    ps = {k: v.detach().clone().requires_grad_(need_dw) for k, v in params.items()}
    out = reni_forward(spec, ps, Zr, D)
    if loss_kind == "mse":
        loss = train_loss(out, target, weight)
        terms = (loss, loss, torch.zeros(()), torch.zeros(()))
    elif loss_kind == "test":
        terms = test_loss(out, target, weight, Zr, alpha, beta)
        loss = terms[0]
    else:
        raise ValueError(loss_kind)
    loss.backward()
    return {
        "out": out.detach(),
        "loss_terms": tuple(float(t.detach()) for t in terms),
        "dZ": Zr.grad.detach(),
        "grads": {k: v.grad.detach() for k, v in ps.items()} if need_dw else {},
    }


def factored_torch_fwd_loss_bwd(spec: DecoderSpec, params: Dict[str, torch.Tensor], Z: torch.Tensor,
                                D: torch.Tensor, target: torch.Tensor, weight: torch.Tensor, need_dw: bool = True):
    """fp32 torch + autograd on the FACTORED algebra (variant B): the per-image constant
    columns of the reference's concatenated input (RENI.py:27,51,59) are folded into a per-image affine map
    a0 = A_b (dx, dy, dz, r, 1), so nothing of width F_in is materialised per sample.  Same layers and loss as
    ``fwd_loss_bwd`` (WeightedMSE); checked against it in tests/test_oracle_golden.py.  This is the "factored" figure
    of bench.py's cpu_baseline: what a CPU does with the algorithm the HIP kernels implement."""
    keys = spec.param_keys()
    Zr = Z.detach().clone().requires_grad_(True)
    ps = {k: v.detach().clone().requires_grad_(need_dw) for k, v in params.items()}
    W0, b0 = ps[keys[0]], ps[keys[1]]
    nd, H = spec.ndims, spec.hidden_features
    B, P = Zr.shape[0], D.shape[1]
    Dd = D if D.shape[0] == B else D.expand(B, P, 3)
    r = torch.sqrt(Dd[..., 0] ** 2 + Dd[..., 2] ** 2)
    X5 = torch.stack((Dd[..., 0], Dd[..., 1], Dd[..., 2], r, torch.ones_like(r)), -1)  # [B,P,5]
    if spec.equivariance == "SO2":
        W_ip, W_G, w_r = W0[:, :nd], W0[:, nd:nd + nd * nd], W0[:, nd + nd * nd]
        W_zy, w_dy = W0[:, nd + nd * nd + 1:2 * nd + nd * nd + 1], W0[:, 2 * nd + nd * nd + 1]
        Zxz = Zr[:, :, [0, 2]]
        U = torch.einsum("hn,bnk->bhk", W_ip, Zxz)  # [B,H,2]
        G = Zxz @ Zxz.transpose(1, 2)
        c = G.reshape(B, -1) @ W_G.T + Zr[:, :, 1] @ W_zy.T + b0
        A = torch.stack((U[..., 0], w_dy.expand(B, H), U[..., 1], w_r.expand(B, H), c), -1)  # [B,H,5]
    else:
        W_ip, W_c = W0[:, :nd], W0[:, nd:]
        U = torch.einsum("hn,bnk->bhk", W_ip, Zr)  # [B,H,3]
        const = (Zr @ Zr.transpose(1, 2)).reshape(B, -1) if spec.equivariance == "SO3" else Zr.reshape(B, -1)
        c = const @ W_c.T + b0
        A = torch.cat((U, torch.zeros(B, H, 1), c.unsqueeze(-1)), -1)
    h = torch.sin(spec.first_omega_0 * (X5 @ A.transpose(1, 2)))
    L = spec.hidden_layers
    for i in range(1, L + 1):
        h = torch.sin(spec.hidden_omega_0 * torch.nn.functional.linear(h, ps[keys[2 * i]], ps[keys[2 * i + 1]]))
    y = torch.nn.functional.linear(h, ps[keys[2 * L + 2]], ps[keys[2 * L + 3]])
    if not spec.last_layer_linear:
        y = torch.sin(spec.hidden_omega_0 * y)
    out = torch.tanh(y) if spec.output_activation == "tanh" else torch.exp(y) if spec.output_activation == "exp" else y
    loss = train_loss(out, target, weight)
    loss.backward()
    return {"out": out.detach(), "loss_terms": (float(loss.detach()),) * 2 + (0.0, 0.0), "dZ": Zr.grad.detach(),
            "grads": {k: v.grad.detach() for k, v in ps.items()} if need_dw else {}}


# --------------------------------------------------------------------------------------
# factored float64 restatement with the hand-derived backward
# --------------------------------------------------------------------------------------


def first_layer_split(spec: DecoderSpec, W0: np.ndarray):
    """Column blocks of the first-layer weight in the reference's concat order
    (RENI.py:27,51,59): returns (W_ip, W_const, w_r, w_dy) with absent pieces = None."""
    nd = spec.ndims
    if spec.equivariance == "SO2":
        o = 0
        W_ip = W0[:, o:o + nd]; o += nd
        W_G = W0[:, o:o + nd * nd]; o += nd * nd
        w_r = W0[:, o]; o += 1
        W_zy = W0[:, o:o + nd]; o += nd
        w_dy = W0[:, o]
        return W_ip, (W_G, W_zy), w_r, w_dy
    if spec.equivariance == "SO3":
        return W0[:, :nd], (W0[:, nd:],), None, None
    return W0[:, :nd], (W0[:, nd:],), None, None  # "None": const block multiplies vec(Z)


def per_image_affine(spec: DecoderSpec, W0: np.ndarray, b0: np.ndarray, Z: np.ndarray) -> np.ndarray:
    """A_b [B, H, 5] such that the first-layer pre-activation is
    a0 = A_b @ (dx, dy, dz, r, 1), r = sqrt(dx^2 + dz^2).  Folds every per-image-constant input
    column (Gram entries, Z_y / vec(Z)) and the bias into column 4, and the inner-product columns
    into the three direction columns."""
    B, nd = Z.shape[0], spec.ndims
    H = W0.shape[0]
    A = np.zeros((B, H, 5), dtype=np.float64)
    W_ip, consts, w_r, w_dy = first_layer_split(spec, W0)
    for b in range(B):
        z = Z[b]
        if spec.equivariance == "SO2":
            zxz = z[:, [0, 2]]
            U = W_ip @ zxz  # H x 2
            A[b, :, 0] = U[:, 0]
            A[b, :, 2] = U[:, 1]
            A[b, :, 1] = w_dy
            A[b, :, 3] = w_r
            G = zxz @ zxz.T
            A[b, :, 4] = consts[0] @ G.reshape(-1) + consts[1] @ z[:, 1] + b0
        elif spec.equivariance == "SO3":
            A[b, :, 0:3] = W_ip @ z
            G = z @ z.T
            A[b, :, 4] = consts[0] @ G.reshape(-1) + b0
        else:
            A[b, :, 0:3] = W_ip @ z
            A[b, :, 4] = consts[0] @ z.reshape(-1) + b0
    return A


def factored_fwd_bwd(spec: DecoderSpec, params: Dict[str, np.ndarray], Z: np.ndarray, D: np.ndarray,
                     target: Optional[np.ndarray] = None, weight: Optional[np.ndarray] = None,
                     dout: Optional[np.ndarray] = None, loss_kind: str = "mse",
                     alpha: float = 0.0, beta: float = 0.0, need_dw: bool = True):
    """float64 factored forward and hand-written backward.

    D may be [1,P,3] (shared grid) or [B,P,3]; weight [1|B, P, 3]; target [B,P,3].
    Returns dict(out, loss_terms, dZ, grads, A, dA).  Either (target, weight) or dout is given."""
    f8 = lambda a: np.asarray(a, dtype=np.float64)
    keys = spec.param_keys()
    P_ = {k: f8(v) for k, v in params.items()}
    Z = f8(Z); D = f8(D)
    B, nd = Z.shape[0], spec.ndims
    P = D.shape[1]
    if D.shape[0] == 1 and B > 1:
        D = np.broadcast_to(D, (B, P, 3))
    L = spec.hidden_layers
    H = spec.hidden_features
    W0, b0 = P_[keys[0]], P_[keys[1]]
    A = per_image_affine(spec, W0, b0, Z)
    r = np.sqrt(D[..., 0] ** 2 + D[..., 2] ** 2)
    X5 = np.stack((D[..., 0], D[..., 1], D[..., 2], r, np.ones_like(r)), -1)  # B,P,5
    omegas = [spec.first_omega_0] + [spec.hidden_omega_0] * L
    pre, hs = [], []
    a = np.einsum("bhk,bpk->bph", A, X5)
    for i in range(L + 1):
        if i > 0:
            a = hs[-1] @ P_[keys[2 * i]].T + P_[keys[2 * i + 1]]
        pre.append(a)
        hs.append(np.sin(omegas[i] * a))
    Wo, bo = P_[keys[-2]], P_[keys[-1]]
    y_lin = hs[-1] @ Wo.T + bo
    y = y_lin if spec.last_layer_linear else np.sin(spec.hidden_omega_0 * y_lin)
    if spec.output_activation == "tanh":
        out = np.tanh(y)
    elif spec.output_activation == "exp":
        out = np.exp(y)
    else:
        out = y
    res = {"out": out, "A": A}
    if target is None and dout is None:
        return res
    # ---- loss and d(loss)/d(out)
    if dout is not None:
        g_out = f8(dout)
        terms = (0.0, 0.0, 0.0, 0.0)
    else:
        t = f8(target); s = f8(weight)
        if s.shape[0] == 1 and B > 1:
            s = np.broadcast_to(s, (B, P, 3))
        mse = float(((out - t) ** 2 * s).reshape(B, -1).mean(1).sum())
        g_out = 2.0 * s * (out - t) / (3.0 * P)
        prior = cosv = 0.0
        if loss_kind == "test":
            prior = alpha * float((Z ** 2).sum())
            so_t = (out * t).sum(1); n_o = np.sqrt((out ** 2).sum(1)); n_t = np.sqrt((t ** 2).sum(1))
            den = np.maximum(n_o * n_t, 1e-20)  # [B,3]
            cs = so_t / den
            s0 = s[:, 0, :]
            cosv = beta * float((1.0 - (cs * s0).mean(1)).sum())
            coef = -beta * s0 / 3.0  # [B,3]
            g_out = g_out + coef[:, None, :] * (t / den[:, None, :]
                                                - cs[:, None, :] * out / (n_o ** 2)[:, None, :])
        terms = (mse + prior + cosv, mse, prior, cosv)
    res["loss_terms"] = terms
    # ---- output activation and head
    if spec.output_activation == "tanh":
        g_y = g_out * (1.0 - out ** 2)
    elif spec.output_activation == "exp":
        g_y = g_out * out
    else:
        g_y = g_out
    if not spec.last_layer_linear:
        g_y = g_y * spec.hidden_omega_0 * np.cos(spec.hidden_omega_0 * y_lin)
    grads = {}
    if need_dw:
        grads[keys[-2]] = np.einsum("bpo,bph->oh", g_y, hs[-1])
        grads[keys[-1]] = g_y.sum((0, 1))
    g_h = g_y @ Wo
    # ---- hidden layers L..1
    for i in range(L, 0, -1):
        g_a = g_h * omegas[i] * np.cos(omegas[i] * pre[i])
        if need_dw:
            grads[keys[2 * i]] = np.einsum("bpo,bpi->oi", g_a, hs[i - 1])
            grads[keys[2 * i + 1]] = g_a.sum((0, 1))
        g_h = g_a @ P_[keys[2 * i]]
    g_a0 = g_h * omegas[0] * np.cos(omegas[0] * pre[0])
    dA = np.einsum("bph,bpk->bhk", g_a0, X5)  # per image, H x 5
    res["dA"] = dA
    # ---- per-image tail: dA -> dZ, dW0, db0
    dZ = np.zeros_like(Z)
    dW0 = np.zeros_like(W0)
    W_ip, consts, w_r, w_dy = first_layer_split(spec, W0)
    for b in range(B):
        z = Z[b]
        g_c = dA[b, :, 4]
        if spec.equivariance == "SO2":
            zxz = z[:, [0, 2]]
            dU = dA[b][:, [0, 2]]  # H x 2
            dZxz = W_ip.T @ dU  # nd x 2
            W_G, W_zy = consts
            dG = (W_G.T @ g_c).reshape(nd, nd)
            dZxz = dZxz + (dG + dG.T) @ zxz
            dZ[b, :, 0] = dZxz[:, 0]; dZ[b, :, 2] = dZxz[:, 1]
            dZ[b, :, 1] = W_zy.T @ g_c
            if need_dw:
                o = 0
                dW0[:, o:o + nd] += dU @ zxz.T; o += nd
                dW0[:, o:o + nd * nd] += np.outer(g_c, (zxz @ zxz.T).reshape(-1)); o += nd * nd
                dW0[:, o] += dA[b, :, 3]; o += 1
                dW0[:, o:o + nd] += np.outer(g_c, z[:, 1]); o += nd
                dW0[:, o] += dA[b, :, 1]
        elif spec.equivariance == "SO3":
            dU = dA[b][:, 0:3]
            dG = (consts[0].T @ g_c).reshape(nd, nd)
            dZ[b] = W_ip.T @ dU + (dG + dG.T) @ z
            if need_dw:
                dW0[:, :nd] += dU @ z.T
                dW0[:, nd:] += np.outer(g_c, (z @ z.T).reshape(-1))
        else:
            dU = dA[b][:, 0:3]
            dZ[b] = W_ip.T @ dU + (consts[0].T @ g_c).reshape(nd, 3)
            if need_dw:
                dW0[:, :nd] += dU @ z.T
                dW0[:, nd:] += np.outer(g_c, z.reshape(-1))
    if dout is None and loss_kind == "test":
        dZ = dZ + 2.0 * alpha * Z
    if need_dw:
        grads[keys[0]] = dW0
        grads[keys[1]] = dA[:, :, 4].sum(0)
    res["dZ"] = dZ
    res["grads"] = grads
    return res


# --------------------------------------------------------------------------------------
# FiLM conditioning  (src/models/RENI.py:407-858)   -- pinned by tests/golden/g11_film_*.npz
# --------------------------------------------------------------------------------------


class FilmSpec:
    """Hyper-parameters of RENIAutoDecoderFiLM / RENIVariationalAutoDecoderFiLM (RENI.py:522-598):
    ``siren_hidden_layers`` is the TOTAL number of FiLM layers (RENI.py:563-568)."""

    def __init__(self, ndims, equivariance="SO2", siren_hidden_features=128, siren_hidden_layers=5,
                 mapping_network_features=128, mapping_network_layers=3, out_features=3, output_activation="tanh"):
        assert equivariance in ("SO2", "SO3"), "the reference's None-FiLM encoding cannot run (RENI.py:449-452 vs :552-555)"
        self.ndims = ndims
        self.equivariance = equivariance
        self.H = siren_hidden_features
        self.n_film = siren_hidden_layers
        self.map_features = mapping_network_features
        self.map_layers = mapping_network_layers
        self.out_features = out_features
        self.output_activation = output_activation
        self.in_features = 2 + ndims if equivariance == "SO2" else ndims            # RENI.py:556-561
        self.mn_in_features = ndims * ndims + ndims if equivariance == "SO2" else ndims * ndims

    def param_keys(self) -> List[str]:
        keys = []
        for i in range(self.n_film):
            keys += [f"net.{i}.layer.weight", f"net.{i}.layer.bias"]
        keys += ["final_layer.weight", "final_layer.bias"]
        for i in range(self.map_layers + 1):
            keys += [f"mapping_network.network.{2 * i}.weight", f"mapping_network.network.{2 * i}.bias"]
        return keys


def film_encode(spec: FilmSpec, Z: torch.Tensor, D: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(Siren_Input [B,P,F0], Mapping_Input [B,M]) -- RENI.py:407-447.  The reference repeats the mapping
    input for every pixel; it is constant per image, so one row per image is returned here."""
    if spec.equivariance == "SO3":
        G = Z @ Z.transpose(1, 2)
        return torch.bmm(D, Z.transpose(1, 2)), G.flatten(start_dim=1)
    Z_xz = torch.stack((Z[:, :, 0], Z[:, :, 2]), -1)
    D_xz = torch.stack((D[:, :, 0], D[:, :, 2]), -1)
    G = torch.bmm(Z_xz, Z_xz.transpose(1, 2))
    innerprod = torch.bmm(D_xz, Z_xz.transpose(1, 2))
    D_xz_norm = torch.sqrt(D[:, :, 0] ** 2 + D[:, :, 2] ** 2).unsqueeze(2)
    D_y = D[:, :, 1].unsqueeze(2)
    return torch.cat((D_xz_norm, D_y, innerprod), 2), torch.cat((G.flatten(start_dim=1), Z[:, :, 1]), 1)


def film_mapping(spec: FilmSpec, params: Dict[str, torch.Tensor], m: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """CustomMappingNetwork.forward (RENI.py:470-505): Linear + LeakyReLU(0.2) x layers, Linear; split in halves."""
    x = m
    for i in range(spec.map_layers):
        x = torch.nn.functional.leaky_relu(
            torch.nn.functional.linear(x, params[f"mapping_network.network.{2 * i}.weight"],
                                       params[f"mapping_network.network.{2 * i}.bias"]), 0.2)
    i = spec.map_layers
    fo = torch.nn.functional.linear(x, params[f"mapping_network.network.{2 * i}.weight"],
                                    params[f"mapping_network.network.{2 * i}.bias"])
    half = fo.shape[-1] // 2
    return fo[..., :half], fo[..., half:]


def film_forward(spec: FilmSpec, params: Dict[str, torch.Tensor], Z: torch.Tensor, D: torch.Tensor) -> torch.Tensor:
    """RENI*FiLM.forward on a latent tensor (RENI.py:654-676): x <- sin(freq . (W x + b) + phase) per FiLM
    layer with freq = 15 f + 30, then final_layer and the output activation."""
    if D.shape[0] == 1 and Z.shape[0] != 1:
        D = D.expand(Z.shape[0], -1, -1)
    x, m = film_encode(spec, Z, D)
    f, ph = film_mapping(spec, params, m)
    f = f * 15 + 30
    H = spec.H
    for i in range(spec.n_film):
        a = torch.nn.functional.linear(x, params[f"net.{i}.layer.weight"], params[f"net.{i}.layer.bias"])
        x = torch.sin(f[:, None, i * H:(i + 1) * H] * a + ph[:, None, i * H:(i + 1) * H])
    y = torch.nn.functional.linear(x, params["final_layer.weight"], params["final_layer.bias"])
    if spec.output_activation == "tanh":
        return torch.tanh(y)
    if spec.output_activation == "exp":
        return torch.exp(y)
    return y


def film_fwd_loss_bwd(spec: FilmSpec, params: Dict[str, torch.Tensor], Z: torch.Tensor, D: torch.Tensor,
                      target: torch.Tensor, weight: torch.Tensor, loss_kind: str = "mse", alpha: float = 0.0,
                      beta: float = 0.0) -> Dict[str, object]:
    """Autograd reference of one FiLM training / latent-fitting step."""
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    Zr = Z.detach().clone().requires_grad_(True)
    out = film_forward(spec, ps, Zr, D)
    B = Z.shape[0]
    if loss_kind == "mse":
        terms = (train_loss(out, target.expand(B, -1, -1), weight.expand(B, -1, -1)),)
    else:
        terms = test_loss(out, target.expand(B, -1, -1), weight.expand(B, -1, -1), Zr, alpha, beta)
    terms[0].backward()
    return {"out": out.detach(), "terms": [float(t.detach()) for t in terms], "dZ": Zr.grad.detach(),
            "grads": {k: (v.grad.detach() if v.grad is not None else torch.zeros_like(v)) for k, v in ps.items()}}


def film_init_params(spec: FilmSpec, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """Random parameters with the reference's distributions (RENI.py:455-498, 574-576); NOT the reference's
    RNG stream (seed parity is the product module's job and is tested against the goldens)."""
    g = generator
    H, out = spec.H, {}

    def U(shape, b):
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    for i in range(spec.n_film):
        fin = spec.in_features if i == 0 else H
        out[f"net.{i}.layer.weight"] = U((H, fin), 1 / fin if i == 0 else math.sqrt(6 / fin) / 25)
        out[f"net.{i}.layer.bias"] = U((H,), 1 / math.sqrt(fin))
    out["final_layer.weight"] = U((spec.out_features, H), math.sqrt(6 / H) / 25)
    out["final_layer.bias"] = U((spec.out_features,), 1 / math.sqrt(H))
    fin = spec.mn_in_features
    for i in range(spec.map_layers + 1):
        fout = spec.map_features if i < spec.map_layers else 2 * spec.n_film * H
        std = math.sqrt(2.0 / (1 + 0.2 ** 2)) / math.sqrt(fin)
        w = torch.randn((fout, fin), generator=g) * std
        out[f"mapping_network.network.{2 * i}.weight"] = w * (0.25 if i == spec.map_layers else 1.0)
        out[f"mapping_network.network.{2 * i}.bias"] = U((fout,), 1 / math.sqrt(fin))
        fin = fout
    return out


# --------------------------------------------------------------------------------------
# helpers shared by tests / bench (synthetic inputs of SURVEY.md section 8d)
# --------------------------------------------------------------------------------------

MINMAX = (-18.0536, 11.4633)  # configs/experiment.yaml:88


def synthetic_images(indices: Sequence[int], height: int, width: int) -> torch.Tensor:
    """[B,3,H,W] minmax-log-normalised synthetic HDR maps: x~LogNormal(-3,2) per pixel-channel,
    t = 2(log x - m0)/(m1-m0) - 1 (src/utils/custom_transforms.py:8-12), seed 1234+index."""
    out = []
    for i in indices:
        g = torch.Generator().manual_seed(1234 + int(i))
        logx = torch.randn(3, height, width, generator=g) * 2.0 - 3.0
        out.append(2.0 * (logx - MINMAX[0]) / (MINMAX[1] - MINMAX[0]) - 1.0)
    return torch.stack(out)


def rel_l2(a, b) -> float:
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# ---------------------------------------------------------------------------------------------
# environment-map Blinn-Phong shading (FIT_INVERSE): src/utils/pytorch3d_envmap_shader.py:46-116
# ---------------------------------------------------------------------------------------------
def interpolate_face_attributes(pix_to_face, bary_coords, face_attrs):
    """pytorch3d.ops.interpolate_face_attributes as documented (pytorch3d is NOT under /root/reference; pinned
    0.7.0 by the reference's environment.yml): out[n,h,w,k,:] = sum_i bary[n,h,w,k,i] * face_attrs[pix_to_face, i, :],
    zeros where pix_to_face < 0.  pix_to_face [N,H,W,K] int64, bary_coords [N,H,W,K,3], face_attrs [F,3,D]."""
    mask = pix_to_face < 0
    idx = pix_to_face.clamp(min=0)
    attrs = face_attrs[idx]                                   # [N,H,W,K,3,D]
    out = (bary_coords[..., None] * attrs).sum(dim=-2)
    return out.masked_fill(mask[..., None], 0.0)


def blinn_phong_gbuffer(pixel_normals, pixel_positions, camera_center, light_directions, light_colors,
                        shininess, kd, ks, dtype=torch.float64):
    """Restatement of blinn_phong_shading_env_map behind the interpolation (pytorch3d_envmap_shader.py:75-115).
    pixel_normals / pixel_positions [NP,3] (interpolated, not normalised), camera_center [3],
    light_directions [B,J,3], light_colors [B,J,3] (= map * sineweight, :41) -> colors [B,NP,3]."""
    N = torch.nn.functional.normalize(pixel_normals.to(dtype), p=2, dim=-1, eps=1e-6)            # :82
    V = torch.nn.functional.normalize(camera_center.to(dtype)[None] - pixel_positions.to(dtype), p=2, dim=-1, eps=1e-6)  # :91-93
    L = light_directions.to(dtype)
    C = light_colors.to(dtype)
    s = torch.as_tensor(shininess, dtype=dtype)
    diffuse = torch.einsum("pk,bjk->bpj", N, L).clamp(0.0, 1.0)                                  # :87-88
    diffuse = torch.einsum("bjk,bpj->bpk", C, diffuse)                                           # :90
    Hv = torch.nn.functional.normalize(V[None, :, None, :] + L[:, None, :, :], p=2, dim=-1, eps=1e-6)  # :105-107
    spec = torch.einsum("pk,bpjk->bpj", N, Hv).clamp(0.0, 1.0) ** s                              # :108-110
    specular = torch.einsum("bjk,bpj->bpk", C, spec)                                             # :111
    norm = (s + 2) / (4 * (2 - torch.exp(-s / 2)))                                               # :112-114
    return kd * diffuse + norm * ks * specular                                                   # :115
