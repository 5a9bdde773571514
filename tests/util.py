"""Shared helpers for the GPU parity tests (oracle = checker only)."""
import os

import numpy as np
import torch

from oracle import reni_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def sd_from(g, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def flat_params(spec: O.DecoderSpec, params) -> torch.Tensor:
    return torch.cat([torch.as_tensor(params[k]).reshape(-1).float() for k in spec.param_keys()])


def unflatten(spec: O.DecoderSpec, flat: torch.Tensor):
    out, o = {}, 0
    shapes = spec.param_shapes()
    for k in spec.param_keys():
        n = int(np.prod(shapes[k]))
        out[k] = flat[o:o + n].reshape(shapes[k])
        o += n
    assert o == flat.numel()
    return out


def make_plan(spec: O.DecoderSpec, dtype="f32"):
    from reni_amd.ops import Plan
    return Plan(spec.equivariance, spec.ndims, spec.hidden_features, spec.hidden_layers, 3,
                spec.last_layer_linear, spec.output_activation, spec.first_omega_0, spec.hidden_omega_0, dtype)


def random_problem(spec: O.DecoderSpec, B, P, seed=0, per_image_dirs=False, grid_w=None):
    g = torch.Generator().manual_seed(seed)
    params = O.init_params(spec, g)
    Z = torch.randn(B, spec.ndims, 3, generator=g)
    if grid_w is not None:
        D = O.get_directions(grid_w)
        W = O.get_sineweight(grid_w)
        P = D.shape[1]
    else:
        nb = B if per_image_dirs else 1
        D = torch.nn.functional.normalize(torch.randn(nb, P, 3, generator=g), dim=-1)
        W = torch.rand(1, P, 3, generator=g)
    T = torch.rand(B, P, 3, generator=g) * 2 - 1
    return params, Z, D, W, T
