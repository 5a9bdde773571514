"""SURVEY.md section 8 row f3 on the GPU: the HDR epilogue reni_unnormalise_srgb / reni_minmax_normalise (reni_tu_image.hip)
against the reference's own outputs (golden G12: src/utils/custom_transforms.py:4-21 and src/utils/utils.py:30-42 run in
the build container) and, at BASELINE's image sizes, against the torch expressions of the same chain."""
import numpy as np
import pytest
import torch

from reni_amd import ops, utils
from reni_amd.custom_transforms import MinMaxNormalise, UnMinMaxNormlise

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _srgb_torch(x):  # the reference chain on the host, fp32 (utils.py:30-42)
    q = torch.quantile(torch.quantile(torch.quantile(x, 0.98, dim=1), 0.98, dim=1), 0.98, dim=1)
    y = torch.clamp(x / q.view(-1, 1, 1, 1), 0.0, 1.0)
    return torch.where(y <= 0.0031308, 12.92 * y, 1.055 * torch.pow(torch.abs(y), 1 / 2.4) - 0.055)


def test_g12_unnormalise_and_srgb_match_the_reference(golden):
    g = golden("g12_transforms.npz")
    mm = [float(g["minmax"][0]), float(g["minmax"][1])]
    n = torch.from_numpy(g["normalised"]).to(DEV)                       # [3,16,32] log-normalised
    lin = UnMinMaxNormlise(mm)(n)
    assert lin.shape == n.shape and lin.is_cuda
    np.testing.assert_allclose(lin.cpu().numpy(), g["unnormalised"], rtol=3e-6, atol=0)
    # the fused call: un-normalise + nested 0.98 quantile + clamp + gamma == sRGB(UnMinMaxNormlise(x)) of the reference
    srgb, lin2 = ops.unnormalise_srgb(n, mm, srgb=True, want_linear=True)
    assert torch.equal(lin2[0], lin)
    np.testing.assert_allclose(srgb.cpu().numpy(), g["srgb1"], rtol=0, atol=2e-6)
    # plain sRGB of a linear batch (two images: per-image quantiles)
    x2 = (torch.rand(2, 3, 8, 16, generator=torch.Generator().manual_seed(13)) * 3.0).to(DEV)
    np.testing.assert_allclose(utils.sRGB(x2).cpu().numpy(), g["srgb2"], rtol=0, atol=2e-6)


def test_g12_minmax_normalise_matches_the_reference(golden):
    g = golden("g12_transforms.npz")
    mm = [float(g["minmax"][0]), float(g["minmax"][1])]
    img = torch.from_numpy(g["img"]).to(DEV)                            # holds a zero, an inf and a 1e4 spike
    n = MinMaxNormalise(mm)(img)
    np.testing.assert_allclose(n.cpu().numpy(), g["normalised"], rtol=0, atol=3e-7)


@pytest.mark.parametrize("B,H,W", [(3, 128, 256), (1, 512, 1024), (2, 33, 70)])
def test_model_output_layout_full_size(B, H, W):
    """A model output [B, H*W, 3] (channel-last) read in place, at BASELINE's grids and a ragged one."""
    g = torch.Generator().manual_seed(B * 1000 + H)
    out = (torch.rand(B, H * W, 3, generator=g) * 2 - 1)
    mm = [-18.0536, 11.4633]
    view = out.to(DEV).view(B, H, W, 3).permute(0, 3, 1, 2)              # strided [B,3,H,W], never copied
    srgb, lin = ops.unnormalise_srgb(view, mm, srgb=True, want_linear=True)
    lin_ref = torch.exp(0.5 * (out + 1) * (mm[1] - mm[0]) + mm[0]).view(B, H, W, 3).permute(0, 3, 1, 2)
    np.testing.assert_allclose(lin.cpu().numpy(), lin_ref.numpy(), rtol=3e-6, atol=0)
    # the quantile of the DEVICE's linear image (exp differs from the host's in the last bit, and a quantile picks elements)
    ref = _srgb_torch(lin.cpu())
    np.testing.assert_allclose(srgb.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)
    # [B,P,3] through the transform class (the FIT_INVERSE call site, RENI_module.py:108)
    y = UnMinMaxNormlise(mm)(out.to(DEV))
    assert y.shape == (B, H * W, 3)
    assert torch.equal(y.view(B, H, W, 3).permute(0, 3, 1, 2), lin)


def test_unnormalise_is_differentiable_on_the_device():
    """FIT_INVERSE back-propagates through the un-normalisation (RENI_module.py:108-112)."""
    mm = [-18.0536, 11.4633]
    x = (torch.rand(2, 50, 3, generator=torch.Generator().manual_seed(5)) * 2 - 1)
    xd = x.to(DEV).requires_grad_(True)
    w = torch.rand(2, 50, 3, generator=torch.Generator().manual_seed(6))
    (UnMinMaxNormlise(mm)(xd) * w.to(DEV)).sum().backward()
    xc = x.clone().requires_grad_(True)
    (torch.exp(0.5 * (xc + 1) * (mm[1] - mm[0]) + mm[0]) * w).sum().backward()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xc.grad.numpy(), rtol=5e-6)


def test_ties_and_constant_images():
    """Order statistics with many equal values (a masked / saturated image): ranks are broken by index, as a sort does."""
    x = torch.ones(2, 3, 16, 24)
    x[1, :, :4] = 5.0
    x[1, 0, 8:, 3] = 0.25
    got = utils.sRGB(x.to(DEV)).cpu()
    np.testing.assert_allclose(got.numpy(), _srgb_torch(x).numpy(), rtol=0, atol=2e-6)
