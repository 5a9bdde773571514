"""Grid helpers with the reference's names (src/utils/utils.py:46-91).

These produce the *inputs* of the hot path (direction grid, sin(theta) weights, inpainting mask);
they are generated once per resolution on the host in fp32 with the reference's operation order so
the grids are bit-identical to the reference's, then live on the GPU.
"""
import numpy as np
import torch


def _pixel_centre_coords(sidelen):
    half = sidelen // 2
    u = (torch.linspace(1, sidelen, steps=sidelen) - 0.5) / half   # utils.py:50
    v = (torch.linspace(1, half, steps=half) - 0.5) / half         # utils.py:51
    return u, v


def get_directions(sidelen):
    """Unit direction of every pixel of a (sidelen/2 x sidelen) equirectangular image,
    shape [1, sidelen/2*sidelen, 3], row-major (utils.py:46-65)."""
    u, v = _pixel_centre_coords(sidelen)
    half = sidelen // 2
    theta = (np.pi * (u - 1)).repeat(half)
    phi = (np.pi * v).repeat_interleave(sidelen)
    sin_phi = torch.sin(phi)
    return torch.stack((sin_phi * torch.sin(theta), torch.cos(phi), -sin_phi * torch.cos(theta)), -1).unsqueeze(0)


def get_sineweight(sidelen):
    """sin(polar angle) sampling-density compensation, [1, P, 3] (utils.py:68-78)."""
    _, v = _pixel_centre_coords(sidelen)
    phi = (np.pi * v).repeat_interleave(sidelen)
    return torch.sin(phi).unsqueeze(1).repeat(1, 3).unsqueeze(0)


def mask_from_array(sidelen, img):
    """A mask image already in memory (uint8 [Hs, Ws] or [Hs, Ws, C]) -> [1, P, 3] in {0,1}: the part of ``get_mask`` behind
    ``Image.open``."""
    m = torch.from_numpy(np.asarray(img).astype(np.float32) / 255.0)
    if m.ndim == 2:
        m = m.unsqueeze(-1)
    if m.shape[-1] == 1:
        m = m.repeat(1, 1, 3)
    m = m[..., :3]
    hs, ws = m.shape[0], m.shape[1]
    ht, wt = sidelen // 2, sidelen
    ri = torch.clamp((torch.arange(ht, dtype=torch.float32) * (hs / ht)).floor().long(), max=hs - 1)
    ci = torch.clamp((torch.arange(wt, dtype=torch.float32) * (ws / wt)).floor().long(), max=ws - 1)
    return m[ri][:, ci].reshape(-1, 3).unsqueeze(0)


def get_mask(sidelen, path):
    """Inpainting mask PNG -> [1, P, 3] in {0,1}, nearest-neighbour resized to (sidelen/2, sidelen)
    (utils.py:81-91; torchvision's Resize(NEAREST) restated with PIL/torch: source index =
    floor(dst * src/dst_size))."""
    from PIL import Image

    return mask_from_array(sidelen, np.asarray(Image.open(path)))


def sRGB(imgs):
    """Linear HDR -> sRGB for viewing (utils.py:30-42).  Device tensors: the HIP epilogue (reni_unnormalise_srgb)."""
    if len(imgs.shape) == 3:
        imgs = imgs.unsqueeze(0)
    if imgs.is_cuda and imgs.shape[1] == 3:  # (the HIP epilogue is written for 3-channel images; anything else: the formula below)
        from . import ops
        return ops.unnormalise_srgb(imgs, None, srgb=True).to(imgs.dtype)
    q = torch.quantile(torch.quantile(torch.quantile(imgs, 0.98, dim=(1)), 0.98, dim=(1)), 0.98, dim=(1))
    imgs = torch.clamp(imgs / q.unsqueeze(1).unsqueeze(2).unsqueeze(3), 0.0, 1.0)
    return torch.where(imgs <= 0.0031308, 12.92 * imgs, 1.055 * torch.pow(torch.abs(imgs), 1 / 2.4) - 0.055)
