"""HDR normalisation transforms with the reference's names (src/utils/custom_transforms.py:4-33).

Tensors that live on the GPU (a model output on its way to the viewer / the FIT_INVERSE renderer, RENI_module.py:108) go
through the HIP epilogue of libreni_hip.so (``reni_unnormalise_srgb`` / ``reni_minmax_normalise``, reni_tu_image.hip).
Host tensors -- the dataset loader normalises every image once on the CPU, src/data/datasets.py:95-101 -- are mapped by
the one-line torch expressions below; that is data preparation in front of the path, not the path.
``transform_builder`` covers the deterministic transforms the reference's configs use (resize, centercrop, to_tensor,
normalize, minmaxnormalise) with torchvision's tensor semantics restated in torch (torchvision is absent here and on the
GPU box); the random augmentations, which no shipped config uses, raise with the transform's name."""
import numpy as np
import torch

from . import ops


class MinMaxNormalise(object):
    """clip to [smallest positive, largest finite] -> log -> affine map of [min, max] to [-1, 1]
    (custom_transforms.py:4-12)."""

    def __init__(self, minmax):
        self.minmax = minmax

    def __call__(self, img):
        if img.is_cuda:
            return ops.minmax_normalise(img, self.minmax).to(img.dtype)
        lo, hi = self.minmax
        positive, finite = img[img > 0.0], img[img < torch.inf]
        return 2 * (torch.clip(img, positive.min(), finite.max()).log() - lo) / (hi - lo) - 1


def _unnormalise_device(img, minmax):
    """y = exp(0.5 (x + 1)(m1 - m0) + m0) of a device tensor through reni_unnormalise_srgb, in the tensor's own layout."""
    if img.dim() in (3, 4) and img.shape[-3] == 3:   # an image batch [B,3,H,W] / [3,H,W], any strides
        return ops.unnormalise_srgb(img, minmax, srgb=False).view(img.shape).to(img.dtype)
    if img.dim() == 3 and img.shape[-1] == 3:        # a model output [B,P,3] (RENI_module.py:108), read in place
        B, P, _ = img.shape
        lin = ops.unnormalise_srgb(img.view(B, 1, P, 3).permute(0, 3, 1, 2), minmax, srgb=False)  # [B,3,1,P]
        return lin.permute(0, 2, 3, 1).reshape(B, P, 3).to(img.dtype)
    flat = img.reshape(1, 1, -1, 1).expand(1, 3, -1, 1)  # any other shape: elementwise, channel axis broadcast
    return ops.unnormalise_srgb(flat, minmax, srgb=False)[0, 0].reshape(img.shape).to(img.dtype)


class _UnnormaliseFn(torch.autograd.Function):
    """FIT_INVERSE differentiates through the un-normalisation (RENI_module.py:108-112): dy/dx = y (m1 - m0) / 2."""

    @staticmethod
    def forward(ctx, img, m0, m1):
        y = _unnormalise_device(img.detach(), (m0, m1))
        ctx.save_for_backward(y)
        ctx.k = 0.5 * (m1 - m0)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return gy * y * ctx.k, None, None


class UnMinMaxNormlise(object):
    """inverse of MinMaxNormalise (the reference's spelling; custom_transforms.py:14-21)."""

    def __init__(self, minmax):
        self.minmax = minmax

    def __call__(self, img):
        lo, hi = float(self.minmax[0]), float(self.minmax[1])
        if img.is_cuda:
            return _UnnormaliseFn.apply(img, lo, hi) if img.requires_grad else _unnormalise_device(img, (lo, hi))
        return torch.exp(0.5 * (img + 1) * (hi - lo) + lo)


class UnNormalise(object):
    """per-channel x * std + mean on a [B, C, H, W] batch, IN PLACE like the reference (custom_transforms.py:23-40)."""

    def __init__(self, mean, std):
        self.mean = mean
        self.std = std

    def __call__(self, tensor):
        tensor = tensor.permute(1, 0, 2, 3)
        for t, m, s in zip(tensor, self.mean, self.std):
            t.mul_(s).add_(m)
        return tensor.permute(1, 0, 2, 3)


class Resize(object):
    """torchvision.transforms.Resize((h, w)) on a float tensor [..., H, W] as the reference's pinned torchvision 0.11 does it
    (environment.yml): bilinear ``F.interpolate(..., align_corners=False)`` without antialiasing.  ``size`` is mutable --
    the datasets' ``double_resolution`` (datasets.py:81-85) doubles it in place."""

    def __init__(self, size):
        self.size = (int(size[0]), int(size[1]))

    def __call__(self, img):
        if tuple(img.shape[-2:]) == tuple(self.size):
            return img
        x = img if img.dim() == 4 else img[None]
        y = torch.nn.functional.interpolate(x, size=tuple(self.size), mode="bilinear", align_corners=False)
        return y if img.dim() == 4 else y[0]


class CenterCrop(object):
    def __init__(self, size):
        self.size = (int(size), int(size)) if isinstance(size, (int, float)) else (int(size[0]), int(size[1]))

    def __call__(self, img):
        h, w = img.shape[-2:]
        th, tw = self.size
        if th > h or tw > w:
            raise ValueError(f"CenterCrop {self.size} of a {h} x {w} image")
        top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return img[..., top:top + th, left:left + tw]


class ToTensor(object):
    """[H, W, C] (or [H, W]) array -> [C, H, W] tensor; uint8 is scaled by 1 / 255, float arrays are passed through
    (torchvision.transforms.ToTensor; the HDR loader feeds it float32, datasets.py:79)."""

    def __call__(self, pic):
        if isinstance(pic, torch.Tensor):
            return pic
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.to(torch.float32).div(255) if a.dtype == np.uint8 else t


class Normalize(object):
    def __init__(self, mean, std):
        self.mean, self.std = list(mean), list(std)

    def __call__(self, img):
        mean = torch.as_tensor(self.mean, dtype=img.dtype, device=img.device).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=img.dtype, device=img.device).view(-1, 1, 1)
        return (img - mean) / std


def get_transform(transform_name, args):
    """custom_transforms.py:35-72"""
    if transform_name == "resize":
        return Resize((args[0], args[1]))
    if transform_name == "centercrop":
        return CenterCrop(args)
    if transform_name == "to_tensor":
        return ToTensor()
    if transform_name == "normalize":
        return Normalize(args[0], args[1])
    if transform_name == "minmaxnormalise":
        return MinMaxNormalise(args)
    raise NotImplementedError(f"transform {transform_name!r} needs torchvision, which this build does not depend on")


class _Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


def transform_builder(transform_config):
    """[(name, args), ...] -> callable (custom_transforms.py:75-80)."""
    return _Compose([get_transform(t, args) for t, args in transform_config])
