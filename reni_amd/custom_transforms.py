"""HDR normalisation transforms with the reference's names (src/utils/custom_transforms.py:4-33).

Tensors that live on the GPU (a model output on its way to the viewer / the FIT_INVERSE renderer, RENI_module.py:108) go
through the HIP epilogue of libreni_hip.so (``reni_unnormalise_srgb`` / ``reni_minmax_normalise``, reni_tu_image.hip).
Host tensors -- the dataset loader normalises every image once on the CPU, src/data/datasets.py:95-101 -- are mapped by
the one-line torch expressions below; that is data preparation in front of the path, not the path.
``transform_builder`` covers the transforms that do not need torchvision (absent here and on the GPU box);
asking for one that does raises with the transform's name."""
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


def get_transform(transform_name, args):
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
