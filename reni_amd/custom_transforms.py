"""HDR normalisation transforms with the reference's names (src/utils/custom_transforms.py:4-33).

Not on the per-sample hot path: they run once per image when a dataset is loaded / a prediction is viewed.
``transform_builder`` covers the transforms that do not need torchvision (absent here and on the GPU box);
asking for one that does raises with the transform's name."""
import torch


class MinMaxNormalise(object):
    """clip to [smallest positive, largest finite] -> log -> affine map of [min, max] to [-1, 1]
    (custom_transforms.py:4-12)."""

    def __init__(self, minmax):
        self.minmax = minmax

    def __call__(self, img):
        img = torch.clip(img, img[img > 0.0].min(), img[img < torch.inf].max())
        img = torch.log(img)
        img = 2 * (img - self.minmax[0]) / (self.minmax[1] - self.minmax[0]) - 1
        return img


class UnMinMaxNormlise(object):
    """inverse of MinMaxNormalise (the reference's spelling; custom_transforms.py:14-21)."""

    def __init__(self, minmax):
        self.minmax = minmax

    def __call__(self, img):
        img = 0.5 * (img + 1) * (self.minmax[1] - self.minmax[0]) + self.minmax[0]
        img = torch.exp(img)
        return img


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
