"""Synthetic stand-in for the reference's HDR dataset (src/data/datasets.py is host I/O that needs
a network download and is out of scope -- SURVEY.md section 2, row 10).

Items have the reference's shape contract: ``dataset[i] -> (img[3,H,W] float32, i)`` with values
distributed like minmax-normalised log-HDR pixels (custom_transforms.py:8-12, minmax from
configs/experiment.yaml:88).
"""
import torch
from torch.utils.data import Dataset

MINMAX = (-18.0536, 11.4633)


class SyntheticEnvMapDataset(Dataset):
    def __init__(self, n_images, height, width, seed_base=1234):
        self.n, self.h, self.w, self.seed_base = n_images, height, width, seed_base
        from .custom_transforms import UnMinMaxNormlise
        self.unnormalise = UnMinMaxNormlise(MINMAX)  # datasets.py:80-86: inverse of the minmax-log transform

    def __len__(self):
        return self.n

    def make(self, i, height=None, width=None):
        g = torch.Generator().manual_seed(self.seed_base + int(i))
        h, w = height or self.h, width or self.w
        logx = torch.randn(3, h, w, generator=g) * 2.0 - 3.0
        return 2.0 * (logx - MINMAX[0]) / (MINMAX[1] - MINMAX[0]) - 1.0

    def __getitem__(self, i):
        return self.make(i), i

    def double_resolution(self):
        """Multi-resolution curriculum hook (src/lightning/callbacks.py:27)."""
        self.h, self.w = 2 * self.h, 2 * self.w
