"""Datasets on the input side of the hot path.

``RENIDatasetHDR`` / ``RENIDatasetLDR`` / ``get_dataset`` mirror src/data/datasets.py:18-175: a directory of .exr (HDR) or
ordinary image files, naturally sorted, ``dataset[i] -> (img[3,H,W] float32, i)`` after the configured transforms.  The
EXR files are read by reni_amd/exr.py (imageio is not installed).  The reference's ``download=True`` branch fetches a
Google-Drive archive; there is no network here, so it raises.

``SyntheticEnvMapDataset`` is the stand-in bench.py and the tests use when no dataset is on disk: values distributed like
minmax-normalised log-HDR pixels (custom_transforms.py:8-12, minmax from configs/experiment.yaml:88).
"""
import os
import re

import numpy as np
import torch
from torch.utils.data import Dataset

from .custom_transforms import MinMaxNormalise, Normalize, Resize, ToTensor, UnMinMaxNormlise, UnNormalise
from .exr import read_exr


def natsorted(names):
    """natural order ("img2" before "img10"), as natsort.natsorted gives for the dataset's file names (datasets.py:47)"""
    return sorted(names, key=lambda s: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)])


class RENIDatasetHDR(Dataset):
    """src/data/datasets.py:18-101"""

    def __init__(self, dataset_path, transforms=None, download=False):
        super().__init__()
        self.dataset_path = dataset_path
        self.transforms = transforms
        if download:
            raise NotImplementedError("download=True fetches the RENI_HDR archive from Google Drive (datasets.py:31-38); "
                                      "there is no network here: unpack the archive under DATASET.RENI_HDR.PATH")
        files = [f for f in os.listdir(self.dataset_path) if f.endswith(".exr")]
        self.img_names = natsorted(files)
        self.unnormalise = None
        # the dataset's min / max in the log domain, for MinMaxNormalise and its inverse (datasets.py:49-64)
        if self.transforms is not None:
            for t in self.transforms.transforms:
                if isinstance(t, MinMaxNormalise):
                    if len(t.minmax) == 0:
                        t.minmax = self.calculate_minmax()
                    self.unnormalise = UnMinMaxNormlise(t.minmax)

    def __len__(self):
        return len(self.img_names)

    def __getitem__(self, idx):
        img = self.get_image(idx)
        img = self.transforms(img)
        img = torch.nan_to_num(img)
        return img, idx

    def get_image(self, idx):
        img = read_exr(os.path.join(self.dataset_path, self.img_names[idx]))
        return ToTensor()(img[:, :, :3] if img.ndim == 3 else img)

    def double_resolution(self):
        if self.transforms is not None:
            for t in self.transforms.transforms:
                if isinstance(t, Resize):
                    t.size = (t.size[0] * 2, t.size[1] * 2)

    def calculate_minmax(self):
        lo, hi = float("inf"), float("-inf")
        for idx in range(len(self)):
            img = self.get_image(idx)
            img = torch.clip(img, img[img > 0.0].min(), img[img < torch.inf].max()).log()
            lo, hi = min(lo, float(img.min())), max(hi, float(img.max()))
        return [lo, hi]


class RENIDatasetLDR(Dataset):
    """src/data/datasets.py:104-156"""

    def __init__(self, dataset_path, transforms=None, download=False):
        super().__init__()
        self.dataset_path = dataset_path
        self.transforms = transforms
        if download:
            raise NotImplementedError("download=True fetches the RENI_LDR archive from Google Drive (datasets.py:117-124); "
                                      "there is no network here: unpack the archive under DATASET.RENI_LDR.PATH")
        self.unnormalise = None
        if self.transforms is not None:
            for t in self.transforms.transforms:
                if isinstance(t, Normalize):
                    self.unnormalise = UnNormalise(t.mean, t.std)
        self.img_names = natsorted(os.listdir(self.dataset_path))

    def __len__(self):
        return len(self.img_names)

    def __getitem__(self, idx):
        img = self.get_image(idx)[:3, :, :]  # no alpha channel
        return self.transforms(img), idx

    def double_resolution(self):
        if self.transforms is not None:
            for t in self.transforms.transforms:
                if isinstance(t, Resize):
                    t.size = (t.size[0] * 2, t.size[1] * 2)

    def get_image(self, idx):
        from PIL import Image
        return ToTensor()(np.asarray(Image.open(os.path.join(self.dataset_path, self.img_names[idx]))))


def get_dataset(dataset_name, dataset_path, transform, is_hdr):
    """src/data/datasets.py:166-170"""
    if dataset_name == "RENI_HDR" or (dataset_name == "CUSTOM" and is_hdr):
        return RENIDatasetHDR(dataset_path, transform, False)
    if dataset_name == "RENI_LDR" or (dataset_name == "CUSTOM" and not is_hdr):
        return RENIDatasetLDR(dataset_path, transform, False)
    raise ValueError(f"unknown DATASET.NAME {dataset_name!r}")

MINMAX = (-18.0536, 11.4633)


class SyntheticEnvMapDataset(Dataset):
    def __init__(self, n_images, height, width, seed_base=1234):
        self.n, self.h, self.w, self.seed_base = n_images, height, width, seed_base
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
