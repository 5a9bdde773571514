"""G20 diagnostic: how far is each kernel's FORWARD image from the reference's fp32 image AT THE REFERENCE'S final latents (no optimisation)?
usage: python profiles/tools/gpu_g20_forward.py [128|256]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_trajectory import _decoder_sd, _psnr  # noqa: E402
from tests.util import load_golden  # noqa: E402
from reni_amd.models import RENIAutoDecoder  # noqa: E402
from reni_amd.utils import get_directions  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
g = load_golden("g14_c4_trajectory.npz")
f = load_golden("g20_concat256_c4_trajectory.npz") if width == 256 else g
W, N = int(g["W"]), 3
masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
D = get_directions(W).to(dev)
for name, env, dtype in (("f32", None, "f32"), ("persistent bf16", None, "bf16"), ("generic bf16", "1", "bf16")):
    if env:
        os.environ["RENI_NO_PERSIST"] = env
    else:
        os.environ.pop("RENI_NO_PERSIST", None)
    if width == 256:
        torch.manual_seed(int(f["seed"]))
        m = RENIAutoDecoder(N, 36, "SO2", 256, 5, 3, True, "tanh", 30.0, 30.0, True)
    else:
        m = RENIAutoDecoder(N, 36, "SO2", 128, 5, 3, True, "tanh", 30.0, 30.0, True)
        m.load_state_dict({"model." + k: v for k, v in _decoder_sd().items()})
    m.set_compute_dtype(dtype).to(dev)
    for zname in ("Z_after_200",):
        Z = torch.from_numpy(f[zname]).to(dev)
        with torch.no_grad():
            img = m(Z, D).float().cpu().numpy()
        ref = f["img_after_200"]
        d = img - ref
        print(f"width {width} {name:16s} forward at the reference's final latents: PSNR masked-out {_psnr(img, ref, masked_out):.2f} dB, kept {_psnr(img, ref, ~masked_out):.2f} dB, "
              f"max |err| {np.abs(d).max():.2e}, rms {np.sqrt((d * d).mean()):.2e}, mean err {d.mean():+.2e}")
