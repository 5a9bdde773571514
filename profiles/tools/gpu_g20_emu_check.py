import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests.test_gpu_trajectory import _run_g16, _psnr, _cos
from tests.util import load_golden
FIX = "g21_concat128_c4_trajectory.npz" if (len(sys.argv) > 1 and sys.argv[1] == "128") else "g20_concat256_c4_trajectory.npz"
SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
for env, name in ((None, "persistent"), ("1", "generic")):
    if env: os.environ["RENI_NO_PERSIST"] = env
    else: os.environ.pop("RENI_NO_PERSIST", None)
    g, f, terms, Z, img = _run_g16("bf16", dev, FIX)
    masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
    for emu in ("persistent", "generic"):
        ei = f[f"img_after_200_emulated_{emu}"].astype(np.float32); ez = f[f"Z_after_200_emulated_{emu}"]
        rel = np.abs(terms[:, 0] - f[f"terms_emulated_{emu}"][:, 0]) / f[f"terms_emulated_{emu}"][:, 0]
        print(f"{name} kernel vs the run on the {emu} network: image PSNR {_psnr(img, ei, masked_out):.2f} / {_psnr(img, ei, ~masked_out):.2f} dB, latent cos {_cos(Z, ez):.4f}, max rel loss dev {rel.max():.2e}")
    ri = f["img_after_200"]
    for emu in ("persistent", "generic"):
        ei = f[f"img_after_200_emulated_{emu}"].astype(np.float32)
        print(f"   (the {emu} emulation itself vs the fp32 run: {_psnr(ei, ri, masked_out):.2f} / {_psnr(ei, ri, ~masked_out):.2f} dB)")
