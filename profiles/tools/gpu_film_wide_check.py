"""FiLM forward at H = 256 on k_reni_wide256<0, FILM> against the generic kernel (RENI_NO_PERSIST) and the fp32 kernels: max abs / rms
difference of the outputs on config 5's shape, several images per call (tables change inside workgroups' walks), odd tile counts."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from reni_amd.film import RENIAutoDecoderFiLM  # noqa: E402
from reni_amd.utils import get_directions  # noqa: E402

dev = torch.device("cuda:0")
for (B, W, nF) in ((4, 1024, 5), (3, 64, 5), (5, 32, 3), (2, 256, 6)):
    D = get_directions(W).to(dev)
    outs = {}
    for name, env, dtype in (("persistent bf16", None, "bf16"), ("generic bf16", "1", "bf16"), ("f32", None, "f32")):
        if env:
            os.environ["RENI_NO_PERSIST"] = env
        else:
            os.environ.pop("RENI_NO_PERSIST", None)
        torch.manual_seed(3)
        m = RENIAutoDecoderFiLM(B, 49, "SO2", 256, nF, 256, 3, 3, "tanh", True)
        with torch.no_grad():
            m.Z.normal_(generator=torch.Generator().manual_seed(4))
        m.set_compute_dtype(dtype).to(dev)
        with torch.no_grad():
            outs[name] = m(torch.arange(B, device=dev), D).float()
        info = m._plan().path_info(B, D.shape[1], need_dw=False) if hasattr(m._plan(), "path_info") else {}
        print(f"B={B} W={W} FiLM layers={nF} {name:16s} persistent_kernels={info.get('persistent_kernels')}", end="  ")
        if name != "persistent bf16":
            d = outs["persistent bf16"] - outs[name]
            print(f"persistent - this: max abs {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e}", end="")
        print()
    d = outs["generic bf16"] - outs["f32"]
    print(f"   (generic bf16 - f32: max abs {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e}; finite: {bool(torch.isfinite(outs['persistent bf16']).all())})")
os.environ.pop("RENI_NO_PERSIST", None)
