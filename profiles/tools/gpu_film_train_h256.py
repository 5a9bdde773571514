"""Training step throughput of the reference's default model -- FiLM, 5 x 256, mapping 3 x 256, ND = 36 -- through the fused per-call path
(fused_loss + backward), B = 32 images at 128 x 256, bf16: k_reni_wide256<2, FILM> against the generic chain (RENI_NO_PERSIST)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from reni_amd.film import RENIAutoDecoderFiLM  # noqa: E402
from reni_amd.utils import get_directions, get_sineweight  # noqa: E402

dev = torch.device("cuda:0")
D, S = get_directions(256).to(dev), get_sineweight(256).to(dev)
B = 32
T = (torch.rand(B, D.shape[1], 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
for fixed in (False, True):
    torch.manual_seed(0)
    m = RENIAutoDecoderFiLM(B, 36, "SO2", 256, 5, 256, 3, 3, "tanh", fixed)
    with torch.no_grad():
        m.Z.normal_()
    m.set_compute_dtype("bf16").to(dev)

    def step():
        m.zero_grad(set_to_none=True)
        Zd = m.Z.detach().clone().requires_grad_(True)
        t = m.fused_loss(Zd, D, T, S)
        t[0].backward()
        return t

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        t = step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("FiLM 5x256 %s decoder, B=32 @128x256, persist=%s: %.3f ms per fwd+loss+bwd = %.1f M samples/s (loss %.5f)" %
          ("frozen" if fixed else "trainable", "RENI_NO_PERSIST" not in os.environ, ms, B * D.shape[1] / ms / 1e3, float(t[0])))
