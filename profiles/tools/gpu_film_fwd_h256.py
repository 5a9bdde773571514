"""Forward (inference) throughput of the reference's default model -- FiLM, 5 x 256, mapping 3 x 256 -- against the concat model on k_reni_wide256<0>,
bf16, config 5's shape (4 x 524 288 directions, ND = 49).  HIP-event timed, 20 calls behind 5."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from reni_amd import ops  # noqa: E402
from reni_amd.film import RENIAutoDecoderFiLM  # noqa: E402
from reni_amd.models import RENIAutoDecoder  # noqa: E402
from reni_amd.utils import get_directions  # noqa: E402

dev = torch.device("cuda:0")
D = get_directions(1024).to(dev)
P = D.shape[1]
for name, m in (("concat 5x256", RENIAutoDecoder(4, 49, "SO2", 256, 5, 3, True, "tanh", 30.0, 30.0, True)),
                ("FiLM 5x256 (mapping 3x256)", RENIAutoDecoderFiLM(4, 49, "SO2", 256, 5, 256, 3, 3, "tanh", True)),
                ("FiLM 5x128 (mapping 3x128)", RENIAutoDecoderFiLM(4, 49, "SO2", 128, 5, 128, 3, 3, "tanh", True))):
    with torch.no_grad():
        m.Z.normal_()
    m.set_compute_dtype("bf16").to(dev)
    idx = torch.arange(4, device=dev)
    with torch.no_grad():
        for _ in range(5):
            m(idx, D)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            out = m(idx, D)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-30s %.3f ms per call = %.3f G samples/s" % (name, ms, 4 * P / ms / 1e6))
