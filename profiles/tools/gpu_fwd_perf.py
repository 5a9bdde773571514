"""Diagnostic (not collected by pytest): bf16 / fp32 inference throughput (reni_forward) at the config-2 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from reni_amd.models import RENIAutoDecoder
from reni_amd.utils import get_directions
dev = torch.device("cuda:0")
D = get_directions(256).to(dev); P = D.shape[1]
B = 32
for dt in ("bf16", "f32"):
    m = RENIAutoDecoder(B, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, True)
    with torch.no_grad():
        m.Z.normal_()
    m.set_compute_dtype(dt).to(dev)
    idx = torch.arange(B, device=dev)
    with torch.no_grad():
        for _ in range(5): out = m(idx, D)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): out = m(idx, D)
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 30
    print(f"forward {dt}: {dtm*1e3:.3f} ms  {B*P/dtm/1e6:.0f} M samples/s  (RENI_NO_PERSIST={os.environ.get('RENI_NO_PERSIST','0')})")
