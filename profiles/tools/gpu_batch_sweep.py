"""Diagnostic (not collected by pytest): the config-2 training step (fused fwd + loss + bwd through the module API, no
optimiser) at the per-GPU batch sizes SURVEY section 8(d) lists."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from reni_amd.models import RENIAutoDecoder
from reni_amd.utils import get_directions, get_sineweight
dev = torch.device("cuda:0")
D = get_directions(256).to(dev); S = get_sineweight(256).to(dev); P = D.shape[1]
for B in (1, 5, 25, 64, 100):
    m = RENIAutoDecoder(B, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, False)
    m.set_compute_dtype("bf16").to(dev)
    T = torch.rand(B, P, 3, device=dev) * 2 - 1
    idx = torch.arange(B, device=dev)
    def step():
        m.zero_grad(set_to_none=True)
        t = m.fused_loss(m.Z[idx], D, T, S)
        t[0].backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d}: {dt*1e3:7.3f} ms/step  {B*P/dt/1e6:7.1f} M samples/s")
