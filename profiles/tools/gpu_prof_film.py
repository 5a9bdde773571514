"""Diagnostic (not collected by pytest): fused FiLM training steps (SO2, ND 36, 5 x 128, mapping 3 x 128; B images of the
128 x 256 grid) for rocprofv3 --kernel-trace.  usage: gpu_prof_film.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from reni_amd.film import RENIAutoDecoderFiLM
from reni_amd.utils import get_directions, get_sineweight

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
D = get_directions(256).to(dev); S = get_sineweight(256).to(dev); P = D.shape[1]
T = (torch.rand(B, P, 3, device=dev) * 2 - 1)
idx = torch.arange(B, device=dev)
m = RENIAutoDecoderFiLM(B, 36, "SO2", 128, 5, 128, 3, 3, "tanh", False)
m.set_compute_dtype("bf16").to(dev)
import time
for k in range(steps + 2):
    if k == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    m.zero_grad(set_to_none=True)
    t = m.fused_loss(m.Z[idx], D, T, S)
    t[0].backward()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"FiLM B={B}: {dt*1e3:.3f} ms/step {B*P/dt/1e6:.1f} M samples/s")
