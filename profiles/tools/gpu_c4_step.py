import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from reni_amd.models import RENIAutoDecoder
from reni_amd.utils import get_directions, get_sineweight
dev = torch.device("cuda:0")
D = get_directions(256).to(dev); S = get_sineweight(256).to(dev); P = D.shape[1]
B = 21
T = (torch.rand(B, P, 3, device=dev) * 2 - 1)
idx = torch.arange(B, device=dev)
m = RENIAutoDecoder(B, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, True)
with torch.no_grad():
    m.Z.normal_()
m.set_compute_dtype("bf16").to(dev)
def step():
    m.zero_grad(set_to_none=True)
    t = m.fused_loss(m.Z[idx], D, T, S, "test", 1e-7, 1e-4)
    t[0].backward()
    return t
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): t = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"C4 (21 images, RENITestLoss with cosine term, fwd stats pass + fwd/bwd latent): {dt*1e3:.3f} ms/step  {B*P/dt/1e6:.1f} M samples/s  loss {float(t[0]):.5f}")
