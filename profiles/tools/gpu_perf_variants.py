"""Diagnostic (not collected by pytest): fused forward+loss+backward throughput of the non-headline variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from reni_amd.models import RENIAutoDecoder
from reni_amd.film import RENIAutoDecoderFiLM
from reni_amd.utils import get_directions, get_sineweight

dev = torch.device("cuda:0")
D = get_directions(256).to(dev); S = get_sineweight(256).to(dev); P = D.shape[1]
B = int(os.environ.get("B", "32"))
T = (torch.rand(B, P, 3, device=dev) * 2 - 1)
idx = torch.arange(B, device=dev)


def run(name, model, dtype, steps=5, fixed=False):
    model.set_compute_dtype(dtype).to(dev)
    lat = model.Z
    def one():
        model.zero_grad(set_to_none=True)
        t = model.fused_loss(lat[idx], D, T, S)
        t[0].backward()
    for _ in range(2):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name:44s} {dtype:5s} {dt*1e3:8.2f} ms/step  {B*P/dt/1e6:8.1f} M samples/s", flush=True)


torch.manual_seed(0)
cases = [
    ("concat SO2 ND36 5x128 (C2, persistent kernel)", lambda: RENIAutoDecoder(B, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, False), ("bf16", "f32")),
    ("concat SO2 ND36 5x128 frozen decoder (C4)", lambda: RENIAutoDecoder(B, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, True), ("bf16",)),
    ("concat SO2 ND49 5x256 (experiment.yaml)", lambda: RENIAutoDecoder(B, 49, "SO2", 256, 5, 3, True, "tanh", 30, 30, False), ("bf16", "f32")),
    ("FiLM SO2 ND36 5x128 map 3x128", lambda: RENIAutoDecoderFiLM(B, 36, "SO2", 128, 5, 128, 3, 3, "tanh", False), ("bf16", "f32")),
    ("FiLM SO2 ND36 5x128 frozen", lambda: RENIAutoDecoderFiLM(B, 36, "SO2", 128, 5, 128, 3, 3, "tanh", True), ("bf16",)),
    ("FiLM SO2 ND49 5x256 map 3x256", lambda: RENIAutoDecoderFiLM(B, 49, "SO2", 256, 5, 256, 3, 3, "tanh", False), ("bf16",)),
]
for name, mk, dts in cases:
    for dt in dts:
        m = mk()
        if m.fixed_decoder:
            with torch.no_grad():
                m.Z.normal_()
        run(name, m, dt)
