"""Diagnostic (not collected by pytest): timing of the environment-map shader at the FIT_INVERSE shapes of
configs/experiment.yaml (128 x 128 render, batch 3) beside the oracle-shaped torch computation on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import reni_oracle as O
from reni_amd import ops
from reni_amd.utils import get_directions, get_sineweight

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (B, side, W) in ((3, 128, 128), (3, 128, 256), (21, 128, 256)):
    NP = side * side
    D = get_directions(W)[0]; J = D.shape[0]
    nrm = torch.randn(NP, 3, generator=g); pos = torch.randn(NP, 3, generator=g) * 0.4
    cam = torch.tensor([0.0, 0.0, 2.0])
    C = torch.exp(torch.randn(B, J, 3, generator=g)) * get_sineweight(W)
    nd, pd, Dd, Cd = nrm.to(dev), pos.to(dev), D.to(dev), C.to(dev)
    w = torch.randn(B, NP, 3, device=dev)
    for _ in range(3):
        ops.envmap_shade(nd, pd, cam, Dd, Cd, 500.0, 0.5, 0.5); ops.envmap_shade_backward(nd, pd, cam, Dd, w, 500.0, 0.5, 0.5)
    torch.cuda.synchronize()
    res = []
    for fn, src in ((ops.envmap_shade, Cd), (ops.envmap_shade_backward, w)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(nd, pd, cam, Dd, src, 500.0, 0.5, 0.5)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    pairs = NP * J
    line = f"B={B} render {side}x{side} map {W//2}x{W} (J={J}): fwd {res[0]*1e3:.0f} us, bwd {res[1]*1e3:.0f} us; " \
           f"{pairs/res[0]/1e6:.1f} G (pixel,texel) pairs/s fwd"
    if B == 3 and W == 128:
        sel = slice(0, 512)  # bounded CPU sample: 512 pixels
        t0 = time.perf_counter()
        O.blinn_phong_gbuffer(nrm[sel], pos[sel], cam, D[None].expand(B, -1, -1), C, 500.0, 0.5, 0.5, dtype=torch.float32)
        dt = time.perf_counter() - t0
        line += f"; host torch (reference-shaped einsums, fp32, {torch.get_num_threads()} threads) {512*J/dt/1e9:.3f} G pairs/s"
    print(line)
