"""Diagnostic (not collected by pytest): a few fused steps of the non-headline variants, for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from reni_amd.film import RENIAutoDecoderFiLM
from reni_amd.models import RENIAutoDecoder
from reni_amd.utils import get_directions, get_sineweight

dev = torch.device("cuda:0")
D = get_directions(256).to(dev); S = get_sineweight(256).to(dev); P = D.shape[1]
B = 32
T = (torch.rand(B, P, 3, device=dev) * 2 - 1)
idx = torch.arange(B, device=dev)
for mk in (lambda: RENIAutoDecoder(B, 49, "SO2", 256, 5, 3, True, "tanh", 30, 30, False),
           lambda: RENIAutoDecoderFiLM(B, 36, "SO2", 128, 5, 128, 3, 3, "tanh", False),
           lambda: RENIAutoDecoderFiLM(B, 49, "SO2", 256, 5, 256, 3, 3, "tanh", False),
           lambda: RENIAutoDecoder(B, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, True)):
    m = mk()
    if m.fixed_decoder:
        with torch.no_grad():
            m.Z.normal_()
    m.set_compute_dtype("bf16").to(dev)
    for _ in range(4):
        m.zero_grad(set_to_none=True)
        t = m.fused_loss(m.Z[idx], D, T, S)
        t[0].backward()
    torch.cuda.synchronize()
