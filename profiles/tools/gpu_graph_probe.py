"""Diagnostic (not collected by pytest): does a captured hipGraph of the fused fwd+loss+bwd call (prologue, training kernel, k_reni_dw1,
side-stream chain, reductions) run faster than the same call launched eagerly?  Static inputs, config-2 shape, B = 64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem

dev = torch.device("cuda:0")
spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
B = int(os.environ.get("B", "64"))
params, Z, D, W, T = random_problem(spec, B, 0, seed=2, grid_w=256)
plan = make_plan(spec, "bf16")
fp = flat_params(spec, params).to(dev)
Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)


def call():
    return plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, need_dw=True)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = timeit(call)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        call()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = call()
    graph = timeit(g.replay)
    ref = call()
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(out[:3], ref[:3]))
    print(f"B={B}: eager {eager:.4f} ms per call, graph replay {graph:.4f} ms per call, results equal: {same}")
except Exception as e:  # noqa: BLE001
    print(f"B={B}: eager {eager:.4f} ms; capture failed: {type(e).__name__}: {str(e)[:300]}")
