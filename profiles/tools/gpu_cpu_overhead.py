import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem
dev = torch.device("cuda:0")
spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
params, Z, D, W, T = random_problem(spec, 1, 0, seed=2, grid_w=256)
plan = make_plan(spec, "bf16")
fp = flat_params(spec, params).to(dev)
Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
for _ in range(5): plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
torch.cuda.synchronize()
n=200
t0=time.perf_counter()
for _ in range(n): plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print(f"ops-level B=1: enqueue {(t1-t0)/n*1e6:.0f} us/call, total {(t2-t0)/n*1e6:.0f} us/call")
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(100): plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
