"""Cycle trace of one wave of workgroup 0 of the FROZEN-decoder instance k_reni_train_bf16<128,false> (needs a -DRENI_TRACE build; -DRENI_TRACE_WAVE=4: a wave of the second tile).
Prints the per-tag deltas of tile `TILE` (default 3: warm) and the per-tag mean over tiles 2.."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem

dev = torch.device("cuda:0")
spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
B = 21
params, Z, D, W, T = random_problem(spec, B, 0, seed=2, grid_w=256)
plan = make_plan(spec, "bf16")
fp = flat_params(spec, params).to(dev)
Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
tr = torch.zeros(1024, dtype=torch.int64, device=dev)
plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, need_dw=False)
torch.cuda.synchronize()
os.environ["RENI_TRACE_PTR"] = str(tr.data_ptr())
plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, need_dw=False)
torch.cuda.synchronize()
words = [w for w in tr.cpu().tolist() if w != 0]
ev = [((w >> 48) & 0xffff, w & ((1 << 48) - 1)) for w in words]
tiles, cur = [], None
for tag, clk in ev:
    if tag == 1:
        cur = []
        tiles.append(cur)
    if cur is not None:
        cur.append((tag, clk))
tiles = [t for t in tiles if len(t) > 10]
print("tiles traced:", len(tiles), " cycles per tile:", [t[-1][1] - t[0][1] for t in tiles])
agg = collections.OrderedDict()
for t in tiles[2:]:
    for (tag0, c0), (tag1, c1) in zip(t, t[1:]):
        agg.setdefault((tag0, tag1), []).append(c1 - c0)
tot = 0
for (a, b), v in agg.items():
    m = sum(v) / len(v)
    tot += m
    print(f"{a:4d} -> {b:4d}  {m:8.0f}   (cum {tot:8.0f})  min {min(v)} max {max(v)}")
