"""Cycle trace of wave 0 of workgroup 0 of k_reni_dw1 (needs a -DRENI_TRACE_DW1 build:
   gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_trace_dw1.py" "-DRENI_TRACE_DW1").
Tags: 100 kernel start, 1 tile start, 2 this tile's prefetched inputs have landed, 3 next tile's loads issued, 4 h_0 rebuilt,
5 barrier, 6 transposition images written, 7 barrier, 8 dW GEMM issued, 101 loop end, 102 partials flushed."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import reni_oracle as O
from tests.util import flat_params, make_plan, random_problem

dev = torch.device("cuda:0")
spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
B = 64
params, Z, D, W, T = random_problem(spec, B, 0, seed=2, grid_w=256)
plan = make_plan(spec, "bf16")
fp = flat_params(spec, params).to(dev)
Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
tr = torch.zeros(1024, dtype=torch.int64, device=dev)
for _ in range(3):
    plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
torch.cuda.synchronize()
os.environ["RENI_TRACE_PTR"] = str(tr.data_ptr())
plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
torch.cuda.synchronize()
if os.environ.get("DW1_WG_TIMES"):  # -DRENI_TRACE_DW1=2 build: [start, end] of every workgroup on the 100 MHz clock
    t = tr.cpu().view(-1, 2)
    t = t[t[:, 0] != 0]
    base = int(t[:, 0].min())
    st, en = (t[:, 0] - base).float() / 100.0, (t[:, 1] - base).float() / 100.0
    print("workgroups:", len(t), " kernel (first start -> last end) %.1f us" % float(en.max()))
    print("start us: min %.1f  median %.1f  max %.1f" % (float(st.min()), float(st.median()), float(st.max())))
    print("end   us: min %.1f  median %.1f  max %.1f" % (float(en.min()), float(en.median()), float(en.max())))
    d = en - st
    print("duration us: min %.1f  median %.1f  max %.1f" % (float(d.min()), float(d.median()), float(d.max())))
    for lo in range(0, len(t), 64):
        print("  wg %3d..%3d: start %.1f..%.1f  end %.1f..%.1f" % (lo, lo + 63, float(st[lo:lo+64].min()), float(st[lo:lo+64].max()), float(en[lo:lo+64].min()), float(en[lo:lo+64].max())))
    sys.exit(0)
words = [w for w in tr.cpu().tolist() if w != 0]
ev = [((w >> 48) & 0xffff, w & ((1 << 48) - 1)) for w in words]
t0 = ev[0][1]
print("events:", len(ev), " whole kernel (wave 0 of workgroup 0):", ev[-1][1] - t0, "cycles")
prev = t0
for tag, clk in ev:
    print(f"{tag:4d}  +{clk - prev:7d}   at {clk - t0:8d}")
    prev = clk
