"""diagnostic: config 4 measured several times in one process (a second dense run in a row was seen at 1.4 ms per step for 0.45 ms of kernels)"""
import sys, os, time
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-also"]
import bench, torch
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
import gc
NOGC = len(sys.argv) > 1 or os.environ.get('NOGC')
for dense in (False, True, True, True, False, True):
    m = bench.run_config("c4", args, 0, 1, dev, dense=dense, steps=20, warmup=10, defer=True)
    if NOGC: gc.collect(); gc.disable()
    r = m()
    r2 = m()
    print("dense " if dense else "sparse", round(r["ms_per_step"], 4), round(r2["ms_per_step"], 4), r["launches_per_step"], round(r["roofline"]["kernel_avg_ms"], 4),
          "allocated MB", torch.cuda.memory_allocated() >> 20, "reserved MB", torch.cuda.memory_reserved() >> 20, flush=True)
    gc.enable()
    del m, r, r2
    gc.collect()
