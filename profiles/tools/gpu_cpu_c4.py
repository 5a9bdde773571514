"""diagnostic: host time per config-4 step (enqueue only, 40 steps, GPU idle-waited before) against the GPU time of the same steps"""
import sys, os, time
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-also"]
import bench, torch
args = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
orig = bench._measure
def probe(shape, args_, world, dev_, dtype, step, barrier, steps, warmup, eng, paths):
    for s in range(10):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(40):
        step(s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(shape["cfg"], "sparse" if (shape.get("sparsity") or {}).get("flag") == "RENI_WEIGHT_SPARSE" else "", "host us / step", round((t1 - t0) / 40 * 1e6, 1), " host + GPU us / step", round((t2 - t0) / 40 * 1e6, 1), flush=True)
    return orig(shape, args_, world, dev_, dtype, step, barrier, steps, warmup, eng, paths)
bench._measure = probe
for cfg, kw in (("c4", {}), ("c4", {"dense": True}), ("c2", {}), ("c2", {"batch": 100, "res": (16, 32)}), ("film", {})):
    bench.run_config(cfg, args, 0, 1, dev, steps=20, warmup=5, **kw)
