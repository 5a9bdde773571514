"""G20 diagnostic: each bf16 kernel's dZ against the exact gradient of ITS OWN network (fp64 autograd emulation) at latents ALONG the
optimisation path (snapshots of the persistent kernels' run after 1, 3, 10, 30, 60, 100, 150, 200 steps)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "256"]
src = open(os.path.join(ROOT, "profiles/tools/gpu_g20_consistency.py")).read().split("Zstar = torch.from_numpy")[0]
exec(src)
from reni_amd.engine import TrainEngine  # noqa: E402

os.environ.pop("RENI_NO_PERSIST", None)
m = model("bf16"); m.set_compute_dtype("bf16").to(dev)
eng = TrainEngine(m, lr=0.1, loss_kind="test", alpha=alpha, beta=beta)
idx = torch.arange(N, device=dev)
Dd, Sd, Td = D1.to(dev), S1.to(dev), T.to(dev)
snaps = {}
for it in range(200):
    eng.step(idx, Td, Sd, Dd)
    if it + 1 in (1, 3, 10, 30, 60, 100, 150, 200):
        snaps[it + 1] = m.Z.detach().cpu().clone()
for t, Z in snaps.items():
    k = kernels(Z)
    ep, eg = emu(Z, "persistent"), emu(Z, "generic")
    print(f"after {t:3d} steps: |Z| {float(Z.norm()):.3f} |dZ| {np.linalg.norm(ep):.3e} | persistent kernel vs its network {rel(k['persistent'], ep):.3e} | generic kernel vs its network {rel(k['generic'], eg):.3e}"
          f" | fp32 kernels vs fp32 network {rel(k['f32'], emu(Z, 'fp32')):.1e}", flush=True)
