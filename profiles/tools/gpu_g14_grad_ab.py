"""Where does the persistent bf16 frozen instance's latent gradient differ from the generic bf16 kernel's?  (round 6)
At the latents the reference's fp32 G14 run passes through (start, after 20 / 100 / 200 steps): loss terms and dZ of
  (a) k_reni_train_bf16<128,false> (+ the statistics instance)      -- the shipped bf16 path
  (b) k_reni_main<bf16,128,FWD_BWD> (RENI_NO_PERSIST at plan creation) -- the generic bf16 kernel
  (c) the fp32 kernels
against the reference's fp32 gradient (fixture), split into the xz columns (the ip / Gram path) and the y column (the z_y path), and the
forward pass alone (model output at those latents) against the fp32 kernels' output."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_trajectory import _decoder_sd  # noqa: E402
from tests.util import load_golden  # noqa: E402
from reni_amd.ops import Plan  # noqa: E402
from reni_amd.utils import get_directions, get_sineweight  # noqa: E402

dev = torch.device("cuda:0")
g = load_golden("g14_c4_trajectory.npz")
N, W = g["imgs"].shape[0], int(g["W"])
sd = _decoder_sd()
keys = ["net.%d.linear.%s" % (i, n) for i in range(6) for n in ("weight", "bias")] + ["net.6.weight", "net.6.bias"]
fp = torch.cat([sd[k].reshape(-1).float() for k in keys]).to(dev)
D = get_directions(W).to(dev)
S = (get_sineweight(W) * torch.from_numpy(g["mask"])).to(dev)
T = torch.from_numpy(g["imgs"]).to(dev).permute(0, 2, 3, 1).reshape(N, -1, 3)
plans = {"persistent bf16": Plan("SO2", 36, 128, 5, 3, True, "tanh", 30.0, 30.0, "bf16")}
os.environ["RENI_NO_PERSIST"] = "1"
plans["generic bf16"] = Plan("SO2", 36, 128, 5, 3, True, "tanh", 30.0, 30.0, "bf16")
del os.environ["RENI_NO_PERSIST"]
plans["f32"] = Plan("SO2", 36, 128, 5, 3, True, "tanh", 30.0, 30.0, "f32")
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for k in (0, 20, 100, 200):
    Z = (torch.zeros(N, 36, 3) if k == 0 else torch.from_numpy(g[f"Z_after_{k}"])).to(dev)
    ref = g[f"dZ_at_{k}"]
    out32 = None
    print(f"-- latents of the reference's step {k}: |dZ_ref| {np.linalg.norm(ref):.3e}   (the reference under autocast-bf16: rel-L2 {rel(g[f'dZ_at_{k}_autocast_bf16'], ref):.3e})")
    res = {}
    for name in ("f32", "generic bf16", "persistent bf16"):
        p = plans[name]
        lt, dZ, _, out = p.forward_loss_backward(Z, D, fp, T, S, loss_kind="test", alpha=float(g["alpha"]), beta=float(g["beta"]), need_dw=False, want_out=True)
        dz = dZ.cpu().numpy(); res[name] = dz
        if name == "f32":
            out32 = out.cpu().numpy()
        oe = out.cpu().numpy() - out32
        print(f"   {name:16s} loss {float(lt[0]):.8f}  dZ rel-L2 {rel(dz, ref):.3e}  xz {rel(dz[:, :, [0, 2]], ref[:, :, [0, 2]]):.3e}  y {rel(dz[:, :, 1], ref[:, :, 1]):.3e}"
              f"  per image {[round(rel(dz[i], ref[i]), 4) for i in range(N)]}  | output vs f32: rms {float(np.sqrt((oe ** 2).mean())):.3e} max {float(np.abs(oe).max()):.3e} mean {float(oe.mean()):+.3e}")
    d = res["persistent bf16"] - res["generic bf16"]
    e_p, e_g = res["persistent bf16"] - ref, res["generic bf16"] - ref
    c = float((e_p * e_g).sum() / (np.linalg.norm(e_p) * np.linalg.norm(e_g)))
    print(f"   persistent - generic: rel-L2 {rel(res['persistent bf16'], res['generic bf16']):.3e}; cosine between the two kernels' ERRORS {c:+.3f}; "
          f"cosine of the persistent error with the gradient itself {float((e_p * ref).sum() / (np.linalg.norm(e_p) * np.linalg.norm(ref))):+.3f} (generic: "
          f"{float((e_g * ref).sum() / (np.linalg.norm(e_g) * np.linalg.norm(ref))):+.3f})")
