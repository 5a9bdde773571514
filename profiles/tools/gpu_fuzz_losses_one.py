"""Diagnostic: one case of tests/test_gpu_fuzz.py::test_fuzz_test_loss_and_upstream_gradients (by seed) -- per-parameter gradient errors of
the persistent and the generic bf16 kernels beside the reference under autocast.  usage: python profiles/tools/gpu_fuzz_losses_one.py SEED"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oracle import reni_oracle as O
from tests.test_gpu_fuzz import _cases_medium
from reni_amd.models import RENIAutoDecoder
seed = int(sys.argv[1])
c = [x for x in _cases_medium(600) if x["seed"] == seed][0]
for kv in [a for a in sys.argv[2:] if '=' in a]:   # overrides: P=2048 B=1 L=5 beta=0.0 ...
    k, v = kv.split("=")
    if k in c: c[k] = type(c[k])(v) if not isinstance(c[k], bool) else v == "1"
BETA = float(dict(kv.split("=") for kv in sys.argv[2:] if '=' in kv).get("beta", 0.05))
print(c)
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(c["seed"] + 13)
B, P, nd, H, L = c["B"], c["P"], c["nd"], c["H"], c["L"]
Z = torch.randn(B, nd, 3, generator=gen) * 0.6
D = torch.nn.functional.normalize(torch.randn(B if c["per_image"] else 1, P, 3, generator=gen), dim=-1).expand(B, P, 3).contiguous()
S = (torch.rand(1, P, 3, generator=gen) + 0.1).expand(B, P, 3).contiguous()
T = torch.rand(B, P, 3, generator=gen) * 2 - 1
_ = torch.randn(B, P, 3, generator=gen)
spec = O.DecoderSpec(nd, c["eq"], H, L, 3, True, c["act"]); params = O.init_params(spec, gen)
if "corr" in sys.argv:   # the test's correlated targets
    with torch.no_grad():
        o0 = O.reni_forward(spec, params, Z, D)
    T = o0 * (1.0 + 0.3 * torch.randn(B, P, 3, generator=gen)) + 0.5 * o0.std() * torch.randn(B, P, 3, generator=gen)
ref = O.fwd_loss_bwd(spec, params, Z, D, T, S, "test", 1e-3, BETA)
with torch.autocast("cpu", dtype=torch.bfloat16):
    rb = O.fwd_loss_bwd(spec, params, Z, D, T, S, "test", 1e-3, BETA)
rows = {"autocast": {k: O.rel_l2(rb["grads"][k].float().numpy(), v.numpy()) for k, v in ref["grads"].items()}}
rows["autocast"]["dZ"] = O.rel_l2(rb["dZ"].float().numpy(), ref["dZ"].numpy())
for name, env in (("persistent", None), ("generic", "1")):
    if env: os.environ["RENI_NO_PERSIST"] = env
    else: os.environ.pop("RENI_NO_PERSIST", None)
    m = RENIAutoDecoder(B, nd, c["eq"], H, L, 3, True, c["act"], 30.0, 30.0, c["frozen"])
    sd = {"model." + k: v for k, v in params.items()}; sd["model.Z"] = torch.zeros(B, nd, 3)
    m.load_state_dict(sd); m.set_compute_dtype("bf16").to(dev)
    Zd = Z.to(dev).requires_grad_(True)
    terms = m.fused_loss(Zd, D.to(dev), T.to(dev), S.to(dev), loss_kind="test", alpha=1e-3, beta=BETA)
    terms[0].backward()
    got = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None and k != "Z"}
    rows[name] = {k: (O.rel_l2(got[k].numpy(), v.numpy()) if k in got else float("nan")) for k, v in ref["grads"].items()}
    rows[name]["dZ"] = O.rel_l2(Zd.grad.cpu().numpy(), ref["dZ"].numpy())
    rows[name]["terms"] = [float(t) for t in terms]
print("reference terms", ref["loss_terms"], "persistent", rows["persistent"].pop("terms"), "generic", rows["generic"].pop("terms"))
for k in [kk for kk in ref["grads"].keys() if kk.endswith("bias") or kk.startswith("net.0")] + ["dZ"]:
    print(f"{k:22s} autocast {rows['autocast'][k]:.4f}  persistent {rows['persistent'][k]:.4f}  generic {rows['generic'][k]:.4f}")
