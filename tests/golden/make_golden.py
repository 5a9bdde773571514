"""Generate the golden vectors under tests/golden/ by IMPORTING the reference implementation.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs are small .npz files of plain arrays (inputs + the reference's outputs).  Nothing of the
reference's source travels.  IDs follow SURVEY.md Appendix C (G1..G10).
"""
import os
import sys
import types

sys.dont_write_bytecode = True  # importing the reference must not write __pycache__ under /root/reference (read-only by rule)

import numpy as np
import torch

REF = os.environ.get("RENI_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

# the reference's utils module imports gdown / torchvision at module top; neither is installed
# and neither is used by get_directions / get_sineweight -> stub the names.
for name in ("gdown", "torchvision", "torchvision.transforms"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.path.insert(0, REF)

from src.models import RENI as ref  # noqa: E402
from src.utils import loss_functions as ref_loss  # noqa: E402
from src.utils import utils as ref_utils  # noqa: E402
from PIL import Image  # noqa: E402


def sd_np(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def synth_targets(B, P, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, P, 3, generator=g) * 2 - 1


# ---------------------------------------------------------------- G1 grids
def g1():
    out = {}
    for W in (32, 64):
        out[f"dir_{W}"] = ref_utils.get_directions(W).numpy()
        out[f"sw_{W}"] = ref_utils.get_sineweight(W).numpy()
    for W in (256, 1024):
        d = ref_utils.get_directions(W)
        s = ref_utils.get_sineweight(W)
        out[f"dir_{W}_head"] = d[0, :64].numpy()
        out[f"dir_{W}_tail"] = d[0, -64:].numpy()
        out[f"dir_{W}_sum64"] = d.double().sum((0, 1)).numpy()
        out[f"sw_{W}_head"] = s[0, :64].numpy()
        out[f"sw_{W}_tail"] = s[0, -64:].numpy()
        out[f"sw_{W}_sum64"] = s.double().sum().numpy()
    save("g1_grids.npz", **out)


# ---------------------------------------------------------------- G2 encodings
def g2():
    out = {}
    for tag, (B, nd, P) in {"a": (2, 3, 5), "b": (2, 9, 16)}.items():
        g = torch.Generator().manual_seed(11)
        Z = torch.randn(B, nd, 3, generator=g)
        D = torch.nn.functional.normalize(torch.randn(B, P, 3, generator=g), dim=-1)
        out[f"Z_{tag}"] = Z.numpy(); out[f"D_{tag}"] = D.numpy()
        out[f"so2_{tag}"] = ref.SO2InvariantRepresentation(Z, D).numpy()
        out[f"so3_{tag}"] = ref.SO3InvariantRepresentation(Z, D).numpy()
        out[f"none_{tag}"] = ref.NoInvariance(Z, D).numpy()
    save("g2_encodings.npz", **out)


# ---------------------------------------------------------------- G3 C1 forward
def g3():
    torch.manual_seed(0)
    m = ref.RENIAutoDecoder(1, 9, "SO2", 64, 3, 3, True, None, 30, 30, False)
    D = ref_utils.get_directions(64)
    with torch.no_grad():
        out = m(0, D)
    arrs = {"sd." + k: v for k, v in sd_np(m).items()}
    save("g3_c1_forward.npz", out=out.numpy(), **arrs)


# ---------------------------------------------------------------- G4 fwd+bwd training
def _fwd_bwd(m, Z, D, t, s):
    for p in m.parameters():
        p.grad = None
    Zr = Z.clone().requires_grad_(True)
    out = m(Zr, D)
    loss = ref_loss.RENITrainLoss()(out, t, s)
    loss.backward()
    return out.detach(), loss.detach(), Zr.grad.detach()


def g4():
    # small: full grads
    torch.manual_seed(4)
    m = ref.RENIAutoDecoder(2, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, False)
    W = 32
    D = ref_utils.get_directions(W).repeat(2, 1, 1)
    s = ref_utils.get_sineweight(W).repeat(2, 1, 1)
    t = synth_targets(2, D.shape[1], 40)
    Z = m.Z.detach().clone()
    out, loss, dZ = _fwd_bwd(m, Z, D, t, s)
    arrs = {"sd." + k: v for k, v in sd_np(m).items()}
    arrs.update({"g." + k: p.grad.numpy().copy() for k, p in m.named_parameters() if k != "Z"})
    save("g4_small.npz", Z=Z.numpy(), target=t.numpy(), out=out.numpy(), loss=loss.numpy(),
         dZ=dZ.numpy(), W=np.int64(W), **arrs)
    # other equivariances / head variants at the small size (forward + dZ + grads)
    for tag, (eq, lll, act) in {"so3": ("SO3", True, "tanh"), "none": ("None", True, None),
                                "sinehead": ("SO2", False, None)}.items():
        torch.manual_seed(5)
        m = ref.RENIAutoDecoder(2, 9, eq, 64, 3, 3, lll, act, 30, 30, False)
        Z = m.Z.detach().clone()
        out, loss, dZ = _fwd_bwd(m, Z, D, t, s)
        arrs = {"sd." + k: v for k, v in sd_np(m).items()}
        arrs.update({"g." + k: p.grad.numpy().copy() for k, p in m.named_parameters() if k != "Z"})
        save(f"g4_small_{tag}.npz", Z=Z.numpy(), target=t.numpy(), out=out.numpy(),
             loss=loss.numpy(), dZ=dZ.numpy(), W=np.int64(W), **arrs)
    # config-2 shape (ND=36,H=128,L=5), B=2, P=2048: norms + heads only, weights by seed recipe
    torch.manual_seed(42)
    m = ref.RENIAutoDecoder(2, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, False)
    W = 64
    D = ref_utils.get_directions(W).repeat(2, 1, 1)
    s = ref_utils.get_sineweight(W).repeat(2, 1, 1)
    t = synth_targets(2, D.shape[1], 41)
    Z = m.Z.detach().clone()
    out, loss, dZ = _fwd_bwd(m, Z, D, t, s)
    arrs = {"sd." + k: v.astype(np.float32) for k, v in sd_np(m).items()}
    for k, p in m.named_parameters():
        if k == "Z":
            continue
        gk = p.grad.numpy()
        arrs["gn." + k] = np.float64(np.linalg.norm(gk.astype(np.float64)))
        arrs["gh." + k] = gk.reshape(-1)[:32].copy()
    save("g4_c2shape.npz", Z=Z.numpy(), target=t.numpy(), out_head=out[:, :256].numpy(),
         out_sum64=out.double().sum().numpy(), loss=loss.numpy(), dZ=dZ.numpy(), W=np.int64(W), **arrs)


# ---------------------------------------------------------------- G5 losses
def g5():
    g = torch.Generator().manual_seed(5)
    o = (torch.rand(2, 64, 3, generator=g) * 2 - 1).requires_grad_(True)
    t = torch.rand(2, 64, 3, generator=g) * 2 - 1
    s = torch.rand(2, 64, 3, generator=g)
    Z = torch.randn(2, 9, 3, generator=g).requires_grad_(True)
    mu = torch.randn(2, 9, 3, generator=g).requires_grad_(True)
    lv = (torch.randn(2, 9, 3, generator=g) - 5).requires_grad_(True)
    res = {"o": o.detach().numpy(), "t": t.numpy(), "s": s.numpy(), "Z": Z.detach().numpy(),
           "mu": mu.detach().numpy(), "lv": lv.detach().numpy()}
    v = ref_loss.WeightedMSE(o, t, s); (go,) = torch.autograd.grad(v, o)
    res["mse"] = v.detach().numpy(); res["mse_go"] = go.numpy()
    v = ref_loss.WeightedCosineSimilarity(o, t, s); (go,) = torch.autograd.grad(v, o)
    res["cos"] = v.detach().numpy(); res["cos_go"] = go.numpy()
    v = ref_loss.KLD(mu, lv, 27); gmu, glv = torch.autograd.grad(v, (mu, lv))
    res["kld"] = v.detach().numpy(); res["kld_gmu"] = gmu.numpy(); res["kld_glv"] = glv.numpy()
    tl = ref_loss.RENITestLoss(1e-7, 1e-1)(o, t, s, Z)
    go, gz = torch.autograd.grad(tl[0], (o, Z))
    res["test"] = np.array([x.item() for x in tl]); res["test_go"] = go.numpy(); res["test_gz"] = gz.numpy()
    vl = ref_loss.RENIVADTrainLoss(1e-4, 27)(o, t, s, mu, lv)
    go, gmu, glv = torch.autograd.grad(vl[0], (o, mu, lv))
    res["vad"] = np.array([x.item() for x in vl]); res["vad_go"] = go.numpy()
    res["vad_gmu"] = gmu.numpy(); res["vad_glv"] = glv.numpy()
    save("g5_losses.npz", **res)


# ---------------------------------------------------------------- G6 training-step re-enactment
def g6():
    """FIT_DECODER / AutoDecoder steps as RENI_module.py:80-146,168-252 prescribes: permute+view of
    [B,3,H,W] images, Z = model.Z[idx], model(Z, D), RENITrainLoss, Adam(lr) over all params."""
    torch.manual_seed(6)
    N, B, W = 4, 2, 32
    m = ref.RENIAutoDecoder(N, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, False)
    sd0 = sd_np(m)
    g = torch.Generator().manual_seed(60)
    imgs_all = torch.rand(N, 3, W // 2, W, generator=g) * 2 - 1
    D1 = ref_utils.get_directions(W); S1 = ref_utils.get_sineweight(W)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crit = ref_loss.RENITrainLoss()
    losses = []
    batches = [[0, 1], [2, 3], [0, 1], [2, 3], [0, 1]]
    for idx in batches:
        idx_t = torch.tensor(idx)
        imgs = imgs_all[idx_t]
        bs = imgs.shape[0]
        t = imgs.permute(0, 2, 3, 1).reshape(bs, -1, 3)
        D = D1.repeat(bs, 1, 1); S = S1.repeat(bs, 1, 1)
        Z = m.Z[idx_t, :, :]
        out = m(Z, D)
        loss = crit(out, t, S)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    arrs = {"sd0." + k: v for k, v in sd0.items()}
    save("g6_train_steps.npz", imgs=imgs_all.numpy(), batches=np.array(batches), losses=np.array(losses),
         Z_final=m.Z.detach().numpy(), W0_final=m.net[0].linear.weight.detach().numpy(),
         Wout_final=m.net[4].weight.detach().numpy(), lr=np.float64(1e-3), W=np.int64(W), **arrs)


# ---------------------------------------------------------------- G7 C4 re-enactment (notebook cell 4)
def g7():
    """Frozen VAD decoder, latents from zero, Mask-3, RENITestLoss(1e-7,1e-1), Adam lr 1e-1."""
    torch.manual_seed(42)
    trained = ref.RENIVariationalAutoDecoder(5, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, False)
    ckpt = {"model." + k: v.clone() for k, v in trained.state_dict().items()}
    N, W = 3, 64
    m = ref.RENIVariationalAutoDecoder(N, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, True)
    m.load_state_dict(ckpt)
    assert float(m.mu.abs().sum()) == 0.0
    g = torch.Generator().manual_seed(70)
    imgs = torch.rand(N, 3, W // 2, W, generator=g) * 2 - 1
    mask_img = np.array(Image.open(os.path.join(REF, "data/Masks/Mask-3.png")))
    # utils.py:81-91: ToTensor (HWC uint8 -> CHW float / 255) -> Resize((W/2, W), NEAREST).  torchvision is absent; its 0.11 tensor
    # path for Resize(NEAREST) is torch.nn.functional.interpolate(img[None], size, mode="nearest")[0], called here directly.
    chw = torch.from_numpy(mask_img[..., :3].astype(np.float32) / 255.0).permute(2, 0, 1)
    mask = torch.nn.functional.interpolate(chw[None], size=(W // 2, W), mode="nearest")[0].permute(1, 2, 0).reshape(1, -1, 3)
    D1 = ref_utils.get_directions(W); S1 = ref_utils.get_sineweight(W) * mask
    opt = torch.optim.Adam(m.parameters(), lr=1e-1)
    crit = ref_loss.RENITestLoss(alpha=1e-7, beta=1e-1)
    terms = []
    idx = torch.arange(N)
    t = imgs.permute(0, 2, 3, 1).reshape(N, -1, 3)
    first_grad = None
    for _ in range(10):
        D = D1.repeat(N, 1, 1); S = S1.repeat(N, 1, 1)
        Z = m.mu[idx, :, :]
        out = m(Z, D)
        opt.zero_grad()
        tl = crit(out, t, S, Z)
        tl[0].backward()
        if first_grad is None:
            first_grad = m.mu.grad.detach().clone().numpy()
        opt.step()
        terms.append([x.item() for x in tl])
    arrs = {"ckpt." + k: v.numpy() for k, v in ckpt.items() if ".net." in k or k.startswith("model.net")}
    save("g7_latent_opt.npz", imgs=imgs.numpy(), mask=mask.numpy(), mask_src=mask_img[..., 0].astype(np.uint8),
         terms=np.array(terms), mu_final=m.mu.detach().numpy(), mu_grad0=first_grad,
         W=np.int64(W), **arrs)


# ---------------------------------------------------------------- G8 VAD
def g8():
    torch.manual_seed(8)
    m = ref.RENIVariationalAutoDecoder(3, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, False)
    idx = torch.tensor([2, 0])
    torch.manual_seed(80)
    Z, mu, lv = m.sample_latent(idx)
    torch.manual_seed(80)
    eps = torch.randn(2, 9, 3)
    W = 32
    D = ref_utils.get_directions(W).repeat(2, 1, 1); S = ref_utils.get_sineweight(W).repeat(2, 1, 1)
    t = synth_targets(2, D.shape[1], 81)
    out = m(Z, D)
    vl = ref_loss.RENIVADTrainLoss(1e-4, 27)(out, t, S, mu, lv)
    vl[0].backward()
    arrs = {"sd." + k: v for k, v in sd_np(m).items()}
    save("g8_vad.npz", idx=idx.numpy(), eps=eps.numpy(), Z=Z.detach().numpy(), target=t.numpy(),
         terms=np.array([x.item() for x in vl]), g_mu=m.mu.grad.numpy(), g_lv=m.log_var.grad.numpy(),
         g_W0=m.net[0].linear.weight.grad.numpy(), W=np.int64(W), **arrs)


# ---------------------------------------------------------------- G9 API
def g9():
    out = {}
    for mt, cls in (("AD", ref.RENIAutoDecoder), ("VAD", ref.RENIVariationalAutoDecoder)):
        for eq in ("SO2", "SO3", "None"):
            torch.manual_seed(9)
            m = cls(3, 9, eq, 64, 3, 3, True, "tanh", 30, 30, False)
            sd = m.state_dict()
            out[f"keys_{mt}_{eq}"] = np.array(list(sd.keys()))
            out[f"shapes_{mt}_{eq}"] = np.array([str(tuple(v.shape)) for v in sd.values()])
            out[f"infeat_{mt}_{eq}"] = np.int64(m.in_features)
    torch.manual_seed(9)
    m = ref.RENIAutoDecoder(3, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, False)
    D = ref_utils.get_directions(16)
    with torch.no_grad():
        out["disp_int"] = m(1, D).numpy()
        out["disp_list"] = m([0, 2], D.repeat(2, 1, 1)).numpy()
        out["disp_idx"] = m(torch.tensor([2, 1]), D.repeat(2, 1, 1)).numpy()
        out["disp_lat"] = m(m.Z[[1]], D).numpy()
    for k, v in sd_np(m).items():
        out["sd." + k] = v
    save("g9_api.npz", **out)


# ---------------------------------------------------------------- G10 equivariance
def g10():
    W = 64
    D = ref_utils.get_directions(W)
    th = 0.7
    Ry = torch.tensor([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], dtype=torch.float32)
    torch.manual_seed(7)
    Q, _ = torch.linalg.qr(torch.randn(3, 3))
    if torch.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    out = {"Ry": Ry.numpy(), "R3": Q.numpy()}
    for eq, R in (("SO2", Ry), ("SO3", Q)):
        torch.manual_seed(10)
        m = ref.RENIAutoDecoder(1, 49, eq, 128, 5, 3, True, "tanh", 30, 30, False)
        with torch.no_grad():
            Z = m.Z[[0]]
            a = m(Z, D)
            b = m(Z @ R.T, D @ R.T)
        out[f"resid_{eq}"] = np.float64((a - b).abs().max())
        out[f"Z_{eq}"] = Z.numpy()
        out[f"out_{eq}_head"] = a[0, :128].numpy()
        out[f"out_{eq}_sum64"] = a.double().sum().numpy()
        for k, v in sd_np(m).items():
            if k != "Z":
                out[f"sd_{eq}." + k] = v.astype(np.float16) if False else v
    save("g10_equivariance.npz", **out)


# ---------------------------------------------------------------- G11 FiLM conditioning
def g11():
    """RENI*FiLM (src/models/RENI.py:407-858): seeded state dicts, per-image frequencies / phase shifts,
    forward, RENITrainLoss gradients of every parameter and of Z, and a RENITestLoss latent gradient."""
    W = 32
    D = ref_utils.get_directions(W).repeat(2, 1, 1)
    s = ref_utils.get_sineweight(W).repeat(2, 1, 1)
    t = synth_targets(2, D.shape[1], 110)
    cases = {
        # tag: (class, eq, H, siren layers, mapping features, mapping layers, activation)
        "so2_ad": (ref.RENIAutoDecoderFiLM, "SO2", 64, 3, 32, 2, "tanh"),
        "so3_vad": (ref.RENIVariationalAutoDecoderFiLM, "SO3", 64, 2, 48, 1, "exp"),
        "so2_one": (ref.RENIAutoDecoderFiLM, "SO2", 32, 1, 16, 1, None),
    }
    for tag, (cls, eq, H, nF, mf, ml, act) in cases.items():
        torch.manual_seed(111)
        m = cls(2, 9, eq, H, nF, mf, ml, 3, act, False)
        Z = (m.Z if hasattr(m, "Z") else m.mu).detach().clone() * 0.5
        for prm in m.parameters():
            prm.grad = None
        Zr = Z.clone().requires_grad_(True)
        si, mi = m.InvariantRepresentation(Zr, D)
        fr, ph = m.mapping_network(mi)
        out = m(Zr, D)
        loss = ref_loss.RENITrainLoss()(out, t, s)
        loss.backward()
        arrs = {"sd." + k: v for k, v in sd_np(m).items()}
        arrs.update({"g." + k: prm.grad.numpy().copy() for k, prm in m.named_parameters()
                     if prm.grad is not None and k not in ("Z", "mu", "log_var")})
        # RENITestLoss on the same model (latent gradient only)
        Z2 = Z.clone().requires_grad_(True)
        tl = ref_loss.RENITestLoss(alpha=1e-3, beta=1e-1)(m(Z2, D), t, s, Z2)
        tl[0].backward()
        save(f"g11_film_{tag}.npz", Z=Z.numpy(), target=t.numpy(), out=out.detach().numpy(), loss=loss.detach().numpy(),
             dZ=Zr.grad.numpy(), freq_raw=fr[:, 0].detach().numpy(), phase=ph[:, 0].detach().numpy(),
             siren_input_head=si[:, :8].detach().numpy(), test_terms=np.array([x.item() for x in tl]),
             test_dZ=Z2.grad.numpy(), W=np.int64(W),
             cfg=np.array([{"SO2": 1, "SO3": 2}[eq], 9, H, nF, mf, ml, {None: 0, "tanh": 1, "exp": 2}[act]]), **arrs)


# ---------------------------------------------------------------- G12 HDR transforms
def g12():
    """MinMaxNormalise / UnMinMaxNormlise / UnNormalise (src/utils/custom_transforms.py) and sRGB (src/utils/utils.py:30-42)
    on a synthetic HDR image with zeros, an inf and a large dynamic range."""
    from src.utils import custom_transforms as ref_ct
    g = torch.Generator().manual_seed(12)
    img = torch.exp(torch.randn(3, 16, 32, generator=g) * 2.0 - 3.0)
    img[0, 0, 0] = 0.0
    img[1, 3, 4] = float("inf")
    img[2, 5, 6] = 1e4
    minmax = [-18.0536, 11.4633]  # configs/experiment.yaml:88
    n = ref_ct.MinMaxNormalise(minmax)(img.clone())
    u = ref_ct.UnMinMaxNormlise(minmax)(n.clone())
    batch = torch.rand(2, 3, 4, 5, generator=g)
    un = ref_ct.UnNormalise([0.1, 0.2, 0.3], [1.5, 2.5, 3.5])(batch.clone())
    srgb1 = ref_utils.sRGB(u.clone())
    srgb2 = ref_utils.sRGB(torch.rand(2, 3, 8, 16, generator=torch.Generator().manual_seed(13)) * 3.0)
    save("g12_transforms.npz", img=img.numpy(), minmax=np.array(minmax), normalised=n.numpy(), unnormalised=u.numpy(),
         batch=batch.numpy(), unnorm_batch=un.numpy(), srgb1=srgb1.numpy(), srgb2=srgb2.numpy())


def g13():
    """G13: the reference's blinn_phong_shading_env_map (pytorch3d_envmap_shader.py:46-116) on a synthetic G-buffer.
    pytorch3d is not installed: its names are stubbed so the module imports, and interpolate_face_attributes -- the
    one pytorch3d function the shading routine calls -- is the oracle's restatement of its documented behaviour.
    Everything after the interpolation (normalisation, einsums, clamp, pow, Blinn-Phong normalisation) is the
    reference's own code running here."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import reni_oracle as O

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {})
    for name in ("pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "pytorch3d.common", "pytorch3d.renderer.utils",
                 "pytorch3d.ops", "pytorch3d.renderer.mesh", "pytorch3d.renderer.mesh.rasterizer", "pytorch3d.io",
                 "pytorch3d.transforms"):
        sys.modules[name] = _Any(name)
    sys.modules["pytorch3d.ops"].interpolate_face_attributes = O.interpolate_face_attributes
    from src.utils import pytorch3d_envmap_shader as ref_shader

    gen = torch.Generator().manual_seed(13)
    Vn, F, Hr, Wr, B = 40, 60, 12, 10, 3
    verts = torch.randn(Vn, 3, generator=gen) * 0.5
    vnorm = torch.nn.functional.normalize(torch.randn(Vn, 3, generator=gen), dim=-1)
    faces = torch.randint(0, Vn, (F, 3), generator=gen)
    pix_to_face = torch.randint(-1, F, (1, Hr, Wr, 1), generator=gen)
    pix_to_face[0, 0, :3, 0] = -1
    bary = torch.rand(1, Hr, Wr, 1, 3, generator=gen)
    bary = bary / bary.sum(-1, keepdim=True)
    D = ref_utils.get_directions(16)          # [1, 8*16, 3]
    sw = ref_utils.get_sineweight(16)
    J = D.shape[1]
    env = torch.exp(torch.randn(B, J, 3, generator=gen))   # positive HDR-like radiance
    cam = torch.tensor([[0.0, 0.0, 2.0]])
    shin, kd = 500.0, 0.5

    class Meshes:
        def verts_packed(self): return verts
        def faces_packed(self): return faces
        def verts_normals_packed(self): return vnorm
    class Frag:
        pass
    fr = Frag(); fr.pix_to_face = pix_to_face; fr.bary_coords = bary
    class Cam:
        def get_camera_center(self): return cam
    class Mat:
        shininess = torch.tensor([shin])
    outs = {}
    for tag, s_ in (("", shin), ("_s20", 20.0)):
        Mat.shininess = torch.tensor([s_])
        envmap = ref_shader.EnvironmentMap(environment_map=env.clone().requires_grad_(True), directions=D.repeat(B, 1, 1),
                                           sineweight=sw.repeat(B, 1, 1))
        colors, pn = ref_shader.blinn_phong_shading_env_map("cpu", Meshes(), fr, envmap, Cam(), Mat(), kd, 1.0 - kd)
        w = torch.randn(colors.shape, generator=torch.Generator().manual_seed(5))
        (gC,) = torch.autograd.grad((colors * w).sum(), envmap.environment_map)
        outs["colors" + tag] = colors.detach().numpy()
        outs["dlight" + tag] = gC.numpy()
        outs["upstream" + tag] = w.numpy()
        outs["pixel_normals"] = pn.detach().numpy()
    save("g13_envmap_shader.npz", verts=verts.numpy(), vnorm=vnorm.numpy(), faces=faces.numpy(),
         pix_to_face=pix_to_face.numpy(), bary=bary.numpy(), directions=D.numpy(), sineweight=sw.numpy(), env=env.numpy(),
         cam=cam.numpy(), shininess=np.float32(shin), kd=np.float32(kd), **outs)


# ---------------------------------------------------------------- G14 / G15 long trajectories at the bench architecture
def _smooth_images(N, Himg, Wimg, seed):
    """Smooth synthetic environment maps in [-1, 1] (a few low-order harmonics of the direction per channel, random per image):
    something a SIREN can actually fit, so that the loss curve falls over hundreds of steps instead of sitting on a noise floor."""
    g = torch.Generator().manual_seed(seed)
    D = ref_utils.get_directions(Wimg)[0]                      # [P, 3]
    th = torch.atan2(D[:, 0], D[:, 2]); y = D[:, 1]
    basis = torch.stack([torch.ones_like(y), y, D[:, 0], D[:, 2], y * y, torch.sin(2 * th) * (1 - y * y), torch.cos(2 * th) * (1 - y * y),
                         y * D[:, 0], y * D[:, 2], torch.sin(3 * th) * (1 - y * y), torch.cos(3 * th) * (1 - y * y), y * y * y], 1)  # [P, 12]
    coef = torch.randn(N, 3, basis.shape[1], generator=g) * torch.tensor([0.3, 0.5, 0.4, 0.4, 0.3, 0.3, 0.3, 0.25, 0.25, 0.2, 0.2, 0.2])
    img = torch.tanh(torch.einsum("nck,pk->ncp", coef, basis))  # [N, 3, P]
    return img.reshape(N, 3, Himg, Wimg).contiguous()


def _c2_decoder():
    """The config-2 decoder g4_c2shape.npz already carries (seed 42, N = 2): G14 / G15 start from ITS weights, so neither fixture
    stores another megabyte of state dict (tests read `sd.net.*` from g4_c2shape.npz)."""
    torch.manual_seed(42)
    return ref.RENIAutoDecoder(2, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, False)


def g14():
    """Config-4 re-enactment at the BENCH architecture (examples.ipynb cell 4; RENI_module.py:126-128; loss_functions.py:60-71):
    ND = 36, 5 x 128, frozen decoder, 3 maps at 64 x 128, the real Mask-3, RENITestLoss(1e-7, 1e-4), Adam(lr 1e-1) on the latents
    from zero, 200 steps.  Recorded: the loss 4-tuple of steps 0, 10, ..., 190 and 199, the latents after 20, 100 and 200 steps, and
    the completed maps model(Z_after_200, D) -- the loop's product -- of the fp32 run and of the autocast-bf16 run (the latter as fp16:
    only a PSNR is taken from it)."""
    src = _c2_decoder()
    ckpt = {"model." + k: v.clone() for k, v in src.state_dict().items()}
    N, W = 3, 128
    m = ref.RENIAutoDecoder(N, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, True)
    m.load_state_dict(ckpt)
    assert float(m.Z.abs().sum()) == 0.0
    imgs = _smooth_images(N, W // 2, W, 140)
    mask_img = np.array(Image.open(os.path.join(REF, "data/Masks/Mask-3.png")))
    chw = torch.from_numpy(mask_img[..., :3].astype(np.float32) / 255.0).permute(2, 0, 1)   # (utils.py:81-91, see g7)
    mask = torch.nn.functional.interpolate(chw[None], size=(W // 2, W), mode="nearest")[0].permute(1, 2, 0).reshape(1, -1, 3)
    D1 = ref_utils.get_directions(W); S1 = ref_utils.get_sineweight(W) * mask
    crit = ref_loss.RENITestLoss(alpha=1e-7, beta=1e-4)
    t = imgs.permute(0, 2, 3, 1).reshape(N, -1, 3)
    D = D1.repeat(N, 1, 1); S = S1.repeat(N, 1, 1)
    idx = torch.arange(N)
    steps = 200

    def run(autocast):
        """autocast: the SAME reference code with its linear layers in bf16 (torch.autocast on the CPU) -- what the reference's own
        arithmetic does to this trajectory at bf16; the tests derive the band of the bf16 HIP kernels from it."""
        with torch.no_grad():
            m.Z.zero_()
        opt = torch.optim.Adam([m.Z], lr=1e-1)
        rec_at, terms, snaps = [], [], {}
        for it in range(steps):
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                Z = m.Z[idx, :, :]
                out = m(Z, D)
                opt.zero_grad()
                tl = crit(out.float(), t, S, Z)
            tl[0].backward()
            opt.step()
            if it % 10 == 0 or it == steps - 1:
                rec_at.append(it); terms.append([x.item() for x in tl])
            if it + 1 in (20, 100, 200):
                snaps[f"Z_after_{it + 1}"] = m.Z.detach().numpy().copy()
        # the PRODUCT of the inpainting loop (examples.ipynb cell 4: `model_output = model(Z, directions)` behind the loop): the
        # completed environment maps the final latents decode to, masked-out region included -- in this run's own arithmetic
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            img = m(m.Z[idx, :, :], D).float().numpy().copy()
        return rec_at, terms, snaps, img

    rec_at, terms, snaps, img_final = run(False)
    _, terms_ac, snaps_ac, img_final_ac = run(True)
    # the latent gradient AT the fp32 trajectory's latents (start, after 20 / 100 / 200 steps), in fp32 and under autocast: a
    # per-step comparison that does not go through 200 steps of Adam (trajectories diverge near stationarity, gradients do not)
    grads = {}
    for name, Zs in [("0", np.zeros_like(snaps["Z_after_20"]))] + [(str(k), snaps[f"Z_after_{k}"]) for k in (20, 100, 200)]:
        for ac in (False, True):
            with torch.no_grad():
                m.Z.copy_(torch.from_numpy(Zs))
            m.Z.grad = None
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=ac):
                Z = m.Z[idx, :, :]
                tl = crit(m(Z, D).float(), t, S, Z)
            tl[0].backward()
            grads[f"dZ_at_{name}" + ("_autocast_bf16" if ac else "")] = m.Z.grad.detach().float().numpy().copy()
    snaps.update(grads)
    snaps.update({k + "_autocast_bf16": v for k, v in snaps_ac.items()})
    save("g14_c4_trajectory.npz", imgs=imgs.numpy(), mask=mask.numpy(), rec_at=np.array(rec_at), terms=np.array(terms),
         terms_autocast_bf16=np.array(terms_ac),
         img_after_200=img_final.astype(np.float32), img_after_200_autocast_bf16=img_final_ac.astype(np.float16),
         W=np.int64(W), steps=np.int64(steps), lr=np.float64(1e-1), alpha=np.float64(1e-7), beta=np.float64(1e-4), **snaps)


def g15():
    """Config-2 training at the bench architecture (RENI_module.py:80-146; run.py): N = 8 images at 32 x 64, batches of 4 in loader
    order, RENITrainLoss, Adam(lr 1e-3) over decoder + latents, 100 steps.  Recorded: the loss of every step, the final latents and
    norms / head values of every final decoder parameter."""
    src = _c2_decoder()
    N, B, W = 8, 4, 64
    m = ref.RENIAutoDecoder(N, 36, "SO2", 128, 5, 3, True, "tanh", 30, 30, False)
    m.net.load_state_dict(src.net.state_dict())
    with torch.no_grad():
        m.Z.copy_(torch.randn(N, 36, 3, generator=torch.Generator().manual_seed(150)))
    Z0 = m.Z.detach().clone()
    net0 = {k: v.clone() for k, v in m.net.state_dict().items()}
    imgs_all = _smooth_images(N, W // 2, W, 151)
    D1 = ref_utils.get_directions(W); S1 = ref_utils.get_sineweight(W)
    crit = ref_loss.RENITrainLoss()
    steps = 100

    def run(autocast):
        with torch.no_grad():
            m.Z.copy_(Z0)
        m.net.load_state_dict(net0)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        losses = []
        for it in range(steps):
            idx_t = torch.arange(B) + (it % (N // B)) * B
            imgs = imgs_all[idx_t]
            t = imgs.permute(0, 2, 3, 1).reshape(B, -1, 3)
            D = D1.repeat(B, 1, 1); S = S1.repeat(B, 1, 1)
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                Z = m.Z[idx_t, :, :]
                out = m(Z, D)
                loss = crit(out.float(), t, S)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses, m.Z.detach().numpy().copy()

    losses_ac, Z_final_ac = run(True)
    losses, _ = run(False)
    arrs = {}
    for k, p in m.named_parameters():
        if k == "Z":
            continue
        v = p.detach().numpy()
        arrs["fn." + k] = np.float64(np.linalg.norm(v.astype(np.float64)))
        arrs["fh." + k] = v.reshape(-1)[:32].copy()
    save("g15_c2_trajectory.npz", imgs=imgs_all.numpy(), Z0=Z0.numpy(), losses=np.array(losses), Z_final=m.Z.detach().numpy(),
         losses_autocast_bf16=np.array(losses_ac), Z_final_autocast_bf16=Z_final_ac,
         W=np.int64(W), B=np.int64(B), steps=np.int64(steps), lr=np.float64(1e-3), **arrs)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15"]
    for w in which:
        globals()[w]()
