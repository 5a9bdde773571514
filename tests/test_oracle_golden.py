"""Pins the CPU oracle (oracle/reni_oracle.py) to the golden vectors generated from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import reni_oracle as O


def _sd(g, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def _spec_from(sd, nd, eq, H, L, lll=True, act="tanh"):
    return O.DecoderSpec(nd, eq, H, L, 3, lll, act, 30.0, 30.0)


def test_g1_grids_bitwise(golden):
    g = golden("g1_grids.npz")
    for W in (32, 64):
        assert np.array_equal(O.get_directions(W).numpy(), g[f"dir_{W}"])
        assert np.array_equal(O.get_sineweight(W).numpy(), g[f"sw_{W}"])
    for W in (256, 1024):
        d = O.get_directions(W); s = O.get_sineweight(W)
        assert np.array_equal(d[0, :64].numpy(), g[f"dir_{W}_head"])
        assert np.array_equal(d[0, -64:].numpy(), g[f"dir_{W}_tail"])
        assert np.array_equal(s[0, -64:].numpy(), g[f"sw_{W}_tail"])
        np.testing.assert_allclose(d.double().sum((0, 1)).numpy(), g[f"dir_{W}_sum64"], atol=1e-9)
        np.testing.assert_allclose(s.double().sum().numpy(), g[f"sw_{W}_sum64"], rtol=1e-12)
    d = O.get_directions(256)
    assert float((d.norm(dim=-1) - 1).abs().max()) < 3e-7
    assert abs(float(O.get_sineweight(256).mean()) - 2 / np.pi) < 1e-4


def test_g2_encodings(golden):
    g = golden("g2_encodings.npz")
    for tag in ("a", "b"):
        Z = torch.from_numpy(g[f"Z_{tag}"]); D = torch.from_numpy(g[f"D_{tag}"])
        for eq, key in (("SO2", "so2"), ("SO3", "so3"), ("None", "none")):
            x = O.encode(eq, Z, D).numpy()
            assert x.shape == g[f"{key}_{tag}"].shape
            assert x.shape[-1] == O.in_features(eq, Z.shape[1])
            np.testing.assert_allclose(x, g[f"{key}_{tag}"], atol=1e-6, rtol=0)


def test_g3_c1_forward(golden):
    g = golden("g3_c1_forward.npz")
    sd = _sd(g)
    spec = O.DecoderSpec(9, "SO2", 64, 3, 3, True, None)
    assert [k for k in sd if k != "Z"] == spec.param_keys()
    out = O.reni_forward(spec, sd, sd["Z"][[0]], O.get_directions(64))
    np.testing.assert_allclose(out.numpy(), g["out"], atol=2e-6, rtol=0)
    # factored float64 restatement agrees too
    f = O.factored_fwd_bwd(spec, {k: v.numpy() for k, v in sd.items() if k != "Z"},
                           sd["Z"][[0]].numpy(), O.get_directions(64).numpy())
    np.testing.assert_allclose(f["out"], g["out"], atol=5e-6, rtol=0)


@pytest.mark.parametrize("name,eq,lll,act", [
    ("g4_small.npz", "SO2", True, "tanh"),
    ("g4_small_so3.npz", "SO3", True, "tanh"),
    ("g4_small_none.npz", "None", True, None),
    ("g4_small_sinehead.npz", "SO2", False, None),
])
def test_g4_small_fwd_bwd(golden, name, eq, lll, act):
    g = golden(name)
    sd = _sd(g)
    params = {k: v for k, v in sd.items() if k != "Z"}
    spec = O.DecoderSpec(9, eq, 64, 3, 3, lll, act)
    W = int(g["W"])
    D = O.get_directions(W); S = O.get_sineweight(W)
    Z = torch.from_numpy(g["Z"]); t = torch.from_numpy(g["target"])
    r = O.fwd_loss_bwd(spec, params, Z, D.repeat(2, 1, 1), t, S.repeat(2, 1, 1))
    np.testing.assert_allclose(r["out"].numpy(), g["out"], atol=2e-6)
    assert abs(r["loss_terms"][0] - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    assert O.rel_l2(r["dZ"].numpy(), g["dZ"]) < 1e-5
    for k in params:
        assert O.rel_l2(r["grads"][k].numpy(), g["g." + k]) < 1e-5, k
    # factored hand-derived backward (float64) vs the reference's autograd
    f = O.factored_fwd_bwd(spec, {k: v.numpy() for k, v in params.items()}, g["Z"], D.numpy(),
                           g["target"], S.numpy())
    np.testing.assert_allclose(f["out"], g["out"], atol=5e-6)
    assert abs(f["loss_terms"][0] - float(g["loss"])) < 2e-6 * abs(float(g["loss"]))
    assert O.rel_l2(f["dZ"], g["dZ"]) < 2e-5
    for k in params:
        assert O.rel_l2(f["grads"][k], g["g." + k]) < 2e-5, k
    # fp32 torch restatement of the factored algebra (bench.py's "factored" CPU figure) vs the reference's goldens
    if act != "exp":
        ft = O.factored_torch_fwd_loss_bwd(spec, params, Z, D, t, S.repeat(2, 1, 1))
        np.testing.assert_allclose(ft["out"].numpy(), g["out"], atol=5e-6)
        assert abs(ft["loss_terms"][0] - float(g["loss"])) < 2e-6 * abs(float(g["loss"]))
        assert O.rel_l2(ft["dZ"].numpy(), g["dZ"]) < 2e-5
        for k in params:
            assert O.rel_l2(ft["grads"][k].numpy(), g["g." + k]) < 2e-5, k


def test_g4_c2_shape(golden):
    g = golden("g4_c2shape.npz")
    sd = _sd(g)
    params = {k: v for k, v in sd.items() if k != "Z"}
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    W = int(g["W"])
    D = O.get_directions(W); S = O.get_sineweight(W)
    f = O.factored_fwd_bwd(spec, {k: v.numpy() for k, v in params.items()}, g["Z"], D.numpy(),
                           g["target"], S.numpy())
    np.testing.assert_allclose(f["out"][:, :256], g["out_head"], atol=5e-6)
    assert abs(f["out"].sum() - float(g["out_sum64"])) < 1e-2
    assert abs(f["loss_terms"][0] - float(g["loss"])) < 2e-6 * abs(float(g["loss"]))
    assert O.rel_l2(f["dZ"], g["dZ"]) < 3e-5
    for k in params:
        assert abs(np.linalg.norm(f["grads"][k]) / float(g["gn." + k]) - 1) < 3e-5, k
        np.testing.assert_allclose(f["grads"][k].reshape(-1)[:32], g["gh." + k],
                                   atol=3e-5 * float(g["gn." + k]))


def test_g5_losses(golden):
    g = golden("g5_losses.npz")
    o = torch.from_numpy(g["o"]).requires_grad_(True)
    t = torch.from_numpy(g["t"]); s = torch.from_numpy(g["s"])
    Z = torch.from_numpy(g["Z"]).requires_grad_(True)
    mu = torch.from_numpy(g["mu"]).requires_grad_(True)
    lv = torch.from_numpy(g["lv"]).requires_grad_(True)
    v = O.weighted_mse(o, t, s); (go,) = torch.autograd.grad(v, o)
    np.testing.assert_allclose(v.item(), g["mse"], rtol=1e-6); np.testing.assert_allclose(go.numpy(), g["mse_go"], atol=1e-8)
    v = O.weighted_cosine(o, t, s); (go,) = torch.autograd.grad(v, o)
    np.testing.assert_allclose(v.item(), g["cos"], rtol=1e-6); np.testing.assert_allclose(go.numpy(), g["cos_go"], atol=1e-8)
    v = O.kld(mu, lv, 27); gm, gl = torch.autograd.grad(v, (mu, lv))
    np.testing.assert_allclose(v.item(), g["kld"], rtol=1e-6)
    np.testing.assert_allclose(gm.numpy(), g["kld_gmu"], atol=1e-7); np.testing.assert_allclose(gl.numpy(), g["kld_glv"], atol=1e-7)
    tl = O.test_loss(o, t, s, Z, 1e-7, 1e-1)
    np.testing.assert_allclose([x.item() for x in tl], g["test"], rtol=1e-6)
    go, gz = torch.autograd.grad(tl[0], (o, Z))
    np.testing.assert_allclose(go.numpy(), g["test_go"], atol=1e-8); np.testing.assert_allclose(gz.numpy(), g["test_gz"], atol=1e-10)
    vl = O.vad_train_loss(o, t, s, mu, lv, 1e-4, 27)
    np.testing.assert_allclose([x.item() for x in vl], g["vad"], rtol=1e-6)


def test_g5_factored_cosine_gradient(golden):
    """The hand-derived cosine-term gradient equals autograd of the reference."""
    g = golden("g5_losses.npz")
    o, t, s = g["o"].astype(np.float64), g["t"].astype(np.float64), g["s"].astype(np.float64)
    beta = 1e-1
    so_t = (o * t).sum(1); n_o = np.sqrt((o ** 2).sum(1)); n_t = np.sqrt((t ** 2).sum(1))
    den = np.maximum(n_o * n_t, 1e-20); cs = so_t / den
    coef = -1.0 * s[:, 0, :] / 3.0
    go = coef[:, None, :] * (t / den[:, None, :] - cs[:, None, :] * o / (n_o ** 2)[:, None, :])
    np.testing.assert_allclose(go, g["cos_go"], atol=1e-8)
    P = o.shape[1]
    full = 2 * s * (o - t) / (3 * P) + beta * go
    np.testing.assert_allclose(full, g["test_go"], atol=1e-8)


def test_g10_invariance_identities(golden):
    g = golden("g10_equivariance.npz")
    D = O.get_directions(64)
    for eq, Rk in (("SO2", "Ry"), ("SO3", "R3")):
        sd = _sd(g, f"sd_{eq}.")
        spec = O.DecoderSpec(49, eq, 128, 5, 3, True, "tanh")
        Z = torch.from_numpy(g[f"Z_{eq}"]); R = torch.from_numpy(g[Rk])
        a = O.reni_forward(spec, sd, Z, D)
        b = O.reni_forward(spec, sd, Z @ R.T, D @ R.T)
        np.testing.assert_allclose(a[0, :128].numpy(), g[f"out_{eq}_head"], atol=3e-6)
        assert float((a - b).abs().max()) < 5e-6
        assert float(g[f"resid_{eq}"]) < 5e-6


# ---- G11: FiLM conditioning ---------------------------------------------------------
@pytest.mark.parametrize("tag", ["so2_ad", "so3_vad", "so2_one"])
def test_g11_film_oracle_matches_reference(golden, tag):
    """film_encode / film_mapping / film_forward / film_fwd_loss_bwd against the reference's RENI*FiLM modules
    (forward, RENITrainLoss gradients of every parameter and of Z, RENITestLoss terms and latent gradient)."""
    g = golden(f"g11_film_{tag}.npz")
    eq, nd, H, nF, mf, ml, act = [int(x) for x in g["cfg"]]
    spec = O.FilmSpec(nd, {1: "SO2", 2: "SO3"}[eq], H, nF, mf, ml, 3, {0: None, 1: "tanh", 2: "exp"}[act])
    params = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.") and k[3:] not in ("Z", "mu", "log_var")}
    assert sorted(params) == sorted(spec.param_keys())
    W = int(g["W"])
    D = O.get_directions(W).repeat(2, 1, 1)
    S = O.get_sineweight(W).repeat(2, 1, 1)
    Z, T = torch.from_numpy(g["Z"]), torch.from_numpy(g["target"])
    si, mi = O.film_encode(spec, Z, D)
    assert np.abs(si[:, :8].numpy() - g["siren_input_head"]).max() <= 1e-6
    f, ph = O.film_mapping(spec, params, mi)
    assert np.abs(f.numpy() - g["freq_raw"]).max() <= 2e-6 and np.abs(ph.numpy() - g["phase"]).max() <= 2e-6
    r = O.film_fwd_loss_bwd(spec, params, Z, D, T, S)
    assert np.abs(r["out"].numpy() - g["out"]).max() <= 1e-6
    assert abs(r["terms"][0] - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    assert O.rel_l2(r["dZ"].numpy(), g["dZ"]) <= 5e-6
    for k in params:
        assert O.rel_l2(r["grads"][k].numpy(), g["g." + k]) <= 5e-6, k
    r2 = O.film_fwd_loss_bwd(spec, params, Z, D, T, S, "test", 1e-3, 1e-1)
    assert np.abs(np.array(r2["terms"]) - g["test_terms"]).max() <= 1e-6 * abs(g["test_terms"][0])
    assert O.rel_l2(r2["dZ"].numpy(), g["test_dZ"]) <= 5e-6


def _g13_gbuffer(g):
    verts, vnorm, faces = (torch.from_numpy(g[k]) for k in ("verts", "vnorm", "faces"))
    p2f, bary = torch.from_numpy(g["pix_to_face"]), torch.from_numpy(g["bary"])
    pn = O.interpolate_face_attributes(p2f, bary, vnorm[faces])[0, :, :, 0, :].reshape(-1, 3)
    pp = O.interpolate_face_attributes(p2f, bary, verts[faces])[0, :, :, 0, :].reshape(-1, 3)
    return pn, pp


@pytest.mark.parametrize("tag,shin,rtol", [("", 500.0, 2e-4), ("_s20", 20.0, 2e-5)])
def test_g13_envmap_shader_oracle_matches_reference(golden, tag, shin, rtol):
    """The fp64 restatement of blinn_phong_shading_env_map against the reference's own fp32 run (colours, the
    gradient w.r.t. the sine-weighted map, the normalised normals).  x ** 500 amplifies fp32 rounding of the
    half-vector dot product 500-fold: 2e-4 of the largest colour is the reference's own fp32 noise floor."""
    g = golden("g13_envmap_shader.npz")
    pn, pp = _g13_gbuffer(g)
    B = g["env"].shape[0]
    L = torch.from_numpy(g["directions"]).repeat(B, 1, 1)
    C = (torch.from_numpy(g["env"]) * torch.from_numpy(g["sineweight"])).double().requires_grad_(True)
    col = O.blinn_phong_gbuffer(pn, pp, torch.from_numpy(g["cam"])[0], L, C, shin, float(g["kd"]), 1.0 - float(g["kd"]))
    ref = torch.from_numpy(g["colors" + tag]).reshape(B, -1, 3).double()
    assert (col - ref).abs().max() <= rtol * ref.abs().max()
    w = torch.from_numpy(g["upstream" + tag]).reshape(B, -1, 3).double()
    (gC,) = torch.autograd.grad((col * w).sum(), C)
    gref = torch.from_numpy(g["dlight" + tag]).double()
    assert (gC - gref).abs().max() <= rtol * gref.abs().max()
    nrm = torch.nn.functional.normalize(pn, dim=-1, eps=1e-6)
    assert np.allclose(nrm.numpy(), g["pixel_normals"][0].reshape(-1, 3), atol=1e-6)
    # background pixels (pix_to_face < 0) render black
    bg = torch.from_numpy(g["pix_to_face"]).reshape(-1) < 0
    assert bg.any() and float(ref[:, bg].abs().max()) == 0.0 and float(col[:, bg].abs().max()) == 0.0


def _adam(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam's update (no amsgrad, no weight decay), in place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.addcdiv_(m / (1 - b1 ** t), (v / (1 - b2 ** t)).sqrt() + eps, value=-lr)


def test_g14_g15_trajectories_start_from_the_oracle(golden):
    """G14 / G15 (the reference's 200- and 100-step runs at the bench architecture): the oracle reproduces the first recorded losses,
    and 20 oracle steps of G14 (frozen decoder, masked RENITestLoss, Adam on the latents) land on the reference's latents."""
    g4 = golden("g4_c2shape.npz")
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    params = {k: v for k, v in _sd(g4).items() if k != "Z"}
    g = golden("g14_c4_trajectory.npz")
    N, W = g["imgs"].shape[0], int(g["W"])
    D = O.get_directions(W).expand(N, -1, 3)
    S = (O.get_sineweight(W) * torch.from_numpy(g["mask"])).expand(N, -1, 3)
    T = torch.from_numpy(g["imgs"]).permute(0, 2, 3, 1).reshape(N, -1, 3)
    Z = torch.zeros(N, 36, 3)
    m, v = torch.zeros_like(Z), torch.zeros_like(Z)
    rec = {int(s): i for i, s in enumerate(g["rec_at"])}
    for it in range(20):
        r = O.fwd_loss_bwd(spec, params, Z, D, T, S, "test", float(g["alpha"]), float(g["beta"]), need_dw=False)
        if it in rec:
            np.testing.assert_allclose(r["loss_terms"], g["terms"][rec[it]], rtol=2e-5, atol=1e-9)
        _adam(Z, r["dZ"], m, v, it + 1, float(g["lr"]))
    assert float((Z - torch.from_numpy(g["Z_after_20"])).abs().max()) <= 2e-3
    # the loop's product: the oracle decodes the reference's final latents to the reference's completed maps (round 5: img_after_200)
    img = O.reni_forward(spec, params, torch.from_numpy(g["Z_after_200"]), D)
    assert float((img - torch.from_numpy(g["img_after_200"])).abs().max()) <= 5e-6
    assert g["img_after_200_autocast_bf16"].shape == g["img_after_200"].shape
    g = golden("g15_c2_trajectory.npz")
    B, W = int(g["B"]), int(g["W"])
    D = O.get_directions(W).expand(B, -1, 3); S = O.get_sineweight(W).expand(B, -1, 3)
    T = torch.from_numpy(g["imgs"][:B]).permute(0, 2, 3, 1).reshape(B, -1, 3)
    r = O.fwd_loss_bwd(spec, params, torch.from_numpy(g["Z0"][:B]), D, T, S)
    assert abs(r["loss_terms"][0] - g["losses"][0]) <= 2e-6 * g["losses"][0]
