"""Long optimisation trajectories at the BENCH architecture (ND = 36, 5 x 128) against the reference's own runs (goldens G14, G15):
what the single-step parity tests cannot show -- whether the bf16 persistent kernels (the ones bench.py times) TRACK the reference's
loss curve when their per-step gradient error (<= 6.5e-3 rel-L2) goes through Adam's normalisation hundreds of times.

G14 = BASELINE config 4 in the small: frozen decoder, 3 maps at 64 x 128, the real Mask-3, RENITestLoss(1e-7, 1e-4), Adam(lr 0.1) on
      the latents from zero, 200 steps (examples.ipynb cell 4; RENI_module.py:126-128; loss_functions.py:60-71).
G15 = BASELINE config 2 in the small: 8 images at 32 x 64, batches of 4, RENITrainLoss, Adam(lr 1e-3) over decoder + latents, 100 steps.

Both fixtures also carry the SAME reference code run under torch.autocast(bfloat16) on the CPU ("the reference's own arithmetic at
bf16").  The bf16 band asserted here is derived from it: at every recorded step the HIP bf16 loss may deviate from the reference's fp32
loss by at most max(FLOOR, 3 x the LARGEST deviation the autocast run shows anywhere on its curve), never more than CAP.

FINDING (round 4, profiles/r04_trajectory.md).  In the latent-optimisation loop the latents themselves do not stay pinned at bf16 -- for
ANY bf16 arithmetic: the reference's own autocast run ends at cosine 0.66 to its fp32 latents, the HIP kernels at 0.48, while the fp32
kernels end at 0.999999.  The cause is not a per-step defect: AT the reference's latents the bf16 kernels' gradient is 3 x closer to the
fp32 gradient than the reference's own bf16 gradient at every checkpoint (asserted below); but the loop runs into a near-stationary
region (|dZ| falls 13-fold by step 100), where the gradient is a small difference of large per-pixel terms and a bf16 forward pass leaves
22 % (HIP) / 71 % (autocast) relative error in it, and Adam's normalisation turns sign flips of near-zero components into full-size
steps.  The loss the loop is run for stays within 0.75 % of the reference's curve.  Latent-level fidelity needs the fp32 kernels.
(Round 6: with the frozen-decoder calls' W^T images made exact multiples of the forward images the bf16 kernels end at cosine 0.76 and
their completed maps at 48.9 / 47.6 dB -- ahead of the autocast run on both counts; FINDING 2 in test_c4_latent_trajectory_bf16_g14.)
"""
import os

import numpy as np
import pytest
import torch

from tests.util import load_golden

pytestmark = pytest.mark.gpu

FLOOR, CAP = 2e-3, 1.5e-2       # relative to the reference loss at that step
F32_BAND = 1e-3                 # fp32 kernels: relative deviation of every recorded loss


def _decoder_sd():
    g4 = load_golden("g4_c2shape.npz")   # seed-42 config-2 decoder: the weights G14 / G15 start from (make_golden._c2_decoder)
    return {k[len("sd."):]: torch.from_numpy(v) for k, v in g4.items() if k.startswith("sd.net.")}


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def _band(ref, autocast):
    dev_ac = float((np.abs(autocast - ref) / np.abs(ref)).max())
    return min(max(FLOOR, 3.0 * dev_ac), CAP)


def _run_g14(dtype, dev, sparse=False, imgs=None):
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    from reni_amd.utils import get_directions, get_sineweight
    g = load_golden("g14_c4_trajectory.npz")
    N, W = g["imgs"].shape[0], int(g["W"])
    m = RENIAutoDecoder(N, 36, "SO2", 128, 5, 3, True, "tanh", 30.0, 30.0, True)
    m.load_state_dict({"model." + k: v for k, v in _decoder_sd().items()})
    assert float(m.Z.detach().abs().sum()) == 0.0
    m.set_compute_dtype(dtype).to(dev)
    D = get_directions(W).to(dev)
    S = (get_sineweight(W) * torch.from_numpy(g["mask"])).to(dev)
    imgs = (torch.from_numpy(g["imgs"]) if imgs is None else imgs).to(dev)   # (imgs: the perturbed targets of the ensemble test)
    P = D.shape[1]
    eng = TrainEngine(m, lr=float(g["lr"]), loss_kind="test", alpha=float(g["alpha"]), beta=float(g["beta"]), sparse_weight=sparse)
    idx = torch.arange(N, device=dev)
    tgt = imgs.permute(0, 2, 3, 1).view(N, P, 3)   # the reference's permute + view (RENI_module.py:83-84), never copied
    terms, snaps = [], {}
    for it in range(int(g["steps"])):
        t = eng.step(idx, tgt, S, D)
        if it in g["rec_at"]:
            terms.append(t.detach().cpu().double().numpy())
        if it + 1 in (20, 100, 200):
            snaps[it + 1] = m.Z.detach().cpu().numpy().copy()
    # the loop's PRODUCT (examples.ipynb cell 4; RENI_module.py:126-128): the completed maps the final latents decode to
    with torch.no_grad():
        snaps["img"] = m(m.Z.data, D).detach().float().cpu().numpy()
    return g, np.array(terms), snaps


def _psnr(x, y, sel):
    """PSNR in dB over the pixels `sel` of [N, P, 3] images in [-1, 1] (peak-to-peak 2)."""
    d = np.asarray(x, np.float64)[:, sel] - np.asarray(y, np.float64)[:, sel]
    return float(10.0 * np.log10(4.0 / np.mean(d * d)))


def test_c4_latent_trajectory_f32_g14():
    dev = torch.device("cuda:0")
    g, terms, snaps = _run_g14("f32", dev)
    ref = g["terms"]
    rel = np.abs(terms[:, 0] - ref[:, 0]) / ref[:, 0]
    print("G14 f32: max rel loss deviation", rel.max(), "final-latent cos", _cos(snaps[200], g["Z_after_200"]))
    assert rel.max() <= F32_BAND, rel
    np.testing.assert_allclose(terms[:, 2], ref[:, 2], rtol=5e-2, atol=1e-7)   # the prior term alpha |Z|^2 follows the latents
    np.testing.assert_allclose(terms[:, 3], ref[:, 3], rtol=1e-4)
    assert _cos(snaps[20], g["Z_after_20"]) >= 0.9999
    # the completed maps -- the only thing a user of the inpainting loop looks at -- against the reference's,
    # every pixel of every channel, the masked-out region included
    # (200 Adam(0.1) steps amplify the last-bit differences of two fp32 arithmetics -- the latents end at cosine 0.999999, not 1 -- so
    # the images agree to ~1e-2 at the worst pixel, not to the 1e-4 of a single forward pass: the bound is in image terms)
    err = np.abs(snaps["img"] - g["img_after_200"])
    allpix = np.ones(err.shape[1], bool)
    print("G14 f32: final image max abs deviation", err.max(), "rms", float(np.sqrt((err ** 2).mean())), "PSNR %.2f dB" % _psnr(snaps["img"], g["img_after_200"], allpix))
    assert err.max() <= 2e-2 and _psnr(snaps["img"], g["img_after_200"], allpix) >= 55.0   # (measured: 8.6e-3, 59.0 dB)


def test_c4_latent_trajectory_bf16_g14():
    """The frozen + statistics instances of the persistent bf16 kernels over 200 Adam(0.1) steps."""
    dev = torch.device("cuda:0")
    g, terms, snaps = _run_g14("bf16", dev)
    ref, ac = g["terms"], g["terms_autocast_bf16"]
    rel = np.abs(terms[:, 0] - ref[:, 0]) / ref[:, 0]
    band = _band(ref[:, 0], ac[:, 0])
    cos = {k: _cos(snaps[k], g[f"Z_after_{k}"]) for k in (20, 100, 200)}
    cos_ac = {k: _cos(g[f"Z_after_{k}_autocast_bf16"], g[f"Z_after_{k}"]) for k in (20, 100, 200)}
    print("G14 bf16: rel loss deviation per recorded step", np.array2string(rel, precision=5), "band", band)
    print("G14 bf16: latent cos vs reference", cos, "| the reference's own autocast-bf16 run:", cos_ac)
    assert (rel <= band).all(), (rel, band)
    # the loss at the end is what the inpainting loop is run for: within 1 % of the reference's
    assert abs(terms[-1, 0] - ref[-1, 0]) <= 1e-2 * ref[-1, 0]
    # the latents: pinned while the gradient is well-conditioned (20 steps); behind that see FINDING in the module docstring.  Kept as a
    # regression guard: at 100 and 200 steps at least half the cosine the reference's own autocast run keeps -- and, since
    # round 6 (the W^T images of a frozen-decoder call are exact multiples of the forward images: FINDING 2 below), MORE than it keeps
    # at the end (measured 0.759 against autocast's 0.664; round 5: 0.485)
    assert cos[20] >= 0.98, cos
    assert cos[100] >= 0.5 * cos_ac[100] and cos[200] >= 0.5 * cos_ac[200], (cos, cos_ac)
    assert cos[200] >= cos_ac[200], (cos, cos_ac)
    # What is ASSERTED about the loop's product: the completed maps against the reference's fp32
    # result over the MASKED-OUT pixels (what inpainting is for) and over the kept ones, no worse than what the reference's own code
    # does under autocast-bf16, less 1 dB.
    # FINDING 2 (round 6, profiles/r06_trajectory.md).  Round 5's kernels missed that by 4-5 dB (43.6 dB against 48.1 dB) although their
    # per-step gradient was 3 x closer to the fp32 gradient than autocast's.  A perturbation ensemble (test below) showed the gap was
    # systematic, not the loop's chaos (43.57 +- 0.15 dB over seven runs), and the generic bf16 kernel -- same per-step gradient error --
    # ended at 51.1 dB.  The cause: the persistent kernels' W^T images were rounded independently of the forward images
    # (bf16(omega W) against bf16(W omega / 2 pi): the ratio 2 pi is no power of two), so the backward pass was the gradient of another
    # network than the one the forward pass evaluates, and the loop's fixed point -- where THAT gradient vanishes -- is not a minimum of
    # the loss the forward pass measures: a first-order image error.  With the W^T images 8 x the forward images (and the rest of the
    # factor on d loss / d y): 48.9 / 47.6 dB.  None of the approximations VERDICT r05 listed (fp16 phase stash, v_cos_f16, unreduced
    # v_sin_f32 arguments) moved the result by more than 1.3 dB in either direction (RENI_ABL 1, 2, 4 in reni_dev_common.inc).
    masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
    ref_img, ac_img = g["img_after_200"], g["img_after_200_autocast_bf16"].astype(np.float32)
    for name, sel in (("masked-out", masked_out), ("kept", ~masked_out)):
        p_hip, p_ac = _psnr(snaps["img"], ref_img, sel), _psnr(ac_img, ref_img, sel)
        print(f"G14 bf16: final image PSNR vs the reference's fp32 image, {name} pixels: HIP {p_hip:.2f} dB, reference under autocast {p_ac:.2f} dB")
        assert p_hip >= p_ac - 1.0 and p_hip >= 46.0, (name, p_hip, p_ac)


def test_c4_latent_trajectory_bf16_ensemble_g14():
    """One trajectory is one draw.  tests/golden/g14_ensemble.npz (make_g14_ensemble.py) holds the reference's own code -- fp32 and under
    autocast(bfloat16) -- on six copies of G14 whose TARGETS carry 1e-6 x N(0, 1) noise: the spread of its final-image PSNR (against
    the unperturbed fp32 run) is what the loop does to a perturbation no arithmetic avoids (autocast: 46.4 .. 48.6 dB masked-out,
    44.6 .. 47.8 dB kept).  The bf16 kernels on the same six targets (rebuilt from the seeds): the MEAN over the ensemble no worse than
    autocast's mean less 1 dB, every member above 45 dB, and the final latents closer to the reference's than autocast's are."""
    dev = torch.device("cuda:0")
    ens = load_golden("g14_ensemble.npz")
    g = load_golden("g14_c4_trajectory.npz")
    rows = ens["rows"]
    masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
    imgs0 = torch.from_numpy(g["imgs"])
    got = []
    for r in rows:
        seed = int(r[0])
        imgs = imgs0 + float(ens["noise"]) * torch.randn(imgs0.shape, generator=torch.Generator().manual_seed(int(ens["seed_base"]) + seed))
        _, _, snaps = _run_g14("bf16", dev, imgs=imgs)
        got.append((_psnr(snaps["img"], g["img_after_200"], masked_out), _psnr(snaps["img"], g["img_after_200"], ~masked_out),
                    _cos(snaps[200], g["Z_after_200"])))
        print("G14 ensemble seed %d: HIP bf16 %.2f / %.2f dB cos %.3f | reference fp32 %.2f / %.2f dB | autocast-bf16 %.2f / %.2f dB cos %.3f"
              % (seed, *got[-1], r[1], r[2], r[4], r[5], r[6]))
    got = np.array(got)
    print("G14 ensemble means: HIP bf16 %.2f / %.2f dB cos %.3f | autocast-bf16 %.2f / %.2f dB cos %.3f"
          % (got[:, 0].mean(), got[:, 1].mean(), got[:, 2].mean(), rows[:, 4].mean(), rows[:, 5].mean(), rows[:, 6].mean()))
    assert got[:, 0].mean() >= rows[:, 4].mean() - 1.0 and got[:, 1].mean() >= rows[:, 5].mean() - 1.0, (got.mean(0), rows.mean(0))
    assert got[:, :2].min() >= 45.0, got
    assert got[:, 2].mean() >= rows[:, 6].mean(), (got[:, 2], rows[:, 6])


def test_c4_latent_trajectory_with_sparse_weight_is_the_same_trajectory_g14():
    """RENI_WEIGHT_SPARSE (what RENI.training_step sets with a mask): Mask-3's zero-weight tiles and the statistics pass of its
    constant cosine term are left out -- and all 200 Adam(0.1) steps, losses and latents, come out EQUAL to the dense run's."""
    dev = torch.device("cuda:0")
    _, terms_d, snaps_d = _run_g14("bf16", dev)
    _, terms_s, snaps_s = _run_g14("bf16", dev, sparse=True)
    assert np.array_equal(terms_s, terms_d)
    for k in (20, 100, 200, "img"):
        assert np.array_equal(snaps_s[k], snaps_d[k]), k


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_c4_latent_gradient_at_the_reference_trajectory_g14(dtype):
    """The per-step statement behind the trajectory: AT the latents the reference's fp32 run passes through (start, after 20, 100, 200
    steps) the kernels' latent gradient against the reference's -- fp32 to 1e-4; bf16 at least as close as the reference's own
    arithmetic under autocast(bfloat16) at the same latents (fixture: dZ_at_k, dZ_at_k_autocast_bf16), and never worse than 0.3 even
    where the gradient has all but vanished (step 100: |dZ| 1.8e-4 against 2.4e-3 at step 20)."""
    from reni_amd.ops import Plan
    from reni_amd.utils import get_directions, get_sineweight
    dev = torch.device("cuda:0")
    g = load_golden("g14_c4_trajectory.npz")
    N, W = g["imgs"].shape[0], int(g["W"])
    sd = _decoder_sd()
    keys = ["net.%d.linear.%s" % (i, n) for i in range(6) for n in ("weight", "bias")] + ["net.6.weight", "net.6.bias"]
    fp = torch.cat([sd[k].reshape(-1).float() for k in keys]).to(dev)
    plan = Plan("SO2", 36, 128, 5, 3, True, "tanh", 30.0, 30.0, dtype)
    D = get_directions(W).to(dev)
    S = (get_sineweight(W) * torch.from_numpy(g["mask"])).to(dev)
    T = torch.from_numpy(g["imgs"]).to(dev).permute(0, 2, 3, 1).reshape(N, -1, 3)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    for k in (0, 20, 100, 200):
        Z = torch.zeros(N, 36, 3) if k == 0 else torch.from_numpy(g[f"Z_after_{k}"])
        _, dZ, _, _ = plan.forward_loss_backward(Z.to(dev), D, fp, T, S, loss_kind="test", alpha=float(g["alpha"]), beta=float(g["beta"]),
                                                 need_dw=False)
        e = rel(dZ.cpu().numpy(), g[f"dZ_at_{k}"])
        e_ac = rel(g[f"dZ_at_{k}_autocast_bf16"], g[f"dZ_at_{k}"])
        print(f"G14 {dtype}: latent gradient at the reference's step {k}: rel-L2 {e:.3e} (the reference under autocast-bf16: {e_ac:.3e})")
        if dtype == "f32":
            assert e <= 1e-4, (k, e)
        else:
            assert e <= min(e_ac, 0.3), (k, e, e_ac)


def _run_g15(dtype, dev):
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    from reni_amd.utils import get_directions, get_sineweight
    g = load_golden("g15_c2_trajectory.npz")
    N, B, W = g["imgs"].shape[0], int(g["B"]), int(g["W"])
    m = RENIAutoDecoder(N, 36, "SO2", 128, 5, 3, True, "tanh", 30.0, 30.0, False)
    sd = {"model." + k: v for k, v in _decoder_sd().items()}
    sd["model.Z"] = torch.from_numpy(g["Z0"])
    m.load_state_dict(sd)
    m.set_compute_dtype(dtype).to(dev)
    D = get_directions(W).to(dev)
    S = get_sineweight(W).to(dev)
    imgs = torch.from_numpy(g["imgs"]).to(dev)
    P = D.shape[1]
    eng = TrainEngine(m, lr=float(g["lr"]))
    losses = []
    for it in range(int(g["steps"])):
        idx = torch.arange(B, device=dev) + (it % (N // B)) * B
        t = eng.step(idx, imgs[idx].permute(0, 2, 3, 1).view(B, P, 3), S, D)
        losses.append(float(t[0]))
    return g, np.array(losses), m


EARLY = 40   # Adam's first steps from zero moments move every weight by ~lr whatever its gradient: sign flips of near-zero gradient
             # components make the first ~30 losses a transient that differs between ANY two arithmetics (the reference's own
             # autocast run: up to 4.4 % there, 0.6 % behind it); the bands treat the two regimes separately


def _check_norms(m, g):
    """The final parameter norms are a sanity bound only (15 %): 100 Adam steps move every weight by up to 100 lr whatever the size
    of its gradient, so weights whose gradient is rounding noise random-walk -- plain fp32 torch on the factored algebra (oracle,
    CPU) ends 3.4 % (W0) and 9 % (b0) away from the reference's fp32 run with losses 0.4 % apart (profiles/r04_trajectory.md)."""
    for k, p in m.named_parameters():
        if k != "Z":
            assert abs(float(p.detach().double().norm()) - float(g["fn." + k])) <= 0.15 * float(g["fn." + k]), k


def _g15_bands(ref, ac, k):
    """(early, late) bands = k x the largest deviation the reference's own autocast-bf16 run shows in that regime (capped)"""
    dev = np.abs(ac - ref) / ref
    return min(k * float(dev[:EARLY].max()), 0.15), min(max(FLOOR, k * float(dev[EARLY:].max())), 2e-2)


def test_c2_training_trajectory_f32_g15():
    dev = torch.device("cuda:0")
    g, losses, m = _run_g15("f32", dev)
    rel = np.abs(losses - g["losses"]) / g["losses"]
    # fp32 against fp32 is NOT bit-close here either: different summation orders feed the same chaos (measured: 4.5 % at step 9,
    # 0.4 % behind step 40); the fp32 kernels get HALF the band of the bf16 ones
    early, late = _g15_bands(g["losses"], g["losses_autocast_bf16"], 1.5)
    print("G15 f32: rel loss deviation", np.array2string(rel, precision=4, max_line_width=200))
    print("G15 f32: max early", rel[:EARLY].max(), "max late", rel[EARLY:].max(), "final-latent cos", _cos(m.Z.detach().cpu().numpy(), g["Z_final"]))
    assert rel[:3].max() <= 1e-4, rel[:3]            # before the chaos: three steps bit-close
    assert rel[:EARLY].max() <= early, (rel[:EARLY], early)
    assert rel[EARLY:].max() <= late, (rel[EARLY:], late)
    assert _cos(m.Z.detach().cpu().numpy(), g["Z_final"]) >= 0.9995
    _check_norms(m, g)


def test_c2_training_trajectory_bf16_g15():
    """The training instance of the persistent bf16 kernel + k_reni_dw1_ring + the fused Adam over 100 steps."""
    dev = torch.device("cuda:0")
    g, losses, m = _run_g15("bf16", dev)
    ref, ac = g["losses"], g["losses_autocast_bf16"]
    rel = np.abs(losses - ref) / ref
    early, late = _g15_bands(ref, ac, 3.0)
    cz = _cos(m.Z.detach().cpu().numpy(), g["Z_final"])
    cz_ac = _cos(g["Z_final_autocast_bf16"], g["Z_final"])
    print("G15 bf16: rel loss deviation", np.array2string(rel, precision=4, max_line_width=200))
    print("G15 bf16: max early", rel[:EARLY].max(), "(band", early, ") max late", rel[EARLY:].max(), "(band", late, ")")
    print("G15 bf16: final-latent cos", cz, "| the reference's own autocast-bf16 run:", cz_ac)
    assert rel[:EARLY].max() <= early and rel[EARLY:].max() <= late, (rel, early, late)
    assert cz >= 0.99
    _check_norms(m, g)


# ---- G16: the same loop with the reference's DEFAULT conditioning (configs/default.py:9: FiLM) -----------------------------------------
def _run_g16(dtype, dev, fixture="g16_film_c4_trajectory.npz"):
    """RENIAutoDecoderFiLM(3, 36, SO2, 128, 5 FiLM layers, mapping 3 x 128, tanh, fixed decoder) from the fixture's seed (the class draws
    the reference's weights bit for bit: tests/test_api_cpu.py), G14's maps / Mask-3 / loss / optimiser, 200 steps through TrainEngine."""
    from reni_amd.engine import TrainEngine
    from reni_amd.film import RENIAutoDecoderFiLM
    from reni_amd.utils import get_directions, get_sineweight
    g, f = load_golden("g14_c4_trajectory.npz"), load_golden(fixture)
    N, W = g["imgs"].shape[0], int(g["W"])
    width = int(f["width"]) if "width" in f else 128
    torch.manual_seed(int(f["seed"]))
    if "kind" in f and str(f["kind"]) == "concat":   # (G20: the concat decoder at 256 features, weights from the seed)
        from reni_amd.models import RENIAutoDecoder
        m = RENIAutoDecoder(N, 36, "SO2", width, 5, 3, True, "tanh", 30.0, 30.0, True)
        if width == 128:   # (G21: G14's decoder -- the seed-42 config-2 weights g4_c2shape.npz carries)
            m.load_state_dict({"model." + k: v for k, v in _decoder_sd().items()})
        w_norm = float(sum(p.detach().double().pow(2).sum() for p in m.net.parameters()).sqrt())
        assert abs(w_norm - float(f["w_norm"])) <= 1e-9 * w_norm, "the class does not draw the reference's weights from this seed"
    else:
        m = RENIAutoDecoderFiLM(N, 36, "SO2", width, 5, width, 3, 3, "tanh", True)
    assert float(m.Z.detach().abs().sum()) == 0.0
    m.set_compute_dtype(dtype).to(dev)
    D = get_directions(W).to(dev)
    S = (get_sineweight(W) * torch.from_numpy(g["mask"])).to(dev)
    imgs = torch.from_numpy(g["imgs"]).to(dev)
    P = D.shape[1]
    eng = TrainEngine(m, lr=float(f["lr"]), loss_kind="test", alpha=float(g["alpha"]), beta=float(g["beta"]))
    idx = torch.arange(N, device=dev)
    tgt = imgs.permute(0, 2, 3, 1).view(N, P, 3)
    terms = []
    for it in range(int(f["steps"])):
        t = eng.step(idx, tgt, S, D)
        if it in f["rec_at"]:
            terms.append(t.detach().cpu().double().numpy())
    with torch.no_grad():
        img = m(m.Z.data, D).detach().float().cpu().numpy()
    return g, f, np.array(terms), m.Z.detach().cpu().numpy(), img


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_film_latent_trajectory_g16(dtype):
    """FiLM is the reference's default conditioning, and FIT_LATENT with it is the notebook's loop (examples.ipynb cell 4): 200 Adam(0.1)
    steps on the FiLM instances of the persistent bf16 kernels / the fp32 kernels against the reference's fp32 run and its own code under
    autocast(bfloat16) (tests/golden/make_g16_film_trajectory.py).  FiLM angles reach tens of revolutions, so every bf16 arithmetic is
    further from fp32 here than on the concat path (autocast: 34.8 / 37.3 dB); the kernels must not be worse than autocast less 1 dB.
    (Round 6 checked this path for the forward / backward inconsistency found on the concat path -- FINDING 2 above: the FiLM instances
    end AHEAD of the generic bf16 kernel, 36.7 / 39.1 dB against 34.8 / 38.4 dB; nothing to fix.)"""
    _check_film_trajectory(dtype, "g16_film_c4_trajectory.npz", "G16 FiLM", 48.0, 1e-3, 0.999)


def _check_film_trajectory(dtype, fixture, tag, f32_psnr, f32_rel, f32_cos, emu_psnr=None, emu_cos=None):
    """emu_psnr / emu_cos (fixtures that carry `*_emulated_*`): the bf16 bars are taken against the reference's fp32-autograd run ON THE
    NETWORK THE KERNEL EVALUATES (hidden weights bf16(W omega / 2 pi) 2 pi / omega for the persistent kernels, bf16(W) for the generic
    ones -- RENI_NO_PERSIST -- head bf16(W_out), sine outputs rounded to bf16), not against the reference under autocast: FINDING 3 below."""
    dev = torch.device("cuda:0")
    g, f, terms, Z, img = _run_g16(dtype, dev, fixture)
    ref = f["terms"]
    rel = np.abs(terms[:, 0] - ref[:, 0]) / ref[:, 0]
    rel_ac = np.abs(f["terms_autocast_bf16"][:, 0] - ref[:, 0]) / ref[:, 0]
    masked_out = (g["mask"].reshape(-1, 3) == 0).all(1)
    ref_img, ac_img = f["img_after_200"], f["img_after_200_autocast_bf16"].astype(np.float32)
    cz, cz_ac = _cos(Z, f["Z_after_200"]), _cos(f["Z_after_200_autocast_bf16"], f["Z_after_200"])
    print(f"{tag} {dtype}: max rel loss deviation {rel.max():.3e} (autocast {rel_ac.max():.3e}), final-latent cos {cz:.4f} (autocast {cz_ac:.4f})")
    for name, sel in (("masked-out", masked_out), ("kept", ~masked_out)):
        p_hip, p_ac = _psnr(img, ref_img, sel), _psnr(ac_img, ref_img, sel)
        print(f"{tag} {dtype}: final image PSNR vs the reference's fp32 image, {name} pixels: HIP {p_hip:.2f} dB, reference under autocast {p_ac:.2f} dB")
        if dtype == "f32":
            assert p_hip >= f32_psnr, (name, p_hip)          # (G16 measured 52.4 / 53.1 dB)
        elif emu_psnr is None:
            assert p_hip >= p_ac - 1.0, (name, p_hip, p_ac)  # (G16 measured 36.7 / 39.1 against 34.8 / 37.3)
        else:
            own = "generic" if os.environ.get("RENI_NO_PERSIST") else "persistent"
            e_img = f[f"img_after_200_emulated_{own}"].astype(np.float32)
            p_own, p_emu_ref = _psnr(img, e_img, sel), _psnr(e_img, ref_img, sel)
            print(f"{tag} {dtype}: {name} pixels against the fp32-autograd run on the {own} network: {p_own:.2f} dB (that run against the fp32 network's: {p_emu_ref:.2f} dB)")
            assert p_own >= emu_psnr and p_hip >= p_emu_ref - 1.5, (name, p_own, p_hip, p_emu_ref)
    if dtype == "f32":
        assert rel.max() <= f32_rel and cz >= f32_cos, (rel.max(), cz)       # (G16 measured 1.6e-4, 0.9995)
    elif emu_psnr is None:
        assert rel.max() <= max(1.5 * rel_ac.max(), 2e-2) and cz >= cz_ac - 0.05, (rel.max(), rel_ac.max(), cz, cz_ac)
    else:
        own = "generic" if os.environ.get("RENI_NO_PERSIST") else "persistent"
        t_e = f[f"terms_emulated_{own}"]
        rel_e = np.abs(terms[:, 0] - t_e[:, 0]) / t_e[:, 0]
        cz_e = _cos(Z, f[f"Z_after_200_emulated_{own}"])
        print(f"{tag} {dtype}: against the run on the {own} network: max rel loss deviation {rel_e.max():.3e}, final-latent cos {cz_e:.4f}")
        assert rel_e.max() <= 5e-3 and cz_e >= emu_cos, (rel_e.max(), cz_e)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_film_latent_trajectory_shipped_width_g18(dtype):
    """G16's loop on the reference's SHIPPED default model (configs/default.py:9-20: FiLM, 256 features, 5 FiLM layers, mapping network
    3 x 256): bf16 runs every step's chain on k_reni_wide256<2, FILM> in front of k_dw_frag<256, true> (d(freq), d(phase) -> the mapping
    network's backward -> dZ), fp32 on the generic kernels.  Same bars as G16: fp32 close to the reference's fp32 run, bf16 no worse than
    the reference's own code under autocast(bfloat16) less 1 dB (tests/golden/make_g16_film_trajectory.py 256)."""
    _check_film_trajectory(dtype, "g18_film256_c4_trajectory.npz", "G18 FiLM-256", 55.0, 5e-4, 0.999)   # (measured 60.0 / 61.1 dB, 8.9e-5, 0.9995; bf16 38.7 / 41.3 dB against autocast's 35.3 / 38.3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_concat_latent_trajectory_shipped_width_g20(dtype):
    """G14's loop on the concat decoder at 256 features (tests/golden/make_g20_concat256_trajectory.py): bf16 runs k_reni_wide256<1>
    (frozen chain, forward / backward consistent weight images since round 6).  The loop that exposed the H = 128 kernels' inconsistent
    images (FINDING 2) had no counterpart at this width.

    FINDING 3 (round 6, profiles/r06_trajectory.md section 8): here the shipped kernel ends 46.5 dB from the reference's fp32 maps, the
    generic bf16 kernel 50.6 dB and the reference under autocast 50.3 dB -- reproducibly over the perturbation ensemble -- and it is NOT a
    kernel error: each kernel's dZ is the exact gradient of the network its forward pass evaluates to 1.4e-4 (fp64 autograd emulation,
    the same for both kernels, along the whole path), and the reference's own fp32-autograd loop ON THOSE NETWORKS lands at 46.2 dB
    (hidden weights bf16(W omega / 2 pi): the persistent kernels' images) and 50.4 dB (bf16(W)).  Which 2^-9 perturbation of the weights
    the loop runs on decides where it ends: over arbitrary scales in front of the rounding 41.9 .. 52.9 dB.  "No worse than the
    reference under autocast" is therefore a lottery, and the bf16 bars of this fixture are: (a) the kernel follows the fp32-autograd
    run on ITS network (measured 54.0 / 55.9 dB on the unperturbed targets -- the ensemble's worst draw; 63 .. 75 dB on the six perturbed
    ones; latent cos 0.989), and (b) it ends no further from the fp32 network's maps than that run does, less 1.5 dB."""
    _check_film_trajectory(dtype, "g20_concat256_c4_trajectory.npz", "G20 concat-256", 80.0, 1e-4, 0.9999, emu_psnr=50.0, emu_cos=0.97)   # (fp32 kernels measured: 117 dB, 8.7e-7, 1.0000)


def test_concat_latent_trajectory_on_its_own_network_g21():
    """The same bars for G14's decoder (5 x 128: the frozen instance of k_reni_train_bf16, BASELINE config 4's kernel): G21 = G14's runs
    again (bit-identical to g14_c4_trajectory.npz) plus the fp32-autograd runs on the two rounded networks
    (tests/golden/make_g20_concat256_trajectory.py 128).  Measured: 56.4 / 58.4 dB, latent cos 0.944 against the run on the persistent
    kernels' network (which itself ends 46.4 / 46.0 dB from the fp32 network's maps; the kernel 48.4 / 47.1)."""
    _check_film_trajectory("bf16", "g21_concat128_c4_trajectory.npz", "G21 concat-128", 0, 0, 0, emu_psnr=50.0, emu_cos=0.9)


# ---- G17: decoder training with the DEFAULT conditioning at the rate the reference trains it at (configs/default.py:9, :25) ------------
@pytest.mark.parametrize("fixture", ["g17_film_c2_trajectory.npz", "g19_film256_c2_trajectory.npz"], ids=["g17", "g19_shipped_width"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_film_training_trajectory_g17(dtype, fixture):
    """FIT_DECODER with FiLM (RENI_module.py:80-146 on RENIAutoDecoderFiLM): 100 Adam(1e-5) steps over net + final_layer + mapping_network
    + latents through TrainEngine (reni_film_model_forward_loss_backward + the fused Adam over the flat buffer) against the reference's
    fp32 run and its own code under autocast(bfloat16) (tests/golden/make_g17_film_training.py).  Weights and latents come from the seed.
    At this rate the run is stable (the loss falls 0.63 -> 0.26; autocast stays within 0.13 % of fp32 everywhere), so the bands are
    tight: fp32 kernels 1e-5 (measured 2.3e-7 over all 100 steps, latents within 2.4e-7), bf16 kernels NO WORSE than the reference's
    own autocast run (measured 2.4e-4 against 1.3e-3; latents 3.6e-5 against 3.3e-4)."""
    from reni_amd.engine import TrainEngine
    from reni_amd.film import RENIAutoDecoderFiLM
    from reni_amd.utils import get_directions, get_sineweight
    dev = torch.device("cuda:0")
    g, f = load_golden("g15_c2_trajectory.npz"), load_golden(fixture)
    N, B, W = g["imgs"].shape[0], int(g["B"]), int(g["W"])
    width = int(f["width"]) if "width" in f else 128   # (G19: the shipped default model, 256 features -- k_reni_wide256<2, FILM> + k_dw_frag<256, true>)
    torch.manual_seed(int(f["seed"]))
    m = RENIAutoDecoderFiLM(N, 36, "SO2", width, 5, width, 3, 3, "tanh", False)
    assert abs(float(m.Z.detach().double().abs().sum()) - float(f["Z0_abs_sum"])) <= 1e-3   # the same draw as the reference's
    m.set_compute_dtype(dtype).to(dev)
    D, S = get_directions(W).to(dev), get_sineweight(W).to(dev)
    imgs = torch.from_numpy(g["imgs"]).to(dev)
    P = D.shape[1]
    eng = TrainEngine(m, lr=float(f["lr"]))
    losses = []
    for it in range(int(f["steps"])):
        idx = torch.arange(B, device=dev) + (it % (N // B)) * B
        t = eng.step(idx, imgs[idx].permute(0, 2, 3, 1).view(B, P, 3), S, D)
        losses.append(float(t[0]))
    losses = np.array(losses)
    ref, ac = f["losses"], f["losses_autocast_bf16"]
    rel, rel_ac = np.abs(losses - ref) / ref, np.abs(ac - ref) / ref
    Zf = m.Z.detach().cpu().numpy()
    dz, dz_ac = np.abs(Zf - f["Z_final"]).max(), np.abs(f["Z_final_autocast_bf16"] - f["Z_final"]).max()
    print(f"{fixture[:3].upper()} FiLM training (width {width}) {dtype}: max rel loss deviation {rel.max():.3e} (first three {rel[:3].max():.3e}; the reference under autocast "
          f"{rel_ac.max():.3e}); final latents max |dZ| {dz:.3e} (autocast {dz_ac:.3e})")
    if dtype == "f32":
        assert rel[:3].max() <= 2e-6 and rel.max() <= 1e-5, rel
        assert dz <= 5e-6, dz                      # 100 steps of 1e-5 move a latent by at most 1e-3
    else:
        assert rel.max() <= rel_ac.max(), (rel.max(), rel_ac.max())
        assert dz <= dz_ac, (dz, dz_ac)
    for k, p in m.named_parameters():
        if k != "Z":
            ref_n = float(f["fn." + k])
            assert abs(float(p.detach().double().norm()) - ref_n) <= 2e-3 * ref_n, k
