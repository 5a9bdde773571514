"""GPU tests of the harness-level semantics (training step, latent optimisation, equivariance) and
of BASELINE.json's full-size configurations through size-independent properties."""
import types

import numpy as np
import pytest
import torch

from oracle import reni_oracle as O
from tests.util import flat_params, load_golden, make_plan, random_problem, sd_from, unflatten

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _task(**k):
    base = dict(LR_START=1e-3, LR_END=1e-3, OPTIMIZER="adam", OPTIMIZER_BETA_1=0.0, OPTIMIZER_BETA_2=0.9,
                SCHEDULER_TYPE="none", SCHEDULER_STEP_SIZE=1, SCHEDULER_GAMMA=1.0, BATCH_SIZE=2, EPOCHS=10,
                MULTI_RES_TRAINING=False, INITAL_RESOLUTION=[16, 32], FINAL_RESOLUTION=[16, 32], CURRICULUM=[1],
                KLD_WEIGHTING=1e-4, COSINE_SIMILARITY_WEIGHT=1e-1, PRIOR_LOSS_WEIGHT=1e-7, APPLY_MASK=False,
                MASK_PATH="")
    base.update(k)
    return types.SimpleNamespace(**base)


def _config(model_type="AutoDecoder", **task_over):
    reni = types.SimpleNamespace(
        CONDITIONING="Cond-by-Concat", MODEL_TYPE=model_type, EQUIVARIANCE="SO2", LATENT_DIMENSION=9, HIDDEN_LAYERS=3,
        HIDDEN_FEATURES=64, OUT_FEATURES=3, LAST_LAYER_LINEAR=True, OUTPUT_ACTIVATION="tanh", FIRST_OMEGA_0=30.0,
        HIDDEN_OMEGA_0=30.0, MAPPING_LAYERS=3, MAPPING_FEATURES=64, FIT_DECODER=_task(**task_over),
        FIT_LATENT=_task(**task_over))
    return types.SimpleNamespace(RENI=reni, TRAINER=types.SimpleNamespace(LOGGER=types.SimpleNamespace(NUMBER_OF_IMAGES=2)),
                                 DATASET=types.SimpleNamespace(NAME="SYNTHETIC"))


class _ListDataset(torch.utils.data.Dataset):
    def __init__(self, imgs):
        self.imgs = imgs
        self.unnormalise = None

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, i):
        return self.imgs[i], i


def test_training_step_reenactment_g6(dev):
    """5 FIT_DECODER steps (AutoDecoder, Adam lr 1e-3 over ALL parameters incl. the whole latent table)
    through RENI.training_step + the fit loop, against the reference re-enactment (golden G6)."""
    from reni_amd.lightning_module import RENI
    from reni_amd import trainer
    g = load_golden("g6_train_steps.npz")
    ds = _ListDataset(torch.from_numpy(g["imgs"]))
    mod = RENI(_config(), "FIT_DECODER", dataset=ds)
    mod.setup()
    mod.model.load_state_dict({"model." + k: v for k, v in sd_from(g, "sd0.").items()})
    mod.model_from_checkpoint = True  # keep the loaded model across fit()'s setup()
    losses = []
    for idx in g["batches"]:
        h = trainer.fit(mod, max_epochs=1, device=dev, batches=[list(map(int, idx))])
        losses.append(h[0]["loss"])
        break
    # fit() builds a fresh optimiser each call, so drive the remaining steps by hand with one optimiser
    mod2 = RENI(_config(), "FIT_DECODER", dataset=ds)
    mod2.setup()
    mod2.model.load_state_dict({"model." + k: v for k, v in sd_from(g, "sd0.").items()})
    mod2.to(dev)
    opt = mod2.configure_optimizers()["optimizer"]
    losses = []
    for bi, idx in enumerate(g["batches"]):
        idx = torch.tensor(idx)
        out = mod2.training_step((ds.imgs[idx].to(dev), idx.to(dev)), bi)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    assert float((mod2.model.Z.detach().cpu() - torch.from_numpy(g["Z_final"])).abs().max()) <= 2e-4
    assert O.rel_l2(mod2.model.net[0].linear.weight.detach().cpu().numpy(), g["W0_final"]) <= 1e-3
    assert O.rel_l2(mod2.model.net[4].weight.detach().cpu().numpy(), g["Wout_final"]) <= 1e-3


def test_latent_optimisation_reenactment_g7(dev, tmp_path):
    """BASELINE config 4 in miniature (notebook cell 4): frozen VAD decoder loaded from a checkpoint,
    latents from zero, Mask-3, RENITestLoss(1e-7, 1e-1), Adam lr 1e-1, 10 steps."""
    from PIL import Image
    from reni_amd.lightning_module import RENI
    g = load_golden("g7_latent_opt.npz")
    Image.fromarray(np.stack([g["mask_src"]] * 3, -1)).save(tmp_path / "mask3.png")
    cfg = _config("VariationalAutoDecoder", LR_START=1e-1, LR_END=1e-1, BATCH_SIZE=3, INITAL_RESOLUTION=[32, 64],
                  FINAL_RESOLUTION=[32, 64], APPLY_MASK=True, MASK_PATH=str(tmp_path / "mask3.png"))
    ds = _ListDataset(torch.from_numpy(g["imgs"]))
    mod = RENI(cfg, "FIT_LATENT", dataset=ds)
    mod.setup()
    ck = {k[len("ckpt."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("ckpt.")}
    mod.load_state_dict(ck)
    assert mod.model.fixed_decoder and float(mod.model.mu.detach().abs().sum()) == 0.0
    mod.to(dev)
    opt = mod.configure_optimizers()["optimizer"]
    assert [id(p) for gr in opt.param_groups for p in gr["params"]] == [id(mod.model.mu)]
    terms = []
    idx = torch.arange(3)
    for s in range(10):
        out = mod.training_step((ds.imgs.to(dev), idx.to(dev)), s)
        opt.zero_grad()
        out["loss"].backward()
        if s == 0:
            assert O.rel_l2(mod.model.mu.grad.cpu().numpy(), g["mu_grad0"]) <= 1e-4
        opt.step()
        terms.append([float(out[k]) for k in ("loss", "mse_loss", "prior_loss", "cosine_loss")])
    np.testing.assert_allclose(np.array(terms), g["terms"], rtol=2e-3, atol=1e-7)
    assert float((mod.model.mu.detach().cpu() - torch.from_numpy(g["mu_final"])).abs().max()) <= 2e-2


def test_vad_training_step_g8(dev):
    from reni_amd.models import RENIVariationalAutoDecoder
    from reni_amd.loss_functions import RENIVADTrainLoss
    g = load_golden("g8_vad.npz")
    m = RENIVariationalAutoDecoder(3, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, False)
    m.load_state_dict({"model." + k: v for k, v in sd_from(g).items()})
    m.to(dev)
    W = int(g["W"])
    D = O.get_directions(W).to(dev); S = O.get_sineweight(W).to(dev)
    idx = torch.from_numpy(g["idx"]).to(dev)
    mu = m.mu[idx]; lv = m.log_var[idx]
    Z = mu + torch.from_numpy(g["eps"]).to(dev) * torch.exp(0.5 * lv)  # the recorded epsilon
    loss, mse, kld = RENIVADTrainLoss(1e-4, 27).fused(m, Z, D, torch.from_numpy(g["target"]).to(dev), S, mu, lv)
    loss.backward()
    np.testing.assert_allclose([float(loss), float(mse), float(kld)], g["terms"], rtol=3e-6)
    assert O.rel_l2(m.mu.grad.cpu().numpy(), g["g_mu"]) <= 1e-5
    assert O.rel_l2(m.log_var.grad.cpu().numpy(), g["g_lv"]) <= 1e-5
    assert O.rel_l2(m.net[0].linear.weight.grad.cpu().numpy(), g["g_W0"]) <= 1e-5


def test_dispatch_forms_g9(dev):
    from reni_amd.models import RENIAutoDecoder
    g = load_golden("g9_api.npz")
    m = RENIAutoDecoder(3, 9, "SO2", 64, 3, 3, True, "tanh", 30, 30, False)
    m.load_state_dict({"model." + k: v for k, v in sd_from(g).items()})
    m.to(dev)
    D = O.get_directions(16).to(dev)
    with torch.no_grad():
        for key, x, d in (("disp_int", 1, D), ("disp_list", [0, 2], D.repeat(2, 1, 1)),
                          ("disp_idx", torch.tensor([2, 1], device=dev), D.repeat(2, 1, 1)), ("disp_lat", m.Z[[1]], D)):
            out = m(x, d)
            assert out.shape == g[key].shape
            assert float((out.cpu() - torch.from_numpy(g[key])).abs().max()) <= 1e-5
    with pytest.raises(AssertionError):
        m([0, 1, 2], D.repeat(2, 1, 1))


def test_equivariance_golden_and_c5_full_size(dev):
    """BASELINE config 5: ND=49 fp32 inference; f(Z R^T, D R^T) == f(Z, D) for a y-rotation (SO2 model)
    and a random SO(3) rotation (SO3 model): first against the reference's outputs (P=2048), then at the
    full 512x1024 grid (B=4) as a size-independent property."""
    g = load_golden("g10_equivariance.npz")
    for eq, Rk in (("SO2", "Ry"), ("SO3", "R3")):
        spec = O.DecoderSpec(49, eq, 128, 5, 3, True, "tanh")
        params = sd_from(g, f"sd_{eq}.")
        plan = make_plan(spec, "f32")
        fp = flat_params(spec, params).to(dev)
        Z = torch.from_numpy(g[f"Z_{eq}"]); R = torch.from_numpy(g[Rk])
        D = O.get_directions(64)
        a = plan.forward(Z.to(dev), D.to(dev), fp)
        b = plan.forward((Z @ R.T).to(dev), (D @ R.T).to(dev), fp)
        assert float((a.cpu()[0, :128] - torch.from_numpy(g[f"out_{eq}_head"])).abs().max()) <= 1e-5
        assert abs(float(a.double().sum()) - float(g[f"out_{eq}_sum64"])) <= 1e-2
        assert float((a - b).abs().max()) <= 2e-5
        # full size: 4 images x 524 288 directions
        Zb = torch.randn(4, 49, 3, generator=torch.Generator().manual_seed(3))
        Dl = O.get_directions(1024)
        a = plan.forward(Zb.to(dev), Dl.to(dev), fp)
        b = plan.forward((Zb @ R.T).to(dev), (Dl @ R.T).to(dev), fp)
        assert a.shape == (4, 524288, 3) and bool(torch.isfinite(a).all())
        assert float((a - b).abs().max()) <= 5e-5


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_c2_full_size_additivity(dev, dtype):
    """BASELINE config 2 at full size (128x256, ND=36, 5x128): gradients are additive over images, the
    per-image latent gradient does not depend on the rest of the batch, and a sub-sampled slice of the
    output agrees with the oracle."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    params, Z, D, W, _ = random_problem(spec, 3, 0, seed=12, grid_w=256)
    T = O.synthetic_images([0, 1, 2], 128, 256).permute(0, 2, 3, 1).reshape(3, -1, 3)
    plan = make_plan(spec, dtype)
    fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), W.to(dev)
    # (the output image comes from a call of its own: a call that wants it takes the generic training instance, the others the SPEC /
    # L0X instance + k_reni_l0_ring, whose layer-0 cosine is fp32 -- batch independence is a property of ONE path, compared like with like)
    _, _, _, out = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, want_out=True)
    out = out.clone()
    lt, dZ, dp, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
    lt, dZ, dp = lt.clone(), dZ.clone(), dp.clone()
    acc = torch.zeros_like(dp); ls = 0.0
    for i in range(3):
        l1, z1, p1, _ = plan.forward_loss_backward(Zd[i:i + 1], Dd, fp, Td[i:i + 1], Wd)
        acc += p1; ls += float(l1[0])
        assert O.rel_l2(z1.cpu().numpy(), dZ[i:i + 1].cpu().numpy()) <= 1e-6
    assert abs(ls - float(lt[0])) <= 1e-5 * ls
    assert O.rel_l2(acc.cpu().numpy(), dp.cpu().numpy()) <= 1e-5
    sl = slice(5000, 5256)
    ref = O.reni_forward(spec, params, Z, D[:, sl].expand(3, -1, 3))
    tol = 1e-5 if dtype == "f32" else 5e-3
    assert float((out[:, sl].cpu() - ref).abs().max()) <= tol


def _oracle_f64(spec, params, Z, D, T, W):
    """float64 factored oracle (pinned to the reference's autograd by tests/test_oracle_golden.py) on numpy copies."""
    return O.factored_fwd_bwd(spec, {k: v.numpy() for k, v in params.items()}, Z.numpy(), D.numpy(), T.numpy(), W.numpy())


def _assert_training_parity(spec, params, lt, dZ, dp, ref, tol_grad, tol_loss):
    assert abs(float(lt[0]) - ref["loss_terms"][0]) <= tol_loss * abs(ref["loss_terms"][0])
    assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"]) <= tol_grad
    gp = unflatten(spec, dp.cpu())
    for k in gp:
        assert O.rel_l2(gp[k].numpy(), ref["grads"][k]) <= tol_grad, (k, O.rel_l2(gp[k].numpy(), ref["grads"][k]))


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_c2_training_instance_against_the_oracle_with_many_tiles_per_workgroup(dev, dtype):
    """The regime bench.py runs the concat training instance in -- many tiles per workgroup, image runs cut inside
    workgroup ranges, the dW accumulators carried across tiles in AGPRs -- compared DIRECTLY with the float64 oracle
    (RENI_module.py:105-118: model call + RENITrainLoss; every other oracle comparison of k_reni_train_bf16<128,true>
    has fewer tiles than workgroups): config-2 architecture, B = 8 images x 32 768 directions = 2 048 tiles, 8 per
    workgroup, every second workgroup range crossing no image boundary and every 32nd crossing one.  Loss, dZ and
    every dW / db.  The fp32 generic kernel on the same problem is the cross-check."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    B = 8
    params, Z, D, W, _ = random_problem(spec, B, 0, seed=31, grid_w=256)
    T = O.synthetic_images(list(range(B)), 128, 256).permute(0, 2, 3, 1).reshape(B, -1, 3)
    ref = _oracle_f64(spec, params, Z, D, T, W.expand(1, -1, 3))
    plan = make_plan(spec, dtype)
    fp = flat_params(spec, params).to(dev)
    lt, dZ, dp, _ = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev))
    tol = (3e-2, 2e-3) if dtype == "bf16" else (1e-5, 2e-6)
    _assert_training_parity(spec, params, lt, dZ, dp, ref, *tol)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_c2_training_instance_against_the_oracle_with_700_one_tile_images(dev, dtype):
    """As above with 700 images of ONE tile each (P = 96 directions): every workgroup's range holds two or three whole
    images, so every tile ends an image run (per-image dA and loss leave the chip each tile) while the weight-gradient
    accumulators carry on across them."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    B, P = 700, 96
    params, Z, D, W, T = random_problem(spec, B, P, seed=32)
    ref = _oracle_f64(spec, params, Z, D, T, W)
    plan = make_plan(spec, dtype)
    fp = flat_params(spec, params).to(dev)
    lt, dZ, dp, _ = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev))
    tol = (3e-2, 2e-3) if dtype == "bf16" else (1e-5, 2e-6)
    _assert_training_parity(spec, params, lt, dZ, dp, ref, *tol)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_h256_training_in_image_chunks(dev, dtype, monkeypatch):
    """The shipped experiment.yaml width (5 x 256): the fragment-stream training path keeps 11 x 64 KB per tile, so a call whose
    stream would exceed the plan's cap runs in chunks of whole images (run_backward_chunked): per-image results in place, loss
    terms and decoder gradient accumulated.  Forced here at a small size (cap 5 MB -> chunks of 2, 2, 1 images, through the
    latent-table entry point): equal to the one-pass result up to the order of the sums, and to the oracle; and the workspace
    the library asks for at the BASELINE batch stays under the cap (16 GB) plus the fixed parts."""
    spec = O.DecoderSpec(9, "SO2", 256, 2, 3, True, "tanh")
    params, Ztab, D, W, T = random_problem(spec, 7, 300, seed=41)
    idx = torch.tensor([6, 1, 3, 0, 4], device=dev)
    Tb = T[:5].to(dev)
    fp = flat_params(spec, params).to(dev)
    res = {}
    for cap in ("100000", "5"):
        monkeypatch.setenv("RENI_FRAG_WS_CAP_MB", cap)
        plan = make_plan(spec, dtype)
        res[cap] = plan.forward_loss_backward(Ztab.to(dev), D.to(dev), fp, Tb, W.to(dev), loss_kind="test", alpha=1e-3, beta=1e-1,
                                              idx=idx, want_out=True)
    one, chunked = res["100000"], res["5"]
    assert torch.equal(one[3], chunked[3]) and torch.equal(one[1], chunked[1])          # per-image results: bit-equal
    assert float((one[0] - chunked[0]).abs().max()) <= 1e-5 * float(one[0][0].abs())
    assert O.rel_l2(chunked[2].cpu().numpy(), one[2].cpu().numpy()) <= 1e-5
    ref = O.fwd_loss_bwd(spec, params, Ztab[idx.cpu()], D.expand(5, -1, 3), T[:5], W.expand(5, -1, 3), "test", 1e-3, 1e-1)
    tol = 3e-2 if dtype == "bf16" else 1e-5
    assert O.rel_l2(chunked[1].cpu().numpy(), ref["dZ"].numpy()) <= tol
    gp = unflatten(spec, chunked[2].cpu())
    for k in gp:
        if gp[k].numel() > 3:
            assert O.rel_l2(gp[k].numpy(), ref["grads"][k].numpy()) <= tol, k
    monkeypatch.delenv("RENI_FRAG_WS_CAP_MB")
    big = make_plan(O.DecoderSpec(49, "SO2", 256, 5, 3, True, "tanh"), dtype)
    from reni_amd import _lib
    # (round 5: the cap is 16 GB of the device's 288 -- bf16 runs the BASELINE batch in one pass, fp32 in two)
    assert big.lib.reni_workspace_bytes(big._h, 64, 32768, _lib.NEED_DW | _lib.NEED_DZ) < 18 * 2 ** 30
    assert big.path_info(64, 32768)["images_per_chunk"] == (64 if dtype == "bf16" else 32)


def test_multires_curriculum_and_exponential_lr(dev):
    """MultiResTrainingCallback semantics (callbacks.py:11-29) and the per-epoch ExponentialLR
    (RENI_module.py:213-214, 243-251) through the fit loop: the grids, the dataset and the kernels' problem size
    double at the curriculum epochs; the learning rate follows lr_start * gamma^epoch with
    gamma = exp(ln(lr_end / lr_start) / epochs); the FiLM model trains through the same loop."""
    from reni_amd import trainer
    from reni_amd.lightning_module import RENI
    for conditioning in ("Cond-by-Concat", "FiLM"):
        cfg = _config(LR_START=1e-2, LR_END=1e-4, SCHEDULER_TYPE="exponential", EPOCHS=4, BATCH_SIZE=2,
                      MULTI_RES_TRAINING=True, INITAL_RESOLUTION=[8, 16], FINAL_RESOLUTION=[32, 64], CURRICULUM=[1, 3])
        cfg.RENI.CONDITIONING = conditioning
        cfg.DATASET = types.SimpleNamespace(NAME="SYNTHETIC", SYNTHETIC=types.SimpleNamespace(N_TRAIN=4, N_TEST=2))
        torch.manual_seed(0)
        mod = RENI(cfg, "FIT_DECODER")
        seen = []
        orig = mod.training_step

        def spy(batch, bi, orig=orig, mod=mod, seen=seen):
            seen.append((mod.current_epoch, tuple(batch[0].shape[-2:]), mod.directions.shape[1]))
            return orig(batch, bi)

        mod.training_step = spy
        hist = trainer.fit(mod, max_epochs=4, device=dev)
        res = {e: (hw, p) for e, hw, p in seen}
        assert res[0] == ((8, 16), 128) and res[1] == ((16, 32), 512) and res[2] == ((16, 32), 512) and res[3] == ((32, 64), 2048)
        assert mod.cur_res == [32, 64] and len(hist) == 4 and all(np.isfinite(h["loss"]) for h in hist)
        opt = mod.configure_optimizers  # a fresh optimiser + scheduler pair to read the schedule
        pair = opt()
        sched, o = pair["lr_scheduler"]["scheduler"], pair["optimizer"]
        gamma = np.exp(np.log(1e-4 / 1e-2) / 4)
        lrs = []
        for _ in range(4):
            lrs.append(o.param_groups[0]["lr"])
            sched.step()
        np.testing.assert_allclose(lrs, [1e-2 * gamma ** e for e in range(4)], rtol=1e-6)


def test_adam_rows_step_is_dense_adam_over_the_table(dev):
    """reni_adam_rows_step == torch.optim.Adam over the WHOLE table with a gradient that is zero outside the batch's rows
    (rows absent from a batch still move by momentum; repeated indices accumulate like index_add_)."""
    from reni_amd import ops
    g = torch.Generator().manual_seed(3)
    N, row = 37, 27
    table = torch.randn(N, 9, 3, generator=g)
    ref = table.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2)
    t = table.clone().to(dev)
    m, v = torch.zeros_like(t), torch.zeros_like(t)
    for step, idx in enumerate(([3, 5, 20], [5, 5, 36, 0], [1]), start=1):
        idx = torch.tensor(idx)
        gr = torch.randn(len(idx), 9, 3, generator=g)
        dense = torch.zeros(N, 9, 3).index_add_(0, idx, gr)
        opt.zero_grad()
        ref.grad = dense * 0.5
        opt.step()
        ops.adam_rows_step(t, gr.to(dev), idx.to(dev), m, v, step, 1e-2, grad_scale=0.5)
    assert float((t.cpu() - ref.detach()).abs().max()) <= 2e-6


def test_c4_full_size_latent_step_bf16(dev):
    """BASELINE config 4 at its real size (examples.ipynb cell 4; RENI_module.py:92-94,126-128): 21 held-out maps x 32 768
    directions, ND = 36, 5 x 128, bf16, frozen decoder, masked weight, RENITestLoss(1e-7, 1e-4) -- the statistics
    instance, the cosine coefficients and the frozen instance of the persistent kernel with a mask, which the G7
    miniature (3 x 2 048, fp32, H = 64) never reaches.  One image is checked against the oracle at full P; the batch
    through additivity over images and run-to-run bit-equality."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    B = 21
    params, Z, D, W, T = random_problem(spec, B, 0, seed=44, grid_w=256)
    P = D.shape[1]
    assert P == 32768
    # the reference's REAL Mask-3 (data/Masks/Mask-3.png; its 256 x 512 source travels as data in g7_latent_opt.npz), resized as
    # utils.get_mask does: rows 20-93 x columns 81-164 kept -- 148 of an image's 256 tiles carry weight, the block cuts through both
    # tiles of every row it touches, pixel 0 is masked (the geometry the tile lists of RENI_WEIGHT_SPARSE actually see)
    from reni_amd.utils import mask_from_array
    keep = mask_from_array(256, load_golden("g7_latent_opt.npz")["mask_src"]).reshape(1, P, 3)
    assert abs(float((keep.reshape(-1, 3) != 0).any(1).float().mean()) - 0.188) < 2e-3
    Wm = W * keep
    plan = make_plan(spec, "bf16")
    fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), Wm.to(dev)
    a, b_ = 1e-7, 1e-4

    def run(sel, sparse=True):   # (RENI_WEIGHT_SPARSE: what RENI.training_step passes with a mask, lightning_module.py)
        lt, dZ, dp, _ = plan.forward_loss_backward(Zd[sel], Dd, fp, Td[sel], Wd, loss_kind="test", alpha=a, beta=b_,
                                                   need_dw=False, sparse_weight=sparse)
        assert dp is None
        return lt.cpu().double(), dZ.cpu()

    lt_all, dZ_all = run(slice(0, B))
    lt_again, dZ_again = run(slice(0, B))
    assert torch.equal(lt_all, lt_again) and torch.equal(dZ_all, dZ_again)          # deterministic reductions
    lt_dense, dZ_dense = run(slice(0, B), sparse=False)                             # every tile + the statistics pass: the same bits
    assert torch.equal(lt_all, lt_dense) and torch.equal(dZ_all, dZ_dense)
    lt_px, dZ_px = run(slice(0, B), sparse="pixels")                                # RENI_WEIGHT_COMPACT: equal to rounding
    assert O.rel_l2(dZ_px.numpy(), dZ_all.numpy()) <= 2e-6 and abs(float(lt_px[0] - lt_all[0])) <= 2e-6 * abs(float(lt_all[0]))
    # one image against the oracle at full P (reference-shaped: 32 768 x 1 370 encoding, autograd)
    k = 7
    ref = O.fwd_loss_bwd(spec, params, Z[k:k + 1], D, T[k:k + 1], Wm, "test", a, b_, need_dw=False)
    lt_k, dZ_k = run(slice(k, k + 1))
    for i in range(4):
        assert abs(float(lt_k[i]) - ref["loss_terms"][i]) <= 3e-3 * abs(ref["loss_terms"][0]), (i, lt_k, ref["loss_terms"])
    assert abs(float(lt_k[3]) - ref["loss_terms"][3]) <= 3e-3 * abs(ref["loss_terms"][3])  # the cosine term on its own scale
    assert O.rel_l2(dZ_k.numpy(), ref["dZ"].numpy()) <= 3e-2
    assert torch.equal(dZ_k[0], dZ_all[k])       # an image's latent gradient does not depend on its batch
    # additivity: every loss term of the batch is the sum over its images (in groups, to keep the test short)
    groups = [slice(0, 5), slice(5, 6), slice(6, 13), slice(13, 21)]
    parts = [run(g) for g in groups]
    lt_sum = sum(p[0] for p in parts)
    for i in range(4):
        assert abs(float(lt_sum[i] - lt_all[i])) <= 2e-6 * abs(float(lt_all[i])) + 1e-12, (i, lt_sum, lt_all)
    assert torch.equal(torch.cat([p[1] for p in parts]), dZ_all)
    assert float(dZ_all.abs().max()) > 0 and torch.isfinite(dZ_all).all()


def test_sparse_weight_leaves_out_only_exact_zeros(dev):
    """RENI_WEIGHT_SPARSE (include/reni_hip.h; RENI.training_step sets it whenever a mask is configured, RENI_module.py:92-94): the
    frozen-decoder persistent kernels visit only the tiles whose weights are not all zero, and run the cosine term's statistics pass
    only for images whose pixel-0 weight is not zero (loss_functions.py:25-32 multiplies the term by that weight).  Every record left
    out is an exact zero in the dense computation: loss terms and latent gradients must be EQUAL, image by image -- here with the
    notebook's block mask (pixel 0 masked), no mask, an all-zero weight, a single live pixel, and a block mask that keeps pixel 0."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    B = 5
    params, Z, D, W, T = random_problem(spec, B, 0, seed=91, grid_w=128)      # 64 x 128 = 8 192 directions = 64 tiles per image
    P = D.shape[1]
    m = torch.zeros(B, 64, 128, 1)
    m[0, 10:46, 40:83] = 1.0            # Mask-3's block at this resolution (rows 20-93 x columns 81-164 of 128 x 256)
    m[1] = 1.0                          # no mask
    m[3, 33, 77] = 1.0                  # one pixel
    m[4, 10:46, 40:83] = 1.0
    m[4, 0, 0] = 1.0                    # the block, and pixel 0: the cosine term is live -> every tile of the image matters
    Wm = (W.view(1, 64, 128, 3) * m).reshape(B, P, 3)
    plan = make_plan(spec, "bf16")
    fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), Wm.to(dev)
    a, b_ = 1e-7, 1e-4

    def run(sparse, sel=slice(0, B)):
        lt, dZ, _, _ = plan.forward_loss_backward(Zd[sel], Dd, fp, Td[sel], Wd[sel], loss_kind="test", alpha=a, beta=b_, need_dw=False,
                                                  sparse_weight=sparse)
        return lt.cpu(), dZ.cpu()

    def where(x, y):  # (which images differ, and by how much: for the assertion messages)
        return [(k, float((x[k] - y[k]).abs().max())) for k in range(x.shape[0]) if not torch.equal(x[k], y[k])]

    lt_d, dZ_d = run(False)
    lt_d2, dZ_d2 = run(False)
    assert torch.equal(lt_d2, lt_d) and torch.equal(dZ_d2, dZ_d), ("dense run to run", lt_d, lt_d2, where(dZ_d2, dZ_d))
    lt_s, dZ_s = run(True)
    assert torch.equal(lt_s, lt_d), ("tiles: loss terms", lt_s, lt_d)
    assert torch.equal(dZ_s, dZ_d), ("tiles: dZ", where(dZ_s, dZ_d))
    assert torch.isfinite(dZ_s).all() and float(dZ_s[0].abs().max()) > 0
    # the all-zero weight: nothing but the prior's gradient, 2 alpha Z
    assert torch.allclose(dZ_s[2], 2 * a * Z[2], rtol=1e-6, atol=0), "all-zero weight"
    for k in range(B):  # image by image (other lists, other workgroups): the same numbers
        lt_k, dZ_k = run(True, slice(k, k + 1))
        assert torch.equal(dZ_k[0], dZ_s[k]), ("tiles: image alone", k, float((dZ_k[0] - dZ_s[k]).abs().max()))
    # and against the oracle (the reference's dense arithmetic) for the block-masked image
    ref = O.fwd_loss_bwd(spec, params, Z[0:1], D, T[0:1], Wm[0:1], "test", a, b_, need_dw=False)
    lt_0, dZ_0 = run(True, slice(0, 1))
    for i in range(4):
        assert abs(float(lt_0[i]) - ref["loss_terms"][i]) <= 3e-3 * abs(ref["loss_terms"][0]), (i, lt_0, ref["loss_terms"])
    assert O.rel_l2(dZ_0.numpy(), ref["dZ"].numpy()) <= 3e-2
    # RENI_WEIGHT_COMPACT: the pixels with weight packed into each image's first tiles -- the same terms in another order: equal to
    # fp32 rounding (not bit for bit), bit-identical run to run, and image by image whatever the batch
    lt_p, dZ_p = run("pixels")
    lt_p2, dZ_p2 = run("pixels")
    assert torch.equal(lt_p, lt_p2) and torch.equal(dZ_p, dZ_p2), ("pixels run to run", lt_p, lt_p2, where(dZ_p, dZ_p2))
    assert torch.allclose(lt_p, lt_d, rtol=2e-6, atol=0), ("pixels: loss terms", lt_p, lt_d)
    for k in range(B):
        ref_k = dZ_d[k]
        assert float((dZ_p[k] - ref_k).norm()) <= 2e-6 * float(ref_k.norm()) + 1e-12, (k, float((dZ_p[k] - ref_k).norm()), float(ref_k.norm()))
        lt_k, dZ_k = run("pixels", slice(k, k + 1))
        assert torch.equal(dZ_k[0], dZ_p[k]), ("pixels: image alone", k, float((dZ_k[0] - dZ_p[k]).abs().max()))
    assert torch.equal(dZ_p[1], dZ_d[1]) and torch.equal(dZ_p[4], dZ_d[4]), "pixels, cosine term live"   # every pixel, in order -- the dense sums
    # an output image is wanted: nothing may be left out (it would have holes) -- the flags are ignored, the image is the dense one
    o_d = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, loss_kind="test", alpha=a, beta=b_, need_dw=False, want_out=True)
    for mode in (True, "pixels"):
        o_s = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, loss_kind="test", alpha=a, beta=b_, need_dw=False, want_out=True, sparse_weight=mode)
        assert torch.equal(o_s[3], o_d[3]) and torch.equal(o_s[1], o_d[1]) and torch.equal(o_s[0], o_d[0])
    # the flag is ignored where it does not apply (training: RENI_NEED_DW) -- same results as without it
    g1 = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, sparse_weight=True)
    g0 = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd)
    assert torch.equal(g1[0], g0[0]) and torch.equal(g1[1], g0[1]) and torch.equal(g1[2], g0[2]), "training call with the flag"


@pytest.mark.parametrize("poison", [0xFF, 0x7F, 0x00])
def test_sparse_weight_with_a_poisoned_workspace(dev, poison):
    """an intermittent failure of the test above inside the full suite (never alone, not reproduced since: 80 stress
    iterations clean in round 5) pointed at a reader of workspace words the sparse path leaves unwritten -- records of unvisited tiles,
    list tails, counts.  Here every byte of the plan's workspace is overwritten (0xFF: NaN floats, -1 indices; 0x7F: NaN floats, huge
    positive indices; zeros) before EACH call: dense, tiles and pixels modes must give the results of the clean run, bit for bit
    (tiles) / to rounding (pixels).  A reader of an unwritten word turns this into a deterministic failure (or a fault)."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    B = 5
    params, Z, D, W, T = random_problem(spec, B, 0, seed=91, grid_w=128)
    P = D.shape[1]
    m = torch.zeros(B, 64, 128, 1)
    m[0, 10:46, 40:83] = 1.0
    m[1] = 1.0
    m[3, 33, 77] = 1.0
    m[4, 10:46, 40:83] = 1.0
    m[4, 0, 0] = 1.0
    Wm = (W.view(1, 64, 128, 3) * m).reshape(B, P, 3)
    plan = make_plan(spec, "bf16")
    fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), Wm.to(dev)

    def run(sparse, dirty, n=B):
        if dirty:
            for ws in plan._ws.values():
                ws.fill_(poison)
        lt, dZ, _, _ = plan.forward_loss_backward(Zd[:n], Dd, fp, Td[:n], Wd[:n], loss_kind="test", alpha=1e-7, beta=1e-4, need_dw=False,
                                                  sparse_weight=sparse)
        return lt.cpu(), dZ.cpu()

    clean = {mode: run(mode, False) for mode in (False, True, "pixels")}
    assert plan._ws, "the plan keeps its workspace"
    for n in (B, 3, 1):   # (odd list lengths, one image: other counts, other tails)
        ref = {mode: run(mode, False, n) for mode in (False, True, "pixels")}
        for mode in (False, True, "pixels"):
            lt, dZ = run(mode, True, n)
            assert torch.isfinite(dZ).all() and torch.isfinite(lt).all(), (mode, n)
            assert torch.equal(lt, ref[mode][0]) and torch.equal(dZ, ref[mode][1]), (mode, n, float((dZ - ref[mode][1]).abs().max()))
    assert torch.equal(clean[True][1], clean[False][1])


@pytest.mark.parametrize("dtype,H,L", [("f32", 64, 3), ("f32", 128, 5), ("bf16", 256, 3), ("bf16", 64, 2)])
def test_sparse_weight_on_the_generic_kernels(dev, dtype, H, L):
    """RENI_WEIGHT_SPARSE / RENI_WEIGHT_COMPACT off the persistent path: fp32 (the parity-grade arithmetic) and the widths without a
    persistent kernel (H = 256 is what the reference ships) walk the same device-built lists in k_reni_main -- tiles: EQUAL to the dense
    call; pixels: equal to rounding, bit-identical run to run."""
    spec = O.DecoderSpec(9, "SO2", H, L, 3, True, "tanh")
    B = 4
    params, Z, D, W, T = random_problem(spec, B, 0, seed=17, grid_w=128)
    P = D.shape[1]
    m = torch.zeros(B, 64, 128, 1)
    m[0, 10:46, 40:83] = 1.0
    m[1] = 1.0
    m[2, 5:9, 100:128] = 1.0
    m[2, 0, 0] = 1.0
    Wm = (W.view(1, 64, 128, 3) * m).reshape(B, P, 3)      # image 3: all-zero weight
    plan = make_plan(spec, dtype)
    fp = flat_params(spec, params).to(dev)
    Zd, Dd, Td, Wd = Z.to(dev), D.to(dev), T.to(dev), Wm.to(dev)

    def run(mode):
        lt, dZ, _, _ = plan.forward_loss_backward(Zd, Dd, fp, Td, Wd, loss_kind="test", alpha=1e-7, beta=1e-4, need_dw=False, sparse_weight=mode)
        return lt.cpu(), dZ.cpu()

    lt_d, dZ_d = run(False)
    lt_s, dZ_s = run(True)
    assert torch.equal(lt_s, lt_d) and torch.equal(dZ_s, dZ_d), (lt_s, lt_d)
    lt_p, dZ_p = run("pixels")
    lt_p2, dZ_p2 = run("pixels")
    assert torch.equal(lt_p, lt_p2) and torch.equal(dZ_p, dZ_p2)
    assert torch.allclose(lt_p, lt_d, rtol=2e-6, atol=0), (lt_p, lt_d)
    for k in range(B):
        assert float((dZ_p[k] - dZ_d[k]).norm()) <= 2e-6 * float(dZ_d[k].norm()) + 1e-12, k
    ref = O.fwd_loss_bwd(spec, params, Z[0:1], D, T[0:1], Wm[0:1], "test", 1e-7, 1e-4, need_dw=False)
    tol = 1e-5 if dtype == "f32" else 3e-2
    assert O.rel_l2(dZ_s[0:1].numpy(), ref["dZ"].numpy()) <= tol
    assert torch.isfinite(dZ_s).all() and float(dZ_s[0].abs().max()) > 0


@pytest.mark.parametrize("dtype,H,L", [("bf16", 128, 5), ("f32", 64, 2)])
def test_sparse_weight_with_ragged_tiles_and_broadcast_strides(dev, dtype, H, L):
    """RENI_WEIGHT_SPARSE / _COMPACT where the bookkeeping has edges: P = 1 000 directions (seven full tiles and one of 104), per-image
    directions, a weight that is ONE [1, P, 1] mask broadcast over batch and channels (strides 0), and a target given as a channel-
    planar view (the reference's permute + view of [B, 3, H, W], RENI_module.py:83-84).  Tiles: equal to dense; pixels: to rounding."""
    spec = O.DecoderSpec(9, "SO2", H, L, 3, True, "tanh")
    B, P = 3, 1000
    params, Z, D, W, T = random_problem(spec, B, P, seed=23, per_image_dirs=True)
    keep = torch.zeros(1, P, 1)
    keep[0, 130:250] = 1.0          # cuts through tiles 1 (pixels 128-255) only ...
    keep[0, 900:1000] = 1.0         # ... and the ragged last tile; pixel 0 masked
    Wb = (keep * 0.7).to(dev).expand(B, P, 3)
    Tp = T.permute(0, 2, 1).contiguous().to(dev).permute(0, 2, 1)       # [B, P, 3] view of a [B, 3, P] buffer
    plan = make_plan(spec, dtype)
    fp = flat_params(spec, params).to(dev)
    Zd, Dd = Z.to(dev), D.to(dev)

    def run(mode):
        lt, dZ, _, _ = plan.forward_loss_backward(Zd, Dd, fp, Tp, Wb, loss_kind="test", alpha=1e-7, beta=1e-4, need_dw=False, sparse_weight=mode)
        return lt.cpu(), dZ.cpu()

    lt_d, dZ_d = run(False)
    lt_s, dZ_s = run(True)
    assert torch.equal(lt_s, lt_d) and torch.equal(dZ_s, dZ_d), (lt_s, lt_d)
    lt_p, dZ_p = run("pixels")
    assert torch.allclose(lt_p, lt_d, rtol=2e-6, atol=0), (lt_p, lt_d)
    for k in range(B):
        assert float((dZ_p[k] - dZ_d[k]).norm()) <= 2e-6 * float(dZ_d[k].norm()) + 1e-12, k
    ref = O.fwd_loss_bwd(spec, params, Z, D, T, (keep * 0.7).expand(B, P, 3), "test", 1e-7, 1e-4, need_dw=False)
    assert O.rel_l2(dZ_s.numpy(), ref["dZ"].numpy()) <= (1e-5 if dtype == "f32" else 3e-2)
    # with pixel 0 kept the cosine term is live: every tile is visited, and the three modes agree bit for bit
    keep[0, 0] = 1.0
    Wb = (keep * 0.7).to(dev).expand(B, P, 3)
    a0, a1, a2 = run(False), run(True), run("pixels")
    assert torch.equal(a0[1], a1[1]) and torch.equal(a0[1], a2[1]) and torch.equal(a0[0], a1[0]) and torch.equal(a0[0], a2[0])


def test_reni_forward_is_the_models_forward(dev):
    """RENI.forward(z) (RENI_module.py:75-78): the module's own inference entry -- the model on the module's grid, for the
    two tensor forms the reference's body (z.size(0)) admits: a latent tensor and a 1-D index tensor."""
    from reni_amd.lightning_module import RENI
    g = load_golden("g6_train_steps.npz")
    ds = _ListDataset(torch.from_numpy(g["imgs"]))
    mod = RENI(_config(), "FIT_DECODER", dataset=ds)
    mod.setup()
    mod.model.load_state_dict({"model." + k: v for k, v in sd_from(g, "sd0.").items()})
    mod.to(dev)
    spec = O.DecoderSpec(9, "SO2", 64, 3, 3, True, "tanh")
    params = {k: v for k, v in sd_from(g, "sd0.").items() if k != "Z"}
    Z0 = sd_from(g, "sd0.")["Z"]
    D = O.get_directions(32)
    with torch.no_grad():
        z = mod.model.Z[:2].detach()
        out_latent = mod(z)
        out_idx = mod(torch.tensor([0, 1], device=dev))
        out_one = mod(torch.tensor([1], device=dev))
    ref = O.reni_forward(spec, params, Z0[:2], D.expand(2, -1, 3))
    assert out_latent.shape == (2, 512, 3)
    assert float((out_latent.cpu() - ref).abs().max()) <= 1e-5
    assert torch.equal(out_idx, out_latent)
    assert torch.equal(out_one, out_latent[1:2])


def test_rows_entry_poisons_an_out_of_range_index_with_nan(dev):
    """reni_forward_loss_backward_rows takes the table's row count; an index outside it (a global / local mix-up with
    sharded tables) reads nothing outside the table and makes the call's loss and gradients NaN -- the reference's Z[idx]
    (RENI_module.py:97-103) raises a device-side assert; silently training on out-of-bounds memory is the one thing not allowed."""
    spec = O.DecoderSpec(9, "SO2", 128, 3, 3, True, "tanh")
    params, Ztab, D, W, T = random_problem(spec, 7, 300, seed=22)
    for dtype in ("f32", "bf16"):
        plan = make_plan(spec, dtype)
        fp = flat_params(spec, params).to(dev)
        good = plan.forward_loss_backward(Ztab.to(dev), D.to(dev), fp, T[:3].to(dev), W.to(dev), idx=torch.tensor([6, 0, 2], device=dev))
        assert bool(torch.isfinite(good[0]).all()) and bool(torch.isfinite(good[1]).all()) and bool(torch.isfinite(good[2]).all())
        for bad_row in (7, -1, 10 ** 9):
            bad = plan.forward_loss_backward(Ztab.to(dev), D.to(dev), fp, T[:3].to(dev), W.to(dev),
                                             idx=torch.tensor([6, bad_row, 2], device=dev))
            assert bool(torch.isnan(bad[0][0])) and bool(torch.isnan(bad[1][1]).all()) and bool(torch.isnan(bad[2]).any())


def test_rows_entry_and_fused_adam_equal_the_separate_calls(dev):
    """reni_forward_loss_backward_rows (the latent gather of RENI_module.py:97-103 inside the prologue kernel) and
    reni_adam_step2 (decoder + latent-table Adam in one launch) are bit-identical to Z[idx] + the separate calls."""
    from reni_amd import ops
    spec = O.DecoderSpec(9, "SO2", 128, 3, 3, True, "tanh")
    params, Ztab, D, W, T = random_problem(spec, 7, 300, seed=21)
    idx = torch.tensor([5, 0, 3, 5], device=dev)              # a repeated row: its gradients accumulate, as index_add_ does
    Tb = T[:4].to(dev)
    for dtype in ("f32", "bf16"):
        plan = make_plan(spec, dtype)
        fp = flat_params(spec, params).to(dev)
        tab = Ztab.to(dev)
        a = plan.forward_loss_backward(tab[idx], D.to(dev), fp, Tb, W.to(dev), loss_kind="test", alpha=1e-3, beta=1e-1)
        b = plan.forward_loss_backward(tab, D.to(dev), fp, Tb, W.to(dev), loss_kind="test", alpha=1e-3, beta=1e-1, idx=idx)
        assert b[1].shape == (4, 9, 3)
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(x, y)
        # fused Adam
        p1, p2, t1, t2 = fp.clone(), fp.clone(), tab.clone(), tab.clone()
        m1, v1, m2, v2 = (torch.rand_like(fp) * 1e-3 for _ in range(4))
        m2.copy_(m1); v2.copy_(v1)
        tm1, tv1 = torch.rand_like(tab) * 1e-3, torch.rand_like(tab) * 1e-3
        tm2, tv2 = tm1.clone(), tv1.clone()
        ops.adam_step(p1, a[2], m1, v1, 3, 1e-2, grad_scale=0.5)
        ops.adam_rows_step(t1, a[1], idx, tm1, tv1, 3, 1e-2, grad_scale=0.5)
        ops.adam_step2(p2, a[2], m2, v2, t2, a[1], idx, tm2, tv2, 3, 1e-2, grad_scale=0.5)
        for x, y in ((p1, p2), (m1, m2), (v1, v2), (t1, t2), (tm1, tm2), (tv1, tv2)):
            assert torch.equal(x, y)
        assert not torch.equal(t1, tab)


def test_fit_decoder_from_exr_files_on_disk(dev, tmp_path):
    """SURVEY 8 f3 end to end: a directory of OpenEXR environment maps -> RENIDatasetHDR (reni_amd/exr.py reader, Resize,
    MinMaxNormalise with the dataset's own log-domain min / max, nan_to_num: src/data/datasets.py:18-101, RENI_module.py:255-281)
    -> the fused training step with the multi-resolution curriculum -> predictions un-normalised + sRGB on the device."""
    from reni_amd import exr, trainer
    from reni_amd.data import RENIDatasetHDR
    from reni_amd.lightning_module import RENI
    from reni_amd.utils import sRGB
    d = tmp_path / "hdr" / "Train"
    d.mkdir(parents=True)
    yy, xx = np.meshgrid(np.arange(32), np.arange(64), indexing="ij")
    for i in range(4):
        sky = np.exp(2.0 * np.cos(np.pi * yy / 32.0) + 0.3 * i)[:, :, None] * np.array([0.6, 0.8, 1.0])
        sun = 500.0 * np.exp(-((yy - 6 - i) ** 2 + (xx - 20 - 5 * i) ** 2) / 6.0)[:, :, None]
        exr.write_exr(str(d / f"env{i + 1}.exr"), (sky + sun).astype(np.float32), pixel_type="half", compression="zip")
    cfg = _config(LR_START=1e-2, LR_END=1e-3, SCHEDULER_TYPE="exponential", EPOCHS=6, BATCH_SIZE=2,
                  MULTI_RES_TRAINING=True, INITAL_RESOLUTION=[16, 32], FINAL_RESOLUTION=[32, 64], CURRICULUM=[3])
    cfg.DATASET = types.SimpleNamespace(NAME="RENI_HDR", RENI_HDR=types.SimpleNamespace(
        PATH=str(tmp_path / "hdr"), TRANSFORMS=[["minmaxnormalise", []]], IS_HDR=True))
    torch.manual_seed(0)
    mod = RENI(cfg, "FIT_DECODER")
    mod.setup()
    assert isinstance(mod.dataset, RENIDatasetHDR) and len(mod.dataset) == 4 and mod.dataset.unnormalise is not None
    img0, i0 = mod.dataset[0]
    assert img0.shape == (3, 16, 32) and float(img0.min()) >= -1 - 1e-6 and float(img0.max()) <= 1 + 1e-6
    hist = trainer.fit(mod, max_epochs=6, device=dev)
    losses = [h["loss"] for h in hist]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] and losses[-1] < losses[3]
    assert mod.cur_res == [32, 64] and mod.dataset[0][0].shape == (3, 32, 64)
    with torch.no_grad():
        pred = mod.forward(mod.model.Z[:2].detach())                       # [2, P, 3] in the normalised log domain
    img = pred.view(2, 32, 64, 3).permute(0, 3, 1, 2)
    lin = mod.dataset.unnormalise(img)                                     # device epilogue (reni_unnormalise_srgb)
    view = sRGB(lin)
    assert lin.is_cuda and bool(torch.isfinite(lin).all()) and float(view.min()) >= 0 and float(view.max()) <= 1


def test_fused_epilogue_shapes_of_the_persistent_training_path(dev):
    """k_tail_a / k_tail_b (round 4: the per-image epilogue of the persistent training path in two launches) at the shapes that stress
    their bookkeeping: ONE image spread over all 256 workgroups (256 image runs in its list), need_dz = False (one block per image, no
    m columns, no dZ blocks), and run-to-run bit-equality of everything they produce."""
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    params, Z, D, W, T = random_problem(spec, 1, 0, seed=77, grid_w=256)   # 1 image x 32 768 directions = 256 tiles
    plan = make_plan(spec, "bf16")
    fp = flat_params(spec, params).to(dev)
    args = (Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev))
    lt, dZ, dp, _ = plan.forward_loss_backward(*args)
    lt2, dZ2, dp2, _ = plan.forward_loss_backward(*args)
    assert torch.equal(lt, lt2) and torch.equal(dZ, dZ2) and torch.equal(dp, dp2)
    ref = O.fwd_loss_bwd(spec, params, Z, D.expand(1, -1, 3), T, W.expand(1, -1, 3))
    assert abs(float(lt[0]) - ref["loss_terms"][0]) <= 3e-3 * abs(ref["loss_terms"][0])
    assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()) <= 3e-2
    gp = unflatten(spec, dp.cpu())
    for k in gp:
        if gp[k].numel() > 3:
            assert O.rel_l2(gp[k].numpy(), ref["grads"][k].numpy()) <= 3e-2, k
    # without the latent gradient: the same decoder gradient and loss, bit for bit
    lt3, dZ3, dp3, _ = plan.forward_loss_backward(*args, need_dz=False)
    assert dZ3 is None and torch.equal(lt3, lt) and torch.equal(dp3, dp)
    # three images whose tile ranges cut through workgroups (3 x 40 tiles over 120 workgroups, then over fewer: 3 x 300 tiles / 256)
    for w in (80, 160):
        params, Z, D, W, T = random_problem(spec, 3, 0, seed=78, grid_w=w)
        fp = flat_params(spec, params).to(dev)
        lt, dZ, dp, _ = plan.forward_loss_backward(Z.to(dev), D.to(dev), fp, T.to(dev), W.to(dev))
        ref = O.fwd_loss_bwd(spec, params, Z, D.expand(3, -1, 3), T, W.expand(3, -1, 3))
        assert abs(float(lt[0]) - ref["loss_terms"][0]) <= 3e-3 * abs(ref["loss_terms"][0])
        assert O.rel_l2(dZ.cpu().numpy(), ref["dZ"].numpy()) <= 3e-2
        gp = unflatten(spec, dp.cpu())
        for k in gp:   # (every layer: below eight tiles per workgroup the partial reduction of layers >= 2 rides in k_tail_a's launch,
            if gp[k].numel() > 3:   # and 300 tiles run on 150 workgroups of two -- the balanced count -- not on 256 of one or two)
                assert O.rel_l2(gp[k].numpy(), ref["grads"][k].numpy()) <= 3e-2, (w, k)
