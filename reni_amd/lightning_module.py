"""``RENI`` -- the training-harness surface of the reference's LightningModule
(/root/reference/src/lightning/RENI_module.py:21-361): ``forward(z)``, ``training_step(batch,
batch_idx) -> dict``, ``configure_optimizers()``, ``load_state_dict`` -- with the step body
re-implemented on the fused HIP path.

pytorch_lightning is not installed here; when it is importable ``RENI`` derives from
``pl.LightningModule`` and can be handed to a ``pl.Trainer``; otherwise it derives from a small
duck-typed base and ``reni_amd.trainer.fit`` drives it.  The FIT_INVERSE task shades through
``reni_amd.envmap_shader`` (HIP); the rasteriser that produces its G-buffer is pytorch3d's and outside this
build, so the renderer is handed in with ``set_renderer`` (a stored G-buffer, or a pytorch3d MeshRenderer).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import dist as rdist
from .custom_transforms import transform_builder
from .data import SyntheticEnvMapDataset, get_dataset
from .loss_functions import RENITestLoss, RENITestLossInverse, RENITrainLoss, RENIVADTrainLoss
from .models import get_model
from .optim import FusedAdam
from .utils import get_directions, get_mask, get_sineweight

try:  # pragma: no cover - not installed in this image
    import pytorch_lightning as pl

    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(torch.nn.Module):
        """The handful of LightningModule attributes the RENI module touches."""

        def __init__(self):
            super().__init__()
            self.global_step = 0
            self.current_epoch = 0
            self.logged = {}

        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, metrics, **kwargs):
            self.logged.update({k: float(v) for k, v in metrics.items()})


class RENI(_Base):
    def __init__(self, config, task, dataset=None):
        super().__init__()
        self.save_hyperparameters()
        self.config = config
        self.task = task
        self.model_from_checkpoint = False
        self._injected_dataset = dataset
        self.example_input_array = torch.randn(
            self.config.TRAINER.LOGGER.NUMBER_OF_IMAGES, self.config.RENI.LATENT_DIMENSION, 3)

    # ------------------------------------------------------------------ setup (RENI_module.py:37-54)
    def setup(self, stage=None):
        if not self.model_from_checkpoint:
            self.setup_dataset()
            self.model = get_model(self.config, len(self.dataset), self.task)
        self.model_type = self.config.RENI.MODEL_TYPE
        self.directions = get_directions(self.cur_res[1])  # (1, H*W, 3)
        self.sineweight = get_sineweight(self.cur_res[1])  # (1, H*W, 3)
        self.setup_for_task(self.task)
        self.mask = None
        self.renderer, self.render_kwargs, self.gt_renders = None, {}, None
        if self.task == "FIT_LATENT" and self.config.RENI.FIT_LATENT.APPLY_MASK:
            self.mask = get_mask(self.cur_res[1], self.config.RENI.FIT_LATENT.MASK_PATH)  # (1, H*W, 3)
        self._grid_cache = {}

    def on_load_checkpoint(self, checkpoint) -> None:
        self.setup_dataset()
        self.model = get_model(self.config, len(self.dataset), self.task)
        self.model_from_checkpoint = True

    def load_state_dict(self, state_dict, strict: bool = True):
        self.model.load_state_dict(state_dict)

    def on_fit_start(self):
        if self.task == "FIT_INVERSE" and self.renderer is None:
            raise NotImplementedError("FIT_INVERSE renders through an environment-map shader: call set_renderer() with "
                                      "reni_amd.envmap_shader.GBufferRenderer (stored G-buffer) or build_renderer's "
                                      "pytorch3d MeshRenderer first")

    # ------------------------------------------------------------------ FIT_INVERSE (RENI_module.py:65-73, 363-396)
    def set_renderer(self, renderer, render_kwargs=None):
        """`renderer(envmap=EnvironmentMap, **render_kwargs) -> (render [B,Hr,Wr,3], normals)`: the call the reference
        makes on its pytorch3d MeshRenderer (RENI_module.py:393-395).  Generates the ground-truth renders."""
        self.renderer = renderer
        self.render_kwargs = dict(render_kwargs or {})
        self.generate_gt_renders()

    def get_render(self, model_output, directions, sineweight):
        from .envmap_shader import EnvironmentMap
        B = model_output.shape[0]
        envmap = EnvironmentMap(environment_map=model_output, directions=directions.expand(B, -1, -1), sineweight=sineweight)
        render, _ = self.renderer(envmap=envmap, **self.render_kwargs)
        return render

    def generate_gt_renders(self, device=None):
        """RENI_module.py:363-384: render every (un-normalised) dataset image once."""
        device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        with torch.no_grad():
            out = []
            for i in range(len(self.dataset)):
                imgs, _ = self.dataset[i]
                imgs = self.dataset.unnormalise(imgs.unsqueeze(0).to(device))
                imgs = imgs.permute(0, 2, 3, 1).reshape(1, -1, 3)
                directions, sineweight = self._grids(imgs)
                out.append(self.get_render(imgs, directions, sineweight))
            self.gt_renders = torch.cat(out, dim=0)

    # ------------------------------------------------------------------ grids on the device
    def _grids(self, like: torch.Tensor):
        """directions / (masked) sineweight on ``like``'s device.  The reference .repeat()s them per
        batch element (RENI_module.py:89-94); the kernels take the single shared grid."""
        key = (like.device, self.cur_res[1], self.mask is not None)
        g = self._grid_cache.get(key)
        if g is None or g[2] is not self.directions:
            d = self.directions.to(like.device, torch.float32)
            s = self.sineweight.to(like.device, torch.float32)
            if self.mask is not None:
                s = s * self.mask.to(like.device, torch.float32)
            g = (d, s, self.directions)
            self._grid_cache = {key: g}
        return g[0], g[1]

    # ------------------------------------------------------------------ inference (RENI_module.py:75-78)
    def forward(self, z):
        directions, _ = self._grids(z)
        return self.model(z, directions)

    # ------------------------------------------------------------------ the step (RENI_module.py:80-146)
    def training_step(self, batch, batch_idx):
        imgs, idx = batch
        batch_size, _, _, _ = imgs.size()
        imgs = imgs.permute(0, 2, 3, 1)  # (B, C, H, W) -> (B, H, W, C): a strided view, never copied
        imgs = imgs.view(batch_size, -1, 3)  # (B, H*W, 3), channel-planar strides
        directions, sineweight = self._grids(imgs)

        if self.model_type == "AutoDecoder":
            Z = self.model.Z[idx, :, :]
        elif self.model_type == "VariationalAutoDecoder":
            if self.task == "FIT_DECODER":
                Z, mu, log_var = self.model.sample_latent(idx)
            else:
                Z = self.model.mu[idx, :, :]

        if self.task == "FIT_INVERSE":  # RENI_module.py:105-112, 135-144
            if self.renderer is None:
                raise NotImplementedError("FIT_INVERSE: call set_renderer() first (see on_fit_start)")
            gt = self.gt_renders[idx, :, :, :]
            model_output = self.model(Z, directions)
            model_output = self.dataset.unnormalise(model_output)
            model_output = self.get_render(model_output, directions, sineweight)
            loss, mse_loss, prior_loss, cosine_loss = self.criterion(model_output, gt, Z)
            return {"loss": loss, "mse_loss": mse_loss, "prior_loss": prior_loss, "cosine_loss": cosine_loss}

        if self.task == "FIT_DECODER":
            if self.model_type == "AutoDecoder":
                loss = self.criterion.fused(self.model, Z, directions, imgs, sineweight)
                log_dict = {"loss": loss}
            elif self.model_type == "VariationalAutoDecoder":
                loss, mse_loss, kld_loss = self.criterion.fused(self.model, Z, directions, imgs, sineweight, mu, log_var)
                log_dict = {"loss": loss, "mse_loss": mse_loss, "kld_loss": kld_loss}
        elif self.task == "FIT_LATENT":
            loss, mse_loss, prior_loss, cosine_loss = self.criterion.fused(self.model, Z, directions, imgs, sineweight,
                                                                           sparse_weight=self.mask is not None)
            log_dict = {"loss": loss, "mse_loss": mse_loss, "prior_loss": prior_loss, "cosine_loss": cosine_loss}
        return log_dict

    def training_epoch_end(self, training_step_outputs):
        metrics = {}
        for key in training_step_outputs[0].keys():
            vals = torch.stack([x[key].detach() for x in training_step_outputs])
            metrics[f"{self.task.lower()}_{key}"] = rdist.allreduce_mean_scalars(torch.mean(vals))
        metrics["step"] = self.current_epoch + 1.0
        self.log_dict(metrics, on_step=False, on_epoch=True, prog_bar=True, batch_size=self.batch_size, sync_dist=True)

    def train_dataloader(self):
        return self.dataloader

    # ------------------------------------------------------------------ optimiser (RENI_module.py:168-252)
    def build_optimizer(self, model, model_type, optimizer_type, learning_rate, beta1, beta2, fixed_decoder=False):
        if fixed_decoder:  # only the latent codes are optimised
            parameters = [model.mu] if model_type == "VariationalAutoDecoder" else [model.Z]
        else:
            parameters = [p for p in model.parameters()]
        if optimizer_type == "adam":
            # the reference builds Adam(parameters, lr) -- configured betas are ignored (:192)
            if parameters and parameters[0].is_cuda:
                return FusedAdam(parameters, lr=learning_rate)
            return torch.optim.Adam(parameters, lr=learning_rate)
        # the reference's sgd / adagrad branches raise TypeError / AttributeError
        raise TypeError(f"optimizer {optimizer_type!r} is not usable in the reference either; use 'adam'")

    def build_scheduler(self, scheduler_type, optimizer, step_size, lr_start=None, lr_end=None, gamma=None, epochs=None):
        if scheduler_type == "step":
            return torch.optim.lr_scheduler.StepLR(optimizer, step_size=step_size, gamma=gamma)
        if scheduler_type == "exponential":
            gamma = np.exp(np.log(lr_end / lr_start) / epochs)
            return torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=gamma)
        if scheduler_type == "plateau":
            return torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=gamma, patience=step_size)
        return None

    def configure_optimizers(self):
        optimizer = self.build_optimizer(self.model, self.config.RENI.MODEL_TYPE, self.optimiser_type, self.lr_start,
                                         self.beta1, self.beta2, self.fixed_decoder)
        scheduler = self.build_scheduler(self.scheduler_type, optimizer, self.step_size, self.lr_start, self.lr_end,
                                         self.gamma, self.epochs)
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": scheduler, "interval": "epoch", "monitor": "loss", "frequency": 1,
                                 "strict": True}}

    # ------------------------------------------------------------------ data (RENI_module.py:254-290)
    def setup_dataset(self):
        tcfg = getattr(self.config.RENI, self.task)
        img_size = tcfg.INITAL_RESOLUTION if tcfg.MULTI_RES_TRAINING else tcfg.FINAL_RESOLUTION
        self.cur_res = list(img_size)
        if self._injected_dataset is not None:
            self.dataset = self._injected_dataset
        else:
            ds = self.config.DATASET
            name = getattr(ds, "NAME", None)
            if name == "SYNTHETIC":  # (not in the reference: the stand-in for boxes without the dataset)
                n = ds.SYNTHETIC.N_TRAIN if self.task == "FIT_DECODER" else ds.SYNTHETIC.N_TEST
                self.dataset = SyntheticEnvMapDataset(n, img_size[0], img_size[1])
            else:  # RENI_HDR / RENI_LDR / CUSTOM on disk: resize, then the configured transforms (RENI_module.py:255-281)
                dcfg = getattr(ds, name)
                self.is_hdr = dcfg.IS_HDR
                transforms = transform_builder([["resize", list(img_size)]] + [list(t) for t in dcfg.TRANSFORMS])
                split = "Train" if self.task == "FIT_DECODER" else "Test"
                self.dataset = get_dataset(name, dcfg.PATH + os.sep + split, transforms, self.is_hdr)
        self.batch_size = tcfg.BATCH_SIZE
        self.dataloader = torch.utils.data.DataLoader(self.dataset, batch_size=self.batch_size)

    def setup_for_task(self, task):
        t = getattr(self.config.RENI, task)
        self.lr_start, self.lr_end = t.LR_START, t.LR_END
        self.beta1, self.beta2 = t.OPTIMIZER_BETA_1, t.OPTIMIZER_BETA_2
        self.optimiser_type = t.OPTIMIZER
        self.scheduler_type = t.SCHEDULER_TYPE
        self.epochs = t.EPOCHS
        self.step_size = t.SCHEDULER_STEP_SIZE
        self.gamma = t.SCHEDULER_GAMMA
        self.multi_res_training = t.MULTI_RES_TRAINING
        self.curriculum = t.CURRICULUM
        h_start, h_end = t.INITAL_RESOLUTION[0], t.FINAL_RESOLUTION[0]
        if task == "FIT_DECODER":
            self.fixed_decoder = False
            if self.model_type == "AutoDecoder":
                self.criterion = RENITrainLoss()
            elif self.model_type == "VariationalAutoDecoder":
                self.criterion = RENIVADTrainLoss(beta=t.KLD_WEIGHTING, Z_dims=3 * self.config.RENI.LATENT_DIMENSION)
        elif task == "FIT_LATENT":
            self.fixed_decoder = True
            self.criterion = RENITestLoss(alpha=t.PRIOR_LOSS_WEIGHT, beta=t.COSINE_SIMILARITY_WEIGHT)
        elif task == "FIT_INVERSE":
            self.fixed_decoder = True
            self.criterion = RENITestLossInverse(alpha=t.PRIOR_LOSS_WEIGHT, beta=t.COSINE_SIMILARITY_WEIGHT)
        # config sanity checks of the reference (RENI_module.py:360-361)
        assert max(self.curriculum) < self.epochs
        assert len(self.curriculum) >= np.log2(h_end / h_start)

    # ------------------------------------------------------------------ multi-res curriculum (callbacks.py:11-29)
    def maybe_double_resolution(self):
        """What MultiResTrainingCallback.on_train_epoch_end does."""
        if self.multi_res_training and self.current_epoch + 1 in self.curriculum:
            self.cur_res = [2 * x for x in self.cur_res]
            self.directions = get_directions(self.cur_res[1])
            self.sineweight = get_sineweight(self.cur_res[1])
            if self.mask is not None:
                self.mask = get_mask(self.cur_res[1], self.config.RENI.FIT_LATENT.MASK_PATH)
            self.dataset.double_resolution()
            self.dataloader = torch.utils.data.DataLoader(self.dataset, batch_size=self.batch_size)
            return True
        return False
