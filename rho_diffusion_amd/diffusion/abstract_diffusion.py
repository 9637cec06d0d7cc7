"""Pipeline base class (reference: rho_diffusion/diffusion/abstract_diffusion.py:51-276).

Same constructor contract (backbone / cond_fn / optimizer resolved by registry name), same helper
methods.  Lightning is optional: when ``lightning`` is importable the class is a LightningModule
(so ``Trainer.fit`` works as in scripts/training.py); otherwise a minimal stand-in supplies the
attributes the pipelines use (``log``, ``device``, ``global_rank``, ``current_epoch``, ``hparams``).
"""
from __future__ import annotations

import math
import types
from inspect import getfullargspec
from logging import getLogger
from typing import Any, Mapping, Union

import torch
from torch import Tensor, nn
from torch.optim import AdamW

from ..registry import registry

try:  # pragma: no cover - lightning is not installed in the build container
    from lightning import pytorch as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    class _Base(nn.Module):
        """LightningModule stand-in (attribute surface only)."""

        def __init__(self):
            super().__init__()
            self.hparams = types.SimpleNamespace()
            self.global_rank = int(__import__("os").environ.get("RANK", "0"))
            self.current_epoch = 0
            self._logged = {}

        def save_hyperparameters(self, d):
            for k, v in d.items():
                setattr(self.hparams, k, v)

        def log(self, name, value, **kwargs):
            self._logged[name] = value

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

__all__ = ["AbstractDiffusionPipeline"]


class AbstractDiffusionPipeline(_Base):
    def __init__(self, backbone, backbone_kwargs: dict, schedule, timesteps: Union[int, Tensor] = 1000,
                 cond_fn=None, cond_fn_kwargs: dict = None, optimizer=None, opt_kwargs: Union[Mapping[str, Any], None] = {}):
        super().__init__()
        if isinstance(backbone, str):
            backbone = registry.get("models", backbone)
        self.backbone = backbone(**backbone_kwargs)
        self.backbone_kwargs = backbone_kwargs
        if isinstance(cond_fn, str):
            cond_fn_class = registry.get("layers", cond_fn)
            self.backbone.cond_fn = cond_fn_class(**(cond_fn_kwargs or {}))
        if optimizer is None:
            optimizer = AdamW          # reference default ("AdamW" string at :72-73 then used as a class)
        elif isinstance(optimizer, str):
            optimizer = registry.get("optimizers", optimizer)
        self.optimizer = optimizer
        self.schedule = schedule
        self.metrics = nn.ModuleDict()
        self.save_hyperparameters({"model_kwargs": backbone_kwargs, "opt_kwargs": dict(opt_kwargs or {})})
        self.timesteps = timesteps
        self._python_logger = getLogger(self.__class__.__name__)

    def configure_optimizers(self, mpi_world_size: int = 1):
        """AdamW defaults merged with the configured kwargs; lr *= sqrt(world) (abstract_diffusion.py:86-148).
        Unlike the reference the stored kwargs are not mutated."""
        opt_kwargs = dict(self.hparams.opt_kwargs)
        spec = getfullargspec(AdamW)
        names = [a for a in spec.args if a not in ("self", "params")]
        for key, value in zip(names, spec.defaults or ()):
            opt_kwargs.setdefault(key, value)
        opt_kwargs.pop("lr_schedule", None)
        opt_kwargs["lr"] = opt_kwargs["lr"] * math.sqrt(mpi_world_size)
        if self.optimizer is AdamW:
            from ..optim import HipAdamW
            opt = HipAdamW(self.parameters(), **{k: v for k, v in opt_kwargs.items()
                                                 if k in ("lr", "betas", "eps", "weight_decay")})
        else:
            opt = self.optimizer(self.parameters(), **opt_kwargs)
        return {"optimizer": opt}

    @property
    def schedule(self):
        return self._schedule

    @schedule.setter
    def schedule(self, _schedule) -> None:
        self._schedule = _schedule

    def random_timesteps(self, num_steps: int) -> Tensor:
        """CPU randint with replacement (abstract_diffusion.py:163-169)."""
        idx = torch.randint(low=0, high=self.timesteps, size=(num_steps,))
        return torch.arange(0, self.timesteps)[idx]

    def reshape_timesteps(self, data: Tensor, t: Tensor) -> Tensor:
        return t.view((-1, *((1,) * (data.ndim - 1))))

    def get_schedule_parameters_at_time(self, data: Tensor, t: torch.LongTensor) -> dict:
        """abstract_diffusion.py:194-220 (kept for API compatibility; the HIP loop reads the
        device-resident coefficient table instead)."""
        result = {}
        for key in ["alpha_t", "beta_t", "alpha_bar_t", "sigma_t"]:
            result[key] = self.reshape_timesteps(data, self.schedule[key].to(data.device)[t])
        return result

    @staticmethod
    def _parse_batch(batch):
        """The batch forms every training_step of the reference accepts (ddpm.py:247-258): [data, labels] list,
        {"data", "label"} dict, bare tensor."""
        if isinstance(batch, list):
            data, labels = batch
        elif isinstance(batch, dict):
            data, labels = batch.get("data"), batch.get("label")
        else:
            data, labels = batch, None
        return data, labels

    def _loss(self, pred: Tensor, target: Tensor) -> Tensor:
        """``loss_func(pred, target)``; the default mean MSELoss runs as the HIP reduction + gradient (rho_mse)."""
        loss_func = getattr(self, "loss_func", None)
        if loss_func is None or (isinstance(loss_func, nn.MSELoss) and loss_func.reduction == "mean"):
            from ..autograd import mse_loss
            return mse_loss(pred, target)
        return loss_func(pred, target)

    def _preembed_conditions(self, cc):
        """Sampling evaluates the backbone T times with the SAME labels: run the label embedding (``cond_fn``, e.g.
        MultiEmbeddings with its host-synchronising table lookup, conditioning.py:115-139) once and hand the UNet the
        pre-embedded ``[B, 4*mc]`` form it also accepts (unet_v2.py:702-719).  Same arithmetic per step, no per-step host
        sync, and the step stays capturable in a HIP graph."""
        fn = getattr(self.backbone, "cond_fn", None)
        mc = getattr(self.backbone, "model_channels", None)
        if cc is None or fn is None or mc is None or not torch.is_tensor(cc):
            return cc
        if cc.dim() == 2 and cc.shape[1] == 4 * mc:
            return cc
        with torch.no_grad():
            emb = fn(cc)
        if emb.dim() == 2 and emb.shape[1] == 4 * mc:
            return emb.float().contiguous()
        return cc

    error_poll_every = 50

    def _check_backbone_errors(self) -> None:
        """Poll the backbone's sticky device error flag (a label outside the parameter space: IndexError like the reference's
        conditioning.py:132).  One host sync: called where a pipeline synchronises anyway (end of a sampling chain, the
        NaN-flag poll of DDPM.training_step), never per step."""
        chk = getattr(self.backbone, "check_errors", None)
        if chk is not None:
            chk()

    def _tick_error_poll(self) -> None:
        """training_step of the pipelines without a NaN-flag poll of their own: every ``error_poll_every`` steps."""
        self._err_ticks = getattr(self, "_err_ticks", 0) + 1
        if self._err_ticks % self.error_poll_every == 0:
            self._check_backbone_errors()

    def forward_process(self, data: Tensor, t: Union[Tensor, None] = None):
        ...

    def reverse_process(self, *args, **kwargs):
        ...

    @staticmethod
    def make_image_grid(batched_image, filename: str = None):
        """The reference tiles the batch into one image with torchvision and optionally saves a PNG
        (abstract_diffusion.py:240-276).  Plotting is out of scope: the samples are handed back as they are,
        so callers that do ``generate(...).cpu().numpy()`` (scripts/inference.py:166-169) keep working."""
        return batched_image
