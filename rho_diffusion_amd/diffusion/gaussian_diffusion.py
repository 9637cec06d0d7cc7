"""GaussianDiffusionPipeline: the pipeline ``scripts/inference.py:122`` instantiates (reference:
rho_diffusion/diffusion/gaussian_diffusion.py:145-1227).  Same constructor; the fixed configuration of the reference
(``diffusion_defaults``, :199-209): cosine betas, x0-prediction (``predict_xstart=True``), fixed-large variance,
MSE loss, no timestep rescaling.  Built here: the coefficient tables (:237-273), ``q_sample`` / ``forward_process``
(:294-312, :1014-1027) and the sampling path ``reverse_process`` (:1029-1099) = DDIM (eta = 0, :654-702) on top of
``p_mean_variance`` with dynamic thresholding (:400-415).

Arithmetic in librho_hip.so:
  * backbone            -> UNet engine (x0 prediction, float32 [B, C, *S] out)
  * dynamic thresholding -> rho_abs_quantile: exact per-sample 0.9-quantile of |x0| (radix select), no sort, no host sync
  * DDIM update          -> rho_ddim_step (clamp / rescale, eps re-derivation, x_{t-1}) in the reference's operation order
  * q_sample             -> rho_q_sample_coef;  noise -> rho_philox_normal
The per-step scalars (four table entries) are kernel arguments computed on the host from the float64 tables exactly as
``_extract_into_tensor`` + the float32 tensor expressions of ``ddim_sample`` do; the loop never synchronises.

``training_step`` (:1153-1210) reproduces the reference's objective as written: the data are noised twice with the same
noise (:1186 then :877) and the backbone output is regressed on the once-noised data.
"""
from __future__ import annotations

import math
import os
from typing import Any, Mapping, Union

import numpy as np
import torch
from torch import Tensor

from .. import hip
from ..engine import ops
from ..registry import registry
from ..utils import sample_from_discrete_parameter_space, save_model_checkpoint
from .abstract_diffusion import AbstractDiffusionPipeline

__all__ = ["GaussianDiffusionPipeline", "get_named_beta_schedule", "betas_for_alpha_bar", "diffusion_tables", "ddim_coefficients"]


def betas_for_alpha_bar(num_diffusion_timesteps: int, alpha_bar, max_beta: float = 0.999) -> np.ndarray:
    """gaussian_diffusion.py:72-89."""
    betas = []
    for i in range(num_diffusion_timesteps):
        t1 = i / num_diffusion_timesteps
        t2 = (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(betas)


def get_named_beta_schedule(schedule_name: str, num_diffusion_timesteps: int) -> np.ndarray:
    """gaussian_diffusion.py:45-69."""
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def diffusion_tables(betas: np.ndarray) -> dict:
    """The float64 coefficient tables of GaussianDiffusionPipeline.__init__ (gaussian_diffusion.py:237-273)."""
    betas = np.array(betas, dtype=np.float64)
    assert len(betas.shape) == 1, "betas must be 1-D"
    assert (betas > 0).all() and (betas <= 1).all()
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": betas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev, "alphas_cumprod_next": np.append(ac[1:], 0.0),
        "sqrt_alphas_cumprod": np.sqrt(ac), "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac), "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1), "posterior_variance": pv,
        "posterior_log_variance_clipped": np.log(np.append(pv[1], pv[1:])),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }


def ddim_coefficients(tables: Mapping[str, np.ndarray], t: int, eta: float = 0.0):
    """The float32 scalars of ddim_sample (:675-697) for a batch-uniform timestep: tables gathered in float64, cast to
    float32 (``_extract_into_tensor``), then combined with float32 arithmetic like the reference's tensor expressions.
    Returns (c_recip, c_recipm1, sqrt_abar_prev, coef_eps, sigma_masked) for rho_ddim_step."""
    f = np.float32
    c_recip, c_recipm1 = f(tables["sqrt_recip_alphas_cumprod"][t]), f(tables["sqrt_recipm1_alphas_cumprod"][t])
    ab, abp = f(tables["alphas_cumprod"][t]), f(tables["alphas_cumprod_prev"][t])
    one = f(1.0)
    sigma = f(eta) * np.sqrt((one - abp) / (one - ab), dtype=f) * np.sqrt(one - ab / abp, dtype=f)
    coef_eps = np.sqrt(one - abp - sigma * sigma, dtype=f)
    mask = f(1.0 if t != 0 else 0.0)
    return float(c_recip), float(c_recipm1), float(np.sqrt(abp, dtype=f)), float(coef_eps), float(mask * sigma)


class GaussianDiffusionPipeline(AbstractDiffusionPipeline):
    def __init__(self, backbone, backbone_kwargs: dict, schedule, loss_func, timesteps: Union[int, Tensor] = 1000,
                 cond_fn: str = None, cond_fn_kwargs: dict = None, optimizer=None,
                 opt_kwargs: Union[Mapping[str, Any], None] = {}, t_checkpoints=None, sampling_batch_size=10,
                 sample_every_n_epochs=5, sample_parameter_space=None, save_checkpoint_every_n_epochs=10):
        super().__init__(backbone=backbone, backbone_kwargs=backbone_kwargs, schedule=schedule, timesteps=timesteps,
                         cond_fn=cond_fn, cond_fn_kwargs=cond_fn_kwargs, optimizer=optimizer, opt_kwargs=opt_kwargs)
        if isinstance(loss_func, str):
            loss_func = registry.get("nn", loss_func)
        if isinstance(loss_func, type):
            loss_func = loss_func()
        self.loss_func = loss_func
        self.t_checkpoints = t_checkpoints
        self.sampling_batch_size = sampling_batch_size
        self.sample_every_n_epochs = sample_every_n_epochs
        self.sample_parameter_space = sample_parameter_space
        self.save_weights_every_n_epochs = save_checkpoint_every_n_epochs
        self.rescale_timesteps = False

        # float64 tables (:237-273)
        self.tables = diffusion_tables(get_named_beta_schedule("cosine", int(timesteps)))
        for k_, v_ in self.tables.items():
            setattr(self, k_, v_)
        self.timesteps = int(self.betas.shape[0])
        self.dynamic_thresholding_percentile = 0.9

        self.noise_seed = int(os.environ.get("RHO_SEED", "777")) + int(os.environ.get("RANK", "0"))
        self._noise_offset = 0
        self._dev_tables = {}
        self._quant_ws = None

    # ------------------------------------------------------------------ noise / q_sample
    def noise(self, data: Tensor) -> Tensor:
        """:1011-1012, Philox4x32-10 on the device (seed + rank; offset advances per draw)."""
        hip.require_gpu(data, "data")
        out = torch.empty(data.shape, dtype=torch.float32, device=data.device)
        ops.philox_normal(out, self.noise_seed, self._noise_offset)
        self._noise_offset += (out.numel() + 3) // 4
        return out

    def _table(self, name: str, device) -> Tensor:
        key = (name, str(device))
        if key not in self._dev_tables:
            self._dev_tables[key] = torch.from_numpy(getattr(self, name)).float().to(device).contiguous()
        return self._dev_tables[key]

    def q_sample(self, x_start: Tensor, t: Tensor, noise: Tensor = None) -> Tensor:
        """:294-312."""
        hip.require_gpu(x_start, "x_start")
        x0 = x_start.float().contiguous()
        if noise is None:
            noise = self.noise(x0)
        assert noise.shape == x_start.shape
        t = t.reshape(-1).to(device=x0.device, dtype=torch.int64).contiguous()
        return ops.q_sample_coef(x0, noise.float().contiguous(), t, self._table("sqrt_alphas_cumprod", x0.device),
                                 self._table("sqrt_one_minus_alphas_cumprod", x0.device))

    def forward_process(self, data: Tensor, t: Union[Tensor, None] = None) -> list:
        """:1014-1027: returns [x_t, noise]."""
        hip.require_gpu(data, "data")
        self.schedule.dtype = data.dtype
        if t is None:
            t = self.random_timesteps(data.size(0))
        noise = self.noise(data)
        return [self.q_sample(x_start=data, t=t, noise=noise), noise]

    # ------------------------------------------------------------------ DDIM coefficients of one step
    def ddim_coefficients(self, t: int, eta: float = 0.0):
        return ddim_coefficients(self.tables, t, eta)

    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def reverse_process(self, x_T: Tensor, conditions=None, t_checkpoints=None, eta: float = 0.0) -> dict:
        """:1029-1099.  ``x_T`` is a shape/device template (:1039 starts from randn_like(x_T))."""
        hip.require_gpu(x_T, "x_T")
        dev = x_T.device
        batch_size = x_T.size(0)
        denoise_steps = len(self.betas)
        x_t = self.noise(x_T).contiguous()
        if t_checkpoints is not None:
            num_checkpoints = len(t_checkpoints)
            buf = torch.zeros((batch_size, num_checkpoints) + tuple(x_T.shape[1:]), dtype=torch.float32, device=dev)
            steps_per_ckpt = denoise_steps // num_checkpoints
        else:
            num_checkpoints, buf, steps_per_ckpt = 0, None, denoise_steps
        if conditions is not None:
            if isinstance(conditions, int):
                cc = torch.full((batch_size,), fill_value=conditions, device=dev, dtype=torch.long)
            elif isinstance(conditions, str) and conditions == "auto":
                cc = torch.randint(0, 10, (batch_size,), device=dev).long()
            elif isinstance(conditions, torch.Tensor):
                cc = conditions
            elif isinstance(conditions, list):
                cc = torch.tensor(conditions).to(dev)
        else:
            cc = None
        cc = self._preembed_conditions(cc)

        engine = self.backbone.engine() if hasattr(self.backbone, "engine") else None
        t_dev = torch.full((1,), denoise_steps - 1, dtype=torch.int32, device=dev)
        quant = torch.empty(batch_size, dtype=torch.float32, device=dev)
        need = hip.lib().rho_abs_quantile_workspace_bytes(batch_size)
        if self._quant_ws is None or self._quant_ws.device != dev or self._quant_ws.numel() * 4 < need:
            self._quant_ws = torch.empty((need + 3) // 4, dtype=torch.int32, device=dev)
        t_idx = 0
        for t in range(denoise_steps - 1, -1, -1):
            if engine is not None:
                x0_hat = engine.forward(x_t, None, cc, t_scalar_dev=t_dev)
            else:
                x0_hat = self.backbone(x_t, torch.full((batch_size,), t, device=dev, dtype=torch.long), cc)
            x0_hat = x0_hat.contiguous()
            c_recip, c_recipm1, sqrt_abp, coef_eps, sig = self.ddim_coefficients(t, eta)
            z = self.noise(x_t) if sig != 0.0 else None          # the reference draws it always; with eta = 0 it is multiplied by 0
            ops.abs_quantile(x0_hat, self.dynamic_thresholding_percentile, out=quant, workspace=self._quant_ws)
            ops.ddim_step(x_t, x0_hat, quant, z, x_t, None, c_recip, c_recipm1, sqrt_abp, coef_eps, sig)
            if buf is not None and t % steps_per_ckpt == 0 and t_idx < num_checkpoints:
                buf[:, t_idx].copy_(x_t)
                t_idx += 1
            ops.step_advance(t_dev, None, 0)
        self._check_backbone_errors()
        return {"buffer": buf, "denoised": x_t}

    # ------------------------------------------------------------------ wrappers
    def generate(self, parameter_space=None, random=False, save_figure_as=None):
        """:1102-1146: zero template of the last training batch shape, else ``[sampling_batch_size, out_channels] + data_shape``
        from the backbone kwargs; labels = rows of the discrete parameter space (``sample_parameter_space`` by default);
        reverse_process.  Returns what ``make_image_grid`` returns: the denoised samples (no figure is drawn)."""
        if hasattr(self, "data_shape"):
            shape = [int(x) for x in self.data_shape]
            shape[0] = self.sampling_batch_size
        else:
            shape = [self.sampling_batch_size, self.backbone_kwargs["out_channels"]] + list(self.backbone_kwargs["data_shape"])
            self.data_dtype = torch.float32
        dev = next(self.backbone.parameters()).device
        if parameter_space is None:
            parameter_space = self.sample_parameter_space
        conditions = None
        if parameter_space is not None:
            conditions = sample_from_discrete_parameter_space(parameter_space, shape[0], random=random, device=dev)
        x_T = torch.zeros(shape, dtype=getattr(self, "data_dtype", torch.float32), device=dev)
        res = self.reverse_process(x_T, conditions=conditions, t_checkpoints=self.t_checkpoints)
        res["conditions"] = conditions
        self.last_samples = res
        return self.make_image_grid(res["denoised"], filename=save_figure_as)

    def save_model_weights(self):
        save_model_checkpoint(self.backbone, "model.pth")

    def training_step(self, batch, batch_idx: int = 0):
        """:1153-1210 with ``training_losses`` (:861-934) for the class's fixed configuration (MSE loss, x0 prediction,
        fixed variance): ``forward_process`` noises the data, ``training_losses`` noises that result AGAIN with the same
        noise (:877) and regresses the backbone output on the once-noised data (START_X target, :922);
        ``mean_flat(.).mean()`` over equally sized samples is the plain mean."""
        data, labels = self._parse_batch(batch)
        self.data_shape = data.shape
        self.data_dtype = data.dtype
        t = self.random_timesteps(data.size(0)).to(data.device)
        x_data, noise = self.forward_process(data, t)
        x_t = self.q_sample(x_data, t, noise=noise)
        self._tick_error_poll()
        if labels is not None:
            out = self.backbone(x_t, t, labels)
        else:
            out = self.backbone(x_t, t)
        from ..autograd import mse_loss
        loss = mse_loss(out, x_data)
        self.log("train_loss", loss, prog_bar=True)
        return loss
