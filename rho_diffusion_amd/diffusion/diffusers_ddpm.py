"""The scheduler-driven pipeline of ``scripts/training.py:85-126`` (reference: rho_diffusion/diffusion/diffusers.py:36-227):
``DiffusersDDPMPipeline(backbone, backbone_kwargs, schedule=<DDPMScheduler>, ...)`` whose ``forward_process`` calls
``schedule.add_noise`` and whose ``reverse_process`` calls ``schedule.step(eps_hat, t, x_t)["prev_sample"]`` per step.

The reference takes the scheduler from the third-party ``diffusers`` package (unpinned, not installable here), so its
arithmetic is restated from the published ``DDPMScheduler`` (PARITY UNPINNED, see oracle/ref_torch.py ``dds_*``) for the
options the reference sets: ``beta_schedule`` linear / squaredcos_cap_v2, ``rescale_betas_zero_snr``, ``prediction_type``
epsilon / sample, ``variance_type`` fixed_small / fixed_large, ``clip_sample`` + ``clip_sample_range``.  ``DDPMScheduler``
here offers that constructor, ``config``, ``timesteps``, ``add_noise`` and ``step``; both run in librho_hip.so
(rho_q_sample_coef, rho_ddpm_sched_step).  ``training_step`` is the epsilon-MSE objective of diffusers.py:70-144 on the
same engine forward / backward as ``DDPM``.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Any, Mapping, Optional, Union

import numpy as np
import torch
from torch import Tensor

from .. import hip
from ..engine import ops
from ..registry import registry
from .abstract_diffusion import AbstractDiffusionPipeline

__all__ = ["DDPMScheduler", "DiffusersDDPMPipeline"]


class DDPMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02,
                 beta_schedule: str = "linear", variance_type: str = "fixed_small", clip_sample: bool = True,
                 prediction_type: str = "epsilon", clip_sample_range: float = 1.0, rescale_betas_zero_snr: bool = False):
        if prediction_type not in ("epsilon", "sample"):
            raise NotImplementedError(f"prediction_type {prediction_type}")
        if variance_type not in ("fixed_small", "fixed_large"):
            raise NotImplementedError(f"variance_type {variance_type}")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, variance_type=variance_type, clip_sample=clip_sample,
                                      prediction_type=prediction_type, clip_sample_range=clip_sample_range,
                                      rescale_betas_zero_snr=rescale_betas_zero_snr)
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "squaredcos_cap_v2":
            ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
            betas = torch.tensor([min(1 - ab((i + 1) / num_train_timesteps) / ab(i / num_train_timesteps), 0.999)
                                  for i in range(num_train_timesteps)], dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule}")
        if rescale_betas_zero_snr:
            alphas = 1.0 - betas
            abs_ = torch.cumprod(alphas, dim=0).sqrt()
            a0, aT = abs_[0].clone(), abs_[-1].clone()
            abs_ = (abs_ - aT) * (a0 / (a0 - aT))
            abar = abs_ ** 2
            alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
            betas = 1 - alphas
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())
        self._dev = {}
        self.noise_seed = int(os.environ.get("RHO_SEED", "777")) + int(os.environ.get("RANK", "0"))
        self._noise_offset = 0

    def __len__(self):
        return self.config.num_train_timesteps

    # ------------------------------------------------------------------ q(x_t | x_0)
    def add_noise(self, original_samples: Tensor, noise: Tensor, timesteps: Tensor) -> Tensor:
        hip.require_gpu(original_samples, "original_samples")
        dev = original_samples.device
        key = str(dev)
        if key not in self._dev:
            ac = self.alphas_cumprod
            self._dev[key] = ((ac ** 0.5).contiguous().to(dev), ((1 - ac) ** 0.5).contiguous().to(dev))
        ca, cb = self._dev[key]
        t = timesteps.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
        return ops.q_sample_coef(original_samples.float().contiguous(), noise.float().contiguous(), t, ca, cb)

    # ------------------------------------------------------------------ p(x_{t-1} | x_t)
    def step_coefficients(self, t: int):
        """float32 scalars of DDPMScheduler.step (0-dim float32 tensor arithmetic, as the published code evaluates it)."""
        ac = self.alphas_cumprod
        alpha_prod_t = ac[t]
        alpha_prod_t_prev = ac[t - 1] if t - 1 >= 0 else self.one
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        current_alpha_t = alpha_prod_t / alpha_prod_t_prev
        current_beta_t = 1 - current_alpha_t
        c0 = (alpha_prod_t_prev ** 0.5 * current_beta_t) / beta_prod_t
        c1 = current_alpha_t ** 0.5 * beta_prod_t_prev / beta_prod_t
        sigma = torch.tensor(0.0)
        if t > 0:
            var = torch.clamp((1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * current_beta_t, min=1e-20)
            if self.config.variance_type == "fixed_large":
                var = current_beta_t
            sigma = var ** 0.5
        return (float(beta_prod_t ** 0.5), float(alpha_prod_t ** 0.5), float(c0), float(c1), float(sigma))

    def step(self, model_output: Tensor, timestep: int, sample: Tensor, generator=None, noise: Optional[Tensor] = None,
             return_dict: bool = True):
        hip.require_gpu(sample, "sample")
        t = int(timestep)
        sb, sa, c0, c1, sigma = self.step_coefficients(t)
        x = sample.float().contiguous()
        m = model_output.float().contiguous()
        if sigma != 0.0 and noise is None:
            noise = torch.empty_like(x)
            ops.philox_normal(noise, self.noise_seed, self._noise_offset)
            self._noise_offset += (x.numel() + 3) // 4
        prev = torch.empty_like(x)
        x0 = torch.empty_like(x)
        clip = float(self.config.clip_sample_range) if self.config.clip_sample else 0.0
        hip.check(hip.lib().rho_ddpm_sched_step(x.data_ptr(), m.data_ptr(), noise.data_ptr() if noise is not None else None,
                                                prev.data_ptr(), x0.data_ptr(), x.numel(),
                                                1 if self.config.prediction_type == "epsilon" else 0, sb, sa, clip, c0, c1, sigma,
                                                torch.cuda.current_stream().cuda_stream), "rho_ddpm_sched_step")
        if not return_dict:
            return (prev,)
        return {"prev_sample": prev, "pred_original_sample": x0}


class DiffusersDDPMPipeline(AbstractDiffusionPipeline):
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
        self.noise_seed = int(os.environ.get("RHO_SEED", "777")) + int(os.environ.get("RANK", "0"))
        self._noise_offset = 0

    def noise(self, data: Tensor) -> Tensor:
        hip.require_gpu(data, "data")
        out = torch.empty(data.shape, dtype=torch.float32, device=data.device)
        ops.philox_normal(out, self.noise_seed, self._noise_offset)
        self._noise_offset += (out.numel() + 3) // 4
        return out

    def forward_process(self, clean_images: Tensor, t: Union[Tensor, None] = None):
        """diffusers.py:146-149: returns (noisy_images, noise)."""
        if t is None:
            t = self.random_timesteps(clean_images.size(0))
        noise = self.noise(clean_images)
        return self.schedule.add_noise(clean_images, noise, t), noise

    @torch.no_grad()
    def reverse_process(self, x_T: Tensor, conditions=None, t_checkpoints=None) -> dict:
        """diffusers.py:152-227."""
        hip.require_gpu(x_T, "x_T")
        dev = x_T.device
        batch_size = x_T.size(0)
        denoise_steps = int(max(self.schedule.timesteps).item())        # (sic) T - 1 steps: t = T-2 .. 0, as the reference
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
        t_idx = 0
        for t in range(denoise_steps - 1, -1, -1):
            if engine is not None:
                out = engine.forward(x_t, None, cc, t_scalar_dev=t_dev)
            else:
                out = self.backbone(x_t, torch.full((batch_size,), t, device=dev, dtype=torch.long), cc)
            noise = self.noise(x_t) if t > 0 else None
            x_t = self.schedule.step(out, t, x_t, noise=noise)["prev_sample"]
            if buf is not None and t % steps_per_ckpt == 0 and t_idx < num_checkpoints:
                buf[:, t_idx].copy_(x_t)
                t_idx += 1
            ops.step_advance(t_dev, None, 0)
        self._check_backbone_errors()
        return {"buffer": buf, "denoised": x_t}

    def training_step(self, batch, batch_idx: int = 0):
        """diffusers.py:70-144: one ``add_noise``; MSE against the noise (``prediction_type`` epsilon) or, as the
        reference writes it (:121-122), against the NOISY images for prediction_type 'sample'.  The reference's
        ``clip_grad_norm_`` (:128) sits before ``backward`` and so acts on the freshly zeroed gradients of the step:
        it has no effect on the update and is not reproduced."""
        data, labels = self._parse_batch(batch)
        self.data_shape = data.shape
        self.data_dtype = data.dtype
        t = self.random_timesteps(data.size(0)).to(data.device)
        noisy_images, noise = self.forward_process(data, t)
        self._tick_error_poll()
        noise_pred = self.backbone(noisy_images, t, labels) if labels is not None else self.backbone(noisy_images, t)
        ptype = self.schedule.config.prediction_type
        if ptype == "epsilon":
            loss = self._loss(noise_pred, noise)
        elif ptype == "sample":
            loss = self._loss(noise_pred, noisy_images)
        else:
            raise Exception("Loss cannot be computed because the prediction type is not understood.")
        self.log("train_loss", loss, prog_bar=True)
        return loss
