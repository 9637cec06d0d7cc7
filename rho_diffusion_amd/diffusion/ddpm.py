"""DDPM pipeline (reference: rho_diffusion/diffusion/ddpm.py:46-371), same constructor and methods:
``forward_process`` (q_sample), ``reverse_process`` (ancestral p_sample loop), ``training_step``,
``p_sample`` / ``generate``, ``save_model_weights``.

The arithmetic runs in the HIP kernels of librho_hip.so:
  * q_sample       -> rho_q_sample (alpha_bar gathered per batch element on the device)
  * noise          -> rho_philox_normal (counter-based, reproducible per (seed, offset))
  * backbone       -> UNet engine (channels-last conv / GroupNorm / attention kernels)
  * reverse update -> rho_p_sample_step; the step index lives on the device (rho_step_advance), the
                      loop issues no host synchronisation and no per-step H2D copies
  * loss           -> rho_mse
Reference quirks are kept deliberately: 0.8*sqrt(beta) noise scale, z = 0 for t <= 1,
no update at t = 0 although the backbone is evaluated, clamp to [-1, 1] after every update,
checkpoint spacing T // 10.
"""
from __future__ import annotations

import os
from typing import Any, Iterable, Mapping, Union

import torch
from torch import Tensor, nn

from .. import hip
from ..engine import ops
from ..registry import registry
from ..utils import sample_from_discrete_parameter_space, save_model_checkpoint
from .abstract_diffusion import AbstractDiffusionPipeline

__all__ = ["DDPM"]


class DDPM(AbstractDiffusionPipeline):
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
        # counter-based RNG state: one stream per rank (seed + rank), offset advances per draw
        self.noise_seed = int(os.environ.get("RHO_SEED", "777")) + int(os.environ.get("RANK", "0"))
        self._noise_offset = 0
        self._nan_flag = None
        self.nan_check_every = 50
        self._steps_seen = 0
        # reverse_process: capture one denoising step (Philox draw, UNet forward, update, device-side step advance) in a
        # HIP graph and replay it; the step index and RNG offset live on the device, so one graph serves every t.  Pays
        # where the step is launch-bound (2-D 64^2: ~190 launches of a few microseconds each); bit-identical to the
        # eager loop.  Falls back to the eager loop if capture is not possible (e.g. label lookups that synchronise).
        self.hip_graph_sampling = os.environ.get("RHO_HIP_GRAPH", "1") != "0"
        # training_step: draw t on the device (rho_randint, Philox stream of this rank) instead of the reference's CPU
        # torch.randint + H2D copy (abstract_diffusion.py:163-169, ddpm.py:264).  Off by default: the CPU draw follows
        # torch.manual_seed like the reference; DPTrainer / bench.py switch it on.
        self.device_timesteps = os.environ.get("RHO_DEVICE_TIMESTEPS", "0") == "1"

    # ------------------------------------------------------------------ noise
    def noise(self, data: Tensor) -> Tensor:
        """Standard normal with the shape of ``data`` (ddpm.py:101-102), Philox4x32-10 on the device."""
        hip.require_gpu(data, "data")
        out = torch.empty(data.shape, dtype=torch.float32, device=data.device)
        ops.philox_normal(out, self.noise_seed, self._noise_offset)
        self._noise_offset += (out.numel() + 3) // 4
        return out

    # ------------------------------------------------------------------ q_sample
    def forward_process(self, data: Tensor, t: Union[Tensor, None] = None) -> list:
        """ddpm.py:104-130: returns [x_t, noise]."""
        hip.require_gpu(data, "data")
        batch_size = data.size(0)
        self.schedule.dtype = data.dtype
        if t is None:
            t = self._draw_timesteps(batch_size, data.device)
        n_rows = len(self.schedule["alpha_bar_t"])
        if not t.is_cuda and t.numel() and (int(t.max()) >= n_rows or int(t.min()) < -n_rows):
            # same failure as the reference's table gather (abstract_diffusion.py:216-220) when `timesteps` exceeds the
            # schedule length; timesteps already on the device are checked by the kernel instead (flag bit 2, see _check_nan)
            raise IndexError(f"index {int(t.max())} is out of bounds for dimension 0 with size {n_rows}")
        t = t.reshape(-1).to(device=data.device, dtype=torch.int64).contiguous()
        x0 = data.float().contiguous()
        noise = self.noise(x0)
        tables = self.schedule.device_tables(data.device)
        if self._nan_flag is None or self._nan_flag.device != data.device:
            self._nan_flag = torch.zeros(1, dtype=torch.int32, device=data.device)
        x_t = ops.q_sample(x0, noise.contiguous(), t, tables["alpha_bar"], nan_flag=self._nan_flag)
        return [x_t.type(data.dtype), noise.type(data.dtype)]

    # ------------------------------------------------------------------ p_sample loop
    @torch.no_grad()  # (reference: inference_mode; no_grad keeps the plan buffers usable in training too)
    def reverse_process(self, x_T: Tensor, conditions=None, t_checkpoints=None) -> dict:
        """ddpm.py:132-229.  ``x_T`` is only a shape/device template (:171)."""
        hip.require_gpu(x_T, "x_T")
        dev = x_T.device
        batch_size = x_T.size(0)
        self.schedule.dtype = x_T.dtype
        num_checkpoints = len(t_checkpoints) if t_checkpoints is not None else 0
        buf = None
        if t_checkpoints is not None:
            buf = torch.zeros((batch_size, num_checkpoints) + tuple(x_T.shape[1:]), dtype=torch.float32, device=dev)

        denoise_steps = len(self.schedule["alpha_bar_t"])
        steps_per_ckpt = denoise_steps // 10
        tables = self.schedule.device_tables(dev)

        x_t = self.noise(x_T).contiguous()

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
        if (self.hip_graph_sampling and engine is not None and denoise_steps > 2 and "noise" not in self.__dict__
                and type(self).noise is DDPM.noise):
            done = self._reverse_process_graph(x_t, cc, engine, tables, t_dev, denoise_steps, buf, steps_per_ckpt, num_checkpoints)
            if done:
                self._check_backbone_errors()
                return {"buffer": buf, "denoised": x_t}
            t_dev.fill_(denoise_steps - 1)
        for t in range(denoise_steps - 1, -1, -1):
            z = self.noise(x_t) if t > 1 else None          # drawn before the backbone call (:196-199)
            if engine is not None:
                pred = engine.forward(x_t, None, cc, t_scalar_dev=t_dev)
            else:
                pred = self.backbone(x_t, torch.full((batch_size,), t, device=dev, dtype=torch.long), cc)
            if t > 0:
                ops.p_sample_step(x_t, pred.contiguous(), z, tables["coef"], t_dev)
            if buf is not None and t % steps_per_ckpt == 0 and t_idx < num_checkpoints:
                buf[:, t_idx].copy_(x_t)
                t_idx += 1
            ops.step_advance(t_dev, None, 0)
        self._check_backbone_errors()
        return {"buffer": buf, "denoised": x_t}

    def _reverse_process_graph(self, x_t, cc, engine, tables, t_dev, denoise_steps, buf, steps_per_ckpt, num_checkpoints) -> bool:
        """The loop of reverse_process with steps t = T-2 .. 0 replayed from one captured HIP graph (step T-1 runs eagerly:
        it builds the engine plan and its buffers).  Same arithmetic and the same Philox stream as the eager loop: z is
        drawn at the offset the eager loop would use (the draws of t <= 1 are ignored by the update kernel either way).
        Returns False (nothing modified but the RNG offset bookkeeping) if the capture fails."""
        dev = x_t.device
        n_elem = x_t.numel()
        delta = (n_elem + 3) // 4
        z = torch.empty_like(x_t)
        off_dev = torch.full((1,), self._noise_offset, dtype=torch.int64, device=dev)
        x_save = x_t.clone()

        def step():
            ops.philox_normal(z, self.noise_seed, 0, offset_dev=off_dev)
            pred = engine.forward(x_t, None, cc, t_scalar_dev=t_dev)
            ops.p_sample_step(x_t, pred, z, tables["coef"], t_dev)
            ops.step_advance(t_dev, off_dev, delta)

        t_idx = 0

        def checkpoint(t):
            nonlocal t_idx
            if buf is not None and t % steps_per_ckpt == 0 and t_idx < num_checkpoints:
                buf[:, t_idx].copy_(x_t)
                t_idx += 1

        try:
            step()                                      # t = T-1, eager
            checkpoint(denoise_steps - 1)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) may call HIP
                step()
        except Exception as exc:  # noqa: BLE001  (capture is an optimisation; any failure -> eager loop)
            torch.cuda.synchronize()
            x_t.copy_(x_save)
            if buf is not None:
                buf.zero_()
            self.hip_graph_sampling = False
            import warnings
            warnings.warn(f"HIP graph capture of the sampling step failed ({type(exc).__name__}: {exc}); using the eager loop")
            return False
        for t in range(denoise_steps - 2, -1, -1):
            graph.replay()
            checkpoint(t)
        self._noise_offset += delta * max(denoise_steps - 2, 0)      # eager-loop bookkeeping: one draw per t > 1
        return True

    # ------------------------------------------------------------------ training
    def _draw_timesteps(self, batch_size: int, device) -> Tensor:
        """random_timesteps on the device when ``device_timesteps`` is set (no H2D copy), else the reference's CPU draw."""
        if self.device_timesteps and torch.device(device).type == "cuda" and "random_timesteps" not in self.__dict__:
            t = ops.randint(batch_size, int(self.timesteps), self.noise_seed ^ 0x5DEECE66D, self._noise_offset, device=device)
            self._noise_offset += (batch_size + 3) // 4
            return t
        return self.random_timesteps(batch_size).to(device)

    def _check_nan(self, force: bool = False) -> None:
        """Device-side form of the per-step host check at ddpm.py:268-272: the flag is set by the
        q_sample kernel and polled every ``nan_check_every`` steps instead of syncing each step.
        Bit 0: NaN in the noised data; bit 2: a timestep outside the schedule table (IndexError in the reference)."""
        self._steps_seen += 1
        due = force or self._steps_seen % self.nan_check_every == 0
        if due:
            self._check_backbone_errors()      # labels of the PREVIOUS steps (the flag is sticky until polled)
        if self._nan_flag is not None and due:
            v = int(self._nan_flag.item())
            if v & 4:
                self._nan_flag.zero_()
                raise IndexError(f"a timestep is out of bounds for the schedule tables of size {len(self.schedule['alpha_bar_t'])}")
            if v & 1:
                print("Error: Noised data contains NaNs. Check your noise scheduler.")
                import sys
                sys.exit(0)

    def training_step(self, batch: Iterable[Any], batch_idx: int = 0):
        """ddpm.py:231-288: eps-prediction objective."""
        if isinstance(batch, list):
            data, labels = batch
        elif isinstance(batch, dict):
            data = batch.get("data")
            labels = batch.get("label")
        else:
            data = batch
            labels = None
        self.data_shape = data.shape
        self.data_dtype = data.dtype
        batch_size = data.size(0)
        t = self._draw_timesteps(batch_size, data.device)
        x_data, noise = self.forward_process(data, t)
        self._check_nan()
        if labels is not None:
            pred_noise = self.backbone(x_data, t, labels)
        else:
            pred_noise = self.backbone(x_data, t)
        if isinstance(self.loss_func, nn.MSELoss) and self.loss_func.reduction == "mean":
            from ..autograd import mse_loss
            loss = mse_loss(pred_noise, noise)           # HIP reduction + gradient (rho_mse)
        else:
            loss = self.loss_func(pred_noise, noise)
        self.log("train_loss", loss, prog_bar=True)
        return loss

    def forward(self, batch):
        return self.backbone(batch)

    def on_train_epoch_end(self) -> None:
        if (self.current_epoch > 0 and self.sample_every_n_epochs > 0
                and self.current_epoch % self.sample_every_n_epochs == 0):
            self.eval()
            self.generate()
        if (self.current_epoch > 0 and self.save_weights_every_n_epochs > 0
                and self.current_epoch % self.save_weights_every_n_epochs == 0):
            self.eval()
            self.save_model_weights()

    def p_sample(self, parameter_space, random=False):
        """ddpm.py:319-355."""
        if hasattr(self, "data_shape"):
            shape = [int(x) for x in self.data_shape]
            shape[0] = self.sampling_batch_size
        else:
            shape = [self.sampling_batch_size, self.backbone_kwargs["out_channels"]] + list(self.backbone_kwargs["data_shape"])
            self.data_dtype = torch.float32
        sample_data = torch.zeros(shape, dtype=self.data_dtype, device=self.device)
        cond = None
        if parameter_space is not None:
            cond = sample_from_discrete_parameter_space(parameter_space, sample_data.shape[0], random=random, device=self.device)
        results = self.reverse_process(x_T=sample_data, conditions=cond, t_checkpoints=self.t_checkpoints)
        self.last_samples = results
        return self.make_image_grid(results["denoised"], filename="output_%d.png" % self.current_epoch)

    def generate(self, parameter_space=None, random=False):
        if parameter_space is None:
            parameter_space = self.sample_parameter_space
        return self.p_sample(parameter_space=parameter_space, random=random)

    def save_model_weights(self):
        print("saving model checkpoints...")
        save_model_checkpoint(self.backbone, "model.pth")

    def validation_step(self, batch, batch_idx: int = 0):
        return 0
