"""fp32 CPU restatement of the rho-diffusion DDPM + UNetv2 hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Functional style: weights come
from a ``state_dict`` that uses the reference's key names, the
network structure is re-derived from the constructor kwargs.  All citations are
relative to ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- schedules
def _sigma(alpha_bar64: Tensor, alpha_bar32: Tensor, beta64: Tensor) -> Tensor:
    """sigma_t = sqrt((1 - abar_{t-1}) / (1 - abar_t) * beta_t), abar_{-1} = 1.

    rho_diffusion/diffusion/schedule.py:79-82 (offset_alpha_bar_t is built from the
    *stored fp32* alpha_bar, padded with 1.0 on the left) and :164-168 / :211-214.
    """
    offset = F.pad(alpha_bar32[:-1], (1, 0), value=1.0)
    return torch.sqrt((1 - offset) / (1 - alpha_bar64) * beta64).to(torch.float32)


def linear_schedule(num_steps: int, beta_1: float = 1.0e-3, beta_T: float = 0.02) -> Dict[str, Tensor]:
    """rho_diffusion/diffusion/schedule.py:141-168 (LinearSchedule.__init__)."""
    scale = 1000 / num_steps
    beta = torch.linspace(scale * beta_1, scale * beta_T, num_steps, dtype=torch.float64)
    alpha = 1.0 - beta
    alpha_bar = alpha.cumprod(0)
    abar32 = alpha_bar.to(torch.float32)
    return {
        "beta_t": beta.to(torch.float32),
        "alpha_t": alpha.to(torch.float32),
        "alpha_bar_t": abar32,
        "sigma_t": _sigma(alpha_bar, abar32, beta),
    }


def cosine_schedule(num_steps: int, offset: float = 0.008) -> Dict[str, Tensor]:
    """rho_diffusion/diffusion/schedule.py:171-214 (CosineBetaSchedule.__init__).

    Quirks kept: T+1 entries; beta is computed from the *unclamped*
    fp64 alpha_bar divided by the fp32 offset alpha_bar (so beta[0] = 1 - abar0/1 -> clip);
    ``clip_`` is in-place so alpha_t = 1 - clipped beta; sigma[0] = sqrt(0/0*..) = NaN.
    """
    t = torch.linspace(0.0, num_steps, num_steps + 1, dtype=torch.float64) / num_steps
    alpha_bar = torch.cos((t + offset) / (1 + offset) * math.pi * 0.5).pow(2.0)
    alpha_bar = alpha_bar.div(alpha_bar[0])
    abar32 = alpha_bar.to(torch.float32)
    abar32[abar32 < 0] = 0
    abar32[abar32 > 1] = 1
    off32 = F.pad(abar32[:-1], (1, 0), value=1.0)
    beta = 1 - (alpha_bar / off32)
    beta = beta.clip_(0.0001, 0.9999)
    return {
        "alpha_bar_t": abar32,
        "beta_t": beta.to(torch.float32),
        "alpha_t": (1 - beta).to(torch.float32),
        "sigma_t": torch.sqrt((1 - off32) / (1 - alpha_bar) * beta).to(torch.float32),
    }


# --------------------------------------------------------------------------- embeddings
def sinusoidal_embedding(t: Tensor, dim: int, wavelength: int = 10000) -> Tensor:
    """rho_diffusion/models/common.py:27-43: interleaved sin/cos, always fp32."""
    assert dim % 2 == 0
    i = torch.arange(dim // 2)
    omega = torch.pow(wavelength, 2 * i / dim)
    pe = torch.empty(len(t), dim)
    pe[:, 2 * i] = torch.sin(t[:, None] / omega[None, :]).float()
    pe[:, 2 * i + 1] = torch.cos(t[:, None] / omega[None, :]).float()
    return pe


def multi_embeddings(y: Tensor, parameter_space: Dict[str, Sequence[float]], sd: Dict[str, Tensor],
                     prefix: str = "cond_fn.") -> Optional[Tensor]:
    """rho_diffusion/models/conditioning.py:115-139 (MultiEmbeddings.forward):
    per key (dict order) look up the position of y[:, i] in parameter_space[key] by exact
    equality, embed, and sum over keys."""
    emb = None
    for i, key in enumerate(parameter_space.keys()):
        yi = y if y.dim() == 1 else y[:, i]
        space = torch.tensor(parameter_space[key])
        categorical = torch.where(yi[:, None] == space[None, :])[1]
        e = F.embedding(categorical, sd[f"{prefix}embedding_layers.{key}.weight"])
        emb = e if emb is None else emb + e
    return emb


# --------------------------------------------------------------------------- layers
def group_norm32(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """rho_diffusion/layers.py:71-74,122-129: GroupNorm(32, C), eps 1e-5, fp32 math."""
    return F.group_norm(x.float(), 32, w, b, eps=1e-5).type(x.dtype)


def conv_nd(dims: int, x: Tensor, w: Tensor, b: Optional[Tensor], stride=1, padding=0) -> Tensor:
    """rho_diffusion/layers.py:77-88."""
    return {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[dims](x, w, b, stride=stride, padding=padding)


def upsample(dims: int, x: Tensor) -> Tensor:
    """rho_diffusion/models/unet_v2.py:122-131: nearest x2; 3-D keeps depth."""
    if dims == 3:
        return F.interpolate(x, (x.shape[2], x.shape[3] * 2, x.shape[4] * 2), mode="nearest")
    return F.interpolate(x, scale_factor=2, mode="nearest")


def avg_pool(dims: int, x: Tensor) -> Tensor:
    """rho_diffusion/models/unet_v2.py:153,165 (Downsample without conv): avg_pool_nd(kernel = stride = 2; 3-D: (1, 2, 2))."""
    if dims == 3:
        return F.avg_pool3d(x, (1, 2, 2), (1, 2, 2))
    return {1: F.avg_pool1d, 2: F.avg_pool2d}[dims](x, 2, 2)


# the elementwise, parameter-free activations of the reference's registry (rho_diffusion/registry.py:162-170; the `activation`
# argument of UNet / ResBlock, unet_v2.py:201,493,518-519), as torch evaluates the nn modules of the same names with default arguments
ACTIVATIONS = {"SiLU": F.silu, "ReLU": F.relu, "GELU": F.gelu, "Tanh": torch.tanh, "Sigmoid": torch.sigmoid, "ELU": F.elu}


def _act(cfg_or_name) -> "callable":
    name = cfg_or_name if isinstance(cfg_or_name, str) else (cfg_or_name or {}).get("activation", "SiLU")
    if not isinstance(name, str):
        name = type(name).__name__
    return ACTIVATIONS[name]


def resblock(dims: int, x: Tensor, emb: Tensor, sd: Dict[str, Tensor], p: str,
             use_scale_shift_norm: bool, updown: Optional[str] = None, act=F.silu, drop=None) -> Tensor:
    """rho_diffusion/models/unet_v2.py:273-293 (ResBlock._forward, dropout 0).  updown = "up" / "down": the activated input and
    the skip input are resampled before the first conv (:277-281; Upsample / Downsample without conv, :221-224).  ``act``: the
    activation the block was built with (:201,214,230,238).  ``drop`` = (mask [N, C, *spatial] of 0 / 1, p): nn.Dropout(p) of
    out_layers in training mode (:239) with the mask INJECTED - h <- act(h) * mask / (1 - p) - as the tests inject noise tapes."""
    dm = (lambda v: v) if drop is None else (lambda v: v * drop[0].to(v.dtype) / (1.0 - drop[1]))
    h = act(group_norm32(x, sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"]))
    if updown is not None:
        rs = (lambda v: upsample(dims, v)) if updown == "up" else (lambda v: avg_pool(dims, v))
        h, x = rs(h), rs(x)
    h = conv_nd(dims, h, sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"], padding=1)
    emb_out = F.linear(act(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"]).type(h.dtype)
    while emb_out.dim() < h.dim():
        emb_out = emb_out[..., None]
    if use_scale_shift_norm:
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = group_norm32(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"]) * (1 + scale) + shift
        h = conv_nd(dims, dm(act(h)), sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    else:
        h = h + emb_out
        h = group_norm32(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"])
        h = conv_nd(dims, dm(act(h)), sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    if (p + "skip_connection.weight") in sd:
        w = sd[p + "skip_connection.weight"]
        pad = 1 if w.shape[-1] == 3 else 0
        x = conv_nd(dims, x, w, sd[p + "skip_connection.bias"], padding=pad)
    return x + h


def qkv_attention(qkv: Tensor, n_heads: int, new_order: bool) -> Tensor:
    """rho_diffusion/models/unet_v2.py:374-393 (legacy) / 409-432 (new order)."""
    bs, width, length = qkv.shape
    assert width % (3 * n_heads) == 0
    ch = width // (3 * n_heads)
    scale = 1 / math.sqrt(math.sqrt(ch))
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q = (q * scale).reshape(bs * n_heads, ch, length)
        k = (k * scale).reshape(bs * n_heads, ch, length)
        v = v.reshape(bs * n_heads, ch, length)
    else:
        q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
        q, k = q * scale, k * scale
    weight = torch.einsum("bct,bcs->bts", q, k)
    weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
    a = torch.einsum("bts,bcs->bct", weight, v)
    return a.reshape(bs, -1, length)


def attention_block(x: Tensor, sd: Dict[str, Tensor], p: str, n_heads: int, new_order: bool) -> Tensor:
    """rho_diffusion/models/unet_v2.py:336-342 (AttentionBlock._forward)."""
    b, c, *spatial = x.shape
    x = x.reshape(b, c, -1)
    qkv = F.conv1d(group_norm32(x, sd[p + "norm.weight"], sd[p + "norm.bias"]), sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    h = qkv_attention(qkv, n_heads, new_order)
    h = F.conv1d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return (x + h).reshape(b, c, *spatial)


# --------------------------------------------------------------------------- UNetv2
def unet_structure(cfg: dict) -> dict:
    """Replays the constructor loops of rho_diffusion/models/unet_v2.py:533-683 and returns,
    per ``input_blocks`` / ``middle_block`` / ``output_blocks`` entry, the list of layer kinds
    with their ``state_dict`` prefixes (incl. resblock_updown = True and conv_resample = False, :575-590, :655-675)."""
    mc = cfg["model_channels"]
    mult = tuple(cfg.get("channel_mult", (1, 2, 4, 8)))
    nres = cfg["num_res_blocks"]
    attn_res = list(cfg.get("attention_resolutions", [16, 8]))
    heads = cfg.get("num_heads", 1)
    nhc = cfg.get("num_head_channels", -1)
    heads_up = cfg.get("num_heads_upsample", -1)
    if heads_up == -1:
        heads_up = heads
    conv_resample = bool(cfg.get("conv_resample", True))
    updown = bool(cfg.get("resblock_updown", False))

    def nh(c, h):
        return h if nhc == -1 else c // nhc

    inp: List[List[tuple]] = [[("conv", "input_blocks.0.0.")]]
    ch = int(mult[0] * mc)
    chans = [ch]
    ds = 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            idx = len(inp)
            layers = [("res", f"input_blocks.{idx}.0.", ch, int(m * mc))]
            ch = int(m * mc)
            if ds in attn_res:
                layers.append(("attn", f"input_blocks.{idx}.1.", ch, nh(ch, heads)))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            idx = len(inp)
            if updown:
                inp.append([("res_down", f"input_blocks.{idx}.0.", ch, ch)])
            else:
                inp.append([("down" if conv_resample else "pool", f"input_blocks.{idx}.0.op.", ch)])
            chans.append(ch)
            ds *= 2
    mid = [("res", "middle_block.0.", ch, ch), ("attn", "middle_block.1.", ch, nh(ch, heads)),
           ("res", "middle_block.2.", ch, ch)]
    out: List[List[tuple]] = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            ich = chans.pop()
            idx = len(out)
            layers = [("res", f"output_blocks.{idx}.0.", ch + ich, int(mc * m))]
            ch = int(mc * m)
            if ds in attn_res:
                layers.append(("attn", f"output_blocks.{idx}.{len(layers)}.", ch, nh(ch, heads_up)))
            if level and i == nres:
                if updown:
                    layers.append(("res_up", f"output_blocks.{idx}.{len(layers)}.", ch, ch))
                else:
                    layers.append(("up" if conv_resample else "up_only", f"output_blocks.{idx}.{len(layers)}.conv.", ch))
                ds //= 2
            out.append(layers)
    return {"input": inp, "middle": mid, "output": out, "final_ch": ch}


def _run_layers(dims, layers, h, emb, sd, cfg):
    ssn = bool(cfg.get("use_scale_shift_norm", False))
    new_order = bool(cfg.get("use_new_attention_order", False))
    act = _act(cfg)
    for layer in layers:
        kind, p = layer[0], layer[1]
        if kind == "conv":
            h = conv_nd(dims, h, sd[p + "weight"], sd[p + "bias"], padding=1)
        elif kind == "res":
            h = resblock(dims, h, emb, sd, p, ssn, act=act, drop=(cfg.get("_drop_masks") or {}).get(p))
        elif kind in ("res_up", "res_down"):
            h = resblock(dims, h, emb, sd, p, ssn, updown=kind[4:], act=act, drop=(cfg.get("_drop_masks") or {}).get(p))
        elif kind == "pool":
            h = avg_pool(dims, h)
        elif kind == "up_only":
            h = upsample(dims, h)
        elif kind == "attn":
            h = attention_block(h, sd, p, layer[3], new_order)
        elif kind == "down":
            stride = 2 if dims != 3 else (1, 2, 2)  # unet_v2.py:153
            h = conv_nd(dims, h, sd[p + "weight"], sd[p + "bias"], stride=stride, padding=1)
        elif kind == "up":
            h = conv_nd(dims, upsample(dims, h), sd[p + "weight"], sd[p + "bias"], padding=1)
    return h


def unet_embedding(sd, cfg, timesteps: Tensor, y: Optional[Tensor] = None,
                   parameter_space=None) -> Tensor:
    """rho_diffusion/models/unet_v2.py:694-719."""
    mc = cfg["model_channels"]
    num_classes = cfg.get("num_classes", None)
    assert (y is not None) == (num_classes is not None)
    e = sinusoidal_embedding(timesteps, mc)
    e = F.linear(e, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    e = F.linear(_act(cfg)(e), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if num_classes is not None:
        if y.dim() == 2 and y.shape == e.shape:
            e = e + y
        else:
            e = e + multi_embeddings(y, parameter_space, sd)
    return e


def unet_forward(sd: Dict[str, Tensor], cfg: dict, x: Tensor, timesteps: Tensor,
                 y: Optional[Tensor] = None, parameter_space=None) -> Tensor:
    """rho_diffusion/models/unet_v2.py:685-732 (UNet.forward)."""
    dims = cfg.get("dims", 2)
    st = unet_structure(cfg)
    emb = unet_embedding(sd, cfg, timesteps, y, parameter_space)
    hs = []
    h = x.float()
    for layers in st["input"]:
        h = _run_layers(dims, layers, h, emb, sd, cfg)
        hs.append(h)
    h = _run_layers(dims, st["middle"], h, emb, sd, cfg)
    for layers in st["output"]:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(dims, layers, h, emb, sd, cfg)
    h = group_norm32(h, sd["out.0.weight"], sd["out.0.bias"])
    return conv_nd(dims, _act(cfg)(h), sd["out.2.weight"], sd["out.2.bias"], padding=1)


# --------------------------------------------------------------------------- DDPM
def _bshape(x: Tensor, v: Tensor) -> Tensor:
    """rho_diffusion/diffusion/abstract_diffusion.py:171-192 (reshape_timesteps)."""
    return v.view((-1, *((1,) * (x.dim() - 1))))


def q_sample(x0: Tensor, t: Tensor, noise: Tensor, alpha_bar: Tensor) -> Tensor:
    """rho_diffusion/diffusion/ddpm.py:104-130 (forward_process) with the noise injected:
    x_t = sqrt(abar[t]) * x0 + sqrt(1 - abar[t]) * eps (abar cast to the data dtype first)."""
    ab = _bshape(x0, alpha_bar.type(x0.dtype)[t])
    return ab.sqrt() * x0 + (1 - ab).sqrt() * noise


def p_sample_step(x_t: Tensor, pred_noise: Tensor, t: int, sched: Dict[str, Tensor], z: Tensor) -> Tensor:
    """rho_diffusion/diffusion/ddpm.py:210-218: one reverse update (callers skip it at t == 0).
    Noise scale is 0.8*sqrt(beta_t) (q2), result clamped to [-1, 1] (q4)."""
    a, b, ab = (sched[k].type(x_t.dtype)[t] for k in ("alpha_t", "beta_t", "alpha_bar_t"))
    x = (1 / a.sqrt()) * (x_t - (b / (1 - ab).sqrt()) * pred_noise) + 0.8 * torch.sqrt(b) * z
    return torch.clamp(x, -1, 1)


def reverse_process(model, x_init: Tensor, sched: Dict[str, Tensor], z_tape: Sequence[Tensor],
                    conditions=None, num_checkpoints: int = 0):
    """rho_diffusion/diffusion/ddpm.py:132-229 with the RNG replaced by a tape: ``x_init`` is
    the first ``randn_like`` draw (:171) and ``z_tape[i]`` the i-th later draw, taken only when
    t > 1 and *before* the backbone call (:196-199).  ``model(x, tt, cond)`` predicts eps.
    Returns (denoised, buffer-or-None) with the q5 checkpoint rule (T // 10 spacing)."""
    T = len(sched["alpha_bar_t"])
    steps_per_ckpt = T // 10
    x_t = x_init.clone()
    B = x_t.shape[0]
    buf = torch.zeros((B, num_checkpoints) + tuple(x_t.shape[1:])) if num_checkpoints else None
    zi = 0
    t_idx = 0
    for t in range(T - 1, -1, -1):
        if t > 1:
            z = z_tape[zi]
            zi += 1
        else:
            z = torch.zeros_like(x_t)
        tt = torch.full((B,), t, dtype=torch.long)
        pred = model(x_t, tt, conditions)
        if t > 0:
            x_t = p_sample_step(x_t, pred, t, sched, z)
        if buf is not None and t % steps_per_ckpt == 0 and t_idx < num_checkpoints:
            buf[:, t_idx] = x_t
            t_idx += 1
    return x_t, buf


def training_loss(model, x0: Tensor, t: Tensor, noise: Tensor, alpha_bar: Tensor, labels=None) -> Tensor:
    """rho_diffusion/diffusion/ddpm.py:231-288 with (t, eps) injected: MSE(eps_hat, eps)."""
    x_t = q_sample(x0, t, noise, alpha_bar)
    pred = model(x_t, t, labels)
    return F.mse_loss(pred, noise)


def discrete_parameter_rows(param_dict: Dict[str, Sequence[float]], batch_size: int) -> Tensor:
    """rho_diffusion/utils.py:213-220 (sample_from_discrete_parameter_space, random=False): the first
    ``batch_size`` rows of itertools.product over the parameter values - the labels DDPM.p_sample / generate
    (rho_diffusion/diffusion/ddpm.py:319-360) hand to reverse_process."""
    import itertools
    _, values = zip(*param_dict.items())
    combos = torch.tensor([v for v in itertools.product(*values)])
    return combos[torch.arange(0, batch_size)]


# --------------------------------------------------------------------------- GaussianDiffusionPipeline
def gd_betas(schedule_name: str, num_steps: int):
    """rho_diffusion/diffusion/gaussian_diffusion.py:45-89 (get_named_beta_schedule / betas_for_alpha_bar), float64 numpy."""
    import numpy as np
    if schedule_name == "linear":
        scale = 1000 / num_steps
        return np.linspace(scale * 0.0001, scale * 0.02, num_steps, dtype=np.float64)
    if schedule_name == "cosine":
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - ab((i + 1) / num_steps) / ab(i / num_steps), 0.999) for i in range(num_steps)])
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def gd_tables(betas) -> Dict[str, "object"]:
    """The float64 coefficient tables of GaussianDiffusionPipeline.__init__ (gaussian_diffusion.py:237-273)."""
    import numpy as np
    betas = np.array(betas, dtype=np.float64)
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


def _gd_extract(arr, t: Tensor, shape) -> Tensor:
    """_extract_into_tensor (gaussian_diffusion.py:91-105): float64 table -> gathered -> float32 -> broadcast."""
    res = torch.from_numpy(arr)[t].float()
    while res.dim() < len(shape):
        res = res[..., None]
    return res.expand(shape)


def gd_q_sample(tab, x_start: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """gaussian_diffusion.py:294-312."""
    return (_gd_extract(tab["sqrt_alphas_cumprod"], t, x_start.shape) * x_start
            + _gd_extract(tab["sqrt_one_minus_alphas_cumprod"], t, x_start.shape) * noise)


def gd_dynamic_threshold(x: Tensor, percentile: float = 0.9) -> Tensor:
    """process_xstart with clip_denoised (gaussian_diffusion.py:400-415): per-sample quantile of |x| over all other
    axes, floored at 1, clamp to +-s and divide by s."""
    s = torch.quantile(x.reshape(x.shape[0], -1).abs(), percentile, dim=-1)
    s.clamp_(min=1.0)
    s = s.view(-1, *((1,) * (x.dim() - 1)))
    return x.clamp(-s, s) / s


def gd_ddim_step(tab, x: Tensor, t: Tensor, model_output: Tensor, noise: Tensor, eta: float = 0.0):
    """ddim_sample for the pipeline's fixed configuration (x0-prediction, fixed-large variance, clip_denoised=True):
    gaussian_diffusion.py:654-702 on top of p_mean_variance :338-443 and _predict_eps_from_xstart :462-466."""
    pred_xstart = gd_dynamic_threshold(model_output)
    eps = (_gd_extract(tab["sqrt_recip_alphas_cumprod"], t, x.shape) * x - pred_xstart) / _gd_extract(
        tab["sqrt_recipm1_alphas_cumprod"], t, x.shape)
    alpha_bar = _gd_extract(tab["alphas_cumprod"], t, x.shape)
    alpha_bar_prev = _gd_extract(tab["alphas_cumprod_prev"], t, x.shape)
    sigma = eta * torch.sqrt((1 - alpha_bar_prev) / (1 - alpha_bar)) * torch.sqrt(1 - alpha_bar / alpha_bar_prev)
    mean_pred = pred_xstart * torch.sqrt(alpha_bar_prev) + torch.sqrt(1 - alpha_bar_prev - sigma ** 2) * eps
    nonzero_mask = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
    return mean_pred + nonzero_mask * sigma * noise, pred_xstart


def gd_reverse_process(model, tab, x_init: Tensor, noise_tape: Sequence[Tensor], y=None, t_checkpoints=None, eta: float = 0.0):
    """GaussianDiffusionPipeline.reverse_process (gaussian_diffusion.py:1029-1099).  ``x_init`` replaces the
    ``randn_like(x_T)`` start (:1039), ``noise_tape[i]`` the ``randn_like(x)`` of loop iteration i (:687).
    model(x, t[B] long, y) -> x0 prediction."""
    T = len(tab["betas"])
    x_t = x_init
    buf = None
    if t_checkpoints is not None:
        ncp = len(t_checkpoints)
        buf = torch.zeros((x_init.shape[0], ncp) + tuple(x_init.shape[1:]), dtype=x_init.dtype)
        per = T // ncp
    else:
        ncp, per = 0, T
    t_idx = 0
    for i, t in enumerate(range(T - 1, -1, -1)):
        tt = torch.full((x_init.shape[0],), t, dtype=torch.long)
        out = model(x_t, tt, y)
        x_t, _ = gd_ddim_step(tab, x_t, tt, out, noise_tape[i], eta)
        if buf is not None and t % per == 0 and t_idx < ncp:
            buf[:, t_idx] = x_t
            t_idx += 1
    return {"buffer": buf, "denoised": x_t}


def gd_training_loss(model, tab, data: Tensor, t: Tensor, noise: Tensor, y=None, predict_xstart: bool = True) -> Tensor:
    """GaussianDiffusionPipeline.training_step (gaussian_diffusion.py:1153-1210): ``forward_process`` noises the data (:1014-1027)
    and ``training_losses`` (:861-934) noises that result AGAIN with the same noise (:877) before the backbone sees it; with
    the class defaults (predict_xstart, :218-219) the MSE target is the once-noised data (:922), otherwise the noise (:923);
    ``mean_flat(...).mean()`` of equally sized samples = the plain mean."""
    x_data = gd_q_sample(tab, data, t, noise)
    x_t = gd_q_sample(tab, x_data, t, noise)
    out = model(x_t, t, y)
    target = x_data if predict_xstart else noise
    return ((target - out) ** 2).mean()


def dds_training_loss(model, sched, data: Tensor, t: Tensor, noise: Tensor, y=None, prediction_type: str = "epsilon") -> Tensor:
    """DiffusersDDPMPipeline.training_step (diffusers.py:70-144; PARITY UNPINNED through ``dds_add_noise``): one ``add_noise``,
    MSE against the noise (epsilon) or (sic, :121-122) against the NOISY images for prediction_type 'sample'.  The
    ``clip_grad_norm_`` of :128 runs before ``backward`` on freshly zeroed gradients: no effect on the update."""
    noisy = dds_add_noise(sched, data, noise, t)
    out = model(noisy, t, y)
    target = noise if prediction_type == "epsilon" else noisy
    return ((out - target) ** 2).mean()


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1=0.9, beta2=0.999,
               eps=1e-8, weight_decay=1e-2):
    """torch.optim.AdamW single-tensor update (abstract_diffusion.py:103-119 builds it with
    PyTorch defaults): decoupled decay, bias-corrected moments.  Returns (p, m, v)."""
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# --------------------------------------------------------------------------- inputs
# --------------------------------------------------------------------------- diffusers.DDPMScheduler subset
# PARITY UNPINNED: the arithmetic lives in the third-party `diffusers` package, which the reference leaves unpinned
# (pyproject.toml) and which is not installed here; this restates the published DDPMScheduler algorithm
# (scheduling_ddpm.py: betas_for_alpha_bar / rescale_zero_terminal_snr / add_noise / step / _get_variance) for the
# configuration scripts/training.py:85-95 builds and the call sites rho_diffusion/diffusion/diffusers.py:146-227.
def dds_tables(num_train_timesteps: int = 1000, beta_schedule: str = "squaredcos_cap_v2", rescale_betas_zero_snr: bool = False,
               beta_start: float = 1e-4, beta_end: float = 0.02) -> Dict[str, Tensor]:
    if beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    elif beta_schedule == "squaredcos_cap_v2":
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        betas = torch.tensor([min(1 - ab((i + 1) / num_train_timesteps) / ab(i / num_train_timesteps), 0.999)
                              for i in range(num_train_timesteps)], dtype=torch.float32)
    else:
        raise NotImplementedError(beta_schedule)
    if rescale_betas_zero_snr:                     # rescale_zero_terminal_snr
        alphas = 1.0 - betas
        abs_ = torch.cumprod(alphas, dim=0).sqrt()
        a0, aT = abs_[0].clone(), abs_[-1].clone()
        abs_ = (abs_ - aT) * (a0 / (a0 - aT))
        abar = abs_ ** 2
        alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
        betas = 1 - alphas
    alphas = 1.0 - betas
    return {"betas": betas, "alphas": alphas, "alphas_cumprod": torch.cumprod(alphas, dim=0)}


def dds_add_noise(tab, x0: Tensor, noise: Tensor, t: Tensor) -> Tensor:
    ac = tab["alphas_cumprod"]
    a = (ac[t] ** 0.5).flatten()
    b = ((1 - ac[t]) ** 0.5).flatten()
    while a.dim() < x0.dim():
        a, b = a.unsqueeze(-1), b.unsqueeze(-1)
    return a * x0 + b * noise


def dds_step(tab, model_output: Tensor, t: int, sample: Tensor, noise: Tensor, prediction_type: str = "epsilon",
             variance_type: str = "fixed_large", clip_sample: bool = True, clip_sample_range: float = 1.0):
    """DDPMScheduler.step for num_inference_steps = None (prev_t = t - 1); returns (prev_sample, pred_original_sample)."""
    ac = tab["alphas_cumprod"]
    one = torch.tensor(1.0)
    alpha_prod_t = ac[t]
    alpha_prod_t_prev = ac[t - 1] if t - 1 >= 0 else one
    beta_prod_t = 1 - alpha_prod_t
    beta_prod_t_prev = 1 - alpha_prod_t_prev
    current_alpha_t = alpha_prod_t / alpha_prod_t_prev
    current_beta_t = 1 - current_alpha_t
    if prediction_type == "epsilon":
        x0 = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
    elif prediction_type == "sample":
        x0 = model_output
    else:
        raise NotImplementedError(prediction_type)
    if clip_sample:
        x0 = x0.clamp(-clip_sample_range, clip_sample_range)
    c0 = (alpha_prod_t_prev ** 0.5 * current_beta_t) / beta_prod_t
    c1 = current_alpha_t ** 0.5 * beta_prod_t_prev / beta_prod_t
    prev = c0 * x0 + c1 * sample
    if t > 0:
        var = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * current_beta_t
        var = torch.clamp(var, min=1e-20)
        if variance_type == "fixed_large":
            var = current_beta_t
        elif variance_type != "fixed_small":
            raise NotImplementedError(variance_type)
        prev = prev + (var ** 0.5) * noise
    return prev, x0


def ema_update(shadow: Tensor, param: Tensor, step_id: int, decay: float = 0.9999) -> float:
    """ExponentialMovingAverage.update for one tensor, in place (rho_diffusion/ema.py:41-60); returns the fraction."""
    frac = decay * (1 - math.exp(-step_id / 2000))
    shadow.sub_((1.0 - frac) * (shadow - param))
    return frac


def spherical_harmonic_field(l: int, m: int, grid: int, dims: int = 3):
    """Density field of rho_diffusion/data/synthetic.py:45-124 on linspace(-2, 2, G)^3:
    |Y_l^{|m|}(theta, phi) * r| min-max normalised; 2-D = central z slice (the reference
    has no 2-D generator; choice stated in SURVEY 8d).  Returns float32 [1, G, G(, G)]."""
    import numpy as np
    try:
        from scipy.special import sph_harm_y

        def sph(mm, ll, th, ph):  # scipy>=1.15: sph_harm_y(n, m, polar, azimuth)
            return sph_harm_y(ll, mm, ph, th)
    except ImportError:  # pragma: no cover
        from scipy.special import sph_harm as sph
    ax = np.linspace(-2, 2, grid)
    xg, yg, zg = np.meshgrid(ax, ax, ax, indexing="xy")
    with np.errstate(divide="ignore", invalid="ignore"):
        theta = np.arctan(np.sqrt(xg ** 2 + yg ** 2) / zg)
        phi = np.arctan(yg / xg)
    radial = np.sqrt(xg ** 2 + yg ** 2 + zg ** 2)
    # reference: sph_harm(|m|, l, theta, phi) with scipy's legacy (m, n, azimuth, polar) order
    sol = sph(abs(m), l, theta, phi) * radial
    sol = (sol - sol.min()) / (sol.max() - sol.min())
    out = np.abs(sol).astype(np.float32)
    if dims == 2:
        out = out[:, :, grid // 2]
    return torch.from_numpy(out)[None]


# --------------------------------------------------------------------------- legacy UNet ("UNet v1")
_V1_ACT = {"ReLU": F.relu, "GELU": F.gelu, "SiLU": F.silu, "Identity": lambda v: v}


def unet_v1_block(x: Tensor, time_pe: Tensor, sd: Dict[str, Tensor], p: str, is_up: bool, act, groups: int = 8) -> Tensor:
    """AbstractUNetBlock.forward, rho_diffusion/models/unet.py:117-135 (2-D): two 3x3 convolutions with the activation after each,
    + residual_conv(x) when present, + the time embedding read out per channel, GroupNorm(groups), activation.  'Up' blocks use
    ConvTranspose2d for conv2 / residual_conv (:57-71, :89-96)."""
    pe = F.linear(time_pe, sd[p + "time_embedding_readout.weight"], sd[p + "time_embedding_readout.bias"])
    second = F.conv_transpose2d if is_up else F.conv2d
    h = act(F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=1, padding=1))
    h = act(second(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], stride=1, padding=1))
    if p + "residual_conv.weight" in sd:
        h = h + second(x, sd[p + "residual_conv.weight"], sd[p + "residual_conv.bias"], stride=1, padding=1)
    h = h + pe[(...,) + (None,) * 2]
    if p + "norm.weight" in sd:
        h = F.group_norm(h, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-5)
    return act(h)


def unet_v1_forward(sd: Dict[str, Tensor], cfg: dict, data: Tensor, t: Tensor) -> Tensor:
    """UNet.forward, rho_diffusion/models/unet.py:239-269, for UNetBlock2d: time_mlp = sinusoid -> Linear (:171-174), input conv
    3x3, the down blocks (outputs pushed), the up blocks on cat(x, popped) - the first pop returns the tensor just pushed -,
    output conv 1x1.  cfg: the constructor kwargs (down_channels, up_channels, time_embedding_dim, activation)."""
    act = _V1_ACT[cfg.get("activation", "ReLU")]
    tdim = cfg.get("time_embedding_dim", 32)
    time_pe = F.linear(sinusoidal_embedding(t, tdim), sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    x = F.conv2d(data, sd["input_conv.weight"], sd["input_conv.bias"], stride=1, padding=1)
    stack: List[Tensor] = []
    for i in range(len(cfg.get("down_channels", [64, 128, 256])) - 1):
        x = unet_v1_block(x, time_pe, sd, f"downsample.{i}.", False, act)
        stack.append(x)
    for i in range(len(cfg.get("up_channels", [256, 128, 64])) - 1):
        x = unet_v1_block(torch.cat((x, stack.pop()), dim=1), time_pe, sd, f"upsample.{i}.", True, act)
    return F.conv2d(x, sd["output_conv.weight"], sd["output_conv.bias"])
