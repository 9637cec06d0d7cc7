"""-m gpu: the legacy UNet (models/unet.py, reference rho_diffusion/models/unet.py:30-269) on the HIP path against the goldens minted
from the reference class (g16: forward, loss, per-parameter gradient digests) and its tail kernels (csrc/unet_v1.hip) against the
oracle.  Tolerances: exact-f32 engine rel-L2 <= 1e-4 forward, gradient norms within 2e-3; bf16 engine <= 3e-2 forward, per-parameter
gradient cosine >= 0.99 against the oracle's gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import V1_CASES, cosine, det_normal, det_state_dict, golden_template, grad_digest_of, load_golden, rel_l2
from gpu_util import DEV, from_cl, to_cl
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu


def _model(case, dtype="fp32"):
    from rho_diffusion_amd.registry import registry
    g = load_golden("g16_unet_v1.npz")
    kw, xshape = V1_CASES[case]
    model = registry.get("models", "UNet")(**dict(kw), compute_dtype=dtype)
    model.load_state_dict(det_state_dict(golden_template(g, case), case))
    x = det_normal(xshape, case + "x")
    t = torch.tensor([(37 * i + 11) % 1000 for i in range(xshape[0])])
    return g, model.to(DEV), x, t


@pytest.mark.parametrize("case", list(V1_CASES))
def test_legacy_unet_forward_and_gradients_fp32_vs_reference_golden(case):
    from rho_diffusion_amd.autograd import mse_loss
    g, model, x, t = _model(case)
    pred = model(x.to(DEV), t.to(DEV))
    assert rel_l2(pred, torch.from_numpy(g[f"{case}/pred"])) < 1e-4
    loss = mse_loss(pred, det_normal(tuple(pred.shape), case + "tgt").to(DEV))
    assert abs(loss.item() - float(g[f"{case}/loss"])) < 2e-4
    loss.backward()
    bad = []
    for name, p in model.named_parameters():
        ref = g[f"{case}/grad/{name}"]
        assert p.grad is not None, name
        d = grad_digest_of(p.grad)
        rms = ref[0] / np.sqrt(p.numel())
        if abs(d[0] - ref[0]) > 2e-3 * ref[0] + 1e-6:
            bad.append((name, "norm", d[0], ref[0]))
        elif np.max(np.abs(d[2:] - ref[2:])) > 2e-2 * max(rms, 1e-7) + 1e-6:
            bad.append((name, "head", d[2:4], ref[2:4]))
    assert not bad, bad[:6]


@pytest.mark.parametrize("case", ["v1_relu", "v1_gelu_rgb"])
def test_legacy_unet_bf16_tracks_the_oracle(case):
    from rho_diffusion_amd.autograd import mse_loss
    g, model, x, t = _model(case, "bf16")
    pred = model(x.to(DEV), t.to(DEV))
    assert rel_l2(pred, torch.from_numpy(g[f"{case}/pred"])) < 3e-2
    tgt = det_normal(tuple(pred.shape), case + "tgt")
    mse_loss(pred, tgt.to(DEV)).backward()
    kw, _ = V1_CASES[case]
    sd = {k: v.clone().requires_grad_(True) for k, v in det_state_dict(golden_template(g, case), case).items()}
    F.mse_loss(R.unet_v1_forward(sd, dict(kw), x, t), tgt).backward()
    bad = []
    for name, p in model.named_parameters():
        c = cosine(p.grad, sd[name].grad)
        rn, dn = float(sd[name].grad.norm()), float(p.grad.norm())
        if c < 0.99 or abs(dn - rn) > 0.05 * rn:
            bad.append((name, round(c, 4), round(dn / rn, 4)))
    assert not bad, bad[:8]


def test_legacy_unet_3d_block_raises_like_the_reference():
    """unet.py:128-129 adds time_pe [B, C, 1, 1] to [B, C, D, H, W]: a broadcast error in the reference, the same class of error here."""
    from rho_diffusion_amd.registry import registry
    m = registry.get("models", "UNet")("UNetBlock3d", 1, [32, 64], [64, 32]).to(DEV)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 1, 4, 8, 8, device=DEV), torch.tensor([1, 2], device=DEV))
    with pytest.raises(Exception):
        registry.get("models", "UNet")("UNetBlock2d", 1, [32, 64], [64, 32], activation="Tanh")(torch.zeros(1, 1, 8, 8, device=DEV),
                                                                                                torch.tensor([1], device=DEV))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("groups,act", [(8, 2), (8, 3), (4, 1), (32, 0), (3, 2)])
def test_groupnorm_any_groups_with_activation_forward_backward(groups, act, dtype):
    """rho_groupnorm_act(_bwd) vs torch.nn.functional.group_norm + activation + autograd (unet.py:109-112,131-135)."""
    from rho_diffusion_amd import hip
    from rho_diffusion_amd.hip import check, ptr
    L = hip.load()
    N, C, H, W = 3, 96, 10, 7
    fn = {0: lambda v: v, 1: F.silu, 2: F.relu, 3: F.gelu}[act]
    x = det_normal((N, C, H, W), f"gx{groups}{act}")
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    gamma = (1 + 0.3 * det_normal((C,), "gg")).requires_grad_(True)
    beta = (0.2 * det_normal((C,), "gb")).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    ref = fn(F.group_norm(xr, groups, gamma, beta, eps=1e-5))
    dy = det_normal(tuple(ref.shape), "gdy")
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    ref.backward(dy)
    xcl, dycl = to_cl(x, dtype), to_cl(dy, dtype)
    y = torch.empty_like(xcl)
    stats = torch.empty(N, groups, 2, device=DEV)
    gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
    code = hip.dtype_code(dtype)
    check(L.rho_groupnorm_act(ptr(xcl), ptr(y), ptr(stats), ptr(gd), ptr(bd), code, N, H * W, C, groups, 1e-5, act, hip.stream()), "fwd")
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert rel_l2(from_cl(y, 2), ref.detach()) < tol
    dx = torch.empty_like(xcl)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    check(L.rho_groupnorm_act_bwd(ptr(xcl), ptr(dycl), ptr(stats), ptr(gd), ptr(bd), ptr(dx), ptr(dg), ptr(db), code, N, H * W, C,
                                  groups, act, hip.stream()), "bwd")
    tolb = 5e-5 if dtype == torch.float32 else 1.5e-2
    assert rel_l2(from_cl(dx, 2), xr.grad) < tolb
    assert rel_l2(dg.cpu(), gamma.grad) < tolb and rel_l2(db.cpu(), beta.grad) < tolb


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_act_add_forward_backward(act):
    """rho_act_add / rho_act_bwd: out = act(x) + r + nc[n, c] and dx = dout * act'(x)  (unet.py:121-129)."""
    from rho_diffusion_amd import hip
    from rho_diffusion_amd.hip import check, ptr
    L = hip.load()
    N, C, H, W = 2, 64, 5, 9
    fn = {0: lambda v: v, 1: F.silu, 2: F.relu, 3: F.gelu}[act]
    x = det_normal((N, C, H, W), f"ax{act}").requires_grad_(True)
    r = det_normal((N, C, H, W), "ar")
    nc = det_normal((N, C), "anc")
    ref = fn(x) + r + nc[:, :, None, None]
    dout = det_normal(tuple(ref.shape), "adout")
    ref.backward(dout)
    xcl, rcl, dcl = to_cl(x.detach(), torch.float32), to_cl(r, torch.float32), to_cl(dout, torch.float32)
    out = torch.empty_like(xcl)
    ncd = nc.to(DEV)
    check(L.rho_act_add(ptr(xcl), ptr(rcl), ptr(ncd), ptr(out), hip.RHO_F32, N, H * W, C, act, hip.stream()), "act_add")
    assert rel_l2(from_cl(out, 2), ref.detach()) < 1e-6
    dx = torch.empty_like(xcl)
    check(L.rho_act_bwd(ptr(xcl), ptr(dcl), ptr(dx), hip.RHO_F32, xcl.numel(), act, hip.stream()), "act_bwd")
    assert rel_l2(from_cl(dx, 2), x.grad) < 1e-5
