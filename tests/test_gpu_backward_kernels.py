"""-m gpu: every backward HIP kernel against torch autograd of the CPU oracle ops on the same inputs.
fp32: rel-L2 <= 5e-5;  bf16: <= 1.5e-2 against autograd evaluated on bf16-rounded operands."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import det_normal, det_uniform, rel_l2
from gpu_util import DEV, from_cl, rnd, to_cl
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def tolb(dtype, f32=5e-5, bf16=1.5e-2):
    return bf16 if dtype == torch.bfloat16 else f32


@pytest.fixture(scope="module")
def ops():
    from rho_diffusion_amd.engine import ops as o
    from rho_diffusion_amd import hip
    hip.load()
    return o


BWD_CASES = [
    # name, dims, N, c1, c2, cout, spatial, k, stride, up
    ("3d_basic", 3, 2, 32, 0, 64, (4, 8, 8), 3, 1, False),
    ("3d_ragged", 3, 1, 64, 0, 32, (5, 6, 7), 3, 1, False),
    ("3d_concat", 3, 2, 64, 32, 64, (4, 8, 8), 3, 1, False),
    # (round 4, k_wgrad's per-tile validity masks: several tiles per axis with overhanging last tiles; whole tiles = the GEO variant)
    ("3d_multi_ragged", 3, 2, 32, 0, 64, (9, 20, 12), 3, 1, False),
    ("3d_multi_whole", 3, 1, 64, 0, 64, (12, 24, 16), 3, 1, False),
    ("2d_multi_ragged", 2, 2, 64, 0, 64, (40, 36), 3, 1, False),
    ("3d_down", 3, 2, 32, 0, 32, (4, 8, 8), 3, (1, 2, 2), False),
    ("3d_down_odd", 3, 1, 32, 0, 64, (3, 7, 9), 3, (1, 2, 2), False),
    ("3d_up", 3, 2, 32, 0, 32, (4, 4, 4), 3, 1, True),
    ("3d_1x1", 3, 2, 96, 0, 64, (3, 5, 7), 1, 1, False),
    ("2d_basic", 2, 3, 32, 0, 64, (12, 10), 3, 1, False),
    ("2d_concat", 2, 2, 128, 64, 128, (16, 16), 3, 1, False),
    ("2d_down", 2, 2, 64, 0, 64, (16, 12), 3, 2, False),
    ("2d_up", 2, 2, 64, 0, 64, (6, 8), 3, 1, True),
    ("1d_basic", 1, 2, 32, 0, 32, (40,), 3, 1, False),
    ("1d_down", 1, 2, 32, 0, 32, (32,), 3, 2, False),
    ("1d_1x1", 1, 2, 64, 0, 192, (16,), 1, 1, False),
]


def _geom(dims, k, stride, up):
    sdims = stride if isinstance(stride, tuple) else (stride,) * dims
    s3 = (1,) * (3 - dims) + tuple(sdims)
    stride_hw = (s3[1], s3[2])
    up_hw = ((1, 1) if dims >= 2 else (0, 1)) if up else (0, 0)
    kernel = (1,) * (3 - dims) + (k,) * dims
    return kernel, stride_hw, up_hw


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_conv_dgrad_and_wgrad(ops, dtype, case):
    name, dims, N, c1, c2, cout, spatial, k, stride, up = case
    cin = c1 + c2
    x = rnd(det_normal((N, cin, *spatial), name + "bx"), dtype).requires_grad_(True)
    w = rnd(det_normal((cout, cin) + (k,) * dims, name + "bw") / math.sqrt(cin * k ** dims), dtype).requires_grad_(True)
    b = torch.zeros(cout)
    xin = R.upsample(dims, x) if up else x
    y = R.conv_nd(dims, xin, w, b, stride=stride, padding=k // 2)
    dy = rnd(det_normal(tuple(y.shape), name + "bdy"), dtype)
    y.backward(dy)
    kernel, stride_hw, up_hw = _geom(dims, k, stride, up)
    xs = x.detach()
    x1, x2 = (xs[:, :c1], xs[:, c1:]) if c2 else (xs, None)
    x1cl, x2cl = to_cl(x1, dtype), (to_cl(x2, dtype) if c2 else None)
    dycl = to_cl(dy, dtype)

    # ---- data gradient: the forward kernel on dY with flipped / transposed weights
    wd = ops.prep_conv_weight_dgrad(w.detach().to(DEV), dtype)          # [taps, ceil32(cin), ceilCK(cout)]
    zb = torch.zeros(wd.shape[1], device=DEV)
    Din, Hin, Win = x1cl.shape[1:4]
    if up:
        # gradient w.r.t. the upsampled input, then the 2x2 / 1x2 sum
        du, _ = ops.conv(dycl, None, wd, zb, kernel=kernel, cout=cin)
        dx = torch.empty(N, Din, Hin, Win, cin, dtype=dtype, device=DEV)
        ops.pool2x_sum(du, dx, up_hw)
        got = from_cl(dx, dims)
    elif stride_hw != (1, 1):
        zs = (int(stride_hw[0] == 2), int(stride_hw[1] == 2))
        y_ = torch.empty(N, Din, Hin, Win, cin, dtype=dtype, device=DEV)
        d = ops.make_conv_desc(dycl, None, wd, zb, kernel=kernel, cout=cin, split=cin, y=y_, y2=None, zs_hw=zs, out_hw=(Hin, Win))
        ops.conv_launch(d)
        got = from_cl(y_, dims)
    elif c2:
        # concatenated input: two channels-last gradients, the second accumulated in place on top of a residual
        y_ = torch.empty(N, Din, Hin, Win, c1, dtype=dtype, device=DEV)
        base = rnd(det_normal((N, c2, *spatial), name + "base"), dtype)
        y2_ = to_cl(base, dtype)
        d = ops.make_conv_desc(dycl, None, wd, zb, kernel=kernel, cout=cin, split=c1, y=y_, y2=y2_, y2_cl=True, res2=y2_)
        ops.conv_launch(d)
        got = torch.cat([from_cl(y_, dims), from_cl(y2_, dims) - base], 1)
    else:
        y_, _ = ops.conv(dycl, None, wd, zb, kernel=kernel, cout=cin)
        got = from_cl(y_, dims)
    assert rel_l2(got, x.grad) < tolb(dtype), f"dgrad {name}"

    # ---- weight gradient
    wf = ops.prep_conv_weight(w.detach().to(DEV), dtype)               # only shapes matter for the descriptor
    xin1, xin2 = x1cl, x2cl
    if up:
        xin1 = ops.upsample2x(x1cl, up_hw)
    zbw = torch.zeros(wf.shape[1], device=DEV)
    dfw = ops.make_conv_desc(xin1, xin2, wf, zbw, kernel=kernel, cout=cout, split=cout,
                             y=dycl, y2=None, stride_hw=stride_hw)
    dwbuf = torch.zeros(wf.shape[0], wf.shape[1], wf.shape[2], dtype=torch.float32, device=DEV)
    dbias = torch.zeros(wf.shape[1], dtype=torch.float32, device=DEV)
    ops.conv_wgrad(dfw, dycl, dwbuf, dbias)
    grad = torch.full(tuple(w.shape), 7.0, device=DEV)
    ops.wgrad_finalize(dwbuf, grad)
    assert rel_l2(grad, w.grad) < tolb(dtype), f"wgrad {name}"
    # bias gradient accumulated by the same kernel = channel sums of dY (as stored in the engine dtype)
    ref_db = dycl.float().reshape(-1, dycl.shape[-1]).sum(0)[:cout]
    assert rel_l2(dbias[:cout], ref_db) < 1e-5 and float(dbias[cout:].abs().max() if dbias.numel() > cout else 0.0) == 0.0, f"dbias {name}"


@pytest.mark.parametrize("c1,c2,pre", [(48, 32, False), (16, 0, False), (64, 16, False), (48, 32, True), (96, 0, True)],
                         ids=["straddle_odd", "single_chunk", "even", "straddle_odd_pre", "pre"])
def test_conv_wgrad_f32_1x1_chunk_pairs(ops, c1, c2, pre):
    """Exact-f32 1x1x1 weight gradient: a workgroup owns TWO 16-channel input chunks (the upper 16 columns of the MFMA's B operand
    carry the second; wgrad.hip PAIRC).  Cases: a pair that straddles the two concat sources with an odd chunk count (the last
    workgroup has no second chunk), a single chunk, an even count; with and without the prologue in the loader; partial last tile."""
    dims, N, cout, spatial = 2, 3, 64, (9, 13)                       # 351 positions: one full 256-position tile + a partial one
    cin = c1 + c2
    x = det_normal((N, cin, *spatial), f"cp{c1}{c2}x")
    w = (det_normal((cout, cin, 1, 1), f"cp{c1}{c2}w") / math.sqrt(cin)).requires_grad_(True)
    a = 1 + 0.3 * det_normal((N, cin), "cpa")
    b = 0.2 * det_normal((N, cin), "cpb")
    sh = (N, cin, 1, 1)
    act = F.silu(a.reshape(sh) * x + b.reshape(sh)) if pre else x
    y = F.conv2d(act, w)
    dy = det_normal(tuple(y.shape), "cpdy")
    y.backward(dy)
    f32 = torch.float32
    wf = ops.prep_conv_weight(w.detach().to(DEV), f32)
    x1cl = to_cl(x[:, :c1], f32)
    x2cl = to_cl(x[:, c1:], f32) if c2 else None
    dycl = to_cl(dy, f32)
    zb, a_d, b_d = torch.zeros(wf.shape[1], device=DEV), a.to(DEV), b.to(DEV)
    d = ops.make_conv_desc(x1cl, x2cl, wf, zb, kernel=(1, 1, 1), cout=cout, split=cout, y=dycl, y2=None,
                           pre_a=a_d if pre else None, pre_b=b_d if pre else None, pre_silu=pre)
    dwbuf = torch.zeros(tuple(wf.shape), dtype=torch.float32, device=DEV)
    dbias = torch.zeros(wf.shape[1], dtype=torch.float32, device=DEV)
    ops.conv_wgrad(d, dycl, dwbuf, dbias)
    grad = torch.zeros(tuple(w.shape), device=DEV)
    ops.wgrad_finalize(dwbuf, grad)
    assert rel_l2(grad, w.grad) < 5e-5
    assert rel_l2(dbias[:cout], dy.sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_wgrad_with_prologue(ops, dtype):
    """wgrad recomputes SiLU(a*x+b) in its loader."""
    dims, N, cin, cout, spatial = 3, 2, 64, 32, (4, 6, 8)
    x = rnd(det_normal((N, cin, *spatial), "px"), dtype)
    a = 1 + 0.3 * det_normal((N, cin), "pa")
    b = 0.2 * det_normal((N, cin), "pb")
    w = rnd(det_normal((cout, cin, 3, 3, 3), "pw") / 40, dtype).requires_grad_(True)
    sh = (N, cin, 1, 1, 1)
    act = rnd(F.silu(a.reshape(sh) * x + b.reshape(sh)), dtype)
    y = F.conv3d(act, w, padding=1)
    dy = rnd(det_normal(tuple(y.shape), "pdy"), dtype)
    y.backward(dy)
    wf = ops.prep_conv_weight(w.detach().to(DEV), dtype)
    dycl = to_cl(dy, dtype)
    # descriptors hold raw pointers: keep every tensor alive for as long as the descriptor is used
    xcl, zb, a_d, b_d = to_cl(x, dtype), torch.zeros(wf.shape[1], device=DEV), a.to(DEV), b.to(DEV)
    d = ops.make_conv_desc(xcl, None, wf, zb, kernel=(3, 3, 3), cout=cout, split=cout,
                           y=dycl, y2=None, pre_a=a_d, pre_b=b_d, pre_silu=True)
    dwbuf = torch.zeros(tuple(wf.shape), dtype=torch.float32, device=DEV)
    ops.conv_wgrad(d, dycl, dwbuf)
    grad = torch.zeros(tuple(w.shape), device=DEV)
    ops.wgrad_finalize(dwbuf, grad)
    assert rel_l2(grad, w.grad) < tolb(dtype)
    # accumulate flag + row permutation (qkv)
    perm = torch.randperm(cout)
    src = perm.to(torch.int32).to(DEV)
    ops.wgrad_finalize(dwbuf, grad, row_src=src, accumulate=True)
    expect = w.grad.clone()
    expect[perm] += w.grad
    assert rel_l2(grad, expect) < tolb(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c1,c2,spatial,silu,film", [(64, 0, (3, 5, 7), True, True), (32, 64, (4, 6, 6), True, False),
                                                     (128, 0, (20, 20), False, False), (256, 128, (9,), True, True)])
def test_groupnorm_backward(ops, dtype, c1, c2, spatial, silu, film):
    N = 3
    C = c1 + c2
    x = rnd(det_normal((N, C, *spatial), "gbx") * 1.5 + 0.3, dtype).requires_grad_(True)
    gamma = (1 + 0.2 * det_uniform((C,), "gbg")).requires_grad_(True)
    beta = (0.1 * det_uniform((C,), "gbb")).requires_grad_(True)
    fl = (det_normal((N, 2 * C), "gbf") * 0.3).requires_grad_(True)
    shape = (N, C) + (1,) * len(spatial)
    y = R.group_norm32(x, gamma, beta)
    if film:
        y = y * (1 + fl[:, :C].reshape(shape)) + fl[:, C:].reshape(shape)
    if silu:
        y = F.silu(y)
    g = rnd(det_normal(tuple(y.shape), "gbdy"), dtype)
    y.backward(g)

    xs = x.detach()
    x1, x2 = (xs[:, :c1], xs[:, c1:]) if c2 else (xs, None)
    x1cl, x2cl = to_cl(x1, dtype), (to_cl(x2, dtype) if c2 else None)
    fl_d = fl.detach().to(DEV)
    a, b, stats = ops.gn_coeffs(x1cl, x2cl, gamma.detach().to(DEV), beta.detach().to(DEV),
                                scale=fl_d if film else None, shift=fl_d[:, C:] if film else None, film_stride=2 * C if film else 0)
    dx1 = torch.empty_like(x1cl)
    base2 = rnd(det_normal((N, c2, *spatial), "gbase"), dtype) if c2 else None
    dx2 = to_cl(base2, dtype) if c2 else None
    dgamma = torch.zeros(C, device=DEV)
    dbeta = torch.zeros(C, device=DEV)
    dfilm = torch.zeros(N, 2 * C, device=DEV)
    ops.gn_bwd(to_cl(g, dtype), x1cl, x2cl, a, b, stats, gamma.detach().to(DEV), beta.detach().to(DEV), silu, dx1, dx2, dgamma, dbeta,
               scale=fl_d if film else None, film_stride=2 * C if film else 0,
               dscale=dfilm if film else None, dshift=dfilm[:, C:].data_ptr() if film else None, dfilm_stride=2 * C,
               acc2=bool(c2))
    dims = len(spatial)
    got = from_cl(dx1, dims)
    if c2:
        got = torch.cat([got, from_cl(dx2, dims) - base2], 1)     # second source accumulates on top of an existing gradient
    t = tolb(dtype)
    assert rel_l2(got, x.grad) < t
    assert rel_l2(dgamma, gamma.grad) < t
    assert rel_l2(dbeta, beta.grad) < t
    if film:
        assert rel_l2(dfilm, fl.grad) < t


@pytest.mark.parametrize("dtype", DTYPES)
def test_chan_sum_upsample_pool(ops, dtype):
    x = rnd(det_normal((3, 64, 4, 6, 8), "csx"), dtype)
    xcl = to_cl(x, dtype)
    out_nc = torch.zeros(3, 80, device=DEV)
    out_c = torch.ones(64, device=DEV)
    ops.chan_sum(xcl, out_nc[:, 8:].data_ptr(), None, nc_stride=80)
    assert rel_l2(out_nc[:, 8:72], x.sum(dim=(2, 3, 4))) < 1e-5
    nc = torch.empty(3, 64, device=DEV)
    ops.chan_sum(xcl, nc, out_c, acc_c=True)
    assert rel_l2(out_c, 1 + x.sum(dim=(0, 2, 3, 4))) < 1e-5
    up = ops.upsample2x(xcl, (1, 1))
    assert torch.equal(from_cl(up, 3), R.upsample(3, x))
    dx = torch.empty_like(xcl)
    ops.pool2x_sum(up, dx, (1, 1))
    assert rel_l2(from_cl(dx, 3), 4 * x) < (1e-6 if dtype == torch.float32 else 4e-3)
    up1 = ops.upsample2x(xcl, (0, 1))
    assert torch.equal(from_cl(up1, 3), x.repeat_interleave(2, dim=4))


def test_linear_backward(ops):
    for (B, K, O, act) in [(5, 64, 256, False), (4, 256, 1000, True)]:
        x = det_normal((B, K), "lbx").requires_grad_(True)
        w = (det_normal((O, K), "lbw") / math.sqrt(K)).requires_grad_(True)
        b = det_normal((O,), "lbb").requires_grad_(True)
        y = F.linear(F.silu(x) if act else x, w, b)
        g = det_normal((B, O), "lbg")
        y.backward(g)
        dw = torch.empty(O, K, device=DEV)
        db = torch.empty(O, device=DEV)
        dx = torch.empty(B, K, device=DEV)
        ops.linear_bwd(g.to(DEV), x.detach().to(DEV), w.detach().to(DEV), dw, db, dx, act_in=act)
        assert rel_l2(dw, w.grad) < 1e-5 and rel_l2(db, b.grad) < 1e-5 and rel_l2(dx, x.grad) < 1e-5
        ops.linear_bwd(g.to(DEV), x.detach().to(DEV), w.detach().to(DEV), dw, db, dx, act_in=act, acc_params=True, acc_dx=True)
        assert rel_l2(dw, 2 * w.grad) < 1e-5 and rel_l2(dx, 2 * x.grad) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,heads,ch", [(2, 64, 4, 16), (2, 256, 2, 32), (1, 100, 1, 32), (2, 4, 1, 256), (1, 300, 4, 64),
                                          (1, 192, 2, 128)])
def test_attention_backward(ops, dtype, B, T, heads, ch):
    C = heads * ch
    q = rnd(det_normal((B, C, T), f"bq{T}{ch}"), dtype).requires_grad_(True)
    k = rnd(det_normal((B, C, T), f"bk{T}{ch}"), dtype).requires_grad_(True)
    v = rnd(det_normal((B, C, T), f"bv{T}{ch}"), dtype).requires_grad_(True)
    out = R.qkv_attention(torch.cat([q, k, v], 1), heads, new_order=True)
    dout = rnd(det_normal(tuple(out.shape), f"bdo{T}{ch}"), dtype)
    out.backward(dout)
    qk = torch.cat([q, k], 1).detach().permute(0, 2, 1).contiguous().to(DEV).to(dtype)
    vt = v.detach().contiguous().to(DEV).to(dtype)
    lse = torch.empty(B, heads, T, device=DEV)
    o = ops.attention(qk, vt, heads, lse=lse)
    # lse = log2-sum-exp of the scaled logits
    logits = torch.einsum("bhct,bhcs->bhts", q.detach().reshape(B, heads, ch, T), k.detach().reshape(B, heads, ch, T)) / math.sqrt(ch)
    ref_lse = torch.logsumexp(logits, dim=-1) / math.log(2.0)
    assert rel_l2(lse, ref_lse) < (1e-5 if dtype == torch.float32 else 5e-3)
    docl = dout.permute(0, 2, 1).contiguous().to(DEV).to(dtype)
    dqkv = ops.attention_bwd(qk, vt, o, docl, lse, heads)
    t = tolb(dtype, f32=5e-5, bf16=2.5e-2)
    dqkv = dqkv.float().cpu().permute(0, 2, 1)
    assert rel_l2(dqkv[:, :C], q.grad) < t, "dq"
    assert rel_l2(dqkv[:, C:2 * C], k.grad) < t, "dk"
    assert rel_l2(dqkv[:, 2 * C:], v.grad) < t, "dv"
