"""-m gpu: every HIP kernel of librho_hip.so against the CPU oracle (oracle/ref_torch.py) on the
same seeded inputs.  Tolerances: fp32 kernels rel-L2 <= 2e-5 (summation order only);
bf16 kernels rel-L2 <= 6e-3 against the oracle evaluated on bf16-rounded operands (fp32
accumulate, one output rounding = 2^-9 relative)."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import det_normal, det_uniform, load_golden, rel_l2
from gpu_util import DEV, bf16_round, from_cl, rnd, to_cl, tol
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from rho_diffusion_amd.engine import ops as o
    from rho_diffusion_amd import hip
    hip.load()
    return o


# ----------------------------------------------------------------------------- diffusion loops
@pytest.mark.parametrize("shape", [(4, 1, 8, 8, 8), (3, 3, 5, 7), (2, 1, 33)])
def test_q_sample(ops, shape):
    sched = R.linear_schedule(1000, 1e-3, 0.02)
    x0 = det_uniform(shape, "qx0", 0, 1)
    eps = det_normal(shape, "qeps")
    t = torch.tensor([(331 * i + 7) % 1000 for i in range(shape[0])])
    ref = R.q_sample(x0, t, eps, sched["alpha_bar_t"])
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = ops.q_sample(x0.to(DEV), eps.to(DEV), t.to(DEV), sched["alpha_bar_t"].to(DEV), nan_flag=flag)
    assert rel_l2(out, ref) < 1e-6
    assert int(flag.item()) == 0
    bad = x0.clone()
    bad.view(-1)[5] = float("nan")
    ops.q_sample(bad.to(DEV), eps.to(DEV), t.to(DEV), sched["alpha_bar_t"].to(DEV), nan_flag=flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("t", [999, 500, 2, 1, 0])
def test_p_sample_step(ops, t):
    from rho_diffusion_amd.diffusion.schedule import LinearSchedule
    s = LinearSchedule(1000, 1e-3, 0.02)
    sched = R.linear_schedule(1000, 1e-3, 0.02)
    shape = (2, 1, 6, 10, 10)
    x = det_normal(shape, "px") * 0.7
    eh = det_normal(shape, "pe")
    z = det_normal(shape, "pz")
    xg = x.to(DEV).clone()
    t_dev = torch.tensor([t], dtype=torch.int32, device=DEV)
    ops.p_sample_step(xg, eh.to(DEV), z.to(DEV), s.device_tables(DEV)["coef"], t_dev)
    if t == 0:
        ref = x                                             # no update at t = 0 (ddpm.py:210)
    else:
        ref = R.p_sample_step(x, eh, t, sched, z if t > 1 else torch.zeros_like(z))
    assert rel_l2(xg, ref) < 2e-6


def test_step_advance(ops):
    t_dev = torch.tensor([5], dtype=torch.int32, device=DEV)
    off = torch.tensor([10], dtype=torch.int64, device=DEV)
    ops.step_advance(t_dev, off, 7)
    assert int(t_dev.item()) == 4 and int(off.item()) == 17


def test_philox_normal(ops):
    n = 1 << 20
    a = torch.empty(n, device=DEV)
    b = torch.empty(n, device=DEV)
    ops.philox_normal(a, 777, 0)
    ops.philox_normal(b, 777, 0)
    assert torch.equal(a, b)                                # reproducible
    # offset continuity: the stream does not depend on how it is cut into launches
    c = torch.empty(n // 2, device=DEV)
    ops.philox_normal(c, 777, n // 8)
    assert torch.equal(c, a[n // 2:])
    ops.philox_normal(b, 778, 0)
    assert not torch.equal(a, b)
    a = a.double().cpu()
    assert abs(a.mean()) < 5e-3 and abs(a.std() - 1) < 5e-3
    assert abs((a ** 3).mean()) < 2e-2 and abs((a ** 4).mean() - 3) < 5e-2
    assert torch.isfinite(a).all()
    od = torch.tensor([n // 8], dtype=torch.int64, device=DEV)
    ops.philox_normal(c, 777, 0, offset_dev=od)
    assert torch.equal(c.cpu().double(), a[n // 2:])


def test_mse(ops):
    a = det_normal((3, 1, 9, 11), "ma")
    b = det_normal((3, 1, 9, 11), "mb")
    loss, grad = ops.mse(a.to(DEV), b.to(DEV), want_grad=True)
    assert abs(loss.item() - F.mse_loss(a, b).item()) < 1e-6
    assert rel_l2(grad, 2 * (a - b) / a.numel()) < 1e-6


def test_adamw_matches_golden(ops):
    g = load_golden("g8_adamw.npz")
    p = det_normal((257,), "adam_p").to(DEV)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step in range(1, 4):
        ops.adamw(p, det_normal((257,), f"adam_g{step}").to(DEV), m, v, 1e-4, 0.9, 0.999, 1e-8, 1e-2, step)
        assert rel_l2(p, torch.from_numpy(g[f"p{step}"])) < 1e-6


def test_timestep_sinusoid_any_t(ops):
    """rho_timestep_embed without the MLP: the interleaved sinusoid for ANY integer t (the r1 engine gathered rows of a
    1024-row table and clamped later steps to t = 1023), vs the oracle and the reference-minted golden g2."""
    g = load_golden("g2_sinusoid.npz")
    tg = torch.from_numpy(g["t"])
    for dim in (32, 64, 128):
        om = ops.sinusoid_frequencies(dim, 10000, DEV)
        out = ops.timestep_embed(om, tg.to(DEV), len(tg))
        # sinf / cosf of the device libm vs the host's: a couple of ulp of the result, measured against |value| <= 1
        assert float((out.cpu() - torch.from_numpy(g[f"dim{dim}"])).abs().max()) < 5e-7
    t = torch.tensor([0, 1, 17, 500, 999, 1023, 1024, 1999, 2000, 3999, 100000])
    ref = R.sinusoidal_embedding(t, 64)
    out = ops.timestep_embed(ops.sinusoid_frequencies(64, 10000, DEV), t.to(DEV), len(t))
    assert float((out.cpu() - ref).abs().max()) < 5e-7
    assert float((out.cpu()[6] - out.cpu()[5]).abs().max()) > 1e-3        # t = 1024 is not t = 1023
    ts = torch.tensor([1999], dtype=torch.int32, device=DEV)
    out = ops.timestep_embed(ops.sinusoid_frequencies(64, 10000, DEV), None, 3, t_scalar_dev=ts)
    assert float((out.cpu() - ref[7].expand(3, -1)).abs().max()) < 5e-7


def test_timestep_embed_mlp_and_linear(ops):
    """Fused sinusoid -> Linear -> SiLU -> Linear (+ cond) vs the oracle's embedding chain (unet_v2.py:699-719)."""
    mc, E, B = 64, 256, 5
    sd = {"time_embed.0.weight": det_normal((E, mc), "tw0") / 8, "time_embed.0.bias": det_normal((E,), "tb0") * 0.1,
          "time_embed.2.weight": det_normal((E, E), "tw2") / 16, "time_embed.2.bias": det_normal((E,), "tb2") * 0.1}
    t = torch.tensor([0, 3, 999, 1500, 2000])
    cond = det_normal((B, E), "tcond")
    ref = R.unet_embedding(sd, dict(model_channels=mc, num_classes=1), t, cond)
    g = {k: v.to(DEV) for k, v in sd.items()}
    pe = torch.empty(B, mc, device=DEV)
    h = torch.empty(B, E, device=DEV)
    out = ops.timestep_embed(ops.sinusoid_frequencies(mc, 10000, DEV), t.to(DEV), B, w0=g["time_embed.0.weight"],
                             b0=g["time_embed.0.bias"], w2=g["time_embed.2.weight"], b2=g["time_embed.2.bias"], cond=cond.to(DEV),
                             pe_out=pe, h_out=h)
    assert rel_l2(out, ref) < 2e-6
    assert rel_l2(pe, R.sinusoidal_embedding(t, mc)) < 1e-6
    assert rel_l2(h, F.linear(R.sinusoidal_embedding(t, mc), sd["time_embed.0.weight"], sd["time_embed.0.bias"])) < 2e-6
    for (B, K, O, ai, ao) in [(5, 64, 256, False, True), (5, 256, 1000, True, False), (3, 1024, 130, False, False)]:
        x = det_normal((B, K), "lx")
        w = det_normal((O, K), "lw") / math.sqrt(K)
        b = det_normal((O,), "lb")
        add = det_normal((B, O), "la")
        ref = F.linear(F.silu(x) if ai else x, w, b) + add
        ref = F.silu(ref) if ao else ref
        out = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), add.to(DEV), act_in=ai, act_out=ao)
        assert rel_l2(out, ref) < 2e-6


# ----------------------------------------------------------------------------- GroupNorm coefficients
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c1,c2,spatial", [(64, 0, (3, 5, 7)), (32, 64, (4, 6, 6)), (128, 0, (40, 40)), (256, 128, (9,)),
                                           (512, 512, (2, 4, 4))])
def test_gn_coeffs(ops, dtype, c1, c2, spatial):
    N = 3
    x1 = rnd(det_normal((N, c1, *spatial), "g1") * 2 + 0.7, dtype)
    x2 = rnd(det_normal((N, c2, *spatial), "g2") * 0.5 - 1.0, dtype) if c2 else None
    C = c1 + c2
    gamma = 1 + 0.2 * det_uniform((C,), "gg")
    beta = 0.1 * det_uniform((C,), "gb")
    film = det_normal((N, 2 * C), "gf") * 0.3
    xcat = torch.cat([x1, x2], 1) if c2 else x1
    a, b, st = ops.gn_coeffs(to_cl(x1, dtype), to_cl(x2, dtype) if c2 else None, gamma.to(DEV), beta.to(DEV),
                             scale=film.to(DEV), shift=film.to(DEV)[:, C:], film_stride=2 * C)
    # oracle: y = GN(x) * (1 + scale) + shift  ==  a * x + b
    ref = R.group_norm32(xcat, gamma, beta)
    shape = (N, C) + (1,) * len(spatial)
    ref = ref * (1 + film[:, :C].reshape(shape)) + film[:, C:].reshape(shape)
    got = a.cpu().reshape(shape) * xcat + b.cpu().reshape(shape)
    assert rel_l2(got, ref) < 1e-5
    a0, b0, _ = ops.gn_coeffs(to_cl(x1, dtype), to_cl(x2, dtype) if c2 else None, gamma.to(DEV), beta.to(DEV))
    got = a0.cpu().reshape(shape) * xcat + b0.cpu().reshape(shape)
    assert rel_l2(got, R.group_norm32(xcat, gamma, beta)) < 1e-5


# ----------------------------------------------------------------------------- convolution
def _oracle_conv(dims, x, w, b, stride, up, pre, pre_silu, dtype):
    xx = x
    if pre is not None:
        a, bb = pre
        shape = a.shape + (1,) * dims
        xx = a.reshape(shape) * xx + bb.reshape(shape)
        if pre_silu:
            xx = F.silu(xx)
        xx = rnd(xx, dtype)          # the loader rounds the activated tile to the MFMA input type
    if up:
        xx = R.upsample(dims, xx)
    return R.conv_nd(dims, xx, w, b, stride=stride, padding=w.shape[-1] // 2)


CONV_CASES = [
    # name, dims, N, cin1, cin2, cout, spatial, k, stride, up, prologue, residual
    ("3d_basic", 3, 2, 32, 0, 64, (4, 8, 8), 3, 1, False, False, False),
    ("3d_ragged", 3, 1, 64, 0, 32, (5, 6, 7), 3, 1, False, True, True),
    ("3d_concat", 3, 2, 64, 32, 64, (4, 8, 8), 3, 1, False, True, True),
    ("3d_wide", 3, 1, 128, 128, 128, (4, 4, 8), 3, 1, False, True, False),
    ("3d_down", 3, 2, 32, 0, 32, (4, 8, 8), 3, (1, 2, 2), False, False, False),
    ("3d_down_odd", 3, 1, 32, 0, 64, (3, 7, 9), 3, (1, 2, 2), False, False, False),
    ("3d_up", 3, 2, 32, 0, 32, (4, 4, 4), 3, 1, True, False, False),
    ("3d_1x1", 3, 2, 96, 0, 64, (3, 5, 7), 1, 1, False, False, False),
    ("2d_basic", 2, 3, 32, 0, 64, (12, 10), 3, 1, False, True, True),
    ("2d_concat", 2, 2, 128, 64, 128, (16, 16), 3, 1, False, True, False),
    ("2d_down", 2, 2, 64, 0, 64, (16, 12), 3, 2, False, False, False),
    ("2d_up", 2, 2, 64, 0, 64, (6, 8), 3, 1, True, False, False),
    ("2d_1x1_concat", 2, 2, 64, 32, 32, (8, 8), 1, 1, False, False, False),
    ("1d_basic", 1, 2, 32, 0, 32, (40,), 3, 1, False, True, True),
    ("1d_down", 1, 2, 32, 0, 32, (32,), 3, 2, False, False, False),
    ("1d_up", 1, 2, 32, 0, 32, (16,), 3, 1, True, False, False),
    ("1d_1x1", 1, 2, 64, 0, 192, (16,), 1, 1, False, True, True),
    ("3d_big_tile", 3, 1, 64, 0, 64, (8, 16, 16), 3, 1, False, True, True),
    ("3d_tile512_ragged", 3, 2, 64, 32, 64, (9, 10, 13), 3, 1, False, True, True),     # 512-position tiles (bf16), ragged edges
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv(ops, dtype, case):
    name, dims, N, c1, c2, cout, spatial, k, stride, up, prologue, residual = case
    cin = c1 + c2
    x1 = rnd(det_normal((N, c1, *spatial), name + "x1"), dtype)
    x2 = rnd(det_normal((N, c2, *spatial), name + "x2"), dtype) if c2 else None
    w = rnd(det_normal((cout, cin) + (k,) * dims, name + "w") / math.sqrt(cin * k ** dims), dtype)
    b = det_normal((cout,), name + "b") * 0.1
    pre = None
    if prologue:
        pre = (1 + 0.3 * det_normal((N, cin), name + "a"), 0.2 * det_normal((N, cin), name + "pb"))
    xcat = torch.cat([x1, x2], 1) if c2 else x1
    ref = _oracle_conv(dims, xcat, w, b, stride, up, pre, True, dtype)
    res = rnd(det_normal(tuple(ref.shape), name + "r"), dtype) if residual else None
    if residual:
        ref = ref + res
    sdims = stride if isinstance(stride, tuple) else (stride,) * dims
    s3 = (1,) * (3 - dims) + tuple(sdims)
    stride_hw = (s3[1], s3[2])
    up_hw = ((1, 1) if dims >= 2 else (0, 1)) if up else (0, 0)
    kernel = (1,) * (3 - dims) + (k,) * dims
    wp = ops.prep_conv_weight(w.to(DEV), dtype)
    bp = b.to(DEV)
    y, _ = ops.conv(to_cl(x1, dtype), to_cl(x2, dtype) if c2 else None, wp, bp, kernel=kernel, cout=cout,
                    stride_hw=stride_hw, up_hw=up_hw, pre_a=pre[0].to(DEV) if pre else None,
                    pre_b=pre[1].to(DEV) if pre else None, pre_silu=True, res=to_cl(res, dtype) if residual else None)
    got = from_cl(y, dims)
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < tol(dtype), name


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_split_and_small_cout(ops, dtype):
    """channel-major second output (attention V^T / final NC* float32 head) and res_add."""
    dims, N, cin, spatial = 2, 2, 64, (8, 8)
    x = rnd(det_normal((N, cin, *spatial), "sx"), dtype)
    # (a) qkv-like: 3C outputs, first 2C channels-last, last C channel-major
    C = 64
    w = rnd(det_normal((3 * C, cin, 1, 1), "sw") / 8, dtype)
    b = det_normal((3 * C,), "sb") * 0.1
    ref = F.conv2d(x, w, b)
    y, y2 = ops.conv(to_cl(x, dtype), None, ops.prep_conv_weight(w.to(DEV), dtype), b.to(DEV), kernel=(1, 1, 1),
                     cout=3 * C, split=2 * C)
    assert rel_l2(from_cl(y, dims), ref[:, :2 * C]) < tol(dtype)
    assert rel_l2(y2.float().cpu().reshape(N, C, *spatial), ref[:, 2 * C:]) < tol(dtype)
    # (b) head-like: 3 output channels, float32 NC* output, + additive per-(n, co) embedding on a 32-ch conv
    w = rnd(det_normal((3, cin, 3, 3), "hw") / 24, dtype)
    b = det_normal((3,), "hb")
    ref = F.conv2d(x, w, b, padding=1)
    wp = ops.prep_conv_weight(w.to(DEV), dtype)
    bp = torch.zeros(wp.shape[1], device=DEV)
    bp[:3] = b.to(DEV)
    _, y2 = ops.conv(to_cl(x, dtype), None, wp, bp, kernel=(1, 3, 3), cout=3, split=0, y2_dtype=torch.float32)
    assert y2.dtype == torch.float32
    assert rel_l2(y2.cpu().reshape(N, 3, *spatial), ref) < tol(dtype, bf16=2e-3)
    w = rnd(det_normal((32, cin, 3, 3), "aw") / 24, dtype)
    b = det_normal((32,), "ab")
    emb = det_normal((N, 40), "ae")
    ref = F.conv2d(x, w, b, padding=1) + emb[:, 4:36].reshape(N, 32, 1, 1)
    y, _ = ops.conv(to_cl(x, dtype), None, ops.prep_conv_weight(w.to(DEV), dtype), b.to(DEV), kernel=(1, 3, 3), cout=32,
                    res_add=emb.to(DEV)[:, 4:], res_add_stride=40)
    assert rel_l2(from_cl(y, dims), ref) < tol(dtype)


def test_prep_conv_weight_layout(ops):
    w = det_normal((5, 3, 3, 3, 3), "pw")
    out = ops.prep_conv_weight(w.to(DEV), torch.float32)            # [27, 32, 16]
    assert tuple(out.shape) == (27, 32, 16)
    ref = torch.zeros(27, 32, 16)
    ref[:, :5, :3] = w.reshape(5, 3, 27).permute(2, 0, 1)
    assert torch.equal(out.cpu(), ref)
    src = torch.tensor([4, 3, 2, 1, 0] + [-1] * 27, dtype=torch.int32, device=DEV)
    out = ops.prep_conv_weight(w.to(DEV), torch.bfloat16, row_src=src)
    assert torch.equal(out.float().cpu()[:, :5, :3], bf16_round(ref[:, [4, 3, 2, 1, 0], :3]))


def test_pack_input(ops):
    x = det_normal((2, 3, 4, 5, 6), "pk")
    for dt in DTYPES:
        y = ops.pack_input(x.to(DEV), dt)
        assert y.shape[-1] == (32 if dt == torch.bfloat16 else 16)
        assert torch.equal(from_cl(y[..., :3], 3), rnd(x, dt))
        assert float(y[..., 3:].float().abs().max()) == 0.0


# ----------------------------------------------------------------------------- attention
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,heads,ch", [(2, 64, 4, 16), (2, 256, 2, 32), (1, 100, 1, 32), (2, 4, 1, 256), (1, 300, 4, 64),
                                          (1, 192, 2, 128), (1, 16, 4, 64)])
def test_attention(ops, dtype, B, T, heads, ch):
    C = heads * ch
    q = rnd(det_normal((B, C, T), f"aq{T}{ch}"), dtype)
    k = rnd(det_normal((B, C, T), f"ak{T}{ch}"), dtype)
    v = rnd(det_normal((B, C, T), f"av{T}{ch}"), dtype)
    ref = R.qkv_attention(torch.cat([q, k, v], 1), heads, new_order=True)          # canonical [Q|K|V] order
    qk = torch.cat([q, k], 1).permute(0, 2, 1).contiguous().to(DEV).to(dtype)      # [B, T, 2C]
    vt = v.contiguous().to(DEV).to(dtype)                                          # [B, C, T]
    out = ops.attention(qk, vt, heads)                                             # [B, T, C]
    got = out.float().cpu().permute(0, 2, 1)
    assert rel_l2(got, ref) < tol(dtype, f32=1e-5, bf16=1e-2)


def test_attention_online_softmax_rescale(ops):
    """Force the running-max rescale branch: one key far above the rest, placed in a late tile."""
    B, T, heads, ch = 1, 256, 1, 32
    q = det_normal((B, ch, T), "rq")
    k = det_normal((B, ch, T), "rk")
    v = det_normal((B, ch, T), "rv")
    k[0, :, 200] = q[0, :, 7] * 6.0
    for dtype in DTYPES:
        qq, kk, vv = rnd(q, dtype), rnd(k, dtype), rnd(v, dtype)
        ref = R.qkv_attention(torch.cat([qq, kk, vv], 1), heads, new_order=True)
        qk = torch.cat([qq, kk], 1).permute(0, 2, 1).contiguous().to(DEV).to(dtype)
        out = ops.attention(qk, vv.contiguous().to(DEV).to(dtype), heads)
        assert rel_l2(out.float().cpu().permute(0, 2, 1), ref) < tol(dtype, f32=1e-5, bf16=1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["3d", "3d_ragged", "1x1"])
def test_conv_fused_groupnorm_statistics(ops, dtype, kind):
    """rho_conv_desc.stats + rho_gn_finalize2 (fused epilogue statistics) against rho_gn_partial + rho_gn_finalize on the
    stored output: same (a, b) up to fp32 summation order, for a single source and for a concat with a plain source."""
    import ctypes as C
    from rho_diffusion_amd import hip
    L = hip.lib()
    N, cin, cout = 2, 32, 64
    spatial = {"3d": (4, 8, 8), "3d_ragged": (5, 6, 7), "1x1": (4, 8, 8)}[kind]
    k = 1 if kind == "1x1" else 3
    x = rnd(det_normal((N, cin, *spatial), kind + "sx"), dtype)
    w = rnd(det_normal((cout, cin, k, k, k), kind + "sw") / math.sqrt(cin * k ** 3), dtype)
    b = det_normal((cout,), kind + "sb") * 0.1
    wp = ops.prep_conv_weight(w.to(DEV), dtype)
    x_cl = to_cl(x, dtype)
    S = spatial[0] * spatial[1] * spatial[2]
    y = torch.empty(N, *spatial, cout, dtype=x_cl.dtype, device=DEV)
    b_dev = b.to(DEV)                      # the descriptor holds raw pointers: keep every operand alive
    d = ops.make_conv_desc(x_cl, None, wp, b_dev, kernel=(k, k, k), cout=cout, split=cout, y=y, y2=None)
    tiles = int(L.rho_conv_stats_tiles(C.byref(d)))
    assert tiles > 0
    sbuf = torch.full((N * tiles * 2 * cout,), float("nan"), dtype=torch.float32, device=DEV)
    d.stats = sbuf.data_ptr()
    ops.conv_launch(d)
    assert torch.isfinite(sbuf).all()
    got = sbuf.view(N, tiles, 2, cout).double().sum(1).cpu()
    yf = y.float().reshape(N, S, cout).double().cpu()
    assert torch.allclose(got[:, 0], yf.sum(1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(got[:, 1], (yf * yf).sum(1), rtol=1e-5, atol=1e-4)
    # finalize: fused statistics vs the separate pass, also as the first half of a concat
    extra = to_cl(rnd(det_normal((N, 32, *spatial), kind + "sx2"), dtype), dtype)
    for x2, c2 in ((None, 0), (extra, 32)):
        Cc = cout + c2
        gamma = (1 + 0.2 * det_normal((Cc,), "sg")).to(DEV)
        beta = (0.1 * det_normal((Cc,), "sbt")).to(DEV)
        a_ref, b_ref, _ = ops.gn_coeffs(y, x2, gamma, beta)
        a2 = torch.empty(N, Cc, device=DEV)
        b2 = torch.empty(N, Cc, device=DEV)
        p2, nb2 = None, 0
        if x2 is not None:
            nb2 = ops.gn_nblk(S)
            p2 = torch.empty(N * nb2 * (c2 // 8) * 16, device=DEV)
            hip.check(L.rho_gn_partial(ops.ptr(x2), c2, None, 0, hip.dtype_code(dtype), N, S, ops.ptr(p2), ops.stream()), "gn_partial")
        hip.check(L.rho_gn_finalize2(ops.ptr(sbuf), 1, tiles, cout, ops.ptr(p2), 0, nb2, c2, N, S, ops.ptr(gamma), ops.ptr(beta),
                                     None, None, 0, None, ops.ptr(a2), ops.ptr(b2), ops.stream()), "gn_finalize2")
        assert rel_l2(a2.cpu(), a_ref.cpu()) < 1e-5 and rel_l2(b2.cpu(), b_ref.cpu()) < 1e-4


def test_im2col_taps_and_tap_gather_sum(ops):
    """The two helpers that let a 1-channel stem / head convolution run as a 1x1x1 GEMM (rho_im2col_taps,
    rho_tap_gather_sum): im2col columns exact (bf16 rounding of the gathered values), gather = fp32 sum in tap order."""
    import ctypes as C
    import torch.nn.functional as F
    from rho_diffusion_amd import hip
    L = hip.lib()
    torch.manual_seed(5)
    for (N, cin, D, H, W, k) in [(2, 1, 5, 6, 7, (3, 3, 3)), (3, 2, 1, 9, 10, (1, 3, 3)), (2, 1, 1, 1, 33, (1, 1, 3))]:
        taps = k[0] * k[1] * k[2]
        x = torch.randn(N, cin, D, H, W)
        out = torch.empty(N, D, H, W, 32, dtype=torch.bfloat16, device=DEV)
        xd = x.to(DEV)
        rc = L.rho_im2col_taps(xd.data_ptr(), out.data_ptr(), 1, N, cin, D, H, W, k[0], k[1], k[2], 32, hip.stream())
        assert rc == 0
        xp = F.pad(x, (k[2] // 2, k[2] // 2, k[1] // 2, k[1] // 2, k[0] // 2, k[0] // 2))
        cols = []
        for ci in range(cin):
            for a in range(k[0]):
                for b in range(k[1]):
                    for c in range(k[2]):
                        cols.append(xp[:, ci, a:a + D, b:b + H, c:c + W])
        ref = torch.stack(cols, dim=-1).to(torch.bfloat16)
        got = out.cpu()
        assert torch.equal(got[..., : cin * taps], ref)
        assert float(got[..., cin * taps:].abs().max()) == 0.0
        # gather-sum: T[q][tap] -> out[pos] = bias + sum_tap T[pos + off(tap)][tap]
        t = torch.randn(N, D, H, W, 32).to(torch.bfloat16)
        bias = torch.tensor([0.37])
        o = torch.empty(N, 1, D * H * W, device=DEV)
        td, bd = t.to(DEV), bias.to(DEV)
        rc = L.rho_tap_gather_sum(td.data_ptr(), 1, N, D, H, W, k[0], k[1], k[2], 32, bd.data_ptr(), o.data_ptr(), hip.stream())
        assert rc == 0
        tp = F.pad(t.float().permute(0, 4, 1, 2, 3), (k[2] // 2, k[2] // 2, k[1] // 2, k[1] // 2, k[0] // 2, k[0] // 2))
        acc = torch.full((N, D, H, W), 0.37)
        tap = 0
        for a in range(k[0]):
            for b in range(k[1]):
                for c in range(k[2]):
                    acc = acc + tp[:, tap, a:a + D, b:b + H, c:c + W]
                    tap += 1
        assert torch.allclose(o.cpu().view(N, D, H, W), acc, rtol=1e-6, atol=1e-6)
    assert L.rho_im2col_taps(xd.data_ptr(), out.data_ptr(), 0, 1, 1, 1, 1, 8, 1, 1, 3, 32, hip.stream()) != 0      # bf16 only


WIDE_CONV = [((1, 1, 3), (2, 1, 1, 4096), 64, 64), ((1, 1, 3), (2, 1, 1, 4096), 96, 64), ((1, 1, 3), (2, 1, 1, 4096), 64, 128),
             ((1, 1, 3), (2, 1, 1, 2048), 256, 128), ((1, 3, 3), (2, 1, 64, 64), 128, 128), ((1, 3, 3), (2, 1, 32, 32), 256, 256),
             ((3, 3, 3), (2, 16, 16, 16), 128, 128), ((3, 3, 3), (1, 8, 16, 16), 256, 128)]


@pytest.mark.parametrize("kernel,shape,cin,cout", WIDE_CONV, ids=[f"{'x'.join(map(str, c[0]))}-{c[2]}to{c[3]}" for c in WIDE_CONV])
def test_conv_bf16_every_tap_count_and_cout_tile(ops, kernel, shape, cin, cout):
    """The 16x16x32 variants by tap count (3 / 9 / 27) and cout tile (64: barrier per tap, 3-slot weight ring; 128: three taps per
    barrier and a 9-slot ring where 9 divides the taps of a chunk, else per tap), several channel chunks deep - against torch's fp32
    convolution of the same bf16-rounded operands.  Tolerance: relative l2 <= 4e-3 (bf16 output rounding is 2^-9 per element)."""
    N, D, H, W = shape
    x = det_normal((N, D, H, W, cin), "wc_x").to(DEV).to(torch.bfloat16)
    wt = det_normal((cout, cin) + tuple(kernel), "wc_w").to(DEV) * 0.05
    b = det_normal((cout,), "wc_b").to(DEV)
    w = ops.prep_conv_weight(wt, torch.bfloat16)
    y = torch.empty(N, D, H, W, cout, device=DEV, dtype=torch.bfloat16)
    d = ops.make_conv_desc(x, None, w, b, kernel=tuple(kernel), cout=cout, split=cout, y=y, y2=None)
    assert "M16=1" in ops.conv_variant(d) and f"BM={min(cout, 128)}" in ops.conv_variant(d)
    ops.conv_launch(d)
    ref = F.conv3d(x.float().permute(0, 4, 1, 2, 3), wt.to(torch.bfloat16).float(), b, padding=tuple(k // 2 for k in kernel)).permute(0, 2, 3, 4, 1)
    assert rel_l2(y.float(), ref) <= 4e-3


PHASE_CASES = [((3, 3, 3), (2, 4, 8, 8), (1, 1), 64, 64), ((3, 3, 3), (1, 6, 8, 16), (1, 1), 128, 128), ((3, 3, 3), (2, 3, 5, 7), (1, 1), 32, 32),
               ((1, 3, 3), (2, 1, 16, 16), (1, 1), 64, 128), ((1, 3, 3), (3, 1, 9, 11), (1, 1), 32, 64), ((1, 1, 3), (2, 1, 1, 128), (0, 1), 64, 64),
               ((1, 1, 3), (2, 1, 1, 100), (0, 1), 32, 32)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kernel,shape,up,cin,cout", PHASE_CASES, ids=[f"{'x'.join(map(str, c[0]))}-{c[3]}to{c[4]}" for c in PHASE_CASES])
def test_upsample_conv_as_subpixel_phases(ops, dtype, kernel, shape, up, cin, cout):
    """Upsample (nearest x2 on the inner axes, unet_v2.py:122-131) + conv as one 2-tap launch per output parity on the source
    tensor (rho_conv_desc.ph_h / ph_w, rho_prep_conv_weight_phase) against torch: interpolate + convNd in fp32 on the same
    operands - output and the fused GroupNorm statistics of the output (3-D).  f32: 2e-5 relative l2; bf16: 6e-3 (the phase
    weights are sums of two / four bf16-rounded-later taps)."""
    N, D, H, W = shape
    x = rnd(det_normal((N, D, H, W, cin), "ph_x").to(DEV), dtype).to(dtype)
    wt = det_normal((cout, cin) + tuple(kernel), "ph_w").to(DEV) * 0.05
    b = det_normal((cout,), "ph_b").to(DEV)
    Ho, Wo = H * (2 if up[0] else 1), W * (2 if up[1] else 1)
    y = torch.full((N, D, Ho, Wo, cout), float("nan"), device=DEV, dtype=dtype)
    descs, keep = [], []
    for a in ((1, 2) if up[0] else (0,)):
        for c in ((1, 2) if up[1] else (0,)):
            wp = ops.prep_conv_weight_phase(wt, dtype, (a, c))
            keep.append(wp)                                   # the descriptor holds a raw pointer
            descs.append(ops.make_conv_desc(x, None, wp, b, kernel=(kernel[0], 2 if a else kernel[1], 2 if c else kernel[2]), cout=cout,
                                            split=cout, y=y, y2=None, phase_hw=(a, c)))
    tiles = ops.conv_stats_tiles(descs[0])
    sbuf = None
    if tiles > 0:
        sbuf = torch.zeros(N, tiles, 2, cout, device=DEV)
        for d in descs:
            d.stats = sbuf.data_ptr()
    for d in descs:
        ops.conv_launch(d)
    xr = x.float().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    xu = F.interpolate(xr, size=(D, Ho, Wo), mode="nearest")
    refc = F.conv3d(xu, wt, b, padding=tuple(k // 2 for k in kernel))
    ref = refc.detach().permute(0, 2, 3, 4, 1)
    assert not torch.isnan(y.float()).any()                                      # every output position written exactly by one phase
    assert rel_l2(y.float(), ref) <= tol(dtype)
    if sbuf is not None:
        assert kernel[0] == 3
        yy = y.float().reshape(N, -1, cout)
        s1, s2 = sbuf[:, :, 0].sum(1), sbuf[:, :, 1].sum(1)
        assert rel_l2(s1, yy.sum(1)) <= 1e-3 and rel_l2(s2, (yy * yy).sum(1)) <= 1e-3
    # data gradient: each phase's share is a 2-tap conv of that parity of dY with the phase's flipped weights (phd_h / phd_w),
    # accumulated in place - against autograd through interpolate + conv
    ck = 32 if dtype == torch.bfloat16 else 16
    dyw = ((cout + ck - 1) // ck) * ck
    dy = torch.zeros(N, D, Ho, Wo, dyw, device=DEV, dtype=dtype)
    dy[..., :cout] = rnd(det_normal((N, D, Ho, Wo, cout), "ph_dy").to(DEV), dtype).to(dtype)
    refc.backward(dy[..., :cout].float().permute(0, 4, 1, 2, 3))
    gx = xr.grad.permute(0, 2, 3, 4, 1)
    dx = torch.full((N, D, H, W, cin), float("nan"), device=DEV, dtype=dtype)
    zb = torch.zeros(((cin + 31) // 32) * 32, device=DEV)
    i = 0
    for a in ((1, 2) if up[0] else (0,)):
        for c in ((1, 2) if up[1] else (0,)):
            wd = ops.prep_conv_weight_phase(wt, dtype, (a, c), dgrad=True)
            keep.append(wd)
            dd = ops.make_conv_desc(dy, None, wd, zb, kernel=(kernel[0], 2 if a else kernel[1], 2 if c else kernel[2]), cout=cin, split=cin,
                                    y=dx, y2=None, res=dx if i > 0 else None, phase_dgrad_hw=(a, c))
            ops.conv_launch(dd)
            i += 1
    assert rel_l2(dx.float(), gx) <= (1.5e-2 if dtype == torch.bfloat16 else 5e-5)
    # weight gradient: per phase a 2-tap weight gradient on the source tensor against that parity of dY, routed to the 3-tap
    # parameter gradient (rho_wgrad_finalize_phase); bias gradient = the channel sums of dY over all parities
    wtr = wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    F.conv3d(F.interpolate(x.float().permute(0, 4, 1, 2, 3), size=(D, Ho, Wo), mode="nearest"), wtr, br,
             padding=tuple(k // 2 for k in kernel)).backward(dy[..., :cout].float().permute(0, 4, 1, 2, 3))
    grad = torch.zeros_like(wt)
    coutp = ((cout + 31) // 32) * 32
    dbias = torch.zeros(max(coutp, dyw), device=DEV)
    from rho_diffusion_amd import hip
    for dsc, wp in zip(descs, keep):
        dsc.stats = None
        dwb = torch.zeros(tuple(wp.shape), dtype=torch.float32, device=DEV)
        ops.conv_wgrad(dsc, dy, dwb, dbias)
        hip.check(hip.lib().rho_wgrad_finalize_phase(dwb.data_ptr(), grad.data_ptr(), cout, cin, kernel[0], kernel[1], kernel[2], dsc.ph_h,
                                                     dsc.ph_w, wp.shape[1], wp.shape[2], 1, hip.stream()), "rho_wgrad_finalize_phase")
    tb = 1.5e-2 if dtype == torch.bfloat16 else 5e-5
    assert rel_l2(grad, wtr.grad) <= tb
    assert rel_l2(dbias[:cout], br.grad) <= 1e-4


S2_CASES = [((2, 4, 8, 8), 64, 64), ((1, 6, 16, 8), 128, 128), ((2, 3, 6, 10), 32, 64)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,cin,cout", S2_CASES, ids=[f"{c[1]}to{c[2]}" for c in S2_CASES])
def test_stride2_conv_as_parity_split(ops, dtype, shape, cin, cout):
    """Downsample's 3-D conv (stride (1, 2, 2), unet_v2.py:153-162) on stride-1 launches: forward = one launch per INPUT parity
    accumulated in place (even rows: tap w1; odd rows: taps w0, w2), data gradient = one launch per parity of dX (dx[2m] = w1 dy[m];
    dx[2m+1] = w2 dy[m] + w0 dy[m+1]) - against torch's strided conv3d and its autograd.  Same tolerances as test_conv."""
    N, D, H, W = shape
    x = rnd(det_normal((N, D, H, W, cin), "s2_x").to(DEV), dtype).to(dtype)
    wt = rnd(det_normal((cout, cin, 3, 3, 3), "s2_w").to(DEV) * 0.05, dtype)
    b = det_normal((cout,), "s2_b").to(DEV)
    xr = x.float().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    refc = F.conv3d(xr, wt, b, stride=(1, 2, 2), padding=1)
    Ho, Wo = H // 2, W // 2
    y = torch.full((N, D, Ho, Wo, cout), float("nan"), device=DEV, dtype=dtype)
    coutp = ((cout + 31) // 32) * 32
    bp = torch.zeros(coutp, device=DEV); bp[:cout] = b
    zb = torch.zeros(coutp, device=DEV)
    keep, i = [], 0
    for a in (0, 1):
        for c in (0, 1):
            sel = ((1,) if a == 0 else (0, 2), (1,) if c == 0 else (0, 2))
            ws = ops.prep_conv_weight_sel(wt, dtype, sel)
            keep.append(ws)
            d = ops.make_conv_desc(x, None, ws, bp if i == 0 else zb, kernel=(3, len(sel[0]), len(sel[1])), cout=cout, split=cout, y=y, y2=None,
                                   res=y if i > 0 else None, phase_dgrad_hw=(a + 1, c + 1))
            ops.conv_launch(d)
            i += 1
    assert rel_l2(y.float(), refc.detach().permute(0, 2, 3, 4, 1)) <= tol(dtype)
    ck = 32 if dtype == torch.bfloat16 else 16
    dyw = ((cout + ck - 1) // ck) * ck
    dy = torch.zeros(N, D, Ho, Wo, dyw, device=DEV, dtype=dtype)
    dy[..., :cout] = rnd(det_normal((N, D, Ho, Wo, cout), "s2_dy").to(DEV), dtype).to(dtype)
    refc.backward(dy[..., :cout].float().permute(0, 4, 1, 2, 3))
    dx = torch.full((N, D, H, W, cin), float("nan"), device=DEV, dtype=dtype)
    zbd = torch.zeros(((cin + 31) // 32) * 32, device=DEV)
    for a in (0, 1):
        for c in (0, 1):
            sel = ((1,) if a == 0 else (2, 0), (1,) if c == 0 else (2, 0))
            wd = ops.prep_conv_weight_sel(wt, dtype, sel, flip_d=True, dgrad=True)
            keep.append(wd)
            dd = ops.make_conv_desc(dy, None, wd, zbd, kernel=(3, len(sel[0]), len(sel[1])), cout=cin, split=cin, y=dx, y2=None,
                                    phase_hw=(a + 1, c + 1))
            ops.conv_launch(dd)
    assert not torch.isnan(dx.float()).any()
    assert rel_l2(dx.float(), xr.grad.permute(0, 2, 3, 4, 1)) <= (1.5e-2 if dtype == torch.bfloat16 else 5e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["3d_concat", "1x1"])
def test_groupnorm_backward_reduction_fused_into_dgrad(ops, dtype, kind):
    """rho_conv_desc.gnb_*: a dgrad launch whose output is d act(a x + b) also writes, per tile and channel, sum dz and sum dz * x
    (dz = output * act'(a x + b)); rho_gn_bwd_finalize(fmt = 1) on those sums must give the same apply coefficients and parameter
    gradients as the separate rho_gn_bwd_reduce pass (fmt = 0) over the same tensors.  (An identity 'weight' makes the launch's
    output a chosen tensor; 3-D with a concatenated forward input, and the 1x1x1 path.)"""
    from rho_diffusion_amd import hip
    L = hip.lib()
    N, C1, C2 = 2, 64, 64
    C = C1 + C2
    shape = (N, 4, 8, 8) if kind == "3d_concat" else (N, 1, 16, 16)
    S = shape[1] * shape[2] * shape[3]
    kernel = (3, 3, 3) if kind == "3d_concat" else (1, 1, 1)
    x1 = rnd(det_normal(shape + (C1,), "gb_x1").to(DEV), dtype).to(dtype)
    x2 = rnd(det_normal(shape + (C2,), "gb_x2").to(DEV) + 0.5, dtype).to(dtype)
    dact_src = rnd(det_normal(shape + (C,), "gb_g").to(DEV), dtype).to(dtype)
    a = (1 + 0.3 * det_normal((N, C), "gb_a")).to(DEV)
    b = (0.2 * det_normal((N, C), "gb_b")).to(DEV)
    gamma, beta = (1 + 0.1 * det_normal((C,), "gb_ga")).to(DEV), (0.1 * det_normal((C,), "gb_be")).to(DEV)
    xcat = torch.cat([x1.float(), x2.float()], -1).reshape(N, S, 32, C // 32)
    mean = xcat.mean((1, 3))
    rstd = (xcat.var((1, 3), unbiased=False) + 1e-5).rsqrt()
    stats = torch.stack([mean, rstd], -1).contiguous()
    # identity weights: centre tap = I
    wt = torch.zeros((C, C) + kernel, device=DEV)
    ctr = tuple(k // 2 for k in kernel)
    wt[(torch.arange(C), torch.arange(C)) + ctr] = 1.0
    w = ops.prep_conv_weight(wt, dtype)
    dact = torch.empty(shape + (C,), device=DEV, dtype=dtype)
    d = ops.make_conv_desc(dact_src, None, w, torch.zeros(C, device=DEV), kernel=kernel, cout=C, split=C, y=dact, y2=None)
    tiles = ops.conv_stats_tiles(d)
    assert tiles > 0
    sbuf = torch.zeros(N * tiles * 2 * C, device=DEV)
    d.stats = sbuf.data_ptr()
    d.gnb_x1, d.gnb_x2, d.gnb_c1 = x1.data_ptr(), x2.data_ptr(), C1
    d.gnb_a, d.gnb_b, d.gnb_silu = a.data_ptr(), b.data_ptr(), 1
    ops.conv_launch(d)
    assert torch.equal(dact, dact_src)
    outs = []
    for fmt in (0, 1):
        work = torch.zeros(2 * N * C, device=DEV)
        cA, cP, cQ = torch.zeros(N, C, device=DEV), torch.zeros(N, 32, device=DEV), torch.zeros(N, 32, device=DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        if fmt == 0:
            nblk = ops.gn_nblk(S)
            part = torch.zeros(N * nblk * (C // 8) * 16, device=DEV)
            hip.check(L.rho_gn_bwd_reduce(dact.data_ptr(), x1.data_ptr(), C1, x2.data_ptr(), C2, hip.dtype_code(dtype), N, S, a.data_ptr(),
                                          b.data_ptr(), stats.data_ptr(), 1, part.data_ptr(), hip.stream()), "reduce")
            src, nb = part, nblk
        else:
            src, nb = sbuf, tiles
        hip.check(L.rho_gn_bwd_finalize(src.data_ptr(), N, C, S, nb, fmt, gamma.data_ptr(), beta.data_ptr(), None, 0, stats.data_ptr(),
                                        work.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, None, None, 0, cA.data_ptr(), cP.data_ptr(),
                                        cQ.data_ptr(), hip.stream()), "finalize")
        outs.append((cA, cP, cQ, dg, db))
    for u, v, nm in zip(outs[0], outs[1], ("cA", "cP", "cQ", "dgamma", "dbeta")):
        assert rel_l2(v, u) <= (2e-3 if dtype == torch.bfloat16 else 1e-4), nm
