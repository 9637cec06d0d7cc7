"""-m gpu: the k-split of small-grid 2-D / 1-D convolutions (rho_conv_desc.ws, conv.hip `k_splitk_reduce`) at the layer
geometries of BASELINE config c1 (2-D 64^2, batch 16: 1024 / 4096 positions at the two deepest levels) against the CPU oracle and
against the same launch without a workspace.  Tolerances as test_gpu_kernels.py (fp32 rel-L2 <= 2e-5, bf16 <= 6e-3); split and
unsplit launches differ by fp32 summation order only (<= 2e-6 fp32; bf16 outputs may differ by one rounding of the last bit)."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import det_normal, rel_l2
from gpu_util import DEV, from_cl, rnd, to_cl, tol
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from rho_diffusion_amd.engine import ops as o
    from rho_diffusion_amd import hip
    hip.load()
    return o


CASES = [
    # name, dims, N, c1, c2, cout, spatial, stride, up, prologue, residual, res_add
    ("c1_mid_512", 2, 16, 512, 0, 512, (8, 8), 1, False, True, True, True),
    ("c1_up_1024_512", 2, 16, 512, 512, 512, (8, 8), 1, False, True, False, True),
    ("c1_lvl2_256", 2, 16, 256, 0, 256, (16, 16), 1, False, True, True, False),
    ("c1_down_256", 2, 16, 256, 0, 256, (16, 16), 2, False, False, False, False),
    ("c1_up_loader", 2, 4, 256, 0, 256, (8, 8), 1, True, False, False, False),
    ("ragged_2d", 2, 3, 192, 64, 128, (7, 9), 1, False, True, True, True),
    ("1d_long_k", 1, 2, 512, 0, 128, (200,), 1, False, True, True, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_ksplit_vs_oracle_and_unsplit(ops, dtype, case):
    name, dims, N, c1, c2, cout, spatial, stride, up, prologue, residual, use_add = case
    cin = c1 + c2
    x1 = rnd(det_normal((N, c1, *spatial), name + "x1"), dtype)
    x2 = rnd(det_normal((N, c2, *spatial), name + "x2"), dtype) if c2 else None
    w = rnd(det_normal((cout, cin) + (3,) * dims, name + "w") / math.sqrt(cin * 3 ** dims), dtype)
    b = det_normal((cout,), name + "b") * 0.1
    pre = (1 + 0.3 * det_normal((N, cin), name + "a"), 0.2 * det_normal((N, cin), name + "pb")) if prologue else None
    xx = torch.cat([x1, x2], 1) if c2 else x1
    if pre is not None:
        shp = (N, cin) + (1,) * dims
        xx = rnd(F.silu(pre[0].reshape(shp) * xx + pre[1].reshape(shp)), dtype)
    if up:
        xx = R.upsample(dims, xx)
    ref = R.conv_nd(dims, xx, w, b, stride=stride, padding=1)
    add = det_normal((N, cout), name + "add") if use_add else None
    if use_add:
        ref = ref + add.reshape((N, cout) + (1,) * dims)
    res = rnd(det_normal(tuple(ref.shape), name + "r"), dtype) if residual else None
    if residual:
        ref = ref + res
    kernel = (1,) * (3 - dims) + (3,) * dims
    stride_hw = ((stride, stride) if dims == 2 else (1, stride))
    up_hw = ((1, 1) if dims == 2 else (0, 1)) if up else (0, 0)
    x1d, x2d = to_cl(x1, dtype), (to_cl(x2, dtype) if c2 else None)
    wp, bp = ops.prep_conv_weight(w.to(DEV), dtype), b.to(DEV)
    N_, Do, Ho, Wo = ops.conv_out_shape(x1d.shape, kernel, stride_hw, up_hw)
    # (the descriptor holds raw pointers: every operand stays referenced here until the launch has run)
    pa, pb = (pre[0].to(DEV), pre[1].to(DEV)) if pre else (None, None)
    resd, addd = (to_cl(res, dtype) if residual else None), (add.to(DEV) if use_add else None)
    outs = []
    for with_ws in (True, False):
        y = torch.full((N_, Do, Ho, Wo, cout), float("nan"), dtype=dtype, device=DEV)
        d = ops.make_conv_desc(x1d, x2d, wp, bp, kernel=kernel, cout=cout, split=cout, y=y, y2=None, stride_hw=stride_hw, up_hw=up_hw,
                               pre_a=pa, pre_b=pb, pre_silu=True, res=resd, res_add=addd)
        want = ops.conv_workspace_bytes(d)
        assert want > 0 and want % (y.numel() * 4) == 0, (name, want)         # whole fp32 slabs, at least two
        assert want // (y.numel() * 4) >= 2
        ws = None
        if with_ws:
            ws = ops.attach_conv_workspace([d], DEV)
            assert ws is not None and d.ws_bytes == want
            ws.fill_(0xFF)                                                    # NaN bit patterns: every slab element must be written
        ops.conv_launch(d)
        torch.cuda.synchronize()
        got = from_cl(y, dims)
        assert got.shape == ref.shape
        assert torch.isfinite(got).all(), (name, with_ws, torch.isnan(got).float().mean().item(), torch.isnan(got).any(1).float().mean().item())
        assert rel_l2(got, ref) < tol(dtype), (name, with_ws)
        outs.append(got)
    assert rel_l2(outs[0], outs[1]) < (2e-6 if dtype == torch.float32 else 3e-3), name


def test_conv_ksplit_smaller_workspace_lowers_the_split_and_is_reproducible(ops):
    """A workspace of two slabs still splits (by two); one slab does not split; repeated launches are bit-identical (fixed order)."""
    dtype = torch.float32
    N, c, sp = 16, 512, (8, 8)
    x = to_cl(det_normal((N, c, *sp), "wsx"), dtype)
    w = det_normal((c, c, 3, 3), "wsw") / math.sqrt(c * 9)
    wp, bp = ops.prep_conv_weight(w.to(DEV), dtype), torch.zeros(c, device=DEV)
    ref = R.conv_nd(2, det_normal((N, c, *sp), "wsx"), w, torch.zeros(c), stride=1, padding=1)
    got = []
    for slabs in (16, 2, 1, 16):
        y = torch.empty(N, 1, *sp, c, dtype=dtype, device=DEV)
        d = ops.make_conv_desc(x, None, wp, bp, kernel=(1, 3, 3), cout=c, split=c, y=y, y2=None)
        assert ops.conv_workspace_bytes(d) == 16 * y.numel() * 4
        ws = torch.empty(slabs * y.numel() * 4, dtype=torch.uint8, device=DEV)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
        ops.conv_launch(d)
        torch.cuda.synchronize()
        got.append(from_cl(y, 2))
        assert rel_l2(got[-1], ref) < 2e-5
    assert torch.equal(got[0], got[3])
    assert rel_l2(got[0], got[2]) < 2e-6 and rel_l2(got[1], got[2]) < 2e-6


ONE_CASES = [
    # name, N, c1, c2, cout, spatial, prologue, residual
    ("c1_skip_1024_512", 16, 512, 512, 512, (8, 8), False, True),
    ("c1_proj_512", 16, 512, 0, 512, (8, 8), False, True),
    ("c1_skip_768_256", 16, 512, 256, 256, (16, 16), False, True),
    ("pre_one_sample_tiles", 4, 256, 0, 128, (16, 16), True, False),      # 256-position tiles inside one sample: LDS coefficients
    ("pre_ragged", 3, 192, 64, 128, (7, 9), True, True),                  # tiles straddle samples: coefficients fetched per row
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ONE_CASES, ids=[c[0] for c in ONE_CASES])
def test_conv1x1_ksplit_vs_oracle_and_unsplit(ops, dtype, case):
    name, N, c1, c2, cout, spatial, prologue, residual = case
    cin = c1 + c2
    x1 = rnd(det_normal((N, c1, *spatial), name + "x1"), dtype)
    x2 = rnd(det_normal((N, c2, *spatial), name + "x2"), dtype) if c2 else None
    w = rnd(det_normal((cout, cin, 1, 1), name + "w") / math.sqrt(cin), dtype)
    b = det_normal((cout,), name + "b") * 0.1
    pre = (1 + 0.3 * det_normal((N, cin), name + "a"), 0.2 * det_normal((N, cin), name + "pb")) if prologue else None
    xx = torch.cat([x1, x2], 1) if c2 else x1
    if pre is not None:
        xx = rnd(F.silu(pre[0].reshape(N, cin, 1, 1) * xx + pre[1].reshape(N, cin, 1, 1)), dtype)
    ref = R.conv_nd(2, xx, w, b, stride=1, padding=0)
    res = rnd(det_normal(tuple(ref.shape), name + "r"), dtype) if residual else None
    if residual:
        ref = ref + res
    x1d, x2d = to_cl(x1, dtype), (to_cl(x2, dtype) if c2 else None)
    wp, bp = ops.prep_conv_weight(w.to(DEV), dtype), b.to(DEV)
    pa, pb = (pre[0].to(DEV), pre[1].to(DEV)) if pre else (None, None)
    resd = to_cl(res, dtype) if residual else None
    outs = []
    for with_ws in (True, False):
        y = torch.full((N, 1, *spatial, cout), float("nan"), dtype=dtype, device=DEV)
        d = ops.make_conv_desc(x1d, x2d, wp, bp, kernel=(1, 1, 1), cout=cout, split=cout, y=y, y2=None, pre_a=pa, pre_b=pb, pre_silu=True, res=resd)
        want = ops.conv_workspace_bytes(d)
        assert want >= 2 * y.numel() * 4 and want % (y.numel() * 4) == 0, (name, want)
        if with_ws:
            ws = ops.attach_conv_workspace([d], DEV)
            ws.fill_(0xFF)
        ops.conv_launch(d)
        torch.cuda.synchronize()
        got = from_cl(y, 2)
        assert torch.isfinite(got).all(), (name, with_ws)
        assert rel_l2(got, ref) < tol(dtype), (name, with_ws)
        outs.append(got)
    assert rel_l2(outs[0], outs[1]) < (2e-6 if dtype == torch.float32 else 3e-3), name


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,cin,cout", [((16, 1, 8, 8), 512, 512), ((3, 1, 9, 11), 256, 128)], ids=["c1_up_512", "ragged"])
def test_upsample_phases_ksplit(ops, dtype, shape, cin, cout):
    """The four sub-pixel phase launches of Upsample + conv (c1: 512 -> 512 from 8^2 to 16^2, 16 workgroups each) with a workspace:
    each phase's reduce scatters to its own parity of the full-resolution output; against interpolate + conv and the unsplit launches."""
    N, D, H, W = shape
    x = rnd(det_normal((N, D, H, W, cin), "sp_x").to(DEV), dtype).to(dtype)
    wt = det_normal((cout, cin, 1, 3, 3), "sp_w").to(DEV) * 0.02
    b = det_normal((cout,), "sp_b").to(DEV)
    res = rnd(det_normal((N, D, 2 * H, 2 * W, cout), "sp_r").to(DEV), dtype).to(dtype)
    xu = F.interpolate(x.float().permute(0, 4, 1, 2, 3), size=(D, 2 * H, 2 * W), mode="nearest")
    ref = F.conv3d(xu, wt, b, padding=(0, 1, 1)).permute(0, 2, 3, 4, 1) + res.float()
    outs = []
    for with_ws in (True, False):
        y = torch.full((N, D, 2 * H, 2 * W, cout), float("nan"), device=DEV, dtype=dtype)
        descs, keep = [], []
        for a in (1, 2):
            for c in (1, 2):
                wp = ops.prep_conv_weight_phase(wt, dtype, (a, c))
                keep.append(wp)
                descs.append(ops.make_conv_desc(x, None, wp, b, kernel=(1, 2, 2), cout=cout, split=cout, y=y, y2=None, res=res, phase_hw=(a, c)))
        assert all(ops.conv_workspace_bytes(d) > 0 for d in descs)
        if with_ws:
            ws = ops.attach_conv_workspace(descs, DEV)
            ws.fill_(0xFF)
        for d in descs:
            ops.conv_launch(d)
        torch.cuda.synchronize()
        assert not torch.isnan(y.float()).any()
        assert rel_l2(y.float(), ref) <= tol(dtype)
        outs.append(y.float())
    assert rel_l2(outs[0], outs[1]) < (2e-6 if dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kernel", [(1, 3, 3), (1, 1, 1)], ids=["3x3", "1x1"])
def test_conv_ksplit_two_channels_last_outputs(ops, dtype, kernel):
    """The data gradient of a conv whose input was a concat writes two channels-last tensors (rho_conv_desc.y2_cl, in-place
    accumulation through res / res2): c1's 512 + 512 -> 512 up-path convs at 8^2.  Output widths 512 + 212: the last 32-wide cout tile is padded."""
    N, cin, sp, w1, w2 = 16, 512, (8, 8), 512, 212
    cout = w1 + w2
    k = kernel[1]
    x = rnd(det_normal((N, cin, *sp), "y2x"), dtype)
    w = rnd(det_normal((cout, cin, k, k), "y2w") / math.sqrt(cin * k * k), dtype)
    ref = R.conv_nd(2, x, w, torch.zeros(cout), stride=1, padding=k // 2)
    r1, r2 = rnd(det_normal((N, w1, *sp), "y2r1"), dtype), rnd(det_normal((N, w2, *sp), "y2r2"), dtype)
    ref1, ref2 = ref[:, :w1] + r1, ref[:, w1:] + r2
    xd, wp = to_cl(x, dtype), ops.prep_conv_weight(w.to(DEV), dtype)
    zb = torch.zeros(wp.shape[1], device=DEV)
    r1d, r2d = to_cl(r1, dtype), to_cl(r2, dtype)
    outs = []
    for with_ws in (True, False):
        y = torch.full((N, 1, *sp, w1), float("nan"), dtype=dtype, device=DEV)
        y2 = torch.full((N, 1, *sp, w2), float("nan"), dtype=dtype, device=DEV)
        d = ops.make_conv_desc(xd, None, wp, zb, kernel=kernel, cout=cout, split=w1, y=y, y2=y2, y2_cl=True, res=r1d, res2=r2d)
        assert ops.conv_workspace_bytes(d) >= 2 * N * 64 * wp.shape[1] * 4
        if with_ws:
            ws = ops.attach_conv_workspace([d], DEV)
            ws.fill_(0xFF)
        ops.conv_launch(d)
        torch.cuda.synchronize()
        g1, g2 = from_cl(y, 2), from_cl(y2, 2)
        assert torch.isfinite(g1).all() and torch.isfinite(g2).all()
        assert rel_l2(g1, ref1) < tol(dtype) and rel_l2(g2, ref2) < tol(dtype)
        outs.append((g1, g2))
    lim = 2e-6 if dtype == torch.float32 else 3e-3
    assert rel_l2(outs[0][0], outs[1][0]) < lim and rel_l2(outs[0][1], outs[1][1]) < lim


def test_conv_ksplit_is_not_offered_where_it_does_not_apply(ops):
    """3-D kernels, channel-major second outputs, short contractions and grids that fill the chip report no workspace."""
    dtype = torch.bfloat16

    def want(shape4, cin, cout, kernel, split=None):
        nd = 3 if kernel[0] > 1 else (2 if kernel[1] > 1 else 1)
        x = torch.zeros(*shape4, cin, dtype=dtype, device=DEV)
        wp = ops.prep_conv_weight(torch.zeros((cout, cin) + kernel[3 - nd:], device=DEV), dtype)
        split = cout if split is None else split
        N, Do, Ho, Wo = ops.conv_out_shape(x.shape, kernel, (1, 1), (0, 0))
        y = torch.empty(N, Do, Ho, Wo, split, dtype=dtype, device=DEV) if split else None
        y2 = torch.empty(N, cout - split, Do * Ho * Wo, dtype=dtype, device=DEV) if split < cout else None
        d = ops.make_conv_desc(x, None, wp, torch.zeros(wp.shape[1], device=DEV), kernel=kernel, cout=cout, split=split, y=y, y2=y2)
        return ops.conv_workspace_bytes(d)

    assert want((16, 1, 8, 8), 512, 512, (1, 3, 3)) > 0
    assert want((2, 8, 8, 8), 512, 512, (3, 3, 3)) == 0               # 3-D: the batch axis is grid z already
    assert want((16, 1, 8, 8), 512, 512, (1, 1, 1)) > 0               # 1x1x1 path: split too (>= 4 chunks per split)
    assert want((16, 1, 8, 8), 128, 512, (1, 1, 1)) == 0              # four bf16 chunks: one split's worth
    assert want((16, 1, 8, 8), 512, 512, (1, 3, 3), split=256) == 0   # channel-major second output
    assert want((16, 1, 64, 64), 128, 128, (1, 3, 3)) == 0            # 256 tiles: the grid fills the chip
    assert want((16, 1, 8, 8), 64, 512, (1, 3, 3)) == 0               # two bf16 chunks: nothing to split
