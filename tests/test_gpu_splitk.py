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


def test_conv_ksplit_is_not_offered_where_it_does_not_apply(ops):
    """3-D kernels, 1x1, channel-major second outputs and grids that fill the chip report no workspace."""
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
    assert want((16, 1, 8, 8), 512, 512, (1, 1, 1)) == 0              # 1x1x1 path
    assert want((16, 1, 8, 8), 512, 512, (1, 3, 3), split=256) == 0   # channel-major second output
    assert want((16, 1, 64, 64), 128, 128, (1, 3, 3)) == 0            # 256 tiles: the grid fills the chip
    assert want((16, 1, 8, 8), 64, 512, (1, 3, 3)) == 0               # two bf16 chunks: nothing to split
