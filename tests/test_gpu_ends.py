"""-m gpu: the 1-channel ends of the 3-D UNet as single launches (csrc/ends.hip: rho_stem_conv3d, rho_head_conv3d) against the CPU
oracle (oracle/ref_torch.py conv_nd on bf16-rounded operands; tolerance of the bf16 kernels: rel-L2 <= 6e-3) and against the GEMM-form
launches they replace in inference plans (rho_im2col_taps + 1x1x1 conv; 1x1x1 conv + rho_tap_gather_sum)."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import det_normal, rel_l2
from gpu_util import DEV, bf16_round, from_cl, to_cl
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from rho_diffusion_amd.engine import ops as o
    from rho_diffusion_amd import hip
    hip.load()
    return o


@pytest.mark.parametrize("cout", [32, 64])
@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (1, 5, 9, 11), (1, 64, 64, 64)], ids=["even", "ragged", "c3_64cube"])
def test_stem_conv3d_vs_oracle_and_statistics(ops, cout, shape):
    N, D, H, W = shape
    x = det_normal((N, 1, D, H, W), f"stem_x{D}")
    w = det_normal((cout, 1, 3, 3, 3), f"stem_w{cout}") / math.sqrt(27.0)
    b = det_normal((cout,), "stem_b") * 0.1
    ref = R.conv_nd(3, bf16_round(x), bf16_round(w), b, stride=1, padding=1)                  # [N, cout, D, H, W]
    wp = ops.prep_conv_weight(w.reshape(cout, 27, 1, 1, 1).to(DEV), BF16)                       # [1, coutp, 32]: the im2col form
    assert tuple(wp.shape) == (1, cout, 32), tuple(wp.shape)
    y = torch.full((N, D, H, W, cout), float("nan"), dtype=BF16, device=DEV)
    tiles = ops.stem_conv3d_tiles(D, H, W)
    assert tiles == -(-D // 4) * -(-H // 8) * -(-W // 8)
    st = torch.full((N, tiles, 2, cout), float("nan"), device=DEV)
    xd, bd = x.to(DEV), b.to(DEV)
    ops.stem_conv3d(xd, wp, bd, y, st)
    torch.cuda.synchronize()
    got = from_cl(y, 3)
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) < 6e-3, rel_l2(got, ref)
    yy = y.float().reshape(N, -1, cout)
    assert rel_l2(st[:, :, 0].sum(1), yy.sum(1)) < 1e-3 and rel_l2(st[:, :, 1].sum(1), (yy * yy).sum(1)) < 1e-3
    # the two launches it replaces (im2col to [pos][32], then the 1x1x1 GEMM): same operands, same rounding points
    from rho_diffusion_amd import hip
    xc = torch.empty(N, D, H, W, 32, dtype=BF16, device=DEV)
    hip.check(hip.lib().rho_im2col_taps(xd.data_ptr(), xc.data_ptr(), 1, N, 1, D, H, W, 3, 3, 3, 32, hip.stream()), "rho_im2col_taps")
    y2, _ = ops.conv(xc, None, wp, bd, kernel=(1, 1, 1), cout=cout)
    assert rel_l2(got, from_cl(y2, 3)) < 3e-3


@pytest.mark.parametrize("C", [32, 64, 128])
@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (1, 5, 9, 11)], ids=["even", "ragged"])
@pytest.mark.parametrize("prologue", [True, False], ids=["gn_silu", "raw"])
def test_head_conv3d_vs_oracle(ops, C, shape, prologue):
    N, D, H, W = shape
    x = bf16_round(det_normal((N, C, D, H, W), f"head_x{C}"))
    w = bf16_round(det_normal((1, C, 3, 3, 3), f"head_w{C}") / math.sqrt(27.0 * C))
    b = det_normal((1,), "head_b") * 0.1
    pre = (1 + 0.3 * det_normal((N, C), "head_a"), 0.2 * det_normal((N, C), "head_pb")) if prologue else None
    xa = x
    if pre is not None:
        xa = bf16_round(F.silu(pre[0].reshape(N, C, 1, 1, 1) * x + pre[1].reshape(N, C, 1, 1, 1)))
    ref = R.conv_nd(3, xa, w, b, stride=1, padding=1)                                          # [N, 1, D, H, W]
    # taps as rows: W[tap][c], prepared like a 1x1x1 conv with 32 output channels (the _HeadAsGemm form)
    wt = torch.zeros(32, C, 1, 1, 1)
    wt[:27, :, 0, 0, 0] = w[0].reshape(C, 27).t()
    wp = ops.prep_conv_weight(wt.to(DEV), BF16)
    assert tuple(wp.shape) == (1, 32, C)
    out = torch.full((N, 1, D, H, W), float("nan"), device=DEV)
    pa, pb = (pre[0].to(DEV), pre[1].to(DEV)) if pre else (None, None)
    ops.head_conv3d(to_cl(x, BF16), pa, pb, True, wp, b.to(DEV), out)
    torch.cuda.synchronize()
    got = out.cpu()
    assert torch.isfinite(got).all()
    # (the 27 partial sums are rounded to bf16 before they are added: the round-2 launches' rounding point)
    assert rel_l2(got, ref) < 6e-3, rel_l2(got, ref)


def test_head_conv3d_at_bench_size_matches_the_gemm_form(ops):
    """c3's head (64 channels at 64^3, one sample): against the two launches of the GEMM form on the same operands."""
    from rho_diffusion_amd import hip
    N, C, D, H, W = 1, 64, 64, 64, 64
    x = to_cl(bf16_round(det_normal((N, C, D, H, W), "headb_x")), BF16)
    w = bf16_round(det_normal((1, C, 3, 3, 3), "headb_w") / math.sqrt(27.0 * C))
    wt = torch.zeros(32, C, 1, 1, 1)
    wt[:27, :, 0, 0, 0] = w[0].reshape(C, 27).t()
    wp = ops.prep_conv_weight(wt.to(DEV), BF16)
    pa = (1 + 0.3 * det_normal((N, C), "headb_a")).to(DEV)
    pb = (0.2 * det_normal((N, C), "headb_pb")).to(DEV)
    bias = torch.full((1,), 0.25, device=DEV)
    out = torch.empty(N, 1, D, H, W, device=DEV)
    ops.head_conv3d(x, pa, pb, True, wp, bias, out)
    tt, _ = ops.conv(x, None, wp, torch.zeros(32, device=DEV), kernel=(1, 1, 1), cout=32, pre_a=pa, pre_b=pb, pre_silu=True)
    ref = torch.empty(N, 1, D * H * W, device=DEV)
    hip.check(hip.lib().rho_tap_gather_sum(tt.data_ptr(), 1, N, D, H, W, 3, 3, 3, 32, bias.data_ptr(), ref.data_ptr(), hip.stream()),
              "rho_tap_gather_sum")
    torch.cuda.synchronize()
    assert rel_l2(out.reshape(-1).cpu(), ref.reshape(-1).cpu()) < 2e-3


def test_ends_refuse_what_they_do_not_cover(ops):
    from rho_diffusion_amd.hip import RhoHipError
    x = torch.zeros(1, 1, 8, 8, 8, device=DEV)
    w = torch.zeros(1, 96, 32, dtype=BF16, device=DEV)
    with pytest.raises(RhoHipError):
        ops.stem_conv3d(x, w, torch.zeros(96, device=DEV), torch.empty(1, 8, 8, 8, 96, dtype=BF16, device=DEV))
    xh = torch.zeros(1, 8, 8, 8, 48, dtype=BF16, device=DEV)
    with pytest.raises(RhoHipError):
        ops.head_conv3d(xh, None, None, False, torch.zeros(1, 32, 48, dtype=BF16, device=DEV), None, torch.empty(1, 1, 8, 8, 8, device=DEV))
