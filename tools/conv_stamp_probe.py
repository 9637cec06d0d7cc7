"""Timing probe (not a product path): phase timestamps of k_conv from an instrumented scratch build (tools/ab_libs/libconv_stamp.so:
wall_clock64() of thread 0 at: 0 start, 1 slots decoded, 2 weight pipeline filled, 3 first halo chunk staged, 4 tap loops done,
5 accumulators staged for the epilogue, 6 outputs stored) - where the per-tile fixed time of a 3x3x3 launch goes.
usage (GPU box): RHO_HIP_LIB=tools/ab_libs/libconv_stamp.so RHO_CONV_NPER=1 python tools/conv_stamp_probe.py"""
import ctypes as C, os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import numpy as np
import torch
from rho_diffusion_amd import hip
from rho_diffusion_amd.engine import ops

dev = "cuda"
lib = hip.lib()
for name, (N, D, H, W, cin, cout, pre) in {"128->128 @64x32x32 plain": (32, 64, 32, 32, 128, 128, False), "128->128 pre": (32, 64, 32, 32, 128, 128, True),
                                           "64->64 @64^3 plain": (32, 64, 64, 64, 64, 64, False), "64->64 pre": (32, 64, 64, 64, 64, 64, True),
                                           "512->512 @64x8x8": (32, 64, 8, 8, 512, 512, False)}.items():
    x = (torch.randn(N, D, H, W, cin, device=dev) * 0.5).to(torch.bfloat16)
    w = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02, torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    y = torch.empty(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
    kw = dict(pre_a=torch.ones(N, cin, device=dev), pre_b=torch.zeros(N, cin, device=dev), pre_silu=True) if pre else {}
    d = ops.make_conv_desc(x, None, w, b, kernel=(3, 3, 3), cout=cout, split=cout, y=y, y2=None, **kw)
    for _ in range(3):
        ops.conv_launch(d)
    torch.cuda.synchronize()
    buf = np.zeros((8192, 8), dtype=np.uint64)
    assert lib.rho_dbg_stamps(C.c_void_p(buf.ctypes.data)) == 0
    st = buf[1024:4096, :7].astype(np.float64) * 10.0 / 1000.0      # 100 MHz ticks -> microseconds; workgroups in the steady state
    dl = np.diff(st, axis=1)
    tot = st[:, 6] - st[:, 0]
    print(name, "us: decode %.2f | wfill %.2f | first halo %.2f | taps %.2f | epi stage %.2f | store %.2f | total %.2f (median over workgroups)" %
          (*np.median(dl, axis=0), np.median(tot)), flush=True)
