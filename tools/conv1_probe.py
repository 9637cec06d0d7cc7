"""Timing probe (not a product path) of the 1x1x1 path of k_conv from debug builds with parts compiled out (RHO_DBG bits:
1 no epilogue, 2 no activation global loads, 4 no prologue, 8 no weight global loads, 16 no MFMA; tools/probe/libconv_dbg<bits>.so
built from a scratch copy of conv.hip, not kept in the tree)."""
import ctypes as C, glob, os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd.engine import ops

dev = "cuda"
N = 32
# name: (D, H, W, cin, cout, split, pre, res)
cases = {"qkv 512->1536 T=4096": (1, 1, 4096, 512, 1536, 1024, True, False),
         "qkv-like, all channels-last": (1, 1, 4096, 512, 1536, 1536, True, False),
         "qkv-like, no prologue": (1, 1, 4096, 512, 1536, 1024, False, False),
         "proj 512->512 T=4096": (1, 1, 4096, 512, 512, 512, False, True),
         "skip 1024->512 @64x8x8": (64, 8, 8, 1024, 512, 512, False, False),
         "skip 192->64 @64^3": (64, 64, 64, 192, 64, 64, False, False),
         "skip 128->64 @64^3": (64, 64, 64, 128, 64, 64, False, False),
         "dgrad-like 64->192 @64^3 (accumulating)": (64, 64, 64, 64, 192, 192, False, True)}
libs = sorted(glob.glob(os.path.join(R0, "tools/probe/libconv_dbg*.so")), key=lambda p: int(p.split("dbg")[-1][:-3]))
for name, (D, H, W, cin, cout, split, pre, res_) in cases.items():
    x = (torch.randn(N, D, H, W, cin, device=dev) * 0.5).to(torch.bfloat16)
    w = ops.prep_conv_weight(torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.02, torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    a = torch.ones(N, cin, device=dev) if pre else None
    bb = torch.zeros(N, cin, device=dev) if pre else None
    res = torch.zeros(N, D, H, W, split, device=dev, dtype=torch.bfloat16) if res_ else None
    y = torch.empty(N, D, H, W, split, device=dev, dtype=torch.bfloat16)
    y2 = torch.empty(N, cout - split, D * H * W, device=dev, dtype=torch.bfloat16) if split < cout else None
    d = ops.make_conv_desc(x, None, w, b, kernel=(1, 1, 1), cout=cout, split=split, y=y, y2=y2, pre_a=a, pre_b=bb, pre_silu=False, res=res)
    fl = 2.0 * N * D * H * W * cin * cout
    row = []
    for lp in libs:
        lib = C.CDLL(lp)
        fn = lib.rho_conv_nd_fwd
        fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            rc = fn(C.byref(d), st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn(C.byref(d), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        row.append((int(lp.split("dbg")[-1][:-3]), dt * 1e3, fl / dt / 1e12))
    print(name, " | ".join(f"dbg{k}: {ms:.3f} ms ({tf:.0f} TF/s)" for k, ms, tf in row), flush=True)
