"""Timing probe (not a product path): rho_conv_nd_wgrad from the reference build vs a TIMING-ONLY build in which every
32x32x16 MFMA is replaced by two 16x16x32 MFMAs of the same FLOPs (wrong numerics): does the weight-gradient kernel hold a
higher clock on that shape as k_conv does?  (tools/probe/libwgrad_*.so: scratch builds, not kept in the tree.)"""
import ctypes as C, os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd.engine import ops

dev = "cuda"
N = 32
cases = {"64->64 @64^3": (64, 64, 64, 64, 64), "128->128 @64x32x32": (64, 32, 32, 128, 128), "512->512 @64x8x8": (64, 8, 8, 512, 512)}
libs = {k: C.CDLL(os.path.join(R0, "tools/probe", f"libwgrad_{k}.so")) for k in ("ref", "hack16")}
for name, (D, H, W, cin, cout) in cases.items():
    x = (torch.randn(N, D, H, W, cin, device=dev) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(N, D, H, W, cout, device=dev) * 0.5).to(torch.bfloat16)
    w = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02, torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    y = torch.empty(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
    d = ops.make_conv_desc(x, None, w, b, kernel=(3, 3, 3), cout=cout, split=cout, y=y, y2=None)
    dw = torch.zeros(27, w.shape[1], cin, device=dev)
    fl = 2.0 * N * D * H * W * cin * cout * 27
    row = []
    for k, lib in libs.items():
        fn = lib.rho_conv_nd_wgrad
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        st = torch.cuda.current_stream().cuda_stream
        assert fn(C.byref(d), dy.data_ptr(), cout, dw.data_ptr(), None, st) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(C.byref(d), dy.data_ptr(), cout, dw.data_ptr(), None, st); torch.cuda.synchronize(); one = time.perf_counter() - t0
        reps = int(2.0 / one) + 1
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(C.byref(d), dy.data_ptr(), cout, dw.data_ptr(), None, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        row.append(f"{k}: {dt * 1e3:.3f} ms ({fl / dt / 1e12:.0f} TF/s)")
        dw.zero_()
    print(name, " | ".join(row), flush=True)
