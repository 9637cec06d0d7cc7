"""Timing probe (not a product path): rho_conv_nd_fwd from debug builds of conv.hip with parts compiled out
(RHO_DBG bits: 1 no epilogue, 2 no halo global loads, 4 no prologue, 8 no weight global loads, 16 no MFMA).
The debug libraries (tools/probe/libconv_dbg<bits>.so, not kept in the tree) were built from a scratch copy of conv.hip
with `if (RHO_DBG & bit)` guards around the halo loads, apply_pre, the weight loads, the MFMA calls and the epilogue
(stores kept behind an impossible runtime condition so nothing is dead code): hipcc -DRHO_DBG=<bits> -shared -fPIC.
Results of the run that guided round 1 are quoted in DESIGN.md section 5."""
import ctypes as C, glob, os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd import hip
from rho_diffusion_amd.engine import ops

dev = "cuda"
cases = {"narrow 64->64 @64^3": (32, 64, 64, 64, 64, 64), "mid 128->64 @64^3": (32, 64, 64, 64, 128, 64),
         "wide 128->128 @64x32x32": (32, 64, 32, 32, 128, 128), "wide 512->512 @64x8x8": (32, 64, 8, 8, 512, 512)}
libs = sorted(glob.glob(os.path.join(R0, "tools/probe/libconv_dbg*.so")), key=lambda p: int(p.split("dbg")[-1][:-3]))
for name, (N, D, H, W, cin, cout) in cases.items():
    x = (torch.randn(N, D, H, W, cin, device=dev) * 0.5).to(torch.bfloat16)
    w = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02, torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    a = torch.ones(N, cin, device=dev); bb = torch.zeros(N, cin, device=dev)
    res = torch.zeros(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
    y = torch.empty(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
    use_res = os.environ.get("RHO_PROBE_RES", "1") != "0"
    d = ops.make_conv_desc(x, None, w, b, kernel=(3, 3, 3), cout=cout, split=cout, y=y, y2=None, pre_a=a, pre_b=bb, pre_silu=True,
                           res=res if use_res else None)
    fl = 2.0 * N * D * H * W * cin * cout * 27
    row = []
    for lp in libs:
        lib = C.CDLL(lp)
        fn = lib.rho_conv_nd_fwd
        fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            rc = fn(C.byref(d), st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn(C.byref(d), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        row.append((int(lp.split("dbg")[-1][:-3]), dt * 1e3, fl / dt / 1e12))
    print(name, " | ".join(f"dbg{k}: {ms:.3f} ms ({tf:.0f} TF/s)" for k, ms, tf in row), flush=True)
