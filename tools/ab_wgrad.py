"""Same-box A/B of rho_conv_nd_wgrad between two builds of the library (not a product path):
tools/probe/librho_head.so (a build of the committed tree; AB_REF overrides) vs the in-tree librho_hip.so, alternated per case on random data,
with the two weight gradients compared.  usage (GPU box): python tools/ab_wgrad.py [B]"""
import ctypes as C, os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd.engine import ops

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cases = {"64->64 @64^3": (64, 64, 64, 64, 64), "192->64 @64^3": (64, 64, 64, 192, 64), "128->128 @64x32x32": (64, 32, 32, 128, 128),
         "256->256 @64x16x16": (64, 16, 16, 256, 256), "512->512 @64x8x8": (64, 8, 8, 512, 512), "1024->512 @64x8x8": (64, 8, 8, 1024, 512)}
libs = {"ref": C.CDLL(os.path.join(R0, os.environ.get("AB_REF", "tools/probe/librho_head.so"))), "new": C.CDLL(os.path.join(R0, "rho_diffusion_amd/librho_hip.so"))}
for extra in os.environ.get("AB_EXTRA", "").split():          # e.g. AB_EXTRA=tools/probe/libwgrad_m16.so (timing-only probes)
    libs[os.path.basename(extra)] = C.CDLL(os.path.join(R0, extra))
from rho_diffusion_amd.hip import check_abi
for _n, _l in libs.items():
    check_abi(_l, _n)      # a probe build with older signatures would be called with shifted arguments
ZERO = os.environ.get("AB_ZERO") == "1"                       # all-zero operands: no data-dependent power draw (clock probe)
tot = {k: 0.0 for k in libs}
for name, (D, H, W, cin, cout) in cases.items():
    x = (torch.randn(N, D, H, W, cin, device=dev) * (0.0 if ZERO else 0.5)).to(torch.bfloat16)
    dy = (torch.randn(N, D, H, W, cout, device=dev) * (0.0 if ZERO else 0.5)).to(torch.bfloat16)
    w = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02, torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    y = torch.empty(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
    d = ops.make_conv_desc(x, None, w, b, kernel=(3, 3, 3), cout=cout, split=cout, y=y, y2=None)
    fl = 2.0 * N * D * H * W * cin * cout * 27
    row, outs = [], {}
    for k, lib in libs.items():
        fn = lib.rho_conv_nd_wgrad
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]; fn.restype = C.c_int
        st = torch.cuda.current_stream().cuda_stream
        dw = torch.zeros(27, w.shape[1], cin, device=dev)
        db = torch.zeros(w.shape[1], device=dev)
        assert fn(C.byref(d), dy.data_ptr(), cout, dw.data_ptr(), db.data_ptr(), st) == 0
        torch.cuda.synchronize()
        outs[k] = (dw.clone(), db.clone())
        for _ in range(3):
            fn(C.byref(d), dy.data_ptr(), cout, dw.data_ptr(), db.data_ptr(), st)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(C.byref(d), dy.data_ptr(), cout, dw.data_ptr(), db.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tot[k] += dt
        row.append(f"{k}: {dt * 1e3:.3f} ms ({fl / dt / 1e12:.0f} TF/s)")
    e = float((outs["ref"][0] - outs["new"][0]).norm() / (outs["ref"][0].norm() + 1e-30))
    eb = float((outs["ref"][1] - outs["new"][1]).norm() / (outs["ref"][1].norm() + 1e-30))
    print(name, " | ".join(row), f"| rel diff dw {e:.2e} db {eb:.2e}", flush=True)
print("sum", {k: round(v * 1e3, 2) for k, v in tot.items()})
