"""Diagnostic (not a product path): in-kernel clock of k_conv under sustained load.  tools/probe/libconv_clk.so is a scratch build
of conv.hip in which thread 0 of every workgroup stamps s_memtime / s_memrealtime (100 MHz) at kernel entry and before the
epilogue into a buffer of its own; clock = d(memtime) / d(memrealtime) * 100 MHz, median over workgroups, after >= 2 s of
back-to-back launches on random data."""
import ctypes as C, os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd.engine import ops

dev = "cuda"
lib = C.CDLL(os.path.join(R0, "tools/probe", os.environ.get("RHO_CLK_LIB", "libconv_clk.so")))
fn = lib.rho_conv_nd_fwd
fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
lib.rho_dbg_set_buf.argtypes = [C.c_void_p]; lib.rho_dbg_set_buf.restype = C.c_int
buf = torch.zeros(2 * 65536, dtype=torch.int64, device=dev)
assert lib.rho_dbg_set_buf(buf.data_ptr()) == 0
N = 32
cases = {"narrow 64->64 @64^3": (64, 64, 64, 64, 64), "wide 128->128 @64x32x32": (64, 32, 32, 128, 128),
         "wide 512->512 @64x8x8": (64, 8, 8, 512, 512)}
for name, (D, H, W, cin, cout) in cases.items():
    x = (torch.randn(N, D, H, W, cin, device=dev) * 0.5).to(torch.bfloat16)
    w = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02, torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    a = torch.ones(N, cin, device=dev); bb = torch.zeros(N, cin, device=dev)
    y = torch.empty(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
    d = ops.make_conv_desc(x, None, w, b, kernel=(3, 3, 3), cout=cout, split=cout, y=y, y2=None, pre_a=a, pre_b=bb, pre_silu=True)
    st = torch.cuda.current_stream().cuda_stream
    fn(C.byref(d), st); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(C.byref(d), st); torch.cuda.synchronize(); one = time.perf_counter() - t0
    reps = int(2.5 / one) + 1
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(C.byref(d), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    v = buf.view(-1, 2).cpu()
    v = v[v[:, 1] > 0].double()
    clk = (v[:, 0] / v[:, 1] * 100.0)
    fl = 2.0 * N * D * H * W * cin * cout * 27
    print(f"{name}: {dt * 1e3:.3f} ms {fl / dt / 1e12:.0f} TF/s | in-kernel clock median {clk.median():.0f} MHz (p10 {clk.quantile(0.1):.0f}, p90 {clk.quantile(0.9):.0f}), "
          f"workgroup main-loop time median {v[:, 1].median() * 10:.0f} ns, n={len(v)}", flush=True)
    buf.zero_()
