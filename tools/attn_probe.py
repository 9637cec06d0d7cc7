"""Timing probe (not a product path): attention forward / backward from the reference build vs a TIMING-ONLY build in which
every 32x32x16 MFMA is two 16x16x32 MFMAs of the same FLOPs (wrong numerics)."""
import ctypes as C, os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd import hip  # noqa: F401  (loads the product library first: same HIP runtime state)

dev = "cuda"
B, T, heads, ch = 32, 4096, 4, 128
Cc = heads * ch
qk = (torch.randn(B, T, 2 * Cc, device=dev) * 0.5).to(torch.bfloat16)
vt = (torch.randn(B, Cc, T, device=dev) * 0.5).to(torch.bfloat16)
out = torch.empty(B, T, Cc, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, heads, T, device=dev)
dout = (torch.randn(B, T, Cc, device=dev) * 0.1).to(torch.bfloat16)
dqkv = torch.empty(B, T, 3 * Cc, device=dev, dtype=torch.bfloat16)
delta = torch.empty(B, heads, T, device=dev)
vp = C.c_void_p
fl_f = 4.0 * B * heads * T * T * ch
for k in ("ref", "hack16"):
    lib = C.CDLL(os.path.join(R0, "tools/probe", f"libattn_{k}.so"))
    f = lib.rho_attention_fwd; f.restype = C.c_int
    f.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, vp]
    g = lib.rho_attention_bwd; g.restype = C.c_int
    g.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, vp]
    st = torch.cuda.current_stream().cuda_stream
    fwd = lambda: f(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), lse.data_ptr(), 1, B, T, heads, ch, st)
    bwd = lambda: g(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), 3 * Cc,
                    dqkv.data_ptr() + 2 * Cc * 2, 3 * Cc, 1, B, T, heads, ch, st)
    for name, fn, fl in (("fwd", fwd, fl_f), ("bwd", bwd, 2.5 * fl_f)):
        assert fn() == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); one = time.perf_counter() - t0
        reps = int(2.0 / one) + 1
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{k} {name}: {dt * 1e3:.3f} ms ({fl / dt / 1e12:.0f} TF/s)", flush=True)
