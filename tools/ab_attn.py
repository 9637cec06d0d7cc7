"""Same-box A/B of rho_attention_fwd / rho_attention_bwd (not a product path): tools/probe/librho_head.so (AB_REF overrides; a build of the committed
tree) vs the in-tree library, c3 (T = 4096, ch = 128, B = 32) and c5 (T = 32768, ch = 64, B = 2) shapes, random data,
outputs compared.  usage: python tools/ab_attn.py"""
import ctypes as C, os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd import hip  # noqa: F401

dev = "cuda"
vp = C.c_void_p
libs = {"ref": C.CDLL(os.path.join(R0, os.environ.get("AB_REF", "tools/probe/librho_head.so"))), "new": C.CDLL(os.path.join(R0, "rho_diffusion_amd/librho_hip.so"))}
from rho_diffusion_amd.hip import check_abi
for _n, _l in libs.items():
    check_abi(_l, _n)      # a probe build with older signatures would be called with shifted arguments
for (B, T, heads, ch) in ((32, 4096, 4, 128), (2, 32768, 4, 64)):
    Cc = heads * ch
    qk = (torch.randn(B, T, 2 * Cc, device=dev) * 0.5).to(torch.bfloat16)
    vt = (torch.randn(B, Cc, T, device=dev) * 0.5).to(torch.bfloat16)
    lse = torch.empty(B, heads, T, device=dev)
    dout = (torch.randn(B, T, Cc, device=dev) * 0.1).to(torch.bfloat16)
    delta = torch.empty(B, heads, T, device=dev)
    fl_f = 4.0 * B * heads * T * T * ch
    res = {}
    for k, lib in libs.items():
        out = torch.empty(B, T, Cc, device=dev, dtype=torch.bfloat16)
        dqkv = torch.empty(B, T, 3 * Cc, device=dev, dtype=torch.bfloat16)
        f = lib.rho_attention_fwd; f.restype = C.c_int
        f.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, vp]
        g = lib.rho_attention_bwd; g.restype = C.c_int
        g.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, vp]
        st = torch.cuda.current_stream().cuda_stream
        fwd = lambda: f(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), lse.data_ptr(), 1, B, T, heads, ch, st)
        bwd = lambda: g(qk.data_ptr(), vt.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), 3 * Cc,
                        dqkv.data_ptr() + 2 * Cc * 2, 3 * Cc, 1, B, T, heads, ch, st)
        for name, fn, fl in (("fwd", fwd, fl_f), ("bwd", bwd, 3.5 * fl_f)):
            assert fn() == 0
            torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); one = time.perf_counter() - t0
            reps = max(3, int(1.0 / one))
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"T={T} ch={ch} {k} {name}: {dt * 1e3:.3f} ms ({fl / dt / 1e12:.0f} TF/s)", flush=True)
        res[k] = (out.float().clone(), dqkv.float().clone())
    print("   rel diff out %.2e dqkv %.2e" % (float((res["ref"][0] - res["new"][0]).norm() / res["ref"][0].norm()),
                                            float((res["ref"][1] - res["new"][1]).norm() / res["ref"][1].norm())), flush=True)
