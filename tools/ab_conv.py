"""Same-box probe of rho_conv_nd_fwd (not a product path): the in-tree library (and optional extra builds, AB_EXTRA) on the c3
layer shapes, on random and on ALL-ZERO operands.  A kernel that runs markedly faster on zeros is limited by the clock the chip
holds under load (power), not by its instruction stream (MI355X_MICROARCH.md, 'DVFS give-back').  usage: python tools/ab_conv.py [B]"""
import ctypes as C, os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd.engine import ops

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
UP = {"128->128 up @64x32x32": (64, 32, 32, 128, 128), "512->512 up @64x8x8": (64, 8, 8, 512, 512)}
cases = {"64->64 @64^3 +pre": (64, 64, 64, 64, 64, True), "64->64 @64^3": (64, 64, 64, 64, 64, False), "192->64 @64^3 +pre": (64, 64, 64, 192, 64, True),
         "128->128 @64x32x32 +pre": (64, 32, 32, 128, 128, True), "128->128 @64x32x32": (64, 32, 32, 128, 128, False),
         "256->256 @64x16x16 +pre": (64, 16, 16, 256, 256, True), "512->512 @64x8x8 +pre": (64, 8, 8, 512, 512, True),
         "512->512 @64x8x8": (64, 8, 8, 512, 512, False), "1024->512 @64x8x8 +pre": (64, 8, 8, 1024, 512, True),
         "768->256 @64x16x16 +pre": (64, 16, 16, 768, 256, True)}
libs = {"tree": C.CDLL(os.path.join(R0, "rho_diffusion_amd/librho_hip.so"))}
libs["tree"].rho_conv_stats_tiles.restype = C.c_int64
libs["tree"].rho_conv_stats_tiles.argtypes = [C.c_void_p]
for extra in os.environ.get("AB_EXTRA", "").split():
    libs[os.path.basename(extra)] = C.CDLL(os.path.join(R0, extra))
from rho_diffusion_amd.hip import check_abi
for _n, _l in libs.items():
    check_abi(_l, _n)      # a probe build with older signatures would be called with shifted arguments
cases.update({k: v + (False,) for k, v in UP.items()})
for name, (D, H, W, cin, cout, pre) in cases.items():
    up = name in UP
    row = []
    for zero in (False, True):
        sc = 0.0 if zero else 0.5
        x = (torch.randn(N, D, H, W, cin, device=dev) * sc).to(torch.bfloat16)
        w = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * (0.0 if zero else 0.02), torch.bfloat16)
        b = torch.zeros(cout, device=dev)
        a_ = (1 + 0.3 * torch.randn(N, cin, device=dev)) * (0.0 if zero else 1.0)
        b_ = 0.2 * torch.randn(N, cin, device=dev) * (0.0 if zero else 1.0)
        m = 2 if up else 1
        y = torch.empty(N, D, H * m, W * m, cout, device=dev, dtype=torch.bfloat16)
        feats = os.environ.get("AB_FEATS", "")                      # "s" = fused output statistics, "r" = residual input
        res = torch.randn_like(y) if "r" in feats else None
        d = ops.make_conv_desc(x, None, w, b, kernel=(3, 3, 3), cout=cout, split=cout, y=y, y2=None, pre_a=a_ if pre else None,
                               pre_b=b_ if pre else None, pre_silu=pre, up_hw=(1, 1) if up else (0, 0), res=res)
        if "s" in feats:
            tiles = int(libs["tree"].rho_conv_stats_tiles(C.byref(d)))
            sbuf = torch.empty(N * tiles * 2 * cout, device=dev)
            d.stats = sbuf.data_ptr()
        fl = 2.0 * N * D * H * W * m * m * cin * cout * 27
        for k, lib in libs.items():
            fn = lib.rho_conv_nd_fwd
            fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
            st = torch.cuda.current_stream().cuda_stream
            assert fn(C.byref(d), st) == 0
            for _ in range(3):
                fn(C.byref(d), st)
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                fn(C.byref(d), st)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            row.append(f"{k}{'/zeros' if zero else '/random'}: {dt * 1e3:.3f} ms ({fl / dt / 1e12:.0f} TF/s)")
    print(name, " | ".join(row), flush=True)
