"""Timing probe (not a product path): what each fused extra of a 3x3x3 forward launch costs on the 64-channel level and the
128-channel level: prologue (GroupNorm affine + SiLU in the loader), residual, per-sample additive embedding, fused output statistics.
usage (GPU box): python tools/conv_epi_probe.py"""
import os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd.engine import ops

dev = "cuda"
for name, (N, D, H, W, cin, cout) in {"64->64 @64^3": (32, 64, 64, 64, 64, 64), "128->128 @64x32x32": (32, 64, 32, 32, 128, 128),
                                      "128->64 @64^3": (32, 64, 64, 64, 128, 64)}.items():
    x = (torch.randn(N, D, H, W, cin, device=dev) * 0.5).to(torch.bfloat16)
    w = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02, torch.bfloat16)
    b = torch.zeros(cout, device=dev)
    y = torch.empty(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
    res = torch.zeros(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
    emb = torch.zeros(N, cout, device=dev)
    pa, pb = torch.ones(N, cin, device=dev), torch.zeros(N, cin, device=dev)
    row = []
    for tag, kw, st in (("plain", {}, False), ("pre", dict(pre_a=pa, pre_b=pb, pre_silu=True), False), ("res", dict(res=res), False),
                        ("emb", dict(res_add=emb), False), ("stats", {}, True),
                        ("pre+emb+stats", dict(pre_a=pa, pre_b=pb, pre_silu=True, res_add=emb), True),
                        ("pre+res+stats", dict(pre_a=pa, pre_b=pb, pre_silu=True, res=res), True)):
        d = ops.make_conv_desc(x, None, w, b, kernel=(3, 3, 3), cout=cout, split=cout, y=y, y2=None, **kw)
        if st:
            nt = ops.conv_stats_tiles(d)
            stats = torch.zeros(N, nt, 2, cout, device=dev)
            d.stats = stats.data_ptr()
        for _ in range(3):
            ops.conv_launch(d)
        torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.conv_launch(d)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        row.append(f"{tag}: {dt * 1e3:.3f} ms ({2.0 * N * D * H * W * cin * cout * 27 / dt / 1e12:.0f} TF/s)")
    print(name, " | ".join(row), flush=True)
