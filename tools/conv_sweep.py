"""Timing probe (not a product path): 3x3x3 forward convolution time against the number of input-channel chunks at a fixed
output shape - separates the per-tile fixed part (set-up, first halo, epilogue) from the per-chunk tap loop.
usage (GPU box): python tools/conv_sweep.py [pre=0|1]"""
import os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
from rho_diffusion_amd.engine import ops

dev = "cuda"
pre = len(sys.argv) > 1 and sys.argv[1] == "1"
shapes = {"cout 64 @64^3": (32, 64, 64, 64, 64), "cout 128 @32^3": (32, 32, 32, 32, 128), "cout 256 @16^3": (32, 16, 16, 16, 256), "cout 512 @8^3": (32, 8, 8, 8, 512)}
for name, (N, D, H, W, cout) in shapes.items():
    pts = []
    for cin in (32, 64, 128, 256, 512):
        x = (torch.randn(N, D, H, W, cin, device=dev) * 0.5).to(torch.bfloat16)
        w = ops.prep_conv_weight(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02, torch.bfloat16)
        b = torch.zeros(cout, device=dev)
        y = torch.empty(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
        kw = {}
        if pre:
            kw = dict(pre_a=torch.ones(N, cin, device=dev), pre_b=torch.zeros(N, cin, device=dev), pre_silu=True)
        d = ops.make_conv_desc(x, None, w, b, kernel=(3, 3, 3), cout=cout, split=cout, y=y, y2=None, **kw)
        for _ in range(3):
            ops.conv_launch(d)
        torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.conv_launch(d)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        fl = 2.0 * N * D * H * W * cin * cout * 27
        pts.append((cin // 32, dt * 1e3, fl / dt / 1e12))
    # least-squares line through the last three points (steady state)
    (x1, y1, _), (x2, y2, _) = pts[-2], pts[-1]
    slope = (y2 - y1) / (x2 - x1)
    fixed = y2 - slope * x2
    print(name, "pre" if pre else "plain", " | ".join(f"{c} chunks: {ms:.3f} ms ({tf:.0f} TF/s)" for c, ms, tf in pts),
          f"| per chunk {slope * 1e3:.1f} us, fixed {fixed * 1e3:.1f} us = {fixed / slope:.2f} chunks", flush=True)
