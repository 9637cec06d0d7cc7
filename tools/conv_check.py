"""Development aid: rho_conv_nd_fwd (bf16) against torch's fp32 convolution on a few 1-D / 2-D / 3-D shapes (relative l2 error)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from rho_diffusion_amd.engine import ops
dev = "cuda"
torch.manual_seed(0)
cases = [((1, 1, 3), (2, 1, 1, 4096), 64, 64), ((1, 1, 3), (2, 1, 1, 4096), 32, 64), ((1, 1, 3), (2, 1, 1, 4096), 96, 64), ((1, 1, 3), (2, 1, 1, 4096), 64, 128),
         ((1, 1, 3), (2, 1, 1, 4096), 256, 128), ((1, 3, 3), (2, 1, 64, 64), 64, 64), ((1, 3, 3), (2, 1, 64, 64), 128, 128), ((3, 3, 3), (2, 16, 16, 16), 64, 64),
         ((3, 3, 3), (2, 16, 16, 16), 128, 128), ((3, 3, 3), (2, 16, 16, 16), 256, 128)]
for kernel, (N, D, H, W), cin, cout in cases:
    x = torch.randn(N, D, H, W, cin, device=dev).to(torch.bfloat16)
    wt = torch.randn(cout, cin, *kernel, device=dev) * 0.05
    w = ops.prep_conv_weight(wt, torch.bfloat16)
    b = torch.randn(cout, device=dev)
    y = torch.empty(N, D, H, W, cout, device=dev, dtype=torch.bfloat16)
    d = ops.make_conv_desc(x, None, w, b, kernel=kernel, cout=cout, split=cout, y=y, y2=None)
    ops.conv_launch(d)
    torch.cuda.synchronize()
    ref = F.conv3d(x.float().permute(0, 4, 1, 2, 3), wt.to(torch.bfloat16).float(), b, padding=tuple(k // 2 for k in kernel)).permute(0, 2, 3, 4, 1)
    err = float((y.float() - ref).norm() / ref.norm())
    print(kernel, (N, D, H, W), cin, cout, ops.conv_variant(d), f"rel l2 {err:.2e}", "BAD" if err > 1e-2 else "")
