import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/rho_diffusion_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from rho_diffusion_amd.engine import ops
dev = "cuda"
N, D, H, W, c = 2, 128, 128, 128, 32
x = (torch.randn(N, D, H, W, c, device=dev) * 0.5).to(torch.bfloat16)
w = ops.prep_conv_weight(torch.randn(c, c, 3, 3, 3, device=dev) * 0.05, torch.bfloat16)
b = torch.zeros(c, device=dev)
y = torch.empty(N, D, H, W, c, device=dev, dtype=torch.bfloat16)
res = torch.zeros_like(y)
pa, pb = torch.ones(N, c, device=dev), torch.zeros(N, c, device=dev)
row = []
for tag, kw, st in (("plain", {}, False), ("pre", dict(pre_a=pa, pre_b=pb, pre_silu=True), False), ("res", dict(res=res), False), ("stats", {}, True),
                    ("pre+res+stats", dict(pre_a=pa, pre_b=pb, pre_silu=True, res=res), True)):
    d = ops.make_conv_desc(x, None, w, b, kernel=(3, 3, 3), cout=c, split=c, y=y, y2=None, **kw)
    if st:
        nt = ops.conv_stats_tiles(d)
        stats = torch.zeros(N, nt, 2, c, device=dev)
        d.stats = stats.data_ptr()
    for _ in range(3):
        ops.conv_launch(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ops.conv_launch(d)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    row.append(f"{tag}: {dt*1e3:.3f} ms ({2.0*N*D*H*W*c*c*27/dt/1e12:.0f} TF/s)")
print(ops.conv_variant(d), " | ".join(row))
