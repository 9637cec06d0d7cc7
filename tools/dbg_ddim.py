import sys, os
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R0, R0 + "/tests", R0 + "/tests/golden"]
import numpy as np, torch
from oracle import ref_torch as R
from helpers import det_normal
from rho_diffusion_amd.engine import ops
from rho_diffusion_amd.diffusion.gaussian_diffusion import GaussianDiffusionPipeline as GP
T = 50
tab = R.gd_tables(R.gd_betas("cosine", T))
xshape = (4, 2, 6, 10)
xt = det_normal(xshape, "gk_xt"); scale = torch.tensor([0.3, 2.5, 3.5, 1.0]).view(-1, 1, 1, 1)
m = det_normal(xshape, "gk_m") * scale; noise = det_normal(xshape, "gk_n")
pipe = GP.__new__(GP)
for k, v in tab.items(): setattr(pipe, k, v)
t, eta = 17, 0.5
tt = torch.full((4,), t, dtype=torch.long)
ref, ref_x0 = R.gd_ddim_step(tab, xt, tt, m, noise, eta)
quant = ops.abs_quantile(m.cuda(), 0.9)
print("quant gpu", quant.cpu().tolist(), "torch", torch.quantile(m.reshape(4, -1).abs(), 0.9, dim=-1).tolist())
out = torch.empty(xshape, device="cuda"); px = torch.empty(xshape, device="cuda")
c = pipe.ddim_coefficients(t, eta)
print("coef", c)
ops.ddim_step(xt.cuda(), m.cuda(), quant, noise.cuda(), out, px, *c)
d = (out.cpu() - ref)
idx = d.nonzero()
print("n diff", len(idx), "of", d.numel())
f = np.float32
for i in idx[:5]:
    i = tuple(i.tolist())
    x0 = f(ref_x0[i]); x = f(xt[i]); nz = f(noise[i])
    ax = f(c[0]) * x; eps = (ax - x0) / f(c[1]); p0 = x0 * f(c[2]); p1 = f(c[3]) * eps; v = (p0 + p1) + f(c[4]) * nz
    print(i, "gpu", float(out.cpu()[i]), "ref", float(ref[i]), "np", float(v), "eps", float(eps), "x0", float(x0), "px", float(px.cpu()[i]))
