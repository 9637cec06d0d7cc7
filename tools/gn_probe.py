"""Bandwidth probe of the elementwise GroupNorm kernels of the training step (gn_apply, gn_bwd_reduce, gn_bwd_apply) at the
activation shapes of BASELINE configs[2]; prints ms and algorithmic TB/s per kernel.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rho_diffusion_amd import hip
from rho_diffusion_amd.engine.ops import check, dtype_code, ptr, stream

DEV = "cuda"


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    L = hip.lib()
    N = 32
    for (c1, c2, S) in [(64, 0, 64 ** 3), (64, 64, 64 ** 3), (128, 0, 64 * 32 * 32), (256, 128, 64 * 16 * 16), (512, 512, 64 * 8 * 8)]:
        C = c1 + c2
        g = torch.randn(N, S, C, device=DEV).bfloat16()
        x1 = torch.randn(N, S, c1, device=DEV).bfloat16()
        x2 = torch.randn(N, S, c2, device=DEV).bfloat16() if c2 else None
        dx1, dx2 = torch.empty_like(x1), (torch.empty_like(x2) if c2 else None)
        y = torch.empty_like(g)
        a = torch.rand(N, C, device=DEV) + 0.5
        b = torch.randn(N, C, device=DEV)
        stats = torch.rand(N, 32, 2, device=DEV) + 0.5
        cA, cP, cQ = torch.randn(N, C, device=DEV), torch.randn(N, 32, device=DEV), torch.randn(N, 32, device=DEV)
        nblk = int(L.rho_gn_nblk(S))
        part = torch.empty(N * nblk * (C // 8) * 16, device=DEV)
        dt = dtype_code(g.dtype)
        el = N * S * C * 2
        t = timed(lambda: check(L.rho_gn_apply(ptr(x1), c1, ptr(x2), c2, dt, N, S, ptr(a), ptr(b), 1, ptr(y), stream()), "a"))
        print(f"C={c1}+{c2} S={S}: gn_apply      {t:7.3f} ms {2 * el / t / 1e9:6.2f} TB/s")
        t = timed(lambda: check(L.rho_gn_partial(ptr(x1), c1, ptr(x2), c2, dt, N, S, ptr(part), stream()), "p"))
        print(f"C={c1}+{c2} S={S}: gn_partial    {t:7.3f} ms {1 * el / t / 1e9:6.2f} TB/s")
        t = timed(lambda: check(L.rho_gn_bwd_reduce(ptr(g), ptr(x1), c1, ptr(x2), c2, dt, N, S, ptr(a), ptr(b), ptr(stats), 1, ptr(part),
                                                    stream()), "r"))
        print(f"C={c1}+{c2} S={S}: gn_bwd_reduce {t:7.3f} ms {2 * el / t / 1e9:6.2f} TB/s")
        for acc in (0, 1):
            t = timed(lambda: check(L.rho_gn_bwd_apply(ptr(g), ptr(x1), c1, ptr(x2), c2, dt, N, S, ptr(a), ptr(b), 1, ptr(cA), ptr(cP),
                                                       ptr(cQ), ptr(dx1), ptr(dx2), acc, acc, None, stream()), "b"))
            print(f"C={c1}+{c2} S={S}: gn_bwd_apply acc={acc} {t:7.3f} ms {(3 + acc) * el / t / 1e9:6.2f} TB/s")
        del g, x1, x2, dx1, dx2, y
        torch.cuda.empty_cache()
    # plain device copy for reference
    src = torch.empty(1 << 30, dtype=torch.uint8, device=DEV); dst = torch.empty_like(src)
    t = timed(lambda: dst.copy_(src))
    print(f"torch copy 1 GiB: {t:.3f} ms {2 * (1 << 30) / t / 1e9:.2f} TB/s")


if __name__ == "__main__":
    main()
