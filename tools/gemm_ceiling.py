"""Practical bf16 MFMA ceiling of the box: the vendor GEMM (hipBLASLt through torch.matmul) on large square problems.
Context for roofline.frac: the 2.5 PFLOP/s datasheet peak assumes the maximum clock on every CU."""
import time
import torch

dev = "cuda"
for n in (4096, 8192, 16384):
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        (a @ b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 10
    for _ in range(it):
        c = a @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print(f"bf16 GEMM {n}^3: {dt*1e3:.3f} ms  {2*n**3/dt/1e12:.1f} TFLOP/s", flush=True)
