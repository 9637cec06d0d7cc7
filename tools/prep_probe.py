"""Time of the weight re-preparation after an optimizer step (development aid): the batched launch (rho_prep_batch) against the
per-tensor launches, on the c3 / c1 model with a training plan built.  usage: python tools/prep_probe.py [c3|c1]"""
import os, sys, time
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import torch
sys.argv = ["bench.py", "--config", sys.argv[1] if len(sys.argv) > 1 else "c3"]
import bench
args = bench.parse()
ddpm, kw = bench.build_model(args, "cuda")
eng = ddpm.backbone.engine()
for cw in eng._convs:
    cw.enable_dgrad()
for mode in ("1", "0", "1", "0"):
    os.environ["RHO_BATCH_PREP"] = mode
    eng._prep_table = None
    eng.refresh_weights(force=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(5):
        eng.refresh_weights(force=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"RHO_BATCH_PREP={mode}: {e0.elapsed_time(e1) / 5:.3f} ms per refresh (GPU), {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms wall", flush=True)
