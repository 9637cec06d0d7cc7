#!/usr/bin/env python
"""Headline benchmark: DDPM denoising steps/s (and training samples/s once --mode train is
selected) on 3-D 64^3 spherical-harmonics density fields, UNetv2 (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" = one pass of the reverse-process hot path over one batch:
Philox noise draw, UNetv2 forward (eps prediction), p_sample update, device-side step advance
(rho_diffusion/diffusion/ddpm.py:195-218).  Sampling shards by independent samples: no data-path
collective, weak scaling (per-GPU batch fixed).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense peaks, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["sample", "train", "ddim", "both"], default="both",
                    help="sample: denoising steps/s; train: training samples/s; both (default): `value` is the denoising rate, "
                         "the training rate rides along in the same JSON line")
    ap.add_argument("--train-steps", type=int, default=None, help="timed training steps in --mode both (default: --steps)")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[2]: 32)")
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--dims", type=int, default=3)
    ap.add_argument("--mc", type=int, default=64)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-graph", action="store_true", help="launch the sampling step eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-checkpoint-leg", action="store_true", help="skip the short use_checkpoint=True training leg")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--dump-ops", default=None, help="write the per-launch profile (kind, shape, ms, TFLOP/s) to this file")
    ap.add_argument("--labels", action="store_true",
                    help="class-conditional backbone: num_classes=25, MultiEmbeddings(embedding_dim=4*mc) over the DeepGalaxy "
                         "parameter space, y [B, 4] (BASELINE configs[4]; needs --mc 32 for the reference's embedding_dim=128)")
    ap.add_argument("--config", choices=sorted(PRESETS), default=None,
                    help="BASELINE.json configuration preset (sets --dims/--grid/--mc/--batch/--dtype/--labels); default = c3")
    args = ap.parse_args()
    if args.config:
        for k, v in PRESETS[args.config].items():
            setattr(args, k, v)
    return args


# BASELINE.json configs (c1 .. c5; c4 = c3 on 8 GPUs, i.e. `--gpus 8` with the c3 preset)
PRESETS = {
    "c1": dict(dims=2, grid=64, mc=64, batch=16, dtype="fp32", labels=False),
    "c2": dict(dims=2, grid=128, mc=64, batch=64, dtype="fp32", labels=False),
    "c3": dict(dims=3, grid=64, mc=64, batch=32, dtype="bf16", labels=False),
    "c4": dict(dims=3, grid=64, mc=64, batch=32, dtype="bf16", labels=False),      # = c3 per GPU; run with --gpus 8 (global batch 256)
    "c5": dict(dims=3, grid=128, mc=32, batch=2, dtype="bf16", labels=True),
}
DEEP_GALAXY_SPACE = {"s": [0.25, 0.5, 0.75, 1, 1.25, 1.5], "m": [0.25, 0.5, 0.75, 1, 1.25, 1.5],
                     "t": list(range(300, 655, 5)), "c": list(range(14))}          # rho_diffusion/data/deep_galaxy.py:41-47


def workload_name(args, world):
    """Derived from the arguments; names the BASELINE config only when the arguments ARE that config."""
    which = None
    for name, pz in PRESETS.items():
        if all(getattr(args, k) == v for k, v in pz.items()):
            which = {"c1": "configs[0]", "c2": "configs[1]", "c3": "configs[2]" if world == 1 else f"configs[3] on {world} GPUs",
                     "c4": "configs[2]" if world == 1 else f"configs[3] on {world} GPUs", "c5": "configs[4] geometry (batch 2/GPU)"}[name]
    tag = f"BASELINE {which}" if which else "custom configuration, not a BASELINE config"
    cond = ", class-conditional (MultiEmbeddings y[B,4])" if args.labels else ""
    return (f"DDPM reverse step, UNetv2 {args.dims}D {args.grid}^{args.dims} mc={args.mc}{cond} ({tag}), batch {args.batch}/GPU, "
            f"{args.dtype}, LinearSchedule(1000,1e-3,0.02)")


def build_model(args, device):
    """UNetv2 with BASELINE hyper-parameters, default init seed 777, zero-initialised
    layers re-randomised N(0, 0.02) so kernels see realistic data (a fresh UNetv2 outputs exactly 0)."""
    from rho_diffusion_amd.diffusion import DDPM, LinearSchedule
    from rho_diffusion_amd.models import UNet
    from torch import nn
    torch.manual_seed(777)
    kw = dict(data_shape=[args.grid] * args.dims, in_channels=1, out_channels=1, model_channels=args.mc,
              num_res_blocks=2, channel_mult=(1, 2, 4, 8), attention_resolutions=[16, 8], num_heads=4,
              use_scale_shift_norm=True, dims=args.dims, activation="SiLU", compute_dtype=args.dtype)
    extra = {}
    if args.labels:
        kw["num_classes"] = 25
        extra = dict(cond_fn="MultiEmbeddings", cond_fn_kwargs={"parameter_space": DEEP_GALAXY_SPACE, "embedding_dim": 4 * args.mc})
    ddpm = DDPM(UNet, kw, LinearSchedule(1000, 1e-3, 0.02), nn.MSELoss, timesteps=1000, **extra)
    with torch.no_grad():
        for p in ddpm.backbone.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02)
    return ddpm.to(device), kw


def hip_ws_bytes(batch):
    from rho_diffusion_amd import hip
    return int(hip.lib().rho_abs_quantile_workspace_bytes(batch))


def labels_for(batch, device):
    keys = list(DEEP_GALAXY_SPACE)
    rows = [[float(DEEP_GALAXY_SPACE[k][(3 * i + 5 * j + 1) % len(DEEP_GALAXY_SPACE[k])]) for j, k in enumerate(keys)] for i in range(batch)]
    return torch.tensor(rows, dtype=torch.float32, device=device)


def cpu_baseline(kw, ddpm, args):
    """The CPU oracle (fp32, stock PyTorch CPU ops == what the reference executes) timed on this host's cores for a bounded
    sample, BOTH metrics: denoising steps and one training step (q_sample + forward + MSE + autograd
    backward + torch.optim.AdamW, what scripts/training_ddp.py:185-206 executes).  Batch: the bench batch itself where a full-batch
    step costs seconds (c1: B = 16, as BASELINE.md prescribes), else B = 1 scaled linearly in time."""
    from oracle import ref_torch as R
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the GPU box hands a one-GPU job a share of its cores while sched_getaffinity / os.cpu_count() still report all 256: with 256
    # ATen threads a B=1 step took 70 s instead of 4 s (measured, r02).  32 threads = the r01 setting; both counts are reported.
    n_threads = max(1, min(avail, 32))
    torch.set_num_threads(n_threads)
    sd = {k: v.detach().float().cpu() for k, v in ddpm.backbone.state_dict().items()}
    cfg = {k: v for k, v in kw.items() if k != "compute_dtype"}
    sched = R.linear_schedule(1000, 1e-3, 0.02)
    space = DEEP_GALAXY_SPACE if args.labels else None

    def sample_steps(bsz, n):
        shape = (bsz, 1) + (args.grid,) * args.dims
        x, z = torch.randn(shape), torch.randn(shape)
        yb = labels_for(bsz, "cpu") if args.labels else None
        times = []
        with torch.no_grad():
            for i in range(n):
                t0 = time.perf_counter()
                t = 999 - i
                pred = R.unet_forward(sd, cfg, x, torch.full((bsz,), t, dtype=torch.long), yb, space)
                x = R.p_sample_step(x, pred, t, sched, z)
                times.append(time.perf_counter() - t0)
        return times

    warm = sample_steps(1, 1)[0]                    # (also the ATen warm-up)
    bsz = args.batch if warm * args.batch <= 4.0 else 1
    times = sample_steps(bsz, args.cpu_steps + (1 if bsz > 1 else 0))
    per = sum(times[1:] if bsz > 1 else times) / max(1, len(times) - (1 if bsz > 1 else 0))
    scale = args.batch // bsz
    out = {"value": 1.0 / (per * scale), "unit": "denoising_steps/s", "cores": n_threads, "host_cpu_count": os.cpu_count(),
           "kind": "port",
           "sample": (f"B={bsz} of {args.batch}, {args.cpu_steps} timed steps after 1 warm-up ({per:.2f} s per B={bsz} step)"
                      + (f", scaled x{scale} in time" if scale > 1 else ", the full batch") + "; fp32 oracle (oracle/ref_torch.py)")}
    # ---- training: one optimizer step of the oracle (leaf copies of the weights, autograd through the functional forward)
    if per * 3.5 > 75.0:
        out["training"] = {"value": None, "unit": "training_samples/s",
                           "sample": f"skipped: a B={bsz} oracle training step is ~{per * 3.5:.0f} s of host time here (forward {per:.1f} s)"}
        return out
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    opt = torch.optim.AdamW([v for v in params.values() if v.requires_grad], lr=1e-4)
    shape = (bsz, 1) + (args.grid,) * args.dims
    x0 = torch.rand(shape) * 2 - 1
    yb = labels_for(bsz, "cpu") if args.labels else None
    n_train = 1 if per > 2.0 else 3                 # (>= 2 timed steps where a step is cheap: BASELINE.md section 3, c1)
    tt = []
    for i in range(n_train + (0 if per > 2.0 else 1)):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        t = torch.randint(0, 1000, (bsz,))
        eps = torch.randn(shape)
        loss = R.training_loss(lambda x_, t_, y_: R.unet_forward(params, cfg, x_, t_, y_, space), x0, t, eps, sched["alpha_bar_t"], yb)
        loss.backward()
        opt.step()
        tt.append(time.perf_counter() - t0)
    tper = sum(tt[-n_train:]) / n_train
    out["training"] = {"value": bsz / tper, "unit": "training_samples/s", "cores": n_threads, "kind": "port",
                       "sample": (f"B={bsz} of {args.batch}: {n_train} timed step(s)" + ("" if per > 2.0 else " after 1 warm-up")
                                  + f" of q_sample + forward + MSE + autograd backward + torch.optim.AdamW ({tper:.2f} s per step); "
                                    "samples/s does not scale with the batch on the host, so the B=" + str(bsz) + " rate is the rate")}
    return out


def by_kind(prof):
    acc = {}
    for p in prof:
        k = acc.setdefault(p["kind"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
        k["ms"] += p["ms"]; k["flops"] += p["flops"]; k["bytes"] += p["bytes"]; k["launches"] += 1
    return acc


def dump_ops(path, prof):
    with open(path, "w") as f:
        for i, p_ in enumerate(prof):
            tf = p_["flops"] / (p_["ms"] * 1e-3) / 1e12 if p_["ms"] > 0 else 0.0
            gb = p_["bytes"] / (p_["ms"] * 1e-3) / 1e9 if p_["ms"] > 0 else 0.0
            f.write(f"{i:4d} {p_['kind']:12s} ms={p_['ms']:8.3f} TF/s={tf:8.1f} GB/s={gb:8.1f} "
                    f"cin={p_.get('cin', '')} cout={p_.get('cout', '')} taps={p_.get('taps', '')} pos={p_.get('positions', '')}\n")


KIND_KERNEL = {
    "conv3": "k_conv (3x3x3 implicit GEMM, LDS halo tile; the three Upsample convs as 2x2-tap sub-pixel phases; ResBlock 1x1x1 skips folded into the out-conv launches)",
    "conv1": "k_conv<.,1,1,1> (1x1x1 projections: qkv / proj_out / stem and head GEMM forms; skips where not folded)",
    "attention": "k_attn_bf16 (flash-style QK^T / PV on MFMA, fp32 online softmax)",
}


def roofline_of(plan, args):
    """Dominant kernel = the MFMA kind with the most time in one step (c3: the 3x3x3 conv; c5 with T = 32768: attention when it
    outweighs the convs): achieved = algorithmic FLOPs of its launches in one step / their summed durations, each launch bracketed
    by HIP events on the launch stream (plan.profile)."""
    prof = plan.profile(repeats=3)
    kinds = by_kind(prof)
    dom = max((k for k in ("conv3", "attention", "conv1") if k in kinds), key=lambda k: kinds[k]["ms"])
    sel = [p for p in prof if p["kind"] == dom]
    fl = sum(p["flops"] for p in sel)
    ms = sum(p["ms"] for p in sel)
    if args.dump_ops:
        dump_ops(args.dump_ops, prof)
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    # `achieved` / `frac` = the algorithmic FLOPs of the dominant kind's OWN operator (conv3: the 27-tap count of the reference's
    # conv_nd calls, 127.56 TFLOP per c3 step) / the summed durations of its launches - ONE quantity round to round.  The 1x1x1
    # ResBlock skips that ride inside the same launches since round 3 are extra work done in that time: reported beside it
    # (`folded_conv1_flops_per_step`, `frac_with_folded_conv1`), never inside `frac`.
    folded = sum(p.get("folded_conv1_flops", 0.0) for p in sel)
    fl_all = fl
    fl = fl - folded
    achieved = fl / (ms * 1e-3) / 1e12
    # HBM traffic + MFMA utilisation of the same launches from PMC counters (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
    # SQ_INSTS_MFMA: separate --pmc passes collected with rocprofv3 on this command, tests/gpu_pmc.sh -> tools/pmc_summary.py).
    # bench.py cannot run the profiler itself, so the summary is only used when it was measured on THIS binary (build id of
    # rho_build_info()) and this workload; otherwise traffic is null and the reason is stated.
    from rho_diffusion_amd import hip
    build = hip.lib().rho_build_info().decode().rsplit("build ", 1)[-1]
    traffic = mfma_util = None
    traffic_note = "no PMC summary under profiles/ for this binary"
    import glob
    # newest first (names carry the round: r03d > r03c > r02f); the first summary of THIS workload decides
    workload = dict(dims=args.dims, grid=args.grid, mc=args.mc, batch=args.batch, dtype=args.dtype, labels=args.labels)
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_traffic_{dom}.json")), reverse=True):
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("workload") != workload:
            continue
        if tj.get("build_id") != build:
            traffic_note = (f"{os.path.relpath(tpath, ROOT)} was measured on build {tj.get('build_id', '(unrecorded)')}, this is {build}: "
                            "not reported (re-run tests/gpu_pmc.sh)")
        else:
            traffic, mfma_util = tj["hbm_bytes_per_launch"], tj.get("mfma_util")
            traffic_note = (f"{os.path.relpath(tpath, ROOT)} (build {build}; PMC FETCH_SIZE x2 + WRITE_SIZE, average per launch of the same "
                            f"{dom} launches)")
        break
    alg_bytes = sum(p["bytes"] for p in sel)
    # Upsample + conv runs as 2-tap sub-pixel phases (12 of the 27 taps in 3-D): `achieved` / `frac` count the ALGORITHMIC FLOPs of
    # the reference's formulation (interpolate, then a 27-tap conv) as the contract asks; `executed_frac` counts the multiply-adds
    # the matrix cores actually execute
    exe = sum(p.get("executed_flops", p["flops"]) for p in sel)
    exe_tf = exe / (ms * 1e-3) / 1e12
    return {
        "bound": "mfma", "kernel": KIND_KERNEL[dom], "kind": dom,
        "achieved": achieved, "peak": peak,
        "unit": "TFLOP/s", "frac": achieved / peak, "frac_with_folded_conv1": fl_all / (ms * 1e-3) / 1e12 / peak,
        "executed_frac": exe_tf / peak,
        "traffic": traffic, "traffic_source": traffic_note, "mfma_util": mfma_util,
        "build_id": build, "executed_flops_per_step": exe, "executed_TFLOPs": exe_tf,
        # ResBlock skip convolutions (1x1x1) contracted inside their out-conv's launch (inference plans): part of the figures above
        "folded_conv1_flops_per_step": folded,
        "algorithmic_bytes_per_launch": alg_bytes / max(1, len(sel)), "algorithmic_flops_per_launch": fl / max(1, len(sel)),
        "launches_per_step": len(sel), "avg_launch_ms": ms / max(1, len(sel)),
        "algorithmic_bytes_per_step": alg_bytes, "algorithmic_flops_per_step": fl, "kernel_ms_per_step": ms,
        "all_kernels_ms_per_step": sum(p["ms"] for p in prof),
        "by_kind_ms": {k: round(v["ms"], 3) for k, v in kinds.items()},
        "by_kind_TFLOPs": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in kinds.items() if v["flops"] > 0 and v["ms"] > 0
                           and k in ("conv3", "conv1", "attention")},
        "hbm_kernels_GBps": hbm_kinds(kinds),
    }


HBM_KINDS = ("gn_partial", "gn_apply", "pack", "stem", "head", "resample", "gn_bwd_reduce", "gn_bwd_apply", "add", "chan_sum", "pool2x",
             "upsample", "avgpool_bwd")


def hbm_kinds(kinds):
    """HBM-regime launches of a plan (GroupNorm / elementwise passes): algorithmic bytes / summed HIP-event time, and
    the fraction of the 8 TB/s HBM3E peak."""
    out = {}
    for k, v in kinds.items():
        if k in HBM_KINDS and v["ms"] > 0 and v["bytes"] > 0:
            gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            out[k] = {"GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 3), "ms": round(v["ms"], 3), "launches": v["launches"]}
    return out


def hbm_pipeline_kernels(fns):
    """The pipeline's own elementwise kernels (q_sample, p_sample, Philox, MSE, AdamW, ...): {name: (callable, algorithmic bytes)}
    -> {name: {GBps, frac, ms}} with a HIP-event pair around 5 back-to-back launches on the launch stream."""
    out = {}
    for name, (fn, nbytes) in fns.items():
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[name] = {"GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 3), "ms": round(ms, 4), "bytes": int(nbytes)}
    return out


def launch_self(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: this process becomes the launcher.  It starts N ranks
    of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would), relays rank 0's JSON
    line and returns non-zero if any rank failed.  The parent never touches the GPU (no HIP call, no exec)."""
    from rho_diffusion_amd.launch import spawn_ranks
    return spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, json_only=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_self(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and not (args.gpus == 1 and "WORLD_SIZE" in os.environ):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    local = local % max(1, torch.cuda.device_count())      # (only differs from LOCAL_RANK when ranks share a GPU in rehearsals)
    if world > 1 or os.environ.get("RHO_BENCH_FORCE_DIST") == "1":      # (the override rehearses the RCCL code path on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        torch.cuda.set_device(local)
        # "nccl" IS RCCL on ROCm; RHO_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals
        dist.init_process_group(os.environ.get("RHO_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    from rho_diffusion_amd.engine import ops
    ddpm, kw = build_model(args, device)
    B = args.batch
    shape = (B, 1) + (args.grid,) * args.dims
    engine = ddpm.backbone.engine()
    tables = ddpm.schedule.device_tables(device)
    ddpm.noise_seed = 777 + rank

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    results = {}
    roofline = None
    ranks_counted = 1
    if world > 1:   # warm-up collective (cf. xpu.py:374-375), not part of the sampling data path; it also counts the ranks
        assert dist.get_world_size() == world, (dist.get_world_size(), world)
        w = torch.ones(8, device=device)
        dist.all_reduce(w)
        ranks_counted = int(round(float(w[0].item())))
        assert ranks_counted == world, f"process group reduced over {ranks_counted} ranks, expected {world}"

    rank_spread = {}

    def timed(step_fn, steps, warmup):
        """(wall seconds for `steps` steps, max over ranks; per-step HIP-event durations in ms on this rank).  The events sit on
        the stream the step is launched on; recording them costs nothing measurable against a >= 1 ms step."""
        for _ in range(warmup):
            step_fn()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        barrier()
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(steps):
            step_fn()
            evs[i + 1].record()
        barrier()
        dt = time.perf_counter() - t0
        per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
        rank_spread.clear()
        if world > 1:
            # every rank's own wall time for the K steps: the maximum is the job's time (the contract); minimum and maximum
            # are both reported so that a straggling rank is visible in the line
            mine = torch.tensor([dt], device=device, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [float(t_.item()) for t_ in allr]
            dt = max(per_rank)
            rank_spread.update(min=1e3 * min(per_rank) / steps, max=1e3 * max(per_rank) / steps,
                               slowest_rank=per_rank.index(max(per_rank)))
        med = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
        return dt, med

    if args.mode in ("sample", "both"):
        x_t = ddpm.noise(torch.empty(shape, device=device))
        z = torch.empty(shape, dtype=torch.float32, device=device)
        t_dev = torch.full((1,), 999, dtype=torch.int32, device=device)
        off_dev = torch.full((1,), 1 << 32, dtype=torch.int64, device=device)
        n_elem = x_t.numel()
        # labels are embedded once per chain (DDPM.reverse_process does the same through _preembed_conditions): the step gets [B, 4 mc]
        cc = ddpm._preembed_conditions(labels_for(B, device)) if args.labels else None

        def sample_step():
            ops.philox_normal(z, ddpm.noise_seed, 0, offset_dev=off_dev)
            pred = engine.forward(x_t, None, cc, t_scalar_dev=t_dev)
            ops.p_sample_step(x_t, pred, z, tables["coef"], t_dev)
            ops.step_advance(t_dev, off_dev, (n_elem + 3) // 4)

        step_fn, graphed = sample_step, False
        if not args.no_graph:
            # the step's state (t, Philox offset) lives on the device, so one captured HIP graph serves every step
            # (DDPM.reverse_process does the same); launch-bound configurations (2-D 64^2) gain 3-4x, c3 nothing
            try:
                sample_step()                                   # builds the engine plan outside the capture
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) stay free to call HIP
                    sample_step()
                step_fn, graphed = g_.replay, True
            except Exception as exc:  # noqa: BLE001
                torch.cuda.synchronize()
                print(f"[bench] HIP graph capture failed ({type(exc).__name__}: {exc}); eager launches", file=sys.stderr)
        dt, med = timed(step_fn, args.steps, args.warmup)
        assert torch.isfinite(x_t).all(), "non-finite state after the timed steps"
        results["sample"] = dict(dt=dt, steps=args.steps, graphed=graphed, median_ms=med, rank_ms=dict(rank_spread))
        hbm_pipe = {}
        if rank == 0 and not args.no_roofline:
            roofline = roofline_of(next(iter(engine._plans.values())), args)
            # the step's own HBM-regime kernels: bytes / HIP-event time against the 8 TB/s peak.  t_dev is parked at 500
            # (a mid-chain step: noise term on) and restored, so the state stays finite
            t_save = t_dev.clone()
            t_dev.fill_(500)
            xs, pred_ = x_t.clone(), torch.zeros_like(x_t)
            hbm_pipe.update(hbm_pipeline_kernels({
                "k_philox_normal": (lambda: ops.philox_normal(z, ddpm.noise_seed, 0, offset_dev=off_dev), 4.0 * n_elem),
                "k_p_sample": (lambda: ops.p_sample_step(xs, pred_, z, tables["coef"], t_dev), 16.0 * n_elem)}))
            t_dev.copy_(t_save)
            del xs, pred_

    if args.mode in ("ddim", "both"):
        # the GaussianDiffusionPipeline step (x0-prediction UNet + dynamic thresholding + DDIM eta = 0),
        # same backbone / engine, reported beside the headline as "ddim_sampling"
        from rho_diffusion_amd.diffusion.gaussian_diffusion import ddim_coefficients, diffusion_tables, get_named_beta_schedule
        gtab = diffusion_tables(get_named_beta_schedule("cosine", 1000))
        xd = ddpm.noise(torch.empty(shape, device=device))
        td = torch.full((1,), 999, dtype=torch.int32, device=device)
        quant = torch.empty(B, dtype=torch.float32, device=device)
        qws = torch.empty((hip_ws_bytes(B) + 3) // 4, dtype=torch.int32, device=device)
        state = {"t": 999}

        ccd = ddpm._preembed_conditions(labels_for(B, device)) if args.labels else None

        def ddim_step():
            x0_hat = engine.forward(xd, None, ccd, t_scalar_dev=td)
            c = ddim_coefficients(gtab, state["t"], 0.0)
            ops.abs_quantile(x0_hat, 0.9, out=quant, workspace=qws)
            ops.ddim_step(xd, x0_hat, quant, None, xd, None, *c)
            ops.step_advance(td, None, 0)
            state["t"] = max(state["t"] - 1, 1)

        dt, med = timed(ddim_step, args.steps, args.warmup)
        assert torch.isfinite(xd).all(), "non-finite DDIM state after the timed steps"
        results["ddim"] = dict(dt=dt, steps=args.steps, median_ms=med)
        if rank == 0 and not args.no_roofline:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            x0_hat = engine.forward(xd, None, ccd, t_scalar_dev=td)
            torch.cuda.synchronize()
            ev[0].record()
            ops.abs_quantile(x0_hat, 0.9, out=quant, workspace=qws)
            ev[1].record()
            ops.ddim_step(xd, x0_hat, quant, None, xd, None, *ddim_coefficients(gtab, 500, 0.0))
            ev[2].record()
            torch.cuda.synchronize()
            results["ddim"]["quantile_ms"] = ev[0].elapsed_time(ev[1])
            results["ddim"]["update_ms"] = ev[1].elapsed_time(ev[2])

    if args.mode in ("train", "both"):
        # synthetic spherical-harmonics density fields (rho_diffusion/data/synthetic.py:45-124) generated ON THE DEVICE
        # (rho_sph_harm_fields), a pool of 64 (SURVEY 8d / BASELINE.md section 3) built once before the timed region; the batch is
        # its first B fields
        from rho_diffusion_amd.data import SphericalHarmonicPool
        data = SphericalHarmonicPool(args.grid, args.dims, size=64, seed=777 + rank, device=device).batch(B)
        batch = [data, labels_for(B, device)] if args.labels else data
        from rho_diffusion_amd.trainer import DPTrainer
        trainer = DPTrainer(ddpm, lr=1e-4)
        last = {}

        def train_step():
            last["loss"] = trainer.step(batch)

        tsteps = args.train_steps or args.steps
        dt, med = timed(train_step, tsteps, max(1, args.warmup))
        assert torch.isfinite(last["loss"]).all(), "non-finite training loss"
        results["train"] = dict(dt=dt, steps=tsteps, loss=float(last["loss"].detach()), median_ms=med, rank_ms=dict(rank_spread),
                                peak_mem_gb=round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
                                plan_gb=round(engine._last_train_plan.nbytes() / 2 ** 30, 1))
        if rank == 0 and not args.no_roofline:
            tp = engine._last_train_plan
            bprof = tp.profile(repeats=1, backward=True)
            fk, bk = by_kind(tp.profile(repeats=1)), by_kind(bprof)
            if args.dump_ops:
                dump_ops(args.dump_ops + ".bwd", bprof)
            # training roofline (SURVEY 8d "per training sample": fwd + dgrad + wgrad + the attention recompute): algorithmic
            # FLOPs of the MFMA kernels of one step against the bf16 / fp32 matrix peak, whole step and per kind
            peak = MFMA_PEAK_TFLOPS[args.dtype]
            mf = {("fwd", k): v for k, v in fk.items() if v["flops"] > 0 and k in ("conv3", "conv1", "attention")}
            mf.update({("bwd", k): v for k, v in bk.items() if v["flops"] > 0 and k in ("wgrad", "dgrad", "attention_bwd")})
            tot_fl = sum(v["flops"] for v in mf.values())
            step_ms = 1e3 * dt / tsteps
            results["train"]["roofline"] = {
                "bound": "mfma", "peak": peak, "unit": "TFLOP/s", "algorithmic_flops_per_step": tot_fl,
                "algorithmic_flops_per_sample": tot_fl / B, "ideal_ms_per_step": tot_fl / (peak * 1e12) * 1e3,
                "achieved": tot_fl / (step_ms * 1e-3) / 1e12, "frac": tot_fl / (step_ms * 1e-3) / 1e12 / peak,
                "per_kind": {f"{d}.{k}": {"ms": round(v["ms"], 2), "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                                          "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak, 3)} for (d, k), v in mf.items()},
                "non_mfma_ms": round(sum(v["ms"] for k, v in fk.items() if ("fwd", k) not in mf)
                                     + sum(v["ms"] for k, v in bk.items() if ("bwd", k) not in mf), 2)}
            results["train"]["breakdown"] = {"fwd": {k: round(v["ms"], 2) for k, v in fk.items()},
                                             "bwd": {k: round(v["ms"], 2) for k, v in bk.items()},
                                             "bwd_TFLOPs": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in bk.items()
                                                            if v["flops"] > 0 and v["ms"] > 0}}
            if roofline is None:
                roofline = roofline_of(tp, args)
            results["train"]["hbm_kernels_GBps"] = {**{f"fwd.{k}": v for k, v in hbm_kinds(fk).items()},
                                                    **{f"bwd.{k}": v for k, v in hbm_kinds(bk).items()}}
            n_el = data.numel()
            tq = torch.randint(0, 1000, (B,), device=device, dtype=torch.int64)
            xq, eq = data.float().contiguous(), torch.randn_like(data, dtype=torch.float32)
            oq = torch.empty_like(xq)
            ar = trainer.opt.build_arena()[0]
            sc = [torch.zeros_like(ar["flat"]) for _ in range(4)]        # scratch copies: the measurement must not move the weights
            results["train"]["hbm_kernels_GBps"].update(hbm_pipeline_kernels({
                "k_q_sample": (lambda: ops.q_sample(xq, eq, tq, tables["alpha_bar"], out=oq), 12.0 * n_el),
                "k_mse": (lambda: ops.mse(xq, eq, want_grad=True), 12.0 * n_el),
                "k_adamw": (lambda: ops.adamw(sc[0], sc[1], sc[2], sc[3], 1e-4, 0.9, 0.999, 1e-8, 1e-2, 1), 28.0 * sc[0].numel())}))
            del sc, xq, eq, oq
            # ---- DP overlap budget: where in the backward each gradient bucket closes, from HIP events on the
            # launch stream - the window its RCCL all-reduce has before the optimizer needs it.  Nothing is sent at N = 1.
            trainer.reducer.trace = []
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            trainer.opt.zero_grad()
            loss_ = ddpm.training_step(batch)
            e0.record()
            loss_.backward()
            e1.record()
            trainer.reducer.finish()
            trainer.opt.step()
            e2.record()
            torch.cuda.synchronize()
            tr_, trainer.reducer.trace = trainer.reducer.trace, None
            bwd_ms = e0.elapsed_time(e1)
            closes = [(b_, nb_, e0.elapsed_time(ev_)) for b_, nb_, ev_ in tr_]
            bw_assumed = 100.0                                            # GB/s of all-reduce bus bandwidth per ring: an ASSUMPTION
            end_prev, fin = 0.0, []
            for b_, nb_, tc_ in closes:
                st_ = max(tc_, end_prev)
                end_prev = st_ + nb_ / (bw_assumed * 1e9) * 1e3 * 2.0 * 7 / 8    # ring all-reduce moves 2 (N-1)/N of the bytes, N = 8
                fin.append(end_prev)
            results["train"]["ddp_overlap"] = {
                "what": "bucket close times inside one backward (HIP events on the launch stream, N=1: nothing sent); the modelled "
                        f"finish assumes {bw_assumed:.0f} GB/s ring bus bandwidth at N=8 (unmeasured: no 8-GPU node yet)",
                "backward_ms": round(bwd_ms, 2), "buckets": len(closes),
                "bucket_MB": [round(nb_ / 2 ** 20, 1) for _, nb_, _ in closes],
                "close_ms_after_backward_start": [round(tc_, 2) for _, _, tc_ in closes],
                "window_ms_before_backward_end": [round(bwd_ms - tc_, 2) for _, _, tc_ in closes],
                "modelled_allreduce_finish_ms": [round(f_, 2) for f_ in fin],
                "modelled_exposed_ms": round(max(0.0, (fin[-1] if fin else 0.0) - bwd_ms), 2)}
            del tp, bprof
        if not args.no_checkpoint_leg:
            # the same step with use_checkpoint=True on every ResBlock (reference: layers.py:153-199, unet_v2.py:266-271): activated
            # conv inputs are re-materialised in backward instead of kept.  Reported beside the main training rate, never as `value`.
            from rho_diffusion_amd.models.unet_v2 import ResBlock
            for m_ in ddpm.backbone.modules():
                if isinstance(m_, ResBlock):
                    m_.use_checkpoint = True
            engine.drop_plans(train_only=True)     # free the 70+ GB of the materialising plan first (the flags are part of the plan key)
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(device)
            dtc_, medc_ = timed(train_step, min(3, tsteps), 1)
            nst_ = min(3, tsteps)
            results["train"]["use_checkpoint"] = {
                "samples_per_sec": world * B * nst_ / dtc_, "ms_per_step": 1e3 * dtc_ / nst_, "steps": nst_,
                "peak_mem_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
                "plan_gb": round(engine._last_train_plan.nbytes() / 2 ** 30, 1),
                "what": "every ResBlock built with use_checkpoint=True: GroupNorm+FiLM+SiLU outputs recomputed in backward, not kept"}
            for m_ in ddpm.backbone.modules():
                if isinstance(m_, ResBlock):
                    m_.use_checkpoint = False

    if "sample" in results or "ddim" in results:
        r = results["sample"] if "sample" in results else results["ddim"]
        metric, unit, value, steps, dt = "denoising_steps_per_sec", "steps/s", world * r["steps"] / r["dt"], r["steps"], r["dt"]
    else:
        r = results["train"]
        metric, unit, value, steps, dt = "training_samples_per_sec", "samples/s", world * B * r["steps"] / r["dt"], r["steps"], r["dt"]

    out = {
        "metric": metric, "value": value, "unit": unit, "n_gpus": dist.get_world_size() if dist.is_initialized() else 1,
        "ranks_counted_by_all_reduce": ranks_counted,
        "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload_name(args, world), "global_batch": B * world,
                   "parallelism": f"independent samples x{world} (sampling: no data-path collective; training: DP gradient "
                                  f"all-reduce over {'RCCL' if not dist.is_initialized() or dist.get_backend() == 'nccl' else dist.get_backend()}, "
                                  f"overlapped with backward)"},
    }
    if "sample" in results:
        out["config"]["hip_graph"] = bool(results["sample"].get("graphed"))
        if results["sample"].get("rank_ms"):
            out["ms_per_step_per_rank"] = results["sample"]["rank_ms"]
        out["ms_per_step_hipevent_median"] = results["sample"]["median_ms"]
        out["config"]["sample_steps_per_sec"] = world * B * results["sample"]["steps"] / results["sample"]["dt"]
    if "ddim" in results:
        r = results["ddim"]
        out["ddim_sampling"] = {"metric": "denoising_steps_per_sec", "value": world * r["steps"] / r["dt"], "unit": "steps/s",
                                "ms_per_step": 1e3 * r["dt"] / r["steps"], "ms_per_step_hipevent_median": r["median_ms"],
                                "step": "UNetv2 x0-prediction + exact per-sample 0.9-quantile of |x0| (radix select) + fused DDIM "
                                        "update (GaussianDiffusionPipeline.reverse_process, eta = 0, cosine betas)"}
        for k_ in ("quantile_ms", "update_ms"):
            if k_ in r:
                out["ddim_sampling"][k_] = round(r[k_], 4)
    if "train" in results:
        r = results["train"]
        out["training"] = {"metric": "training_samples_per_sec", "value": world * B * r["steps"] / r["dt"], "unit": "samples/s",
                           "steps": r["steps"], "ms_per_step": 1e3 * r["dt"] / r["steps"], "ms_per_step_hipevent_median": r["median_ms"],
                           "loss": r["loss"], **({"ms_per_step_per_rank": r["rank_ms"]} if r.get("rank_ms") else {}),
                           "peak_mem_gb": r.get("peak_mem_gb"), "plan_gb": r.get("plan_gb"),
                           "step": "q_sample + UNetv2 fwd + MSE + bwd + "
                                   + ((("RCCL" if dist.get_backend() == "nccl" else dist.get_backend()) + " grad all-reduce + ") if world > 1 else "")
                                   + "fused AdamW"}

    if roofline is not None:
        out["roofline"] = roofline
        if "sample" in results and hbm_pipe:
            roofline["hbm_kernels_GBps"].update(hbm_pipe)
    if "train" in results and results["train"].get("use_checkpoint"):
        out["training"]["use_checkpoint"] = results["train"]["use_checkpoint"]
    if "train" in results and results["train"].get("breakdown"):
        out["training"]["by_kind_ms"] = results["train"]["breakdown"]
        out["training"]["roofline"] = results["train"]["roofline"]
        out["training"]["roofline"]["hbm_kernels_GBps"] = results["train"].get("hbm_kernels_GBps", {})
        if results["train"].get("ddp_overlap"):
            out["training"]["ddp_overlap"] = results["train"]["ddp_overlap"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(kw, ddpm, args)
    elif rank == 0 and world > 1:
        # the CPU oracle is timed by the N = 1 run only (the contract: rank 0 at N = 1); the multi-GPU line points at it
        out["cpu_baseline"] = {"value": None, "kind": "port", "sample": "see the N=1 line of this bench (python bench.py --gpus 1): "
                               "the CPU oracle is timed there, on rank 0's host cores"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
