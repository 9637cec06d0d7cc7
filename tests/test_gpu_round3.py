"""-m gpu, round 3: the device error flag of the label lookup surfaces as the reference's IndexError on every path (UNet.forward,
training_step, graph-captured sampling); the stand-alone MultiEmbeddings trains under autograd; the BASELINE configurations c5
(3-D 128^3 mc 32 conditioned, bf16) and c2 (2-D 128^2 mc 64 batch 64, fp32) RUN whole at full size (property checks: the full-size
oracle is minutes of CPU, the per-layer / attention oracles at these shapes live in test_gpu_bench_shapes.py)."""
import numpy as np
import math

import pytest
import torch
from torch import nn

from helpers import (DEEP_GALAXY_SPACE, PARAM_SPACE, UNET_CASES, case_inputs, det_normal, det_state_dict, det_uniform, galaxy_labels,
                     golden_template, load_golden, rel_l2)
from gpu_util import DEV
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu


def _cond_ddpm(T=50, dtype="fp32"):      # (LinearSchedule scales beta by 1000 / T: below T = 20 beta_T exceeds 1 and q_sample is NaN)
    from rho_diffusion_amd.diffusion import DDPM, LinearSchedule
    from rho_diffusion_amd.models import MultiEmbeddings, UNet
    g4 = load_golden("g4_unet.npz")
    case = next(c for c, (_, _, yk) in UNET_CASES.items() if yk == "multi")
    cfg, xshape, _ = UNET_CASES[case]
    ddpm = DDPM(UNet, dict(cfg, compute_dtype=dtype), LinearSchedule(T, 1e-3, 0.02), nn.MSELoss, timesteps=T)
    ddpm.backbone.cond_fn = MultiEmbeddings(parameter_space=PARAM_SPACE, embedding_dim=4 * cfg["model_channels"])
    ddpm.backbone.load_state_dict(det_state_dict(golden_template(g4, case), case))
    return ddpm.to(DEV), case, xshape


# ----------------------------------------------------------------------------- unknown labels (engine err flag was never polled)
def test_unknown_label_raises_from_unet_forward():
    """conditioning.py:132: a label value outside the parameter space is an IndexError.  The engine resolves labels on the device
    (flag bit 1); an inference call of UNet.forward polls it at once."""
    ddpm, case, xshape = _cond_ddpm()
    cfg, x, t, y = case_inputs(case)
    model = ddpm.backbone.eval()
    with torch.no_grad():
        model(x.to(DEV), t.to(DEV), y.to(DEV))                       # good labels: fine
        bad = y.clone().float()
        bad[0, 0] = 123.456
        with pytest.raises(IndexError):
            model(x.to(DEV), t.to(DEV), bad.to(DEV))
        model(x.to(DEV), t.to(DEV), y.to(DEV))                       # the flag was cleared by the raise


def test_unknown_label_raises_from_training_step_at_the_poll():
    """training_step does not sync per step (the reference's NaN check did): the sticky flag is polled every nan_check_every steps."""
    ddpm, case, xshape = _cond_ddpm()
    ddpm.train()
    ddpm.nan_check_every = 3
    cfg, x, t, y = case_inputs(case)
    data = det_uniform(xshape, "r3data", 0, 1).to(DEV)
    bad = y.clone().float()
    bad[1, 0] = -77.0
    with pytest.raises(IndexError):
        for _ in range(4):                                            # surfaces at the latest at step 3's poll (+1: flag of the previous steps)
            ddpm.training_step([data, bad.to(DEV)]).backward()
    ddpm.zero_grad()
    for _ in range(4):                                                # good labels keep training
        ddpm.training_step([data, y.to(DEV)]).backward()


def test_unknown_label_raises_from_graph_captured_sampling():
    """reverse_process pre-embeds the labels once per chain (stand-alone MultiEmbeddings call: raises before the first step); labels
    handed over already on the engine path (the conditions reach engine.forward as [B, k]) surface at the end of the chain."""
    ddpm, case, xshape = _cond_ddpm()
    cfg, x, t, y = case_inputs(case)
    bad = y.clone().float()
    bad[0, 1] = 9e9
    with pytest.raises(IndexError):
        ddpm.reverse_process(torch.zeros(xshape, device=DEV), conditions=bad.to(DEV))
    # engine path: skip the pre-embedding, the captured step itself resolves the labels
    ddpm._preembed_conditions = lambda cc: cc
    assert ddpm.hip_graph_sampling
    with pytest.raises(IndexError):
        ddpm.reverse_process(torch.zeros(xshape, device=DEV), conditions=bad.to(DEV))
    out = ddpm.reverse_process(torch.zeros(xshape, device=DEV), conditions=y.to(DEV))["denoised"]
    assert torch.isfinite(out).all()


def test_standalone_multi_embeddings_trains_under_autograd():
    """MultiEmbeddings.forward outside the UNet engine: gradients reach the embedding tables like the reference's
    nn.Embedding sum (conditioning.py:115-139), checked against stock autograd on the oracle."""
    from rho_diffusion_amd.models import MultiEmbeddings
    dim, B = 128, 6
    me = MultiEmbeddings(parameter_space=DEEP_GALAXY_SPACE, embedding_dim=dim)
    me.load_state_dict(det_state_dict(me.state_dict(), "r3me"))
    sd = {f"cond_fn.{k}": v.clone().requires_grad_(True) for k, v in me.state_dict().items()}
    y = torch.tensor(galaxy_labels(B), dtype=torch.float32)
    y[3] = y[0]                                                       # a repeated row: gradients must accumulate
    wgt = det_normal((B, dim), "r3mew")
    (R.multi_embeddings(y, DEEP_GALAXY_SPACE, sd) * wgt).sum().backward()
    me = me.to(DEV)
    out = me(y.to(DEV))
    assert out.requires_grad
    (out * wgt.to(DEV)).sum().backward()
    for k, layer in me.embedding_layers.items():
        ref = sd[f"cond_fn.embedding_layers.{k}.weight"].grad
        assert layer.weight.grad is not None, k
        assert torch.allclose(layer.weight.grad.cpu(), ref, atol=1e-6), k
    # the device tables are cached on the weights' storage and rebuilt when it moves
    t1 = me._device_tables(torch.device(DEV))
    assert me._device_tables(torch.device(DEV)) is t1
    with torch.no_grad():
        for layer in me.embedding_layers.values():
            layer.weight.data = layer.weight.data.clone()
    assert me._device_tables(torch.device(DEV)) is not t1
    assert torch.equal(me(y.to(DEV)).detach(), out.detach())


# ----------------------------------------------------------------------------- BASELINE configs run whole
def _bench_unet(dims, grid, mc, dtype, labels):
    from rho_diffusion_amd.models import MultiEmbeddings, UNet
    torch.manual_seed(777)
    kw = dict(data_shape=[grid] * dims, in_channels=1, out_channels=1, model_channels=mc, num_res_blocks=2, channel_mult=(1, 2, 4, 8),
              attention_resolutions=[16, 8], num_heads=4, use_scale_shift_norm=True, dims=dims, activation="SiLU", compute_dtype=dtype)
    if labels:
        kw["num_classes"] = 25
    model = UNet(**kw)
    if labels:
        model.cond_fn = MultiEmbeddings(parameter_space=DEEP_GALAXY_SPACE, embedding_dim=4 * mc)
    with torch.no_grad():
        for p in model.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0.0, 0.02)
    return model.to(DEV).eval()


def test_c5_whole_network_runs_at_128_cubed_conditioned():
    """BASELINE configs[4]: 3-D 128^3, mc 32, num_classes 25 + MultiEmbeddings(128) over the DeepGalaxy space, T = 32768 attention
    (unet_v2.py:365-436, examples/config_deep_galaxy.json).  One forward at [1, 1, 128, 128, 128]: finite, label-sensitive, and the
    bf16 engine within 3e-2 rel-L2 of the exact-f32 engine of the same weights (the f32 engine is pinned to the reference by the
    cond3d goldens and the 128^3 single-layer oracles)."""
    model = _bench_unet(3, 128, 32, "bf16", True)
    x = det_normal((1, 1, 128, 128, 128), "r3c5x").to(DEV)
    t = torch.tensor([500], device=DEV)
    y = torch.tensor(galaxy_labels(2), dtype=torch.float32).to(DEV)
    with torch.no_grad():
        p_bf = model(x, t, y[:1])
        assert p_bf.shape == x.shape and torch.isfinite(p_bf).all()
        p_bf2 = model(x, t, y[1:2])
        assert float((p_bf - p_bf2).abs().max()) > 0.0                # the label reaches the output
        model._engines.clear()
        torch.cuda.empty_cache()
        p_32 = model.set_compute_dtype("fp32")(x, t, y[:1])
    assert torch.isfinite(p_32).all()
    sl = (slice(None), slice(None), slice(40, 72))                   # a depth slab (2 M voxels)
    e_all, e_sl = rel_l2(p_bf, p_32), rel_l2(p_bf[sl], p_32[sl])
    assert e_all < 3e-2 and e_sl < 3e-2, (e_all, e_sl)


def test_c5_whole_network_forward_vs_oracle_at_128_cubed():
    """BASELINE configs[4] against the CPU ORACLE at its real geometry (round 4; the test above compares the two engines with each
    other): one conditioned forward at [1, 1, 128, 128, 128] - T = 32768 attention at the 16-fold level, 128^3 convolutions at
    mc = 32, MultiEmbeddings over the DeepGalaxy space (unet_v2.py:365-436,685-732, conditioning.py:31-139,
    examples/config_deep_galaxy.json).  Exact-f32 engine rel-L2 <= 1e-4, bf16 engine <= 3e-2, both against ``R.unet_forward`` on the
    same weights, input and label row (about a minute of host time: the oracle's [4, 32768, 32768] attention matrix is 17 GB)."""
    model = _bench_unet(3, 128, 32, "fp32", True)
    x = det_normal((1, 1, 128, 128, 128), "r4c5x")
    t = torch.tensor([377])
    y = torch.tensor(galaxy_labels(1), dtype=torch.float32)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cfg = dict(data_shape=[128, 128, 128], in_channels=1, out_channels=1, model_channels=32, num_res_blocks=2, channel_mult=(1, 2, 4, 8),
               attention_resolutions=[16, 8], num_heads=4, use_scale_shift_norm=True, dims=3, activation="SiLU", num_classes=25)
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = R.unet_forward(sd, cfg, x, t, y, DEEP_GALAXY_SPACE)
        p32 = model(x.to(DEV), t.to(DEV), y.to(DEV))
        e32 = rel_l2(p32, ref)
        model._engines.clear()
        torch.cuda.empty_cache()
        pbf = model.set_compute_dtype("bf16")(x.to(DEV), t.to(DEV), y.to(DEV))
    ebf = rel_l2(pbf, ref)
    assert torch.isfinite(p32).all() and torch.isfinite(pbf).all()
    assert e32 < 1e-4 and ebf < 3e-2, (e32, ebf)


def test_c2_whole_network_runs_at_full_size():
    """BASELINE configs[1]: 2-D 128^2, mc 64, fp32 engine, batch 64: one forward; sample 5 of the batch equals the same sample run
    alone bit for bit or to 1e-5 (per-sample GroupNorm / attention: no cross-sample coupling, layers.py:71-74), and the B = 2 head of
    the batch matches the CPU oracle to the fp32 tolerance."""
    model = _bench_unet(2, 128, 64, "fp32", False)
    B = 64
    x = det_normal((B, 1, 128, 128), "r3c2x").to(DEV)
    t = (torch.arange(B, device=DEV) * 15) % 1000
    with torch.no_grad():
        p = model(x, t)
        assert p.shape == x.shape and torch.isfinite(p).all()
        p5 = model(x[5:6].contiguous(), t[5:6].contiguous())
    assert rel_l2(p[5:6], p5) < 1e-5
    torch.set_num_threads(16)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cfg = dict(data_shape=[128, 128], in_channels=1, out_channels=1, model_channels=64, num_res_blocks=2, channel_mult=(1, 2, 4, 8),
               attention_resolutions=[16, 8], num_heads=4, use_scale_shift_norm=True, dims=2, activation="SiLU")
    with torch.no_grad():
        ref = R.unet_forward(sd, cfg, x[:2].cpu(), t[:2].cpu())
    assert rel_l2(p[:2], ref) < 1e-4


# ----------------------------------------------------------------------------- use_checkpoint (layers.py:153-199, unet_v2.py:266-271)
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["tiny3d", "tiny2d"])
def test_use_checkpoint_recomputes_in_backward_same_gradients_less_memory(case, dtype):
    """use_checkpoint=True trades the kept intermediates of every ResBlock for recomputation in backward, with identical results
    (the reference's CheckpointFunction re-runs the block).  Here: the activated conv inputs act(GroupNorm(x) * (1 + scale) + shift)
    are not kept but re-materialised by backward - same kernels on the same values, so the loss is bit-identical and every gradient
    equal up to the summation order of the weight-gradient kernel's fp32 atomics (1e-5 relative: the run-to-run noise of one plan),
    and the plan must own fewer bytes."""
    from rho_diffusion_amd.autograd import mse_loss
    from rho_diffusion_amd.models import UNet
    if case not in UNET_CASES:
        pytest.skip(f"no golden case {case}")
    g = load_golden("g4_unet.npz")
    kw, xshape, ykind = UNET_CASES[case]
    assert ykind is None
    cfg, x, t, _ = case_inputs(case)
    res = {}
    for ck in (False, True):
        model = UNet(**dict(kw, use_checkpoint=ck), compute_dtype=dtype)
        model.load_state_dict(det_state_dict(golden_template(g, case), case))
        model = model.to(DEV).train()
        assert all(b.use_checkpoint == ck for b in model.modules() if type(b).__name__ == "ResBlock")
        pred = model(x.to(DEV), t.to(DEV))
        loss = mse_loss(pred, det_normal(tuple(pred.shape), case + "tgt").to(DEV))
        loss.backward()
        plan = model.engine()._last_train_plan
        # (activation bytes: the weight-gradient arena of round 4 - one fp32 region per parameter - is the same in both plans)
        res[ck] = (float(loss), {n: p.grad.clone() for n, p in model.named_parameters()}, plan.nbytes() - getattr(plan, "arena_bytes", 0))
    assert res[True][0] == res[False][0]
    for n, gr in res[False][1].items():
        d = float((res[True][1][n].double() - gr.double()).norm())
        assert d <= 1e-5 * float(gr.double().norm()) + 1e-9, (n, d)
    # (tiny nets: 0.81 in round 3, 0.91 since the round-4 training plans keep a few more fixed-size buffers in BOTH plans - the
    #  head's activated input, the im2col operands of the 1-channel ends; c3 at B = 32: see DESIGN.md)
    assert res[True][2] < 0.93 * res[False][2], (res[True][2], res[False][2])


# ----------------------------------------------------------------------------- HDF5 replay of the synthetic dataset
def test_spherical_harmonic_dataset_to_hdf5_and_replay(tmp_path):
    """synthetic.py:307-348: fields generated on the device are serialised (density / l / m / seed) and replayed from disk: the
    replayed items equal the generated ones bit for bit, carry the same label embeddings, and feed a training step."""
    from rho_diffusion_amd import h5io
    from rho_diffusion_amd.data import SphericalHarmonicDataset, spherical_harmonic_fields
    if not h5io.available():
        pytest.skip("libhdf5 not found on this machine")
    ds = SphericalHarmonicDataset(3, length=6, random_seed=11, grid_el=16, device=DEV)
    ds.to_hdf5(tmp_path / "sh")                                      # '.h5' is appended (synthetic.py:320-321)
    path = tmp_path / "sh.h5"
    assert h5io.shape(path, "density") == (6, 16, 16, 16) and h5io.read_attr(path, "seed") == 11
    l, m = h5io.read(path, "l"), h5io.read(path, "m")
    assert all(0 <= a <= 3 and abs(b) <= a for a, b in zip(l, m))
    ref = spherical_harmonic_fields(list(zip(l.tolist(), m.tolist())), 16, 3, DEV)
    rep = SphericalHarmonicDataset.from_hdf5(path, device=DEV)
    assert len(rep) == 6
    for i in (0, 5):
        x, emb = rep[i]
        assert x.is_cuda and torch.equal(x, ref[i]) and emb.shape == (256,)
    data, labels = rep.batch(4)
    assert torch.equal(data, ref[:4]) and labels.shape == (4, 256)
    # the replayed batch trains: q_sample + UNet forward / backward on the HIP path
    from rho_diffusion_amd.diffusion import DDPM, LinearSchedule
    from rho_diffusion_amd.models import UNet
    kw = dict(data_shape=[16, 16, 16], in_channels=1, out_channels=1, model_channels=32, num_res_blocks=1, channel_mult=(1, 2),
              attention_resolutions=[], num_heads=2, use_scale_shift_norm=True, dims=3, compute_dtype="bf16")
    ddpm = DDPM(UNet, kw, LinearSchedule(50, 1e-3, 0.02), nn.MSELoss, timesteps=50).to(DEV).train()
    loss = ddpm.training_step(data)
    loss.backward()
    assert torch.isfinite(loss)


# ----------------------------------------------------------------------------- the headline geometry against the oracle, whole network
def test_c3_whole_network_forward_vs_oracle_at_64_cubed():
    """BASELINE configs[2]'s network (3-D 64^3, mc = 64: 166.8 M parameters, T = 4096 attention) at batch 1 against the CPU oracle
    on the same weights and input - the whole forward at the benchmarked geometry, not a small-grid stand-in: exact-f32 engine
    rel-L2 <= 1e-4, bf16 engine <= 3e-2; and sample independence at batch 2 (per-sample GroupNorm / attention, layers.py:71-74)."""
    model = _bench_unet(3, 64, 64, "fp32", False)
    x = det_normal((2, 1, 64, 64, 64), "r3c3x")
    t = torch.tensor([741, 12])
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cfg = dict(data_shape=[64, 64, 64], in_channels=1, out_channels=1, model_channels=64, num_res_blocks=2, channel_mult=(1, 2, 4, 8),
               attention_resolutions=[16, 8], num_heads=4, use_scale_shift_norm=True, dims=3, activation="SiLU")
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = R.unet_forward(sd, cfg, x[:1], t[:1])
        p32 = model(x[:1].to(DEV), t[:1].to(DEV))
        e32 = rel_l2(p32, ref)
        model._engines.clear()
        torch.cuda.empty_cache()
        model.set_compute_dtype("bf16")
        pbf = model(x.to(DEV), t.to(DEV))
        pbf0 = model(x[:1].to(DEV).contiguous(), t[:1].to(DEV))
    ebf = rel_l2(pbf[:1], ref)
    assert e32 < 1e-4 and ebf < 3e-2, (e32, ebf)
    assert rel_l2(pbf[:1], pbf0) < 2e-3           # the same sample alone (fused-statistics tiles are summed in another order)


@pytest.mark.parametrize("train", [False, True], ids=["inference", "training"])
def test_folded_skips_and_direct_ends_agree_with_the_unfused_plan(monkeypatch, train):
    """The same bf16 network (3-D, mc 32 at 32^3: skips of 64 - 256 couts, 1-channel stem and head) planned with the round-3 fusions
    (ResBlock skip inside the out-conv launch, stem / head as single launches) and without them: outputs agree to the bf16
    rounding of the tensors that no longer exist (`sk`, the im2col / tap-partial intermediates), the fused plan launches fewer
    kernels, and (training) every parameter gradient agrees - the backward graph of the fused plan is the unfused one."""
    x = det_normal((2, 1, 32, 32, 32), "r3fsx").to(DEV)
    t = torch.tensor([321, 45], device=DEV)
    outs, kinds, grads = [], [], []
    for on in ("1", "0"):
        monkeypatch.setenv("RHO_FOLD_SKIP", on)
        monkeypatch.setenv("RHO_DIRECT_ENDS", on)
        model = _bench_unet(3, 32, 32, "bf16", False)
        if train:
            model.train()
            pred = model(x, t)
            (pred.float() ** 2).mean().backward()
            grads.append({n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})
        else:
            with torch.no_grad():
                pred = model(x, t)
        plan = next(iter(model.engine()._plans.values()))
        kinds.append([i["kind"] for i in plan.info])
        outs.append(pred.detach().float().clone())
        del model, plan
        torch.cuda.empty_cache()
    assert torch.isfinite(outs[0]).all()
    # (two bf16 plans that round different intermediates: 0.8 - 1.0e-2 on this network, whichever kernels run the 32-channel level)
    assert rel_l2(outs[0], outs[1]) < 1.5e-2, rel_l2(outs[0], outs[1])
    fused, plain = kinds
    assert fused.count("conv1") < plain.count("conv1")                      # the skip launches are gone
    if not train:
        assert "stem" in fused and "head" in fused and "tap_sum" not in fused and "pack" not in fused
        assert "tap_sum" in plain and "stem" not in plain
    else:
        assert set(grads[0]) == set(grads[1])
        gtot = math.sqrt(sum(float(g.double().norm()) ** 2 for g in grads[1].values()))
        bad = []
        for n in grads[0]:
            a, b = grads[0][n].flatten().double(), grads[1][n].flatten().double()
            if float(b.norm()) < 1e-4 * gtot:          # numerically zero (e.g. the key bias of an attention block): magnitude only
                if float(a.norm()) > 1e-3 * gtot:
                    bad.append((n, "should be ~0", float(a.norm()), float(b.norm())))
                continue
            c = float(torch.dot(a, b) / (a.norm() * b.norm()))
            if c < 0.98 or abs(float(a.norm()) - float(b.norm())) > 0.1 * float(b.norm()):
                bad.append((n, round(c, 4), float(a.norm()), float(b.norm())))
        assert not bad, bad[:8]
