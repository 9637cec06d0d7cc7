"""CPU: host-side mirror of the reference interface (registry, schedules, module tree / state_dict
layout, conditioning, helper functions).  No kernels are launched."""
import numpy as np
import pytest
import torch
from pytest import approx
from torch import nn

from helpers import PARAM_SPACE, UNET_CASES, det_normal, det_state_dict, golden_template, load_golden
from oracle import ref_torch as R

import rho_diffusion_amd as RA
from rho_diffusion_amd import registry
from rho_diffusion_amd.diffusion import schedule as s


def test_linear_schedule_known_answers():
    """Same assertions as the reference's tests/pipeline/test_schedule.py:28-46."""
    schedule = s.LinearSchedule(100, 1e-4, 0.02)
    beta_t = schedule.beta_t
    assert len(beta_t) == 100 and torch.is_floating_point(beta_t)
    assert beta_t[0] == 0.001 and beta_t[-1] == 0.2
    assert schedule.alpha_t[0] == 0.999 and schedule.alpha_t[-1] == 0.8
    assert schedule.sigma_t[0] == 0.0
    assert approx(schedule.sigma_t[-1], 1e-4) == 0.4472


def test_schedules_in_registry_and_bit_exact():
    for name in s.__all__:
        assert registry.get("schedules", name)
    g = load_golden("g1_schedules.npz")
    for key in g.files:
        name, arr = key.split("/")
        parts = name.split("_")
        sch = s.LinearSchedule(int(parts[1]), float(parts[2]), float(parts[3])) if parts[0] == "lin" else s.CosineBetaSchedule(int(parts[1]))
        np.testing.assert_array_equal(sch[arr].numpy(), g[key], err_msg=key)
    with pytest.raises(NotImplementedError):
        s.SigmoidSchedule(10)


def test_schedule_dtype_protocol():
    sch = s.LinearSchedule(1000)
    sch.dtype = torch.bfloat16
    assert sch["alpha_bar_t"].dtype == torch.bfloat16 and len(sch["alpha_bar_t"]) == 1000
    sch.dtype = None
    assert sch["beta_t"].dtype == torch.float32
    assert len(s.CosineBetaSchedule(1000)["alpha_bar_t"]) == 1001   # T + 1 entries


def test_registry_contract():
    assert registry.get("models", "UNetv2") is RA.models.UNet
    assert registry.get("layers", "MultiEmbeddings") is RA.models.MultiEmbeddings
    assert registry.get("layers", "GroupNorm32") is RA.layers.GroupNorm32
    assert registry.get("layers", "conv_nd") is RA.layers.conv_nd
    assert registry.get("optimizers", "AdamW") is torch.optim.AdamW
    assert registry.get("nn", "MSELoss") is nn.MSELoss
    assert registry.get("activations", "SiLU") is nn.SiLU
    with pytest.raises(KeyError):
        registry.get("models", "nope")
    with pytest.raises(AssertionError):
        registry.get("nope", "UNetv2")
    with pytest.raises(ValueError):
        RA.layers.conv_nd(4, 1, 1, 3)


@pytest.mark.parametrize("case", list(UNET_CASES.keys()))
def test_state_dict_layout_matches_reference(case):
    g = load_golden("g4_unet.npz")
    kw, _, ykind = UNET_CASES[case]
    model = RA.models.UNet(**dict(kw))
    if ykind == "multi":
        model.cond_fn = RA.models.MultiEmbeddings(parameter_space=PARAM_SPACE, embedding_dim=4 * kw["model_channels"])
    ours = [f"{k}|{','.join(map(str, v.shape))}" for k, v in model.state_dict().items()]
    assert ours == [str(x) for x in g[f"{case}/keys"]]
    # zero_module quirk
    assert float(model.out[2].weight.abs().max()) == 0.0
    # structure replay used by the oracle agrees with the module tree
    st = R.unet_structure(dict(kw))
    assert len(st["input"]) == len(model.input_blocks) and len(st["output"]) == len(model.output_blocks)


def test_unet_asserts_label_contract():
    kw, xshape, _ = UNET_CASES["tiny2d"]
    model = RA.models.UNet(**dict(kw))
    with pytest.raises(AssertionError):
        model(torch.zeros(xshape), torch.zeros(2, dtype=torch.long), torch.zeros(2))   # y given, not class-conditional


def test_multi_embeddings_layout_and_no_cpu_path():
    """Table layout = the reference's (conditioning.py:58-60); the lookup itself is a HIP kernel (tests/test_gpu_round2.py)."""
    from rho_diffusion_amd.hip import RhoHipError
    me = RA.models.MultiEmbeddings(parameter_space=PARAM_SPACE, embedding_dim=16)
    assert [(k, tuple(v.shape)) for k, v in me.state_dict().items()] == \
        [(f"embedding_layers.{k}.weight", (len(v), 16)) for k, v in PARAM_SPACE.items()]
    with pytest.raises(RhoHipError):
        me(torch.tensor([[1.0, 2.5]]))
    assert RA.models.MultiEmbeddings(parameter_space=None)(torch.zeros(2, 2)) is None


def test_qkv_row_permutation_makes_orders_canonical():
    """The engine's row gather turns the legacy per-head [q,k,v] interleave into [Q|K|V]; the oracle's
    new-order attention on permuted rows must equal legacy attention on the original rows."""
    from rho_diffusion_amd.engine.unet_engine import UNetEngine
    c, heads, T = 64, 4, 10
    blk = RA.models.AttentionBlock(c, num_heads=heads)
    fake = type("E", (), {"device": "cpu"})()
    src = UNetEngine._qkv_row_src(fake, blk).long()
    qkv = det_normal((2, 3 * c, T), "perm")
    assert torch.allclose(R.qkv_attention(qkv[:, src], heads, new_order=True), R.qkv_attention(qkv, heads, new_order=False))
    blk2 = RA.models.AttentionBlock(c, num_heads=heads, use_new_attention_order=True)
    assert torch.equal(UNetEngine._qkv_row_src(fake, blk2).long(), torch.arange(3 * c))


def test_utils():
    from rho_diffusion_amd import utils
    space = {"l": [1, 2], "m": [3, 4, 5]}
    c = utils.sample_from_discrete_parameter_space(space, 4, random=False)
    assert c.tolist() == [[1, 3], [1, 4], [1, 5], [2, 3]]
    e = utils.calculate_sha512_embedding({"l": 1}, 256)
    assert e.shape == (256,) and float(e.max()) < 1.0
    assert utils.number_cast_dict({"a": "3", "b": ["1.5", "x"]}) == {"a": 3, "b": [1.5, "x"]}


def test_ddpm_constructor_and_optimizer_contract():
    kw, _, _ = UNET_CASES["tiny2d"]
    ddpm = RA.diffusion.DDPM("UNetv2", dict(kw), s.LinearSchedule(1000, 1e-4, 0.02), "MSELoss",
                             opt_kwargs={"lr": 1e-4})
    assert isinstance(ddpm.backbone, RA.models.UNet) and isinstance(ddpm.loss_func, nn.MSELoss)
    t = ddpm.random_timesteps(16)
    assert t.shape == (16,) and int(t.max()) < 1000
    assert ddpm.reshape_timesteps(torch.zeros(4, 1, 8, 8), torch.arange(4)).shape == (4, 1, 1, 1)
    opt = ddpm.configure_optimizers(mpi_world_size=4)["optimizer"]
    assert opt.param_groups[0]["lr"] == pytest.approx(2e-4)          # lr * sqrt(world)  (abstract_diffusion.py:118)
    assert opt.param_groups[0]["weight_decay"] == 1e-2 and opt.param_groups[0]["betas"] == (0.9, 0.999)
    assert ddpm.hparams.opt_kwargs == {"lr": 1e-4}                    # not mutated


def test_experiment_config_loader(tmp_path):
    """config.py: the attribute surface of the reference's ExperimentConfig (config.py:80-110) from plain JSON; booleans stay
    booleans, numeric strings are cast (utils.py:223-244), unknown training keys are dropped, missing sections fail."""
    import json
    from rho_diffusion_amd.config import ExperimentConfig
    cfg = {"experiment": "x", "model": {"name": "UNetv2", "kwargs": {"dims": "3", "use_new_attention_order": False, "lr": "1e-4",
                                                                      "data_shape": [8, "16", 16.0], "flag": True, "act": "SiLU"}},
           "dataset": {"name": "d", "kwargs": {}}, "optimizer": {"name": "AdamW", "kwargs": {"lr": 0.0001}},
           "lr_scheduler": {"name": "s", "kwargs": {}}, "noise_schedule": {"name": "LinearSchedule", "kwargs": {"num_steps": 1000.0}},
           "training": {"device": "xpu", "np": 1, "benchmark_mode": True, "batch_size": "16"},
           "inference": {"device": "xpu", "checkpoint": "model.pth", "parameter_space": {"l": [1, 2]}}}
    p = tmp_path / "c.json"
    p.write_text(json.dumps(cfg))
    c = ExperimentConfig.from_json(str(p))
    kw = c.model.kwargs
    assert kw["dims"] == 3 and isinstance(kw["dims"], int) and kw["lr"] == 1e-4 and kw["data_shape"] == [8, 16, 16]
    assert kw["use_new_attention_order"] is False and kw["flag"] is True and kw["act"] == "SiLU"
    assert c.noise_schedule.kwargs["num_steps"] == 1000 and isinstance(c.noise_schedule.kwargs["num_steps"], int)
    assert c.training.batch_size == 16 and c.training.loss_fn == "MSELoss" and not hasattr(c.training, "np")
    assert c.inference.parameter_space == {"l": [1, 2]} and c.inference.cache_file is None
    with pytest.raises(FileNotFoundError):
        ExperimentConfig.from_json(str(tmp_path / "missing.json"))
    del cfg["noise_schedule"]
    p.write_text(json.dumps(cfg))
    with pytest.raises(ValueError):
        ExperimentConfig.from_json(str(p))


def test_alias_makes_reference_imports_resolve_here():
    import sys
    RA.install_alias()
    import rho_diffusion
    from rho_diffusion.registry import registry as reg2
    from rho_diffusion.diffusion import DDPM
    from rho_diffusion.diffusion.diffusers import DiffusersDDPMPipeline
    assert rho_diffusion is RA and reg2 is registry and DDPM is RA.diffusion.DDPM and DiffusersDDPMPipeline is RA.diffusion.DiffusersDDPMPipeline
    assert registry.get("datasets", "SphericalHarmonicDataset") is RA.data.SphericalHarmonicDataset
    for k in [k for k in sys.modules if k == "rho_diffusion" or k.startswith("rho_diffusion.")]:
        del sys.modules[k]


def test_unetv2_refuses_dropout_and_unbuilt_activations_naming_the_reference_lines():
    """The two constructor arguments of the reference's UNetv2 that this engine does not build (VERDICT r3 missing #2): the refusal is
    loud, at construction, and says which reference lines it stands for."""
    from rho_diffusion_amd.models import UNet
    kw = dict(data_shape=[16, 16], in_channels=1, out_channels=1, model_channels=32, num_res_blocks=1, channel_mult=(1, 2),
              attention_resolutions=[], num_heads=2, dims=2)
    with pytest.raises(ValueError, match=r"unet_v2\.py:239"):
        UNet(**kw, dropout=1.0)
    assert UNet(**kw, dropout=0.1).input_blocks[1][0].dropout == 0.1          # built since round 4 (Philox masks, tests/test_gpu_round4.py)
    for bad in ("PReLU", "Softmax", "LogSoftmax"):
        with pytest.raises(NotImplementedError, match=r"unet_v2\.py:518-519"):
            UNet(**kw, activation=bad)
    with pytest.raises(NotImplementedError, match=r"unet_v2\.py:518-519"):
        UNet(**kw, activation=torch.nn.GELU(approximate="tanh"))
    for good, code in (("SiLU", 1), ("ReLU", 2), ("GELU", 3), ("Tanh", 4), ("Sigmoid", 5), ("ELU", 6)):
        assert UNet(**kw, dropout=0.0, activation=good).act_code == code


def test_head_dgrad_weight_layout_is_the_mirrored_taps_as_contraction_form():
    """engine._HeadDgradW (round 4): w[0][c][t] = weight[0][c][26 - t] for t < 27, zero beyond - what rho_stem_conv3d needs to compute
    the data gradient of the one-output-channel head conv from dpred (host-side index table, checked without a GPU)."""
    from rho_diffusion_amd.engine.unet_engine import _HeadDgradW
    w = torch.nn.Parameter(torch.arange(1 * 4 * 27, dtype=torch.float32).reshape(1, 4, 3, 3, 3))
    hd = _HeadDgradW(w, torch.float32)
    assert hd.w.shape == (1, 4, 32)
    for c in range(4):
        for t in range(27):
            assert float(hd.w[0, c, t]) == float(w[0, c].reshape(-1)[26 - t])
        assert float(hd.w[0, c, 27:].abs().max()) == 0.0


def test_prep_and_finalize_table_records_match_the_header_structs():
    """ctypes mirrors of rho_prep_op / rho_wfin_op (include/rho_hip.h): sizes and field offsets the kernels index by."""
    import ctypes as C
    from rho_diffusion_amd import hip
    assert C.sizeof(hip.PrepOp) == 128 and hip.PrepOp.total.offset == 56 and hip.PrepOp.kind.offset == 64 and hip.PrepOp.blk0.offset == 116
    assert C.sizeof(hip.WfinOp) == 104 and hip.WfinOp.phase_stride.offset == 64 and hip.WfinOp.kind.offset == 72 and hip.WfinOp.blk0.offset == 96
