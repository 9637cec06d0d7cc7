"""UNetv2 constructor kwargs / input shapes shared by make_golden.py and the tests.
value = (kwargs, x_shape, y_kind) with y_kind in {None, "multi", "preemb"}."""

PARAM_SPACE = {"l": [-2.0, -1.0, 1.0, 2.0], "m": [0.5, 1.5, 2.5]}

_tiny = dict(in_channels=1, out_channels=1, model_channels=32, num_res_blocks=1, channel_mult=(1, 2),
             attention_resolutions=[2], num_heads=2, use_scale_shift_norm=True)

UNET_CASES = {
    # small structure, every block kind present (res same/wide, attn, down, up, concat)
    "tiny2d": (dict(_tiny, dims=2, data_shape=[16, 16]), (2, 1, 16, 16), None),
    "tiny3d": (dict(_tiny, dims=3, data_shape=[4, 8, 8]), (2, 1, 4, 8, 8), None),
    "tiny1d": (dict(_tiny, dims=1, data_shape=[32]), (2, 1, 32), None),
    "tiny2d_neworder": (dict(_tiny, dims=2, data_shape=[16, 16], use_new_attention_order=True), (2, 1, 16, 16), None),
    "tiny2d_multi": (dict(_tiny, dims=2, data_shape=[16, 16], num_classes=12), (3, 1, 16, 16), "multi"),
    "tiny2d_preemb": (dict(_tiny, dims=2, data_shape=[16, 16], num_classes=20), (2, 1, 16, 16), "preemb"),
    # the reference's own test fixture (tests/models/test_unet.py:36-43, test_lightning_ddpm.py:36-43):
    # 3 channels, defaults => additive emb (no scale-shift), 1 head, 4 levels
    "reftest2d": (dict(in_channels=3, out_channels=3, model_channels=32, num_res_blocks=2, data_shape=[16, 16]),
                  (2, 3, 16, 16), None),
    # BASELINE hyper-parameters at mc=32, non-cubic to catch axis mix-ups
    "full2d": (dict(in_channels=1, out_channels=1, model_channels=32, num_res_blocks=2, dims=2, data_shape=[32, 32],
                    attention_resolutions=[16, 8], num_heads=4, use_scale_shift_norm=True), (2, 1, 32, 32), None),
    "full3d": (dict(in_channels=1, out_channels=1, model_channels=32, num_res_blocks=2, dims=3, data_shape=[8, 16, 16],
                    attention_resolutions=[16, 8], num_heads=4, use_scale_shift_norm=True), (2, 1, 8, 16, 16), None),
}

# ---- cases at the widths / structures the bench configurations run: recorded in g12_wide.npz
# c5's parameter space (rho_diffusion/data/deep_galaxy.py:41-47): y [B, 4] = (s, m, t, c)
DEEP_GALAXY_SPACE = {"s": [0.25, 0.5, 0.75, 1, 1.25, 1.5], "m": [0.25, 0.5, 0.75, 1, 1.25, 1.5],
                     "t": list(range(300, 655, 5)), "c": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]}

_base = dict(in_channels=1, out_channels=1, num_res_blocks=2, attention_resolutions=[16, 8], num_heads=4,
             use_scale_shift_norm=True)
WIDE_CASES = {
    # c3's network (mc = 64: 512-channel levels, 1024-channel concatenations, ch = 128 heads) on a small grid
    "wide3d": (dict(_base, model_channels=64, dims=3, data_shape=[4, 32, 32]), (1, 1, 4, 32, 32), None),
    # c1 / c2's network on c1's grid
    "wide2d": (dict(_base, model_channels=64, dims=2, data_shape=[64, 64]), (2, 1, 64, 64), None),
    # c5's structure: 3-D mc = 32, conditioned through MultiEmbeddings(embedding_dim = 128) on y [B, 4]
    "cond3d": (dict(_base, model_channels=32, dims=3, data_shape=[4, 16, 16], num_classes=25), (2, 1, 4, 16, 16), "galaxy"),
}


def galaxy_labels(B):
    keys = list(DEEP_GALAXY_SPACE.keys())
    return [[float(DEEP_GALAXY_SPACE[k][(3 * i + 5 * j + 1) % len(DEEP_GALAXY_SPACE[k])]) for j, k in enumerate(keys)] for i in range(B)]


# ---- K12: ResBlock(up / down) and conv-less (average pool / nearest) resampling (unet_v2.py:165,221-224,277-281); g15_updown.npz
UPDOWN_CASES = {
    "updown2d": (dict(_tiny, dims=2, data_shape=[16, 16], resblock_updown=True), (2, 1, 16, 16), None),
    "updown3d_add": (dict(_tiny, dims=3, data_shape=[4, 8, 8], resblock_updown=True, use_scale_shift_norm=False), (2, 1, 4, 8, 8), None),
    "avgpool3d": (dict(_tiny, dims=3, data_shape=[4, 8, 8], conv_resample=False), (2, 1, 4, 8, 8), None),
    "avgpool1d": (dict(_tiny, dims=1, data_shape=[32], conv_resample=False), (2, 1, 32), None),
    "updown2d_3lvl": (dict(in_channels=1, out_channels=1, model_channels=32, num_res_blocks=1, channel_mult=(1, 2, 2), attention_resolutions=[4],
                           num_heads=2, use_scale_shift_norm=True, dims=2, data_shape=[16, 24], resblock_updown=True, conv_resample=False),
                      (2, 1, 16, 24), None),
}


# legacy UNet ("UNet v1", models/unet.py): kwargs, input shape.  Small channel lists (multiples of 32: the HIP conv's channels-last
# granularity) and the reference's own defaults for everything else; one GELU case (the block's default activation), one without
# residual convolutions.
V1_CASES = {
    "v1_relu": (dict(block_type="UNetBlock2d", input_channels=1, down_channels=[32, 64, 96], up_channels=[96, 64, 32],
                     time_embedding_dim=32, activation="ReLU", residual=True), (2, 1, 16, 24)),
    "v1_gelu_rgb": (dict(block_type="UNetBlock2d", input_channels=3, down_channels=[32, 64], up_channels=[64, 32],
                         time_embedding_dim=32, activation="GELU", residual=True), (2, 3, 16, 16)),
    "v1_plain": (dict(block_type="UNetBlock2d", input_channels=1, down_channels=[32, 64], up_channels=[64, 32],
                      time_embedding_dim=16, activation="SiLU", residual=False), (3, 1, 8, 8)),
}


# ---- UNetv2 built with the registry's other elementwise activations (rho_diffusion/registry.py:162-170, unet_v2.py:518-519): g17_activations.npz
ACT_CASES = {
    "relu3d": (dict(_tiny, dims=3, data_shape=[4, 8, 8], activation="ReLU"), (2, 1, 4, 8, 8), None),
    "gelu2d": (dict(_tiny, dims=2, data_shape=[16, 16], activation="GELU"), (2, 1, 16, 16), None),
    "tanh2d_add": (dict(_tiny, dims=2, data_shape=[16, 16], activation="Tanh", use_scale_shift_norm=False), (2, 1, 16, 16), None),
    "sigmoid1d": (dict(_tiny, dims=1, data_shape=[32], activation="Sigmoid"), (2, 1, 32), None),
    "elu3d_updown": (dict(_tiny, dims=3, data_shape=[4, 8, 8], activation="ELU", resblock_updown=True), (2, 1, 4, 8, 8), None),
}
