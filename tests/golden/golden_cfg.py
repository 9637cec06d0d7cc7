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
