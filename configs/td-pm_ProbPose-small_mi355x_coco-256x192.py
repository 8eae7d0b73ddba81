# ProbPose-small on MI355X, inference settings. Mirrors the model/codec/test keys of the reference's
# configs/body_2d_keypoint/topdown_probmap/coco/td-pm_ProbPose-small_8xb64-210e_coco-256x192.py
# (:48 codec, :51-93 model, :183-184 score threshold); training/optimizer/dataset entries are
# not part of the inference hot path and are left out. With a real MMPose install, the reference
# config itself can be used by adding `custom_imports = dict(imports=["probpose_code_amd"])` and
# exporting PROBPOSE_MI355X_OVERRIDE=1.
custom_imports = dict(imports=["probpose_code_amd"], allow_failed_imports=False)
default_scope = "mmpose"

INPUT_PADDING = 1.25
TEST_BATCH_SIZE = 64

codec = dict(type="ProbMap", input_size=(192, 256), heatmap_size=(48, 64), sigma=-1)

model = dict(
    type="TopdownPoseEstimator",
    # MI355X-only key: operand precision of the MFMA kernels. "f16x3" (split-fp16 operands, three fp16 MFMAs per product) is
    # the mode that matches the fp32 reference within 1e-3; "bf16" is ~3x faster and does not (0.3 - 0.45 px); "f32" = exact
    # fp32 products, slowest
    precision="f16x3",
    data_preprocessor=dict(
        type="PoseDataPreprocessor", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], bgr_to_rgb=True
    ),
    backbone=dict(
        type="mmpretrain.VisionTransformer",
        arch={"embed_dims": 384, "num_layers": 12, "num_heads": 12, "feedforward_channels": 384 * 4},
        img_size=(256, 192),
        patch_size=16,
        qkv_bias=True,
        drop_path_rate=0.1,
        with_cls_token=False,
        out_type="featmap",
        patch_cfg=dict(padding=2),
        init_cfg=None,
    ),
    head=dict(
        type="ProbMapHead",
        in_channels=384,
        out_channels=17,
        deconv_out_channels=(256, 256),
        deconv_kernel_sizes=(4, 4),
        keypoint_loss=dict(type="OKSHeatmapLoss", use_target_weight=True, smoothing_weight=0.05),
        probability_loss=dict(type="BCELoss", use_target_weight=True, use_sigmoid=True),
        visibility_loss=dict(type="BCELoss", use_target_weight=True, use_sigmoid=True),
        oks_loss=dict(type="MSELoss", use_target_weight=True),
        error_loss=dict(type="L1LogLoss", use_target_weight=True),
        detach_probability=True,
        detach_visibility=True,
        normalize=1.0,
        freeze_error=True,
        freeze_oks=False,
        decoder=codec,
    ),
    test_cfg=dict(flip_test=True, flip_mode="heatmap", shift_heatmap=False),
)

val_pipeline = [
    dict(type="LoadImage", pad_to_aspect_ratio=False),
    dict(type="GetBBoxCenterScale"),
    dict(type="TopdownAffine", input_size=codec["input_size"], use_udp=True, input_padding=INPUT_PADDING),
    dict(type="PackPoseInputs"),
]
test_dataloader = dict(batch_size=TEST_BATCH_SIZE, dataset=dict(pipeline=val_pipeline))
