# Co-DINO 5-scale, ResNet-50 -- model definition (inference subset).
# Mirrors the `model = dict(...)` section of the reference's base config
# (reference configs/co_dino_5scale_r50_lsj_8xb2_1x_coco.py:14-275).  The auxiliary training heads
# (rpn_head / roi_head / bbox_head), losses' training-only knobs, optimiser, schedule and data
# pipelines are not on the inference path and are left out; a reference config file that still
# carries them loads fine -- `CoDETR` ignores those keys exactly like the reference does.
_base_ = 'mmdet::common/ssj_scp_270k_coco-instance.py'

num_dec_layer = 6
num_classes = 80

model = dict(
    type='CoDETR',
    use_lsj=True,
    eval_module='detr',
    data_preprocessor=dict(
        type='DetDataPreprocessor',
        mean=[123.675, 116.28, 103.53],
        std=[58.395, 57.12, 57.375],
        bgr_to_rgb=True,
        pad_mask=True),
    backbone=dict(
        type='ResNet',
        depth=50,
        num_stages=4,
        out_indices=(0, 1, 2, 3),
        frozen_stages=1,
        norm_cfg=dict(type='BN', requires_grad=False),
        norm_eval=True,
        style='pytorch'),
    neck=dict(
        type='ChannelMapper',
        in_channels=[256, 512, 1024, 2048],
        kernel_size=1,
        out_channels=256,
        act_cfg=None,
        norm_cfg=dict(type='GN', num_groups=32),
        num_outs=5),
    query_head=dict(
        type='CoDINOHead',
        num_query=900,
        num_classes=num_classes,
        in_channels=2048,
        as_two_stage=True,
        transformer=dict(
            type='CoDinoTransformer',
            with_coord_feat=False,
            num_co_heads=2,
            num_feature_levels=5,
            encoder=dict(
                type='DetrTransformerEncoder',
                num_layers=6,
                with_cp=4,
                transformerlayers=dict(
                    type='BaseTransformerLayer',
                    attn_cfgs=dict(type='MultiScaleDeformableAttention', embed_dims=256, num_levels=5, dropout=0.0),
                    feedforward_channels=2048,
                    ffn_dropout=0.0,
                    operation_order=('self_attn', 'norm', 'ffn', 'norm'))),
            decoder=dict(
                type='DinoTransformerDecoder',
                num_layers=num_dec_layer,
                return_intermediate=True,
                transformerlayers=dict(
                    type='DetrTransformerDecoderLayer',
                    attn_cfgs=[
                        dict(type='MultiheadAttention', embed_dims=256, num_heads=8, dropout=0.0),
                        dict(type='MultiScaleDeformableAttention', embed_dims=256, num_levels=5, dropout=0.0),
                    ],
                    feedforward_channels=2048,
                    ffn_dropout=0.0,
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')))),
        positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, temperature=20, normalize=True),
        loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=1.0),
        loss_bbox=dict(type='L1Loss', loss_weight=5.0),
        loss_iou=dict(type='GIoULoss', loss_weight=2.0)),
    train_cfg=[None, None],
    test_cfg=[
        # the Inferencer applies hard per-class NMS with this IoU threshold (reference inferencer.py:66-71)
        dict(max_per_img=300, nms=dict(type='soft_nms', iou_threshold=0.8)),
    ])
