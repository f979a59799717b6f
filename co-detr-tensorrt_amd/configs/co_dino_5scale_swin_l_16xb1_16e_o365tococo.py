# Co-DINO 5-scale, Swin-L (Objects365 -> COCO) -- the headline model
# (reference configs/co_dino_5scale_swin_l_16xb1_16e_o365tococo.py:7-32, 89-98).
_base_ = ['co_dino_5scale_r50_8xb2_1x_coco.py']

model = dict(
    backbone=dict(
        _delete_=True,
        type='SwinTransformer',
        pretrain_img_size=384,
        embed_dims=192,
        depths=[2, 2, 18, 2],
        num_heads=[6, 12, 24, 48],
        window_size=12,
        mlp_ratio=4,
        qkv_bias=True,
        qk_scale=None,
        drop_rate=0.,
        attn_drop_rate=0.,
        drop_path_rate=0.3,
        patch_norm=True,
        out_indices=(0, 1, 2, 3),
        with_cp=True,
        convert_weights=True),
    neck=dict(in_channels=[192, 384, 768, 1536]),
    query_head=dict(transformer=dict(encoder=dict(with_cp=6))))

test_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='Resize', scale=(1152, 768), keep_ratio=True),
    dict(type='Pad', size=(1152, 768)),
    dict(type='PackDetInputs',
         meta_keys=('img_id', 'img_path', 'ori_shape', 'img_shape', 'scale_factor', 'img_unpadded_shape')),
]
val_dataloader = dict(dataset=dict(pipeline=test_pipeline))
test_dataloader = val_dataloader
