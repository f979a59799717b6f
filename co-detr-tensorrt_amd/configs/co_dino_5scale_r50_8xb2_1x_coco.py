# Non-LSJ variant (reference configs/co_dino_5scale_r50_8xb2_1x_coco.py): same model, plain test pipeline.
_base_ = './co_dino_5scale_r50_lsj_8xb2_1x_coco.py'

model = dict(use_lsj=False, data_preprocessor=dict(pad_mask=False, batch_augments=None))

test_pipeline = [
    dict(type='LoadImageFromFile', backend_args=_base_.backend_args),
    dict(type='Resize', scale=(1333, 800), keep_ratio=True),
    dict(type='PackDetInputs', meta_keys=('img_id', 'img_path', 'ori_shape', 'img_shape', 'scale_factor')),
]
val_dataloader = dict(dataset=dict(pipeline=test_pipeline))
test_dataloader = val_dataloader
