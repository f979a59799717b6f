# Vendored stand-in for mmdet's `configs/common/ssj_scp_270k_coco-instance.py` (mmdet v3.3.0), which
# the reference's base config inherits through `_base_ = 'mmdet::common/...'` (reference configs
# co_dino_5scale_r50_lsj_8xb2_1x_coco.py:1) and which is not part of the reference tree.  Only the
# names that child configs READ are provided; inference never touches the training schedule.
dataset_type = 'CocoDataset'
data_root = 'data/coco/'
image_size = (1024, 1024)
backend_args = None

train_dataloader = dict(batch_size=2, num_workers=2, dataset=dict(type=dataset_type, data_root=data_root))
val_dataloader = dict(batch_size=1, num_workers=2, dataset=dict(type=dataset_type, data_root=data_root, test_mode=True))
test_dataloader = val_dataloader
val_evaluator = dict(type='CocoMetric', metric=['bbox'])
test_evaluator = val_evaluator
optim_wrapper = dict(type='OptimWrapper', optimizer=dict(type='AdamW', lr=2e-4, weight_decay=1e-4))
train_cfg = dict(type='EpochBasedTrainLoop', max_epochs=12, val_interval=1)
val_cfg = dict(type='ValLoop')
test_cfg = dict(type='TestLoop')
param_scheduler = []
default_scope = 'mmdet'
