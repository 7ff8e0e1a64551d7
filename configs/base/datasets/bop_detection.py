# data pipeline keys consumed by the hot path (assigner + batch layout); image IO / augmentation are out of scope
dataset_type = 'BOPDataset'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
label_assignment = dict(
    type='LabelAssignment',
    anchor_generator_cfg=dict(type='AnchorGenerator', ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                              strides=[8, 16, 32, 64, 128]),
    neg_threshold=0.2, positive_num=10, adapt_positive_num=False, balance_sample=True)
train_pipeline = [dict(type='GenerateDistanceMap'), label_assignment]
data = dict(samples_per_gpu=16, workers_per_gpu=4)
