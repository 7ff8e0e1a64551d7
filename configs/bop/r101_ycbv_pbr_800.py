_base_ = ['./r50_ycbv_pbr.py']
# BASELINE.json config 5: ResNet-101 backbone, 800x800 input, bs 2 / GPU
model = dict(backbone=dict(depth=101))
data = dict(samples_per_gpu=2)
