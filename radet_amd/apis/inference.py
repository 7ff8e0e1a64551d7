"""init_detector / inference_detector with the reference's signatures (radet/apis/inference.py), for inputs that are
already normalised NCHW tensors (image decoding / resizing is the data pipeline's job and out of scope here)."""
import numpy as np
import torch

from ..models import build_detector
from ..utils import Config
from .train import load_checkpoint


def init_detector(config, checkpoint=None, device="cuda:0", cfg_options=None):
    if isinstance(config, str):
        config = Config.fromfile(config)
    if cfg_options:
        config.merge_from_dict(cfg_options)
    config.model["pretrained"] = None
    model = build_detector(config.model, train_cfg=config.get("train_cfg"), test_cfg=config.get("test_cfg"))
    if checkpoint is not None:
        meta, _ = load_checkpoint(model, checkpoint)
        model.CLASSES = meta.get("CLASSES")
    model.cfg = config
    return model.to(device).eval()


def inference_detector(model, imgs, scale_factor=None):
    """imgs: f32[B,3,H,W] (normalised). Returns list[B] of list[num_classes] of ndarray[k,5]."""
    if imgs.dim() == 3:
        imgs = imgs[None]
    B, _, H, W = imgs.shape
    sf = np.ones(4, np.float32) if scale_factor is None else np.asarray(scale_factor, np.float32)
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=sf, flip=False) for _ in range(B)]
    with torch.no_grad():
        return model(img=[imgs], img_metas=[metas], return_loss=False, rescale=True)
