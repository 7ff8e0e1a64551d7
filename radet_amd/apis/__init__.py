from .train import OneCycleLR, load_checkpoint, save_checkpoint, train_detector, wrap_fp16_model
from .inference import inference_detector, init_detector

__all__ = ["OneCycleLR", "train_detector", "wrap_fp16_model", "save_checkpoint", "load_checkpoint", "init_detector", "inference_detector"]
