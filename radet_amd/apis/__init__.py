from .train import OneCycleLR, load_checkpoint, save_checkpoint, train_detector
from .inference import inference_detector, init_detector

__all__ = ["OneCycleLR", "train_detector", "save_checkpoint", "load_checkpoint", "init_detector", "inference_detector"]
