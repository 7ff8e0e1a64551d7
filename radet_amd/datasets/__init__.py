from .bop import DATASETS, BOPDataset, YcbvDataset, build_dataset
from .pipelines import PIPELINES, GenerateDistanceMap, LabelAssignment, build_pipeline

__all__ = ["PIPELINES", "LabelAssignment", "GenerateDistanceMap", "build_pipeline", "DATASETS", "BOPDataset", "YcbvDataset",
           "build_dataset"]
