from .architectures import HigherHRNet
from .grouping import MPPEHeatmapParser
from .model import InferenceKeypointsModel
from .results import InferenceKeypointsResult

__all__ = ["HigherHRNet", "MPPEHeatmapParser", "InferenceKeypointsModel", "InferenceKeypointsResult"]
