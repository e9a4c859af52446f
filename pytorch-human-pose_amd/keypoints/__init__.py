from .architectures import HigherHRNet
from .grouping import MPPEHeatmapParser
from .model import InferenceKeypointsModel
from .results import InferenceKeypointsResult

__all__ = ["HigherHRNet", "MPPEHeatmapParser", "InferenceKeypointsModel", "InferenceKeypointsResult"]
from .loss import AEGroupingLoss, AEKeypointsLoss, HeatmapsLoss
from . import coco_eval, evaluation, targets
