from .architectures import HigherHRNet
from .grouping import MPPEHeatmapParser
from .model import InferenceKeypointsModel, KeypointsModel, KeypointsModule
from .results import InferenceKeypointsResult, KeypointsResult

__all__ = ["HigherHRNet", "MPPEHeatmapParser", "InferenceKeypointsModel", "InferenceKeypointsResult", "KeypointsModel", "KeypointsModule"]
from .loss import AEGroupingLoss, AEKeypointsLoss, HeatmapsLoss
from . import coco_eval, evaluation, targets
