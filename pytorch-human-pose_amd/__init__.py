"""MI355X-native HigherHRNet forward + associative-embedding decode (drop-in for the
`src.keypoints` model/inference API of thawro/pytorch-human-pose).  See DESIGN.md."""
from . import _lib, synth
from .classification.architectures import ClassificationHRNet
from . import keypoints
from .keypoints import AEKeypointsLoss, HigherHRNet, InferenceKeypointsModel, InferenceKeypointsResult, MPPEHeatmapParser

__all__ = ["ClassificationHRNet", "HigherHRNet", "MPPEHeatmapParser", "InferenceKeypointsModel", "InferenceKeypointsResult", "synth", "_lib"]
