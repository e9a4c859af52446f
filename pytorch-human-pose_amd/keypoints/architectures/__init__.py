from .higher_hrnet import HigherHRNet

__all__ = ["HigherHRNet"]
