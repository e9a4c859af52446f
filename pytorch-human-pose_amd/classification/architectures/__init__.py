from .hrnet import ClassificationHRNet

__all__ = ["ClassificationHRNet"]
