# Import-only stand-in: the reference constructs MLVLROIQueryModule (model/layers.py) but GROVE never
# executes it (bboxes is always None). Container-only, golden generation only.
import torch.nn as nn


class _RoiLayerStub(nn.Module):
    output_size = (14, 14)


class BaseRoIExtractor(nn.Module):
    def __init__(self, roi_layer=None, out_channels=None, featmap_strides=None, init_cfg=None):
        super().__init__()
        self.roi_layers = nn.ModuleList([_RoiLayerStub() for _ in (featmap_strides or [1])])
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
