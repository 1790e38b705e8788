import torch.nn as nn

Linear = nn.Linear


class ConvModule(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None, **kw):
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False),
                         nn.GroupNorm(min(64, out_channels), out_channels), nn.ReLU())
