"""TXT-branch heads of `BEV_TXT` (reference: src/modules.py:133-254).

OUT OF THE HOT PATH (<1 % of the FLOPs, "stays stock
PyTorch on ROCm").  They exist here only so that `BEV_TXT` is constructible with
the reference's `state_dict` layout; nothing in csrc/ touches them.
"""
import torch
from torch import nn
from torch.nn import functional as F


class BevPost(nn.Module):
    def __init__(self, in_channels=4, out_channels=8):
        super().__init__()
        self.post = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=(2, 1), padding=1, bias=False),
            nn.BatchNorm2d(out_channels), nn.ReLU(), nn.MaxPool2d(kernel_size=(5, 4), padding=0))

    def forward(self, x):
        return self.post(x)


def _conv_bn_relu(cin, cout, k, **kw):
    return [nn.Conv2d(cin, cout, k, bias=False, **kw), nn.BatchNorm2d(cout), nn.ReLU()]


class ASPPConv(nn.Sequential):
    def __init__(self, in_channels, out_channels, dilation):
        super().__init__(*_conv_bn_relu(in_channels, out_channels, 3, padding=dilation, dilation=dilation))


class ASPPPooling(nn.Sequential):
    def __init__(self, in_channels, out_channels):
        super().__init__(nn.AdaptiveAvgPool2d(1), *_conv_bn_relu(in_channels, out_channels, 1))

    def forward(self, x):
        size = x.shape[-2:]
        for mod in self:
            x = mod(x)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


class ASPP(nn.Module):
    def __init__(self, in_channels, atrous_rates, out_channels=256):
        super().__init__()
        branches = [nn.Sequential(*_conv_bn_relu(in_channels, out_channels, 1))]
        branches += [ASPPConv(in_channels, out_channels, r) for r in tuple(atrous_rates)]
        branches.append(ASPPPooling(in_channels, out_channels))
        self.convs = nn.ModuleList(branches)
        self.project = nn.Sequential(*_conv_bn_relu(len(self.convs) * out_channels, out_channels, 1),
                                     nn.Dropout(0.5))

    def forward(self, x):
        return self.project(torch.cat([conv(x) for conv in self.convs], dim=1))


class SceneUnder(nn.Sequential):
    def __init__(self, in_channels=512):
        super().__init__(ASPP(in_channels, [12, 24, 36]))


class Embedder_lr1(nn.Sequential):
    def __init__(self, in_channels, out_channels):
        super().__init__(*_conv_bn_relu(in_channels, out_channels, 3, padding=1))


class Embedder_f1(Embedder_lr1):
    pass


class Embedder_lr2(nn.Sequential):
    def __init__(self, out_channels):
        # 22 x 8 = the trunk feature map of a 352 x 128 image
        super().__init__(nn.Flatten(), nn.Linear(out_channels * 22 * 8, out_channels, bias=True))


class Embedder_f2(Embedder_lr2):
    pass


class Predictor(nn.Sequential):
    def __init__(self, num_in, classes):
        super().__init__(nn.Linear(num_in, classes, bias=True))
